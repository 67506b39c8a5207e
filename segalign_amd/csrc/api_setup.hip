// api_setup.hip -- C-ABI, set-up side: InitializeInterface / InitializeProcessor / ShutdownProcessor, target and query upload,
// GenerateShapePos / GenerateSeedPosTable (common/seed_filter_interface.cu, common/seed_pos_table.cu, src/seed_filter.cu:830-940).
#include "engine_internal.h"

namespace sa {

// Every slot issues its kernels on a stream of its own, next to the upload stream.  The HIP runtime multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4), and two slots that share a queue run one after the other: with four slots the
// small kernels of one call then wait behind another call's filter kernel instead of overlapping it (0.94 -> 1.03 Gbp/s on the
// default workload with 8 queues).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process.
// A library has no business changing its host's environment (rounds 2-4 did, from a load-time constructor): the requirement is
// documented (export GPU_MAX_HW_QUEUES=8, as bench.py and the C++ hosts do for themselves), and
// InitializeProcessor says so ONCE on stderr whenever the slots of a device PLUS its upload stream outnumber the hardware queues --
// which includes the default case, four slots on an unset variable (four queues): that is the configuration measured at -6 %.
static void report_hw_queues(bool debug) {
    const char* v = getenv("GPU_MAX_HW_QUEUES");
    const int have = v ? atoi(v) : 4;
    static bool warned = false;
    if (debug)
        fprintf(stderr, "engine: %d slot(s) per device, GPU_MAX_HW_QUEUES=%s\n", SLOTS_PER_DEVICE, v ? v : "unset (the runtime's default is 4)");
    if (!warned && have < 8 && SLOTS_PER_DEVICE + 1 > have) {  // (+ 1: the upload stream shares the queues)
        warned = true;
        fprintf(stderr, "segalign_amd: %d engine slots per device + the upload stream on %d hardware queues: streams that share a queue run one "
                        "after the other; export GPU_MAX_HW_QUEUES=8 before the process initialises HIP (INTEGRATION.md)\n", SLOTS_PER_DEVICE, have);
    }
}

// ---- ASCII upload through the pinned ring (see DevCtx) -------------------------------------------------------------
constexpr size_t UP_CHUNK = (size_t)32 << 20;
const uint8_t* upload_ascii(DevCtx* dc, const char* src, size_t len, const char* tag) {
    hipStream_t st = dc->admin;
    dc->up_tmp.ensure(len + 64, tag);
    for (int k = 0; k < 2; k++)
        if (!dc->up_pinned[k]) {
            if (hipHostMalloc((void**)&dc->up_pinned[k], UP_CHUNK) != hipSuccess || hipEventCreateWithFlags(&dc->up_ev[k], hipEventDisableTiming) != hipSuccess) {
                fprintf(stderr, "Error: hipHostMalloc for the upload ring failed\n");
                exit(12);
            }
        }
    size_t i = 0;
    for (size_t off = 0; off < len; off += UP_CHUNK, i++) {
        const int k = (int)(i & 1);
        const size_t n = std::min(UP_CHUNK, len - off);
        if (i >= 2) hipEventSynchronize(dc->up_ev[k]);  // the DMA that last used this pinned buffer has finished
        memcpy(dc->up_pinned[k], src + off, n);
        check_memcpy(hipMemcpyAsync(dc->up_tmp.p + off, dc->up_pinned[k], n, hipMemcpyHostToDevice, st), tag);
        hipEventRecord(dc->up_ev[k], st);
    }
    return dc->up_tmp.p;
}

// code presence of a freshly encoded block -> *host_mask (one small D2H; the callers synchronise the admin stream anyway)
void presence_of(DevCtx* dc, const uint8_t* codes, uint32_t len, int slot, uint32_t* host_mask) {
    if (!dc->d_present) dc->d_present = (uint32_t*)dev_malloc((1 + SA_BUFFER_DEPTH) * sizeof(uint32_t), "code presence");
    check_memcpy(hipMemsetAsync(dc->d_present + slot, 0, sizeof(uint32_t), dc->admin), "code presence");
    launch_code_presence(codes, len, dc->d_present + slot, dc->admin);
    check_memcpy(hipMemcpyAsync(host_mask, dc->d_present + slot, sizeof(uint32_t), hipMemcpyDeviceToHost, dc->admin), "code presence");
}

}  // namespace sa

using namespace sa;

extern "C" {

const char* sa_version(void) { return "segalign_amd 0.1 (gfx950)"; }

void sa_select_devices(const int* ids, int n) {
    g_selected.clear();
    for (int i = 0; i < n; i++) g_selected.push_back(ids[i]);
}

void sa_shutdown_processor(void);
static void destroy_interface();

int sa_initialize_interface(int num_gpu) {  // seed_filter_interface.cu:49-80
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0) {
        fprintf(stderr, "Error: No GPU device found!\n");
        exit(1);
    }
    if (!g_selected.empty()) {
        // (an ordinal may be named more than once: every entry becomes an engine device of its own -- contexts, streams, tables,
        //  arena, token-pool slots -- on that GPU.  That is how the multi-device paths are exercised on a one-GPU box.)
        for (int id : g_selected)
            if (id < 0 || id >= n) {
                fprintf(stderr, "Requested GPUs greater than available GPUs\n");
                exit(10);
            }
        n = (int)g_selected.size();
    }
    int use;
    if (num_gpu == -1) use = n;
    else if (num_gpu <= n) use = num_gpu;
    else {
        fprintf(stderr, "Requested GPUs greater than available GPUs\n");
        exit(10);
    }
    fprintf(stderr, "Using %d GPU(s)\n", use);
    if (!g_dev.empty()) destroy_interface();  // re-initialisation: release the previous contexts first
    g_ndev = use;
    for (int g = 0; g < use; g++) {
        const int ord = g_selected.empty() ? g : g_selected[g];
        check_set_device(ord, "InitializeInterface");
        int twin = 0;  // how many earlier engine devices sit on the same ordinal (sa_select_devices with a repeated id)
        for (int h = 0; h < g; h++) if (!g_selected.empty() && g_selected[h] == ord) twin++;
        DevCtx* dc = new DevCtx(arena_of(ord + 64 * twin, ord), arena_of(1024 + ord + 64 * twin, ord));
        dc->dev = ord;
        dc->index = g;
        hipStreamCreateWithFlags(&dc->admin, hipStreamNonBlocking);
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, ord);
        dc->total_mem = prop.totalGlobalMem;
        g_dev.push_back(dc);
    }
    return use;
}

void sa_initialize_processor(int transition, uint32_t wga_chunk, uint32_t seed_size, const int* sub_mat, int xdrop,
                             int hspthresh, int noentropy) {  // src/seed_filter.cu:830-897
    require_init("InitializeProcessor");
    resolve_options();
    if (xdrop >= (1 << 25) || xdrop <= -(1 << 25)) {
        fprintf(stderr, "Error: |xdrop| must be below 2^25\n");
        exit(1);
    }
    g_transition = transition ? 1 : 0;
    g_wga_chunk = wga_chunk;
    g_max_seeds = transition ? 13ll * wga_chunk : (int64_t)wga_chunk;  // :836-839
    if (!g_max_hits_overridden) g_max_hits = max_hits_for_mem(g_dev[0]->total_mem);  // :832-841 (device 0)
    g_seed_size = seed_size;
    memcpy(g_sub_mat, sub_mat, sizeof(g_sub_mat));
    g_xdrop = xdrop;
    g_hspthresh = hspthresh;
    g_noentropy = noentropy ? 1 : 0;
    {
        int mx = g_sub_mat[0];
        for (int i = 1; i < 64; i++) mx = std::max(mx, g_sub_mat[i]);
        g_fast_filter = (xdrop >= 0 && (int64_t)7 * std::max(mx, 0) <= (int64_t)xdrop) ? 1 : 0;
        // int16 score arithmetic: the best of a side (<= max(M) * long_cap rounded up to whole 64-base windows) and xdrop itself must stay well inside
        // the saturation range, or a walk could never satisfy the drop test and every hit would become a candidate
        // the 4-bit query copies carry PACK_PAD bytes = 2 * PACK_PAD bases of padding: a capped walk must end inside it
        if (g_long_cap > 2 * PACK_PAD) g_long_cap = 2 * PACK_PAD;
        g_packed_filter = (xdrop >= 0 && xdrop <= 16383 && (int64_t)std::max(mx, 0) * (((int64_t)g_long_cap + 63) / 64 * 64) <= 16383 &&
                           !opt_value("no_packed_filter")) ? 1 : 0;
        if (opt_value("no_fast_filter")) { g_fast_filter = 0; g_packed_filter = 0; }
    }
    report_hw_queues(opt_value("debug") != 0);
    std::lock_guard<std::mutex> lk(g_mu);
    g_tokens.clear();
    for (int g = 0; g < g_ndev; g++) {
        DevCtx* dc = g_dev[g];
        check_set_device(dc->dev, "InitializeProcessor");
        if (!dc->d_sub_mat) dc->d_sub_mat = (int*)dev_malloc(64 * sizeof(int), "sub_mat");
        check_memcpy(hipMemcpy(dc->d_sub_mat, g_sub_mat, 64 * sizeof(int), hipMemcpyHostToDevice), "sub_mat");
        // the work arena first (it is what the first calls need), then the table arena; both are mapped in the background while the
        // host reads its FASTA files (src/main.cpp:300-549) and both stay with the process
        const size_t work_per_slot = (size_t)opt_value("work_gb") << 30;
        if (work_per_slot && g_td && g_packed_filter) arena_request(dc->work_arena, work_per_slot * (size_t)SLOTS_PER_DEVICE);
        for (int k = 0; k < SLOTS_PER_DEVICE; k++) {
            // a slot that exists already (no ShutdownProcessor in between) keeps its region unless the layout changes (option work_gb,
            // or the slot count moved its offset): then its buffers still point into the OLD layout, where a neighbour's new region
            // would carve over them -- the slot is torn down and set up again before it is re-based
            Slot& sk = dc->slots[k];
            const size_t off_k = (size_t)k * work_per_slot;
            if (sk.stream && (sk.work.size != work_per_slot || sk.work.off != off_k || sk.work.arena != (work_per_slot ? &dc->work_arena : nullptr))) {
                hipDeviceSynchronize();
                slot_destroy(sk);
            }
            if (!sk.stream) {
                slot_init(sk, dc);
                sk.work.used = 0;
            }
            dc->slots[k].work.arena = work_per_slot ? &dc->work_arena : nullptr;
            dc->slots[k].work.off = off_k;
            dc->slots[k].work.size = work_per_slot;
            dc->slots[k].seeds.ensure((size_t)g_max_seeds, "seed_offsets");
        }
        if (!opt_value("arena_vmm") && arena_mapped(dc->arena) == 0) {  // (A/B switch: one plain hipMalloc per growth instead of mapped chunks)
            std::lock_guard<std::mutex> alk(dc->arena.mu);
            dc->arena.vmm = false;
        }
        // start mapping the table arena now: the host still has its FASTA files to read (src/main.cpp:300-549)
        if (g_arena_gb > 0 && g_td && g_ctx && g_packed_filter) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                const size_t reserve = ((size_t)8 << 30) + ((size_t)4 << 30) * (size_t)SLOTS_PER_DEVICE;
                const size_t have = arena_mapped(dc->arena);  // (no worker is running: ShutdownProcessor / process start)
                const size_t room = free_b + have > reserve ? free_b + have - reserve : 0;
                arena_request(dc->arena, std::min<size_t>((size_t)g_arena_gb << 30, room));
            }
        }
    }
    // LIFO pool like available_gpus (:895): slot-major so that concurrent callers spread over devices first
    for (int k = SLOTS_PER_DEVICE - 1; k >= 0; k--)
        for (int g = g_ndev - 1; g >= 0; g--) g_tokens.push_back({g, k});
    g_proc_init = true;
}

// everything the engine holds on one device except the context itself (what the reference's cudaDeviceReset() wipes, :939)
static void release_device_state(DevCtx* dc) {
    check_set_device(dc->dev, "ShutdownProcessor");
    hipDeviceSynchronize();
    for (int k = 0; k < MAX_SLOTS_PER_DEVICE; k++) if (dc->slots[k].stream) slot_destroy(dc->slots[k]);
    dc->ref.release("d_ref_seq");
    dc->ref8.release("d_ref_seq rows");
    dc->ref2.release("d_ref_seq 2-bit");
    dc->ref_rc.release("d_seq_rc");
    dc->ref4.release("d_seq 4-bit");
    dc->ref4_rc.release("d_seq_rc 4-bit");
    dc->refq2.release("d_seq 2-bit shifted");
    dc->refq2_rc.release("d_seq_rc 2-bit shifted");
    dev_free(dc->d_present, "code presence");
    dc->d_present = nullptr;
    dc->ref_host_ptr = nullptr;
    nbr_release(dc);
    // (the table arena stays mapped: it is a process-wide cache of cleared device pages that cost seconds to get; its background
    //  worker is stopped here.  Option arena_gb = 0 gives the pages back now, sa_release_arena() whenever the host wants them)
    if (g_arena_gb == 0) { arena_destroy(dc->arena); arena_destroy(dc->work_arena); }
    dc->keep_bucket.release("d_index_table");
    dc->keep_pos.release("d_pos_table");
    dc->keep_nbr_start.release("nbr_start");
    dc->bucket_start = dc->pos_table = nullptr;
    dc->num_index = 0;
    for (int b = 0; b < SA_BUFFER_DEPTH; b++) {
        dc->query[b].release("d_query_seq");
        dc->query_rc[b].release("d_query_rc_seq");
        dc->query4[b].release("d_query_seq 4-bit");
        dc->query4_rc[b].release("d_query_rc_seq 4-bit");
        dc->query2[b].release("d_query_seq 2-bit");
        dc->query2_rc[b].release("d_query_rc_seq 2-bit");
    }
    dc->up_tmp.release("upload staging");
    for (int k = 0; k < 2; k++) {
        if (dc->up_pinned[k]) hipHostFree(dc->up_pinned[k]);
        if (dc->up_ev[k]) hipEventDestroy(dc->up_ev[k]);
        dc->up_pinned[k] = nullptr;
        dc->up_ev[k] = nullptr;
    }
    dev_free(dc->d_sub_mat, "sub_mat");
    dc->d_sub_mat = nullptr;
}

// g_ShutdownProcessor (src/seed_filter.cu:932-940): the reference clears its device vectors and resets the device, which
// also drops the target and the tables.  Same here; the INTERFACE (device list, contexts) stays, so a host may run
// InitializeProcessor / SendRefWriteRequest again without a second InitializeInterface.
void sa_shutdown_processor(void) {
    arena_stop_all();
    for (auto* dc : g_dev) release_device_state(dc);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_tokens.clear();
    }
    g_proc_init = false;
    for (uint32_t b = 0; b < SA_BUFFER_DEPTH; b++) g_query_len[b] = 0;
}

// The reference's cudaDeviceReset() (src/seed_filter.cu:939) frees everything; the engine keeps the table arena as a cache (see
// release_device_state).  A host that wants the memory back -- before handing the GPU to LASTZ's gapped stage, say -- calls this
// after ShutdownProcessor.
void sa_release_arena(void) {
    if (g_proc_init) {
        fprintf(stderr, "Error: ReleaseArena while the processor is initialised (call ShutdownProcessor first)\n");
        exit(1);
    }
    arena_release_all();
}

static void destroy_interface() {  // re-initialisation of the interface: contexts go too
    sa_shutdown_processor();
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "InitializeInterface");
        if (dc->admin) hipStreamDestroy(dc->admin);
        delete dc;
    }
    g_dev.clear();
    g_ndev = 0;
}

// ---- target ---------------------------------------------------------------------------------------------------------
void sa_send_ref_write_request(const char* seq, size_t addr, uint32_t len) {  // seed_filter_interface.cu:82-101
    require_init("SendRefWriteRequest");
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendRefWriteRequest");
        const uint8_t* tmp = upload_ascii(dc, seq + addr, len, "ref_seq");
        dc->ref.create(len, "ref_seq", dc->admin);
        launch_encode(tmp, dc->ref.codes, len, dc->admin);
        dc->ref8.create(len, "ref_seq rows", dc->admin, true);
        launch_row_code(dc->ref.codes, dc->ref8.codes, len, dc->admin);
        dc->ref2.create(dc->ref.codes, len, 2, "ref_seq 2-bit", dc->admin);
        presence_of(dc, dc->ref.codes, len, 0, &dc->ref_present);
        check_launch("compress_string");
        check_sync(dc->admin, "SendRefWriteRequest");
        dc->ref_host_ptr = seq + addr;
        nbr_release(dc);  // a neighbourhood table built for another block must not survive (its context records are target bases)
    }
}

void sa_clear_ref(void) {  // seed_filter_interface.cu:103-113
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "ClearRef");
        dc->ref.release("d_ref_seq");
        dc->ref8.release("d_ref_seq rows");
        dc->ref2.release("d_ref_seq 2-bit");
        dc->ref_host_ptr = nullptr;
        nbr_release(dc);
        // The tables are forgotten; their MEMORY stays for the next block (keep_bucket, keep_pos, keep_nbr_start: a fresh allocation pays
        // first-touch page clearing inside the next GenerateSeedPosTable, 0.37 -> 0.11 s per 500 Mbp block).  That departs from the
        // reference, whose clearRef frees both tables: option clear_ref_frees = 1 restores it; ShutdownProcessor frees them either way.
        dc->bucket_start = dc->pos_table = nullptr;
        dc->num_index = 0;
        if (opt_value("clear_ref_frees")) {
            dc->keep_bucket.release("d_index_table");
            dc->keep_pos.release("d_pos_table");
            dc->keep_nbr_start.release("nbr_start");
        }
    }
}

int sa_generate_shape_pos(const char* shape) {  // ntcoding.cpp:21-37
    SeedShape sh;
    memset(&sh, 0, sizeof(sh));
    int n = 0, span = 0;
    for (int i = 0; shape[i] != '\0'; i++, span++) {
        if (shape[i] == '1' || shape[i] == 'T') {
            if (n >= MAX_CARE) {
                fprintf(stderr, "Error: seed weight above %d is not supported\n", MAX_CARE - 1);
                exit(1);
            }
            sh.pos[n] = (uint8_t)i;
            if (shape[i] == 'T') sh.transition_mask |= 1u << n;
            n++;
        }
    }
    if (span > 32) {
        fprintf(stderr, "Error: seed span above 32 is not supported\n");
        exit(1);
    }
    sh.weight = n;
    sh.span = span;
    g_shape = sh;
    return n;
}

void sa_generate_seed_pos_table(const char* ref_str, size_t start_addr, uint32_t ref_length, uint32_t step, int shape_size,
                                int kmer_size) {  // seed_pos_table.cu:49-109
    require_init("GenerateSeedPosTable");
    if (!(kmer_size <= 15 && kmer_size > 3)) {  // asserts at :51-52
        fprintf(stderr, "Error: GenerateSeedPosTable requires 3 < kmer_size <= 15\n");
        exit(1);
    }
    if (step == 0) step = 1;
    const uint32_t offset = (uint32_t)(shape_size + 1) % step;                       // :58
    const uint32_t start_offset = step - offset;                                     // :59
    const uint32_t nkeys = (uint32_t)1 << (2 * kmer_size);                           // :61
    const uint32_t num_steps = ref_length >= (uint32_t)shape_size ? (ref_length - (uint32_t)shape_size + offset) / step : 0;  // :64
    SeedShape sh = g_shape;
    sh.span = shape_size;
    // every device builds its own copy of the tables (the reference builds once on the host and replicates, seed_pos_table.cu:
    // 33-47); the builds are independent, so with several devices they run CONCURRENTLY, one host thread per device
    auto build_on = [&](DevCtx* dc) {
        check_set_device(dc->dev, "GenerateSeedPosTable");
        hipStream_t st = dc->admin;
        const uint8_t* codes = dc->ref.codes;
        SeqBuf tmp_codes;
        if (!(dc->ref.codes && dc->ref_host_ptr == ref_str + start_addr && dc->ref.len == ref_length)) {
            // not the resident block: encode a private copy
            const uint8_t* tmp = upload_ascii(dc, ref_str + start_addr, ref_length, "table seq");
            tmp_codes.create(ref_length, "table codes", st);
            launch_encode(tmp, tmp_codes.codes, ref_length, st);
            check_sync(st, "table encode");
            codes = tmp_codes.codes;
        }
        nbr_release(dc);
        dc->pos_table = nullptr;
        dc->keep_bucket.ensure((size_t)nkeys + 1, "index_table");
        dc->bucket_start = dc->keep_bucket.p;
        uint32_t num_index = 0;
        bool built = false;
        // Scratch of the build.  A 500 Mbp block needs ~10 GB of it (keys, four pair arrays); from hipMalloc every block paid
        // first-touch page clearing for it inside GenerateSeedPosTable (25-60 ms per GiB: 0.25 s of a 0.37 s build).  The table
        // arena is mapped (cleared) memory that holds nothing at this point -- the neighbourhood table it is there for is filled
        // after the build has been synchronised -- so the scratch is carved from its base; hipMalloc only without a VMM arena.
        uint8_t* scratch = nullptr;
        size_t scratch_off = 0, scratch_cap = 0;
        auto carve = [&](size_t bytes, const char* tag) -> void* {
            bytes = (bytes + 255) & ~(size_t)255;
            if (scratch && scratch_off + bytes <= scratch_cap) { void* p = scratch + scratch_off; scratch_off += bytes; return p; }
            return dev_malloc(bytes, tag);
        };
        auto uncarve = [&](void* p, const char* tag) {
            if (p && !(scratch && (uint8_t*)p >= scratch && (uint8_t*)p < scratch + scratch_cap)) dev_free(p, tag);
        };
        if (table_partition_build_supported(kmer_size) && !g_table_atomic) {
            // PARTITION build (table.hip): keys + coarse histogram -> offsets of the 4096 coarse partitions -> two LDS-staged
            // partition passes -> one workgroup per partition finishes its slice of bucket_start and pos_table in LDS
            const size_t pw = table_partition_part_start_words();
            const size_t fw = table_partition_fine_words(kmer_size);  // (keys above 24 bits -- 14of22 -- take a third partition level)
            {
                const size_t want = (size_t)std::max<uint32_t>(num_steps, 1) * sizeof(uint32_t) * 5 + 3 * pw * sizeof(uint32_t) + 4096 + fw * sizeof(uint32_t) +
                                    scan_temp_bytes(std::max<size_t>(pw, (size_t)1 << 18)) + ((size_t)1 << 20);
                if (g_table_scratch_arena && g_arena_gb != 0 && dc->arena.vmm && arena_wait(dc->arena, want)) {
                    scratch = dc->arena.base;
                    scratch_cap = want;
                }
            }
            uint32_t* keys = (uint32_t*)carve((size_t)std::max<uint32_t>(num_steps, 1) * sizeof(uint32_t), "kmer keys");
            uint32_t* coarse = (uint32_t*)carve(3 * pw * sizeof(uint32_t) + 4096, "coarse histogram");  // hist | part_start | cursor | flags
            uint32_t* part_start = coarse + pw;
            uint32_t* cursor = part_start + pw;
            uint8_t* part_unsorted = reinterpret_cast<uint8_t*>(cursor + pw);
            uint32_t* fine = fw ? (uint32_t*)carve(fw * sizeof(uint32_t), "fine partitions") : nullptr;
            void* scan_tmp = carve(scan_temp_bytes(std::max<size_t>(pw, (size_t)1 << 18)), "scan temp");
            check_memcpy(hipMemsetAsync(coarse, 0, pw * sizeof(uint32_t), st), "coarse histogram");
            launch_table_keys(codes, num_steps, start_offset, step, sh, keys, coarse, st);
            launch_exclusive_scan_u32(coarse, part_start, pw - 1, scan_tmp, st);
            check_launch("table keys/scan");
            check_memcpy(hipMemcpyAsync(&num_index, part_start + (pw - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st), "num_index");
            check_sync(st, "table keys");
            const size_t np = std::max<uint32_t>(num_index, 1);
            dc->keep_pos.ensure(np + np / 32, "pos_table");  // (a little headroom: the next block of the same size fits without a new allocation)
            dc->pos_table = dc->keep_pos.p;
            uint32_t* pairs = (uint32_t*)carve(4 * np * sizeof(uint32_t), "partition pairs");  // key_a | pos_a | key_b | pos_b
            uint32_t* d_err = nullptr;
            uint32_t err = 0;
            launch_table_partition_build(keys, num_steps, start_offset, step, kmer_size, part_start, num_index, cursor, pairs, pairs + np,
                                         pairs + 2 * np, pairs + 3 * np, part_unsorted, dc->bucket_start, dc->pos_table, fine, scan_tmp, &d_err, st);
            check_launch("table partition");
            if (d_err) check_memcpy(hipMemcpyAsync(&err, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "partition flag");
            check_sync(st, "table partition");
            uncarve(pairs, "partition pairs");
            uncarve(keys, "kmer keys");
            uncarve(coarse, "coarse histogram");
            uncarve(fine, "fine partitions");
            uncarve(scan_tmp, "scan temp");
            built = err == 0;  // (a tile of the third level spanned too many coarse groups -- a tiny or wildly skewed table: atomic build)
            if (!built) {
                dc->pos_table = nullptr;
                if (opt_value("debug")) fprintf(stderr, "seed table: the partition build gave up on this table, atomic build instead\n");
            }
        }
        if (!built) {
            // ATOMIC build: histogram + scatter with one global atomic per position (any seed weight)
            uint32_t* hist = (uint32_t*)dev_malloc(((size_t)nkeys + 1) * sizeof(uint32_t), "kmer histogram");
            void* scan_tmp = dev_malloc(scan_temp_bytes(nkeys), "scan temp");
            check_memcpy(hipMemsetAsync(hist, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), st), "histogram");
            launch_table_count(codes, num_steps, start_offset, step, sh, hist, st);
            launch_exclusive_scan_u32(hist, dc->bucket_start, nkeys, scan_tmp, st);
            check_launch("table count/scan");
            check_memcpy(hipMemcpyAsync(&num_index, dc->bucket_start + nkeys, sizeof(uint32_t), hipMemcpyDeviceToHost, st),
                         "num_index");
            check_sync(st, "table count");
            dc->keep_pos.ensure((size_t)std::max<uint32_t>(num_index, 1) + num_index / 32, "pos_table");
            dc->pos_table = dc->keep_pos.p;
            check_memcpy(hipMemsetAsync(hist, 0, ((size_t)nkeys + 1) * sizeof(uint32_t), st), "cursor");
            launch_table_fill(codes, num_steps, start_offset, step, sh, dc->bucket_start, hist, dc->pos_table, st);
            launch_table_sort_buckets(dc->bucket_start, nkeys, dc->pos_table, st);
            check_launch("table fill/sort");
            check_sync(st, "table fill");
            dev_free(hist, "kmer histogram");
            dev_free(scan_tmp, "scan temp");
        }
        tmp_codes.release("table codes");
        dc->num_index = num_index;
        dc->nkeys = nkeys;
        // the neighbourhood table belongs to the table build when the processor parameters are already known (the reference
        // calls InitializeProcessor first, src/main.cpp:298 before :621); otherwise the first table-direct call builds it
        if (g_proc_init && g_packed_filter) ensure_nbr(dc);
    };
    if (g_dev.size() <= 1) {
        for (auto* dc : g_dev) build_on(dc);
    } else {
        std::vector<std::thread> builders;
        for (auto* dc : g_dev) builders.emplace_back(build_on, dc);
        for (auto& t : builders) t.join();
    }
}

// ---- query ----------------------------------------------------------------------------------------------------------
void sa_send_query_write_request(const char* query_buffer, size_t addr, uint32_t len, uint32_t buffer) {  // :899-919
    require_init("SendQueryWriteRequest");
    if (buffer >= SA_BUFFER_DEPTH) {
        fprintf(stderr, "Error: query buffer %u out of range\n", buffer);
        exit(1);
    }
    g_query_len[buffer] = len;
    for (auto* dc : g_dev) {
        check_set_device(dc->dev, "SendQueryWriteRequest");
        hipStream_t st = dc->admin;
        const uint8_t* tmp = upload_ascii(dc, query_buffer + addr, len, "query_seq");
        dc->query[buffer].create(len, "query_seq", st);
        dc->query_rc[buffer].create(len, "query_rc_seq", st);
        launch_encode_rev_comp(tmp, dc->query[buffer].codes, dc->query_rc[buffer].codes, len, st);
        dc->query4[buffer].create(dc->query[buffer].codes, len, 4, "query_seq 4-bit", st);
        dc->query4_rc[buffer].create(dc->query_rc[buffer].codes, len, 4, "query_rc_seq 4-bit", st);
        dc->query2[buffer].create_q2(dc->query[buffer].codes, len, "query_seq 2-bit", st);
        dc->query2_rc[buffer].create_q2(dc->query_rc[buffer].codes, len, "query_rc_seq 2-bit", st);
        presence_of(dc, dc->query[buffer].codes, len, 1 + (int)buffer, &dc->query_present[buffer]);
        check_launch("compress_string_rev_comp");
        check_sync(st, "SendQueryWriteRequest");
    }
}

void sa_clear_query(uint32_t buffer) {  // :921-930
    if (buffer >= SA_BUFFER_DEPTH) return;
    for (auto* dc : g_dev) {
        // the reference frees here (:921-930); the engine keeps the allocations for the next block of this buffer: a
        // hipFree would synchronise the device under the calls that are running on the OTHER query buffer
        dc->query[buffer].clear();
        dc->query_rc[buffer].clear();
        dc->query4[buffer].clear();
        dc->query4_rc[buffer].clear();
        dc->query2[buffer].clear();
        dc->query2_rc[buffer].clear();
    }
}

}  // extern "C"
