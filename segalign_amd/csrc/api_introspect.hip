// api_introspect.hip -- C-ABI, introspection: call statistics, filter / lookup mode, copies of device state (tests, bench).
#include "engine_internal.h"

#include <vector>

using namespace sa;

extern "C" {

// ---- introspection --------------------------------------------------------------------------------------------------
void sa_get_last_call_stats(sa_call_stats* o) { *o = t_stats; }
void sa_set_count_examined(int on) { g_count_examined = on != 0; }
int sa_get_filter_mode(void) {  // which X-drop filter kernel the next plain (non repeat-masker) call uses
    if (g_count_examined) return g_fast_filter ? 1 : 0;
    return g_packed_filter ? 3 : g_fast_filter;
}
int sa_get_lookup_mode(void) {  // how device-seeded calls look seeds up on device 0 right now (builds the table if needed)
    if (g_ndev <= 0 || !g_proc_init) return 0;
    DevCtx* dc = g_dev[0];
    check_set_device(dc->dev, "lookup mode");
    if (!(g_td && g_packed_filter && !g_count_examined && dc->ref2.base && ensure_nbr(dc))) return 0;
    return dc->nbr_ctx ? 2 : 1;
}
uint64_t sa_get_neighbourhood_entries(void) { return (g_ndev > 0 && g_dev[0]->nbr_state == 1) ? g_dev[0]->nbr_total : 0; }

uint32_t sa_get_ref_len(void) { return g_ndev ? g_dev[0]->ref.len : 0; }
uint32_t sa_get_num_index(void) { return g_ndev ? g_dev[0]->num_index : 0; }
uint32_t sa_get_index_table_size(void) { return g_ndev ? g_dev[0]->nkeys : 0; }
uint32_t sa_get_query_len(uint32_t buffer) { return buffer < SA_BUFFER_DEPTH ? g_query_len[buffer] : 0; }

static DevCtx* ctx_of(int dev) {
    if (dev < 0 || dev >= g_ndev) {
        fprintf(stderr, "Error: device %d out of range\n", dev);
        exit(11);
    }
    check_set_device(g_dev[dev]->dev, "copy");
    return g_dev[dev];
}
void sa_copy_ref_codes(int dev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->ref.codes, dc->ref.len, hipMemcpyDeviceToHost), "ref codes");
}
void sa_copy_index_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->bucket_start + 1, (size_t)dc->nkeys * sizeof(uint32_t), hipMemcpyDeviceToHost), "index table");
}
void sa_copy_pos_table(int dev, uint32_t* dst) {
    DevCtx* dc = ctx_of(dev);
    check_memcpy(hipMemcpy(dst, dc->pos_table, (size_t)dc->num_index * sizeof(uint32_t), hipMemcpyDeviceToHost), "pos table");
}
void sa_copy_query_codes(int dev, uint32_t buffer, int rev, uint8_t* dst) {
    DevCtx* dc = ctx_of(dev);
    SeqBuf& b = rev ? dc->query_rc[buffer] : dc->query[buffer];
    check_memcpy(hipMemcpy(dst, b.codes, b.len, hipMemcpyDeviceToHost), "query codes");
}

// The ordering stage alone (dedup.hip) on records the host hands in, as ONE dedup scope: what the reference does to the anchors of an
// iteration -- stable_sort(hspComp) -> unique_copy(hspEqual) -> stable_sort(hspCompLastz), src/seed_filter.cu:776-782; rm != 0: the
// repeat masker's chain, repeat_masker_src/seed_filter.cu:819-831.  path 0: the per-segment LDS chain (dedup_seg_kernel; plain
// chain only, at most its 2048 records), path 1: rocprim merge sorts + the adjacent-pair unique kernels (what oversized segments
// and the repeat masker take).  Test entry (tests/test_gpu_thrust_order.py holds both paths against rocThrust's stable_sort /
// unique_copy and against the CPU restatement): own buffers, device 0, default stream.  Returns the number of records in *out.
size_t sa_order_hsps(const sa_segment_pair* in, size_t n, int rm, int path, sa_segment_pair** out) {
    *out = nullptr;
    if (g_ndev <= 0) {
        fprintf(stderr, "Error: sa_order_hsps before InitializeInterface\n");
        exit(11);
    }
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFull || (path == 0 && (rm || n > 2048))) {
        fprintf(stderr, "Error: sa_order_hsps: %zu records do not fit path %d\n", n, path);
        exit(1);
    }
    ctx_of(0);
    std::vector<HspRec> h(n);
    for (size_t i = 0; i < n; i++) { h[i].ref_start = in[i].ref_start; h[i].query_start = in[i].query_start; h[i].len = in[i].len; h[i].score = in[i].score; h[i].seg = 0; }
    HspRec *a = nullptr, *b = nullptr;
    uint4* o16 = nullptr;
    uint32_t* misc = nullptr;  // [0]: counter | seg_info
    void *stmp = nullptr, *utmp = nullptr;
    const size_t sbytes = sort_temp_bytes(n), ubytes = unique_temp_bytes((uint32_t)n), iwords = dedup_seg_info_words() + 4;
    check_memcpy(hipMalloc((void**)&a, n * sizeof(HspRec)), "order: records");
    check_memcpy(hipMalloc((void**)&b, n * sizeof(HspRec)), "order: records");
    check_memcpy(hipMalloc((void**)&o16, n * sizeof(uint4)), "order: output");
    check_memcpy(hipMalloc((void**)&misc, iwords * sizeof(uint32_t)), "order: counters");
    check_memcpy(hipMalloc(&stmp, sbytes), "order: sort temp");
    check_memcpy(hipMalloc(&utmp, ubytes), "order: unique temp");
    check_memcpy(hipMemcpy(a, h.data(), n * sizeof(HspRec), hipMemcpyHostToDevice), "order: upload");
    check_memcpy(hipMemset(misc, 0, iwords * sizeof(uint32_t)), "order: counters");
    hipStream_t st = 0;
    size_t m = 0;
    auto count = [&]() { uint32_t c = 0; check_memcpy(hipMemcpy(&c, misc, sizeof(c), hipMemcpyDeviceToHost), "order: count"); return (size_t)c; };
    if (path == 0) {
        uint32_t* info = misc + 4;
        launch_dedup_seg(a, (uint32_t)n, nullptr, 1, o16, info, 0, 0, st);
        check_launch("order: dedup seg");
        std::vector<uint32_t> hi(dedup_seg_info_words());
        check_memcpy(hipMemcpy(hi.data(), info, hi.size() * sizeof(uint32_t), hipMemcpyDeviceToHost), "order: segment info");
        if (hi[hi.size() - 1] != 0) { fprintf(stderr, "Error: sa_order_hsps: the LDS chain refused the segment\n"); exit(15); }
        m = hi[0];
    } else {
        if (!rm) {
            launch_sort(a, b, n, ORDER_DIAG, stmp, sbytes, st);
            launch_unique(b, a, (uint32_t)n, 0, misc, utmp, st);
            m = count();
            launch_sort(a, b, m, ORDER_LASTZ, stmp, sbytes, st);
        } else {
            launch_sort(a, b, n, ORDER_RM_FIRST, stmp, sbytes, st);
            launch_unique(b, a, (uint32_t)n, 1, misc, utmp, st);
            const size_t m1 = count();
            launch_sort(a, b, m1, ORDER_RM_DIAG, stmp, sbytes, st);
            launch_unique(b, a, (uint32_t)m1, 0, misc, utmp, st);
            m = count();
            launch_sort(a, b, m, ORDER_RM_FINAL, stmp, sbytes, st);
        }
        launch_strip(b, (uint32_t)m, o16, nullptr, st);
        check_launch("order: sort chain");
    }
    sa_segment_pair* res = (sa_segment_pair*)malloc(std::max<size_t>(m, 1) * sizeof(sa_segment_pair));
    if (m) check_memcpy(hipMemcpy(res, o16, m * sizeof(sa_segment_pair), hipMemcpyDeviceToHost), "order: download");
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(o16); (void)hipFree(misc); (void)hipFree(stmp); (void)hipFree(utmp);
    *out = res;
    return m;
}

}  // extern "C"
