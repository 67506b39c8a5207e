// arena.hip -- the table arena: device memory of the neighbourhood table, obtained through the virtual-memory API in 1 GiB
// chunks that a background thread maps behind each other into one reserved address range (struct Arena, engine_internal.h).
#include "engine_internal.h"

namespace sa {

static void arena_worker(Arena* A) {
    hipSetDevice(A->dev);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = A->dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (opt_value("debug")) {
        size_t gmin = 0, grec = 0;
        hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
        hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
        fprintf(stderr, "table arena: allocation granularity minimum %zu, recommended %zu bytes; chunks of %zu bytes\n", gmin, grec, (size_t)ARENA_CHUNK);
    }
    std::unique_lock<std::mutex> lk(A->mu);
    while (!A->stop && !A->failed && A->mapped < A->goal && A->mapped + ARENA_CHUNK <= A->va_bytes) {
        uint8_t* at = A->base + A->mapped;
        lk.unlock();
        hipMemGenericAllocationHandle_t h;
        bool ok = hipMemCreate(&h, ARENA_CHUNK, &prop, 0) == hipSuccess;
        if (ok && hipMemMap(at, ARENA_CHUNK, 0, h, 0) != hipSuccess) { hipMemRelease(h); ok = false; }
        if (ok && hipMemSetAccess(at, ARENA_CHUNK, &acc, 1) != hipSuccess) { hipMemUnmap(at, ARENA_CHUNK); hipMemRelease(h); ok = false; }
        lk.lock();
        if (ok) {
            A->chunks.push_back(h);
            A->mapped += ARENA_CHUNK;
        } else {
            (void)hipGetLastError();
            A->failed = true;
        }
        A->cv.notify_all();
    }
    A->busy = false;
    A->cv.notify_all();
}

// ask for `bytes` usable bytes (asynchronously); never shrinks
void arena_request(Arena& A, size_t bytes) {
    std::unique_lock<std::mutex> lk(A.mu);
    if (!A.vmm) return;
    if (!A.base) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)288 << 30;
        A.va_bytes = (total_b + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
        void* va = nullptr;
        if (hipMemAddressReserve(&va, A.va_bytes, 0, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            A.vmm = false;
            A.va_bytes = 0;
            return;
        }
        A.base = (uint8_t*)va;
    }
    const size_t want = std::min(A.va_bytes, (bytes + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK);
    if (want > A.goal) {
        A.goal = want;
        A.failed = false;
    } else if (A.failed && want > A.mapped) {
        A.failed = false;  // (memory may have been given back since the last attempt)
    }
    if (!A.busy && !A.failed && A.mapped < A.goal) {
        if (A.worker.joinable()) { lk.unlock(); A.worker.join(); lk.lock(); }
        A.busy = true;
        A.stop = false;
        A.worker = std::thread(arena_worker, &A);
    }
}
// block until `bytes` are usable; false: they cannot be had (out of memory)
bool arena_wait(Arena& A, size_t bytes) {
    arena_request(A, bytes);
    std::unique_lock<std::mutex> lk(A.mu);
    if (!A.vmm) {  // fallback: a plain allocation of exactly what is needed, kept while it is large enough
        if (A.mapped >= bytes) return true;
        if (A.base) { hipFree(A.base); A.base = nullptr; A.mapped = 0; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        A.base = (uint8_t*)p;
        A.mapped = bytes;
        return true;
    }
    A.cv.wait(lk, [&] { return A.mapped >= bytes || A.failed || !A.busy; });
    return A.mapped >= bytes;
}
// The block's need is known and mapped: stop mapping ahead (the default goal is a guess made before any sequence was seen; what is
// mapped stays).  Mapping is page clearing on the device: left running it takes memory bandwidth from the first query pass.
void arena_settle(Arena& A, size_t need) {
    std::lock_guard<std::mutex> lk(A.mu);
    const size_t n = (need + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK;
    if (A.goal > std::max(n, A.mapped)) A.goal = std::max(n, A.mapped);
}
// give everything beyond `keep` bytes back to the device (the worker is stopped first)
void arena_trim(Arena& A, size_t keep) {
    std::unique_lock<std::mutex> lk(A.mu);
    A.stop = true;
    A.goal = std::min(A.goal, (keep + ARENA_CHUNK - 1) / ARENA_CHUNK * ARENA_CHUNK);
    if (A.worker.joinable()) { lk.unlock(); A.worker.join(); lk.lock(); }
    A.stop = false;
    if (!A.vmm) {
        if (keep == 0 && A.base) { hipFree(A.base); A.base = nullptr; A.mapped = 0; }
        return;
    }
    while (A.mapped >= ARENA_CHUNK && A.mapped - ARENA_CHUNK >= keep) {
        A.mapped -= ARENA_CHUNK;
        hipMemUnmap(A.base + A.mapped, ARENA_CHUNK);
        hipMemRelease(A.chunks.back());
        A.chunks.pop_back();
    }
    // On this runtime (ROCm 7.2) the pages of an unmapped + released chunk only go back to the device when the ADDRESS RANGE is
    // freed (tools/micro/vmm_info2.hip: every teardown order leaves hipMemGetInfo and the number of creatable chunks unchanged
    // until hipMemAddressFree).  So giving everything back means giving the range back too; the next request reserves a new one.
    if (A.mapped == 0 && A.base) {
        hipMemAddressFree(A.base, A.va_bytes);
        A.base = nullptr;
        A.va_bytes = 0;
        A.goal = 0;
    }
    A.failed = false;
}
void arena_destroy(Arena& A) {
    arena_trim(A, 0);
    std::lock_guard<std::mutex> lk(A.mu);
    A.vmm = true;
}
size_t arena_mapped(Arena& A) {
    std::lock_guard<std::mutex> lk(A.mu);
    return A.mapped;
}
// One arena per engine device for the life of the process: cleared device pages cost seconds to get, so they are kept across
// ShutdownProcessor / InitializeInterface cycles (option arena_gb = 0 gives them back at ShutdownProcessor, sa_release_arena() at any
// time; the background workers are stopped at ShutdownProcessor and, through an atexit hook, before the HIP runtime unloads).
// key: the device ordinal, + 64 per earlier engine device on the same ordinal (sa_select_devices with a repeated id).
static std::mutex g_arenas_mu;
static std::vector<Arena*> g_arenas;  // (never destroyed: a worker thread may outlive static destruction)

void* WorkRegion::take(size_t bytes) {
    if (!arena || !size) return nullptr;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (used + need > size) return nullptr;
    std::lock_guard<std::mutex> lk(arena->mu);
    if (!arena->vmm || !arena->base || arena->mapped < off + used + need) return nullptr;  // (not there yet: the caller allocates)
    void* p = arena->base + off + used;
    used += need;
    return p;
}

static void arena_stop_worker(Arena& A) {  // the worker finishes the chunk it is on and goes away; what is mapped stays
    std::unique_lock<std::mutex> lk(A.mu);
    A.stop = true;
    A.goal = std::min(A.goal, A.mapped);
    if (A.worker.joinable()) { lk.unlock(); A.worker.join(); lk.lock(); }
    A.stop = false;
}
void arena_stop_all() {
    std::vector<Arena*> v;
    { std::lock_guard<std::mutex> lk(g_arenas_mu); v = g_arenas; }
    for (Arena* a : v) if (a) arena_stop_worker(*a);
}
void arena_release_all() {
    std::vector<Arena*> v;
    { std::lock_guard<std::mutex> lk(g_arenas_mu); v = g_arenas; }
    for (Arena* a : v) if (a) { hipSetDevice(a->dev); arena_destroy(*a); }
}
Arena& arena_of(int key, int ordinal) {
    std::lock_guard<std::mutex> lk(g_arenas_mu);
    static bool hooked = false;
    if (!hooked) {  // registered after the HIP runtime's own exit handlers, so it runs before them: no thread of ours is inside
        hooked = true;  // hipMemCreate / hipMemMap while the runtime goes away
        std::atexit(arena_stop_all);
    }
    if ((int)g_arenas.size() <= key) g_arenas.resize((size_t)key + 1, nullptr);
    if (!g_arenas[key]) {
        g_arenas[key] = new Arena();
        g_arenas[key]->dev = ordinal;
    }
    return *g_arenas[key];
}

}  // namespace sa
