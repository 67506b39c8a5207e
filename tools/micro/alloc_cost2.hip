// alloc_cost2.hip -- one allocation strategy per PROCESS (argv[1] = malloc | chunks | vmm), 40 GiB each: the first allocation of a
// process pays for the physical pages; later ones in the same process are served from what the runtime kept.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "malloc";
    const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 40;
    hipSetDevice(0);
    hipFree(0);
    double t0 = now();
    if (!strcmp(mode, "malloc")) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, gb << 30);
        printf("malloc %zu GiB: %.1f ms (%s)\n", gb, now() - t0, hipGetErrorString(e));
    } else if (!strcmp(mode, "chunks")) {
        for (size_t i = 0; i < gb; i++) { void* p; hipMalloc(&p, (size_t)1 << 30); }
        printf("chunks %zu x 1 GiB: %.1f ms\n", gb, now() - t0);
    } else {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        void* va = nullptr;
        hipMemAddressReserve(&va, gb << 30, 0, nullptr, 0);
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        const size_t chunk = argc > 3 ? ((size_t)atoi(argv[3]) << 20) : ((size_t)1 << 30);
        double first = 0;
        for (size_t off = 0; off < (gb << 30); off += chunk) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("create failed\n"); break; }
            hipMemMap((char*)va + off, chunk, 0, h, 0);
            hipMemSetAccess((char*)va + off, chunk, &acc, 1);
            if (off == 0) first = now() - t0;
        }
        printf("vmm %zu GiB in %zu MiB chunks: %.1f ms (first chunk %.1f ms)\n", gb, chunk >> 20, now() - t0, first);
    }
    return 0;
}
