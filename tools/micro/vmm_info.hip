// Does hipMemGetInfo see memory held through the VMM API (hipMemCreate / hipMemMap)?  The table arena's bookkeeping depends on it.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/vmm_info.hip -o /tmp/vmm_info && /tmp/vmm_info
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double free_gb() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f / 1e9; }
int main() {
    CK(hipSetDevice(0));
    printf("start                      free %.2f GB\n", free_gb());
    const size_t chunk = (size_t)1 << 30, n = 8;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void* base = nullptr;
    CK(hipMemAddressReserve(&base, chunk * n, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < n; i++) {
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char*)base + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
    }
    printf("8 GiB created + mapped     free %.2f GB\n", free_gb());
    CK(hipMemset(base, 1, chunk * n));
    CK(hipDeviceSynchronize());
    printf("... and touched            free %.2f GB\n", free_gb());
    for (size_t i = 0; i < n; i++) {
        CK(hipMemUnmap((char*)base + i * chunk, chunk));
        CK(hipMemRelease(h[i]));
    }
    CK(hipDeviceSynchronize());
    printf("unmapped + released        free %.2f GB\n", free_gb());
    // is the released memory really back, whatever hipMemGetInfo says?  Ask for more than the reported free amount.
    {
        size_t f = 0, t = 0;
        (void)hipMemGetInfo(&f, &t);
        void* big = nullptr;
        hipError_t e = hipMalloc(&big, f + 6 * chunk);
        printf("hipMalloc(reported free + 6 GiB) -> %s\n", hipGetErrorString(e));
        if (e == hipSuccess) {
            (void)hipMemset(big, 0, f + 6 * chunk);
            printf("   memset of it -> %s\n", hipGetErrorString(hipDeviceSynchronize()));
            (void)hipFree(big);
        } else (void)hipGetLastError();
        printf("after that                 free %.2f GB\n", free_gb());
    }
    {   // how many 1 GiB chunks can still be created?  (288 GB device: ~287 if the 8 released ones are really back, ~279 if not)
        std::vector<hipMemGenericAllocationHandle_t> all;
        for (;;) {
            hipMemGenericAllocationHandle_t x;
            if (hipMemCreate(&x, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            all.push_back(x);
        }
        printf("1 GiB chunks creatable now: %zu    free %.2f GB\n", all.size(), free_gb());
        for (auto x : all) (void)hipMemRelease(x);
        printf("all released               free %.2f GB\n", free_gb());
    }
    void* p = nullptr;
    CK(hipMalloc(&p, chunk * n));
    printf("hipMalloc 8 GiB            free %.2f GB\n", free_gb());
    CK(hipFree(p));
    printf("hipFree                    free %.2f GB\n", free_gb());
    CK(hipMemAddressFree(base, chunk * n));
    return 0;
}
