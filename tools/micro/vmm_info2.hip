// Which VMM teardown order really returns the memory on this runtime?  (vmm_info.hip: unmap + release after use does not.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double free_gb() { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); return f / 1e9; }
static int variant(int v) {
    const size_t chunk = (size_t)1 << 30, n = 8;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* base = nullptr;
    CK(hipMemAddressReserve(&base, chunk * n, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    const double f0 = free_gb();
    for (size_t i = 0; i < n; i++) {
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char*)base + i * chunk, chunk, 0, h[i], 0));
        if (v == 1) CK(hipMemRelease(h[i]));  // the mapping keeps the memory alive
        CK(hipMemSetAccess((char*)base + i * chunk, chunk, &acc, 1));
    }
    CK(hipMemset(base, 1, chunk * n));
    CK(hipDeviceSynchronize());
    const double f1 = free_gb();
    if (v == 3) {  // one unmap for the whole range
        CK(hipMemUnmap(base, chunk * n));
        for (size_t i = 0; i < n; i++) CK(hipMemRelease(h[i]));
    } else {
        for (size_t i = 0; i < n; i++) {
            if (v == 2) { CK(hipMemRelease(h[i])); CK(hipMemUnmap((char*)base + i * chunk, chunk)); }
            else { CK(hipMemUnmap((char*)base + i * chunk, chunk)); if (v == 0) CK(hipMemRelease(h[i])); }
        }
    }
    CK(hipDeviceSynchronize());
    const double f2 = free_gb();
    CK(hipMemAddressFree(base, chunk * n));
    const double f3 = free_gb();
    printf("variant %d: before %.2f, mapped %.2f, torn down %.2f, address range freed %.2f GB\n", v, f0, f1, f2, f3);
    return 0;
}
int main() {
    CK(hipSetDevice(0));
    for (int v = 0; v < 4; v++) if (variant(v)) return 1;
    return 0;
}
