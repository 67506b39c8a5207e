// alloc_cost.hip -- what does device memory cost to GET on this box?  hipMalloc / hipFree by size, the same through the virtual
// memory API in 1 GiB chunks (hipMemAddressReserve + hipMemCreate + hipMemMap + hipMemSetAccess), and hipMallocAsync from a pool.
// Decides how the neighbourhood-table arena is obtained (engine.hip).  Build: hipcc --offload-arch=gfx950 -O2 alloc_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main() {
    hipSetDevice(0);
    hipFree(0);
    for (size_t gb : {1, 4, 16, 64, 160}) {
        void* p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, gb << 30);
        double t1 = now();
        if (e != hipSuccess) { printf("hipMalloc %zu GiB failed: %s\n", gb, hipGetErrorString(e)); continue; }
        hipLaunchKernelGGL(touch, dim3((unsigned)(((gb << 30) / 4096 + 255) / 256)), dim3(256), 0, 0, (char*)p, gb << 30);
        hipDeviceSynchronize();
        double t2 = now();
        hipFree(p);
        double t3 = now();
        printf("hipMalloc %3zu GiB: %.1f ms (%.2f ms/GiB)  first touch %.1f ms  hipFree %.1f ms\n", gb, t1 - t0, (t1 - t0) / gb, t2 - t1, t3 - t2);
    }
    // second round: does the runtime cache anything?
    for (int r = 0; r < 2; r++) {
        void* p = nullptr;
        double t0 = now();
        hipMalloc(&p, (size_t)40 << 30);
        double t1 = now();
        hipFree(p);
        printf("hipMalloc 40 GiB again: %.1f ms, free %.1f ms\n", t1 - t0, now() - t1);
    }
    // virtual memory API
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("VMM granularity: %zu (%s)\n", gran, hipGetErrorString(e));
        const size_t total = (size_t)64 << 30, chunk = (size_t)1 << 30;
        void* va = nullptr;
        double t0 = now();
        e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
        printf("reserve 64 GiB VA: %.2f ms (%s)\n", now() - t0, hipGetErrorString(e));
        if (e == hipSuccess) {
            std::vector<hipMemGenericAllocationHandle_t> hs;
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            double tc = 0, tm = 0, ta = 0;
            for (size_t off = 0; off < total; off += chunk) {
                hipMemGenericAllocationHandle_t h;
                double a = now();
                if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", off >> 30); break; }
                double b = now();
                if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) { printf("map failed\n"); break; }
                double c = now();
                if (hipMemSetAccess((char*)va + off, chunk, &acc, 1) != hipSuccess) { printf("access failed\n"); break; }
                double d = now();
                tc += b - a; tm += c - b; ta += d - c;
                hs.push_back(h);
            }
            printf("VMM 64 x 1 GiB: create %.1f ms, map %.1f ms, set access %.1f ms  (%.2f ms/GiB total)\n", tc, tm, ta, (tc + tm + ta) / 64);
            double t1 = now();
            hipLaunchKernelGGL(touch, dim3((unsigned)((total / 4096 + 255) / 256)), dim3(256), 0, 0, (char*)va, total);
            hipError_t es = hipDeviceSynchronize();
            printf("first touch of the mapped range: %.1f ms (%s)\n", now() - t1, hipGetErrorString(es));
            double t2 = now();
            for (size_t i = 0; i < hs.size(); i++) { hipMemUnmap((char*)va + i * chunk, chunk); hipMemRelease(hs[i]); }
            hipMemAddressFree(va, total);
            printf("unmap + release: %.1f ms\n", now() - t2);
        }
    }
    // stream-ordered pool
    {
        hipStream_t s;
        hipStreamCreate(&s);
        hipMemPool_t pool;
        hipDeviceGetDefaultMemPool(&pool, 0);
        uint64_t thr = UINT64_MAX;
        hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
        for (int r = 0; r < 3; r++) {
            void* p = nullptr;
            double t0 = now();
            hipError_t e = hipMallocAsync(&p, (size_t)40 << 30, s);
            hipStreamSynchronize(s);
            double t1 = now();
            if (e == hipSuccess) hipFreeAsync(p, s);
            hipStreamSynchronize(s);
            printf("hipMallocAsync 40 GiB round %d: %.1f ms (%s), free %.1f ms\n", r, t1 - t0, hipGetErrorString(e), now() - t1);
        }
    }
    return 0;
}
