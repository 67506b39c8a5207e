// gather_sector.hip -- is a random gather priced per 128-byte LINE or per 64-byte SECTOR on MI355X?
// mode 0: one 16-byte load per lane in a random line                                  (baseline: N lines)
// mode 1: two 16-byte loads per lane, BOTH halves (bytes 0.. and 64..) of ONE random line  (N lines, 2N sectors)
// mode 2: two 16-byte loads per lane in TWO random lines                                (2N lines, 2N sectors)
// mode 3: two 16-byte loads per lane in the SAME 64-byte half of one random line      (N lines, N sectors)
// If t(1) ~ t(0) and t(3) ~ t(0): the memory system moves whole lines (the second half is free).
// If t(1) ~ t(2): it moves sectors.  Run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum` for the request counts.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_sector gather_sector.hip ; usage: gather_sector [MB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint32_t lines, uint32_t iters, uint32_t* out) {
    uint32_t acc = 0;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t it = 0; it < iters; it++) {
        s = hash32(s + 1);
        const uint32_t l0 = (uint32_t)(((uint64_t)s * lines) >> 32);
        s = hash32(s + 2);
        const uint32_t l1 = (uint32_t)(((uint64_t)s * lines) >> 32);
        const uint32_t sub = s & 3;  // 16-byte slot inside a 64-byte half
        uint4 a = buf[(size_t)l0 * 8 + sub], b = {0, 0, 0, 0};
        if (MODE == 1) b = buf[(size_t)l0 * 8 + 4 + sub];
        if (MODE == 2) b = buf[(size_t)l1 * 8 + sub];
        if (MODE == 3) b = buf[(size_t)l0 * 8 + ((sub + 1) & 3)];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const uint4* buf, uint32_t lines, uint32_t* out, const char* what) {
    const int blocks = 256 * 4;  // 4 waves per SIMD
    const uint32_t iters = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lines, 4u, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters;
    printf("mode %d (%s): %.3f ms, %.1f G lane-iterations/s\n", MODE, what, ms, n / ms / 1e6);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 2048;
    const size_t bytes = mb << 20;
    uint4* buf;
    uint32_t* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    const uint32_t lines = (uint32_t)(bytes / 128);
    printf("buffer %zu MB\n", mb);
    run<0>(buf, lines, out, "1 load, 1 line");
    run<1>(buf, lines, out, "2 loads, both halves of 1 line");
    run<2>(buf, lines, out, "2 loads, 2 lines");
    run<3>(buf, lines, out, "2 loads, same half of 1 line");
    return 0;
}
