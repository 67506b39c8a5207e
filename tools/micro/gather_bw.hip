// gather_bw.hip -- how many random 128-byte lines per second can the memory system deliver?
// Each lane reads 16 bytes at a pseudo-random line of a `mb`-megabyte buffer; ILP independent loads in flight per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_bw gather_bw.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int ILP>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ buf, uint32_t lines, uint32_t iters, uint32_t* out) {
    uint32_t acc = 0;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            s = hash32(s + j + 1);
            const uint32_t line = (uint32_t)(((uint64_t)s * lines) >> 32);
            v[j] = buf[(size_t)line * 8 + (s & 7)];  // 16 bytes somewhere in the line
        }
#pragma unroll
        for (int j = 0; j < ILP; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int ILP>
static void run(const uint4* buf, uint32_t lines, int blocks, uint32_t iters, uint32_t* out, const char* tag) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(gather<ILP>, dim3(blocks), dim3(256), 0, 0, buf, lines, 4u, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather<ILP>, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * ILP;
    printf("%s ilp=%d blocks=%d: %.1f G lines/s  = %.2f TB/s of 128-B lines  (%.3f ms)\n", tag, ILP, blocks, n / ms / 1e6,
           n * 128 / ms / 1e9, ms);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 100;
    const size_t bytes = mb << 20;
    uint4* buf;
    uint32_t* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    const uint32_t lines = (uint32_t)(bytes / 128);
    for (int wps = 2; wps <= 8; wps *= 2) {  // waves per SIMD
        const int blocks = 256 * wps;        // 256 CUs x (wps*4 waves / 4 waves per block)
        char tag[64];
        snprintf(tag, sizeof tag, "%zu MB, %d waves/SIMD", mb, wps);
        run<1>(buf, lines, blocks, 256, out, tag);
        run<2>(buf, lines, blocks, 128, out, tag);
        run<4>(buf, lines, blocks, 64, out, tag);
    }
    return 0;
}
