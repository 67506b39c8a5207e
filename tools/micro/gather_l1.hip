// gather_l1.hip -- throughput of DIVERGENT 16-byte loads (every lane its own address) as a function of where the data
// lives: L1-resident (8 KB), L2-resident (1 MB), fabric (100 MB).  Prints lane-loads per ns per CU and per clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint4 load16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

template <int ILP, bool SAME_LINE>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ buf, uint32_t bytes, uint32_t iters, uint32_t* out) {
    uint32_t acc = 0;
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t it = 0; it < iters; it++) {
        uint4 v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            s = hash32(s + j + 1);
            uint32_t off = (uint32_t)(((uint64_t)s * (bytes - 64)) >> 32);
            if (SAME_LINE) off = (__builtin_amdgcn_readfirstlane(off) & ~127u) + (lane & 7) * 16;  // whole wave in one line
            v[j] = load16u(buf + off);
        }
#pragma unroll
        for (int j = 0; j < ILP; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int ILP, bool SAME>
static void run(const uint8_t* buf, uint32_t bytes, int blocks, uint32_t iters, uint32_t* out, const char* tag) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<ILP, SAME>), dim3(blocks), dim3(256), 0, 0, buf, bytes, 4u, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((gather<ILP, SAME>), dim3(blocks), dim3(256), 0, 0, buf, bytes, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * ILP;  // lane-loads
    printf("%-28s ilp=%d %s: %.2f lane-loads/ns/CU (%.3f per clk @2.4GHz), wave-load every %.1f clk/CU, %.2f TB/s useful\n", tag, ILP,
           SAME ? "same-line" : "divergent", n / (ms * 1e6) / 256, n / (ms * 1e6) / 256 / 2.4, 64.0 / (n / (ms * 1e6) / 256 / 2.4),
           n * 16 / ms / 1e9);
}

int main() {
    const size_t cap = 128u << 20;
    uint8_t* buf;
    uint32_t* out;
    (void)hipMalloc(&buf, cap);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 1, cap);
    const uint32_t sizes[] = {8u << 10, 1u << 20, 100u << 20};
    const char* names[] = {"8 KB (L1)", "1 MB (L2)", "100 MB (fabric)"};
    for (int k = 0; k < 3; k++)
        for (int wps = 4; wps <= 8; wps *= 2) {
            char tag[64];
            snprintf(tag, sizeof tag, "%s %d w/SIMD", names[k], wps);
            run<1, false>(buf, sizes[k], 256 * wps, 512, out, tag);
            run<2, false>(buf, sizes[k], 256 * wps, 256, out, tag);
            run<2, true>(buf, sizes[k], 256 * wps, 256, out, tag);
        }
    return 0;
}
