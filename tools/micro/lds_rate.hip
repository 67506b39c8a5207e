// lds_rate.hip -- what does one ds_read_b32 wave-instruction cost a CU when the 64 lanes read RANDOM entries of a table (the class
// filter's lookups), against conflict-free and broadcast reads, B reads per s_waitcnt, W waves per SIMD, with and without VALU work
// on the returned values?  Cycles per DS wave-instruction per CU = kernel cycles x CUs / instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE /*0 random, 1 lane-linear (conflict free), 2 broadcast*/, int B /*reads in flight*/, int WORK /*VALU ops per read*/, int WIDTH /*4, 8, 16 bytes*/>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, uint32_t tab_bytes, uint32_t seed) {
    extern __shared__ uint32_t s_tab[];
    for (uint32_t i = threadIdx.x; i < tab_bytes / 4; i += blockDim.x) s_tab[i] = i * 2654435761u + seed;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t mask = (tab_bytes - 1u) & ~(uint32_t)(WIDTH - 1);
    uint32_t x = (threadIdx.x + blockIdx.x * blockDim.x) * 747796405u + seed;
    uint32_t acc = 0, P = 0;
    // addresses: B per lane, advanced by a per-lane random stride each round (add + and: cheap, so that the LDS and not the address
    // arithmetic sets the pace); lane-linear / broadcast: the same row walk for every lane of the wave
    uint32_t addr[B], inc[B];
#pragma unroll
    for (int j = 0; j < B; j++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = (x >> 7) & mask, r2 = ((x >> 3) | 1u) * (uint32_t)WIDTH;
        if (MODE == 0) { addr[j] = r; inc[j] = r2 & mask; }
        else if (MODE == 1) { addr[j] = (__builtin_amdgcn_readfirstlane((int)r) & ~(uint32_t)(64 * WIDTH - 1)) + lane * WIDTH; inc[j] = (uint32_t)(64 * WIDTH) * (1u + 2u * j); }
        else { addr[j] = __builtin_amdgcn_readfirstlane((int)r); inc[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r2 & mask)); }
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < B; j++) addr[j] = (addr[j] + inc[j]) & (MODE == 1 ? (tab_bytes - 1u) : mask);
        uint32_t v[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (WIDTH == 4) v[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(s_tab) + addr[j]);
            else if (WIDTH == 8) { const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_tab) + addr[j]); v[j] = t.x ^ t.y; }
            else { const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(s_tab) + addr[j]); v[j] = t.x ^ t.y ^ t.z ^ t.w; }
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (WORK == 0) acc ^= v[j];
            else {
#pragma unroll
                for (int w = 0; w < WORK; w++) {
                    asm volatile("v_pk_add_i16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(P) : "v"(P), "v"(v[j]));
                }
                acc ^= P;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int B, int WORK, int WIDTH>
static void run(const char* name, uint32_t* out, int threads, int blocks_per_cu, uint32_t tab_bytes) {
    const int iters = 2048;
    const int blocks = 256 * blocks_per_cu;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, B, WORK, WIDTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tab_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, B, WORK, WIDTH>), dim3(blocks), dim3(threads), tab_bytes, 0, out, 8, tab_bytes, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, B, WORK, WIDTH>), dim3(blocks), dim3(threads), tab_bytes, 0, out, iters, tab_bytes, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)blocks_per_cu * (threads / 64) * iters * B;  // DS wave-instructions per CU
    printf("%-44s %2d waves/CU, table %3u KB: %8.3f ms -> %6.2f cycles per DS wave-instruction per CU (2.4 GHz)\n", name, blocks_per_cu * threads / 64, tab_bytes >> 10, ms,
           ms * 1e-3 * 2.4e9 / per_cu);
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 4);
    run<2, 8, 0, 4>("b32 broadcast, 8 per wait", out, 1024, 2, 16384);
    run<1, 8, 0, 4>("b32 lane-linear, 8 per wait", out, 1024, 2, 16384);
    run<1, 16, 0, 4>("b32 lane-linear, 16 per wait", out, 1024, 2, 16384);
    run<0, 8, 0, 4>("b32 random, 8 per wait", out, 1024, 2, 16384);
    run<0, 16, 0, 4>("b32 random, 16 per wait", out, 1024, 2, 16384);
    run<0, 4, 0, 4>("b32 random, 4 per wait", out, 1024, 2, 16384);
    run<0, 8, 0, 4>("b32 random, 8 per wait", out, 1024, 1, 16384);
    run<0, 8, 0, 4>("b32 random, 8 per wait, 64 KB table", out, 1024, 2, 65536);
    run<0, 8, 0, 4>("b32 random, 8 per wait, 4 KB table", out, 1024, 2, 4096);
    run<0, 8, 3, 4>("b32 random, 8 per wait, 3 pk ops per read", out, 1024, 2, 16384);
    run<0, 8, 6, 4>("b32 random, 8 per wait, 6 pk ops per read", out, 1024, 2, 16384);
    run<1, 8, 3, 4>("b32 lane-linear, 8 per wait, 3 pk ops", out, 1024, 2, 16384);
    run<0, 8, 0, 8>("b64 random, 8 per wait", out, 1024, 2, 32768);
    run<0, 8, 0, 16>("b128 random, 8 per wait", out, 1024, 2, 65536);
    run<1, 8, 0, 16>("b128 lane-linear, 8 per wait", out, 1024, 2, 65536);
    return 0;
}
