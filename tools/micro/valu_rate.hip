// valu_rate.hip -- issue cost of the VALU instructions the X-drop filters are made of (MI355X, wave64).
// Every kernel runs N iterations of 8 INDEPENDENT copies of one instruction (8 accumulators), W waves per SIMD, and
// reports cycles per wave-instruction per SIMD = (kernel time x clock) / (instructions issued per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 77u;
    const uint32_t b = seed ^ 0x01020304u;
    for (int it = 0; it < iters; it++) {
#define ONE(i)                                                                                                               \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                                 \
    if (OP == 1) asm volatile("v_pk_add_i16 %0, %0, %1 op_sel:[1,0] clamp" : "+v"(a[i]) : "v"(b));                           \
    if (OP == 2) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                              \
    if (OP == 3) asm volatile("v_lshrrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(2)); \
    if (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(0x06010400u));                          \
    if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                                 \
    if (OP == 6) asm volatile("v_bfe_u32 %0, %0, 4, 4" : "+v"(a[i]));                                                        \
    if (OP == 7) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));                                          \
    if (OP == 8) asm volatile("v_max_i16_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]));            \
    if (OP == 9) asm volatile("v_pk_add_i16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(a[i]) : "v"(b));                  \
    if (OP == 10) asm volatile("v_min_i16_sdwa %0, %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b)); \
    if (OP == 11) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                             \
    if (OP == 12) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));                                        \
    if (OP == 13) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(0x7Cu));                             \
    if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                                \
    if (OP == 15) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a[i]));
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u) out[0] = s;
}

template <int OP>
static void run(const char* name, uint32_t* out, double ghz) {
    const int waves_per_simd = 4, blocks = 256 * waves_per_simd, iters = 4096;  // 256 CUs x 4 SIMDs, 4 waves of 64 per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 16, 3u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves_per_simd * iters * 32;  // wave-instructions issued on one SIMD
    printf("%-22s %.3f ms  -> %.2f cycles per wave-instruction per SIMD at %.1f GHz\n", name, ms, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
}

int main() {
    uint32_t* out;
    hipMalloc(&out, 4);
    const double ghz = 2.4;
    run<0>("v_add_u32", out, ghz);
    run<1>("v_pk_add_i16 clamp", out, ghz);
    run<2>("v_pk_max_i16", out, ghz);
    run<3>("v_lshrrev_b32_sdwa", out, ghz);
    run<4>("v_perm_b32", out, ghz);
    run<5>("v_and_b32", out, ghz);
    run<6>("v_bfe_u32", out, ghz);
    run<7>("v_lshl_or_b32", out, ghz);
    run<8>("v_max_i16_sdwa", out, ghz);
    run<9>("v_pk_add_i16 op_sel", out, ghz);
    run<10>("v_min_i16_sdwa preserve", out, ghz);
    run<11>("v_pk_min_i16", out, ghz);
    run<12>("v_alignbit_b32", out, ghz);
    run<13>("v_and_or_b32", out, ghz);
    run<14>("v_xor_b32", out, ghz);
    run<15>("v_lshrrev_b32", out, ghz);
    return 0;
}
