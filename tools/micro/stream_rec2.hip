// stream_rec2.hip -- follow-up to stream_rec.hip: what limits a wave-per-2-KB record stream at ~5.8 TB/s, and does the SHAPE of the
// loads matter?  A wave reads consecutive 2 KB blocks (64 records of 32 bytes).
//   mode 0  record-shaped: lane L loads bytes [32 L, 32 L + 16) and [32 L + 16, 32 L + 32)   (every 128-byte line is requested by BOTH loads)
//   mode 1  line-shaped:   lane L loads bytes [16 L, 16 L + 16) and [1024 + 16 L, ... + 16)    (every line requested once)
//   mode 2  mode 0, next block's loads issued before the current block is consumed (two register sets)
//   mode 3  mode 1, same pipelining
//   mode 4  mode 0 with non-temporal loads
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_rec2 stream_rec2.hip ; usage: stream_rec2 [GB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__device__ __forceinline__ void load2(const uint8_t* __restrict__ blk, int lane, uint4& a, uint4& b) {
    const uint4* p = reinterpret_cast<const uint4*>(blk);
    if (MODE == 1 || MODE == 3) { a = p[lane]; b = p[64 + lane]; }
    else if (MODE == 4) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* q = reinterpret_cast<const u32x4*>(blk);
        const u32x4 va = __builtin_nontemporal_load(q + 2 * lane), vb = __builtin_nontemporal_load(q + 2 * lane + 1);
        a = make_uint4(va.x, va.y, va.z, va.w);
        b = make_uint4(vb.x, vb.y, vb.z, vb.w);
    }
    else { a = p[2 * lane]; b = p[2 * lane + 1]; }
}

template <int MODE>
__global__ __launch_bounds__(1024) void stream(const uint8_t* __restrict__ buf, uint64_t nblk, uint32_t per_wave, uint32_t* out) {
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t b0 = wid * per_wave;
    uint32_t acc = 0;
    if (b0 >= nblk) return;
    const uint64_t b1 = b0 + per_wave < nblk ? b0 + per_wave : nblk;
    if (MODE == 2 || MODE == 3) {
        uint4 a0, c0, a1, c1;
        load2<MODE>(buf + b0 * 2048, lane, a0, c0);
        for (uint64_t b = b0; b < b1; b += 2) {
            load2<MODE>(buf + (b + 1) * 2048, lane, a1, c1);  // (one block of slack behind the buffer)
            acc += a0.x ^ a0.y ^ a0.z ^ a0.w ^ c0.x ^ c0.y ^ c0.z ^ c0.w;
            load2<MODE>(buf + (b + 2) * 2048, lane, a0, c0);
            acc += a1.x ^ a1.y ^ a1.z ^ a1.w ^ c1.x ^ c1.y ^ c1.z ^ c1.w;
        }
    } else {
        for (uint64_t b = b0; b < b1; b++) {
            uint4 a, c;
            load2<MODE>(buf + b * 2048, lane, a, c);
            acc += a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const uint8_t* buf, size_t bytes, uint32_t per_wave, int threads, uint32_t* out) {
    const uint64_t nblk = bytes / 2048;
    const uint64_t waves = (nblk + per_wave - 1) / per_wave;
    const int wpb = threads / 64;
    const uint32_t blocks = (uint32_t)((waves + wpb - 1) / wpb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<MODE>, dim3(blocks), dim3(threads), 0, 0, buf, nblk, per_wave, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(stream<MODE>, dim3(blocks), dim3(threads), 0, 0, buf, nblk, per_wave, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("mode %d  blocks/wave=%u threads=%d : %.3f ms  %.2f TB/s\n", MODE, per_wave, threads, ms, nblk * 2048.0 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 8.3;
    const size_t bytes = (size_t)(gb * 1e9);
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, bytes + 3 * 2048); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes + 3 * 2048);
    for (uint32_t per_wave : {64u, 512u}) {
        run<0>(buf, bytes, per_wave, 1024, out);
        run<1>(buf, bytes, per_wave, 1024, out);
        run<2>(buf, bytes, per_wave, 1024, out);
        run<3>(buf, bytes, per_wave, 1024, out);
        run<4>(buf, bytes, per_wave, 1024, out);
    }
    return 0;
}
