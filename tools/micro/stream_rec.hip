// stream_rec.hip -- what does the access pattern of the context filter cost when NOTHING is computed?
// Every wave reads `chunk` consecutive records of REC bytes (one record per lane and step, two 16-byte loads per lane for
// REC = 32, a 16-byte + a 12-byte load for REC = 28), chunks are dealt to waves in order.  Modes: waves per SIMD 4 / 8.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_rec stream_rec.hip ; usage: stream_rec [GB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int REC>
__global__ __launch_bounds__(1024) void stream(const uint8_t* __restrict__ buf, uint64_t nrec, uint32_t chunk, uint32_t* out) {
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint64_t r0 = wid * chunk;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < chunk; i += 64) {
        const uint64_t r = r0 + i + lane;
        if (r >= nrec) break;
        const uint8_t* p = buf + r * REC;
        uint4 a, b = {0, 0, 0, 0};
        __builtin_memcpy(&a, p, 16);
        if (REC == 32) __builtin_memcpy(&b, p + 16, 16);
        else __builtin_memcpy(&b, p + 16, 12);
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int REC>
static void run(const uint8_t* buf, size_t bytes, uint32_t chunk, int threads, uint32_t* out) {
    const uint64_t nrec = bytes / REC;
    const uint64_t waves = (nrec + chunk - 1) / chunk;
    const int wpb = threads / 64;
    const uint32_t blocks = (uint32_t)((waves + wpb - 1) / wpb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<REC>, dim3(blocks), dim3(threads), 0, 0, buf, nrec, chunk, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 5; k++) hipLaunchKernelGGL(stream<REC>, dim3(blocks), dim3(threads), 0, 0, buf, nrec, chunk, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("REC=%d chunk=%u threads=%d : %.3f ms  %.2f TB/s  %.1f G records/s\n", REC, chunk, threads, ms, nrec * REC / (ms * 1e-3) / 1e12,
           nrec / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 5.5;
    const size_t bytes = (size_t)(gb * 1e9);
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, bytes + 64); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes + 64);
    for (int threads : {512, 1024}) {
        for (uint32_t chunk : {1024u, 4096u}) {
            run<32>(buf, bytes, chunk, threads, out);
            run<28>(buf, bytes, chunk, threads, out);
        }
    }
    return 0;
}
