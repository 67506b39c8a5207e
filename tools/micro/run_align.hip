// run_align.hip -- round-5 verdict item 6: what would laying the context-record runs of a table WITHOUT transitions (one bucket per key:
// ~4.9 records of 32 bytes per run on the 100 Mbp stand-in) on 128-byte lines buy the class filter's record stream?
//
// The filter reads the runs of a call's query positions in QUERY order = random key order; a run of n records is 32 n contiguous bytes
// that start wherever the runs of the smaller keys end, i.e. at a random 32-byte-aligned offset inside a 128-byte line: it touches
// 1 + (32 n - 32) / 128 lines on average (1.98 for the Poisson(4.9) lengths of the stand-in) although ceil(32 n / 128) would do (1.62).
// Three layouts of the SAME runs (lengths Poisson(lambda), n >= 1), walked the way the filter walks them (a wave takes 64 consecutive
// hits per step, lane L loads the 32-byte record of hit g0 + L as two 16-byte loads; the record index of every hit comes from a
// coalesced 4-byte side array, the same in every layout):
//   0  packed     runs back to back (today's table)
//   1  no-straddle  a run that would cross a line boundary it does not have to cross starts on the next line
//   2  aligned    every run starts on a line
// Prints ms, G hits/s, the table's size and the lines per run each layout touches (computed from the offsets).
// usage: run_align [lambda = 4.9] [runs in the table = 16.7 M] [positions walked = 22 M]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ hit_rec, uint64_t hits, uint32_t steps_per_wave, uint32_t* out) {
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps_per_wave; s++) {
        const uint64_t g = (wid * steps_per_wave + s) * 64 + (uint64_t)lane;
        if (g >= hits) break;
        const uint8_t* p = buf + (uint64_t)hit_rec[g] * 32ull;
        const uint4 a = *reinterpret_cast<const uint4*>(p), c = *reinterpret_cast<const uint4*>(p + 16);
        acc += a.x ^ a.w ^ c.x ^ c.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const double lambda = argc > 1 ? atof(argv[1]) : 4.9;
    const uint32_t nruns = argc > 2 ? (uint32_t)atoll(argv[2]) : (1u << 24);
    const uint32_t npos = argc > 3 ? (uint32_t)atoll(argv[3]) : 22000000u;
    std::mt19937_64 rng(4242);
    std::poisson_distribution<int> pois(lambda);
    std::vector<uint32_t> len(nruns);
    for (auto& n : len) { int v; do v = pois(rng); while (v < 1); n = (uint32_t)v; }
    std::vector<uint32_t> pos_run(npos);
    uint64_t hits = 0;
    for (auto& r : pos_run) { r = (uint32_t)(rng() % nruns); hits += len[r]; }
    printf("%u runs of Poisson(%.1f) >= 1 records (32 B each), %u query positions walked in random key order = %llu hits\n", nruns, lambda, npos, (unsigned long long)hits);
    uint32_t* out;
    hipMalloc(&out, 4);
    for (int layout = 0; layout < 3; layout++) {
        std::vector<uint32_t> start(nruns);  // first record index of every run
        uint64_t rec = 0;
        for (uint32_t r = 0; r < nruns; r++) {
            const uint32_t in_line = (uint32_t)(rec & 3u), n = len[r];
            if (layout == 2 && in_line) rec += 4 - in_line;
            else if (layout == 1 && in_line && (in_line + n + 3) / 4 > (n + 3) / 4) rec += 4 - in_line;  // it would touch one line more than it needs
            start[r] = (uint32_t)rec;
            rec += n;
        }
        const size_t bytes = (size_t)rec * 32 + 4096;
        std::vector<uint32_t> hit_rec;
        hit_rec.reserve(hits);
        double lines = 0;
        for (uint32_t r : pos_run) {
            for (uint32_t k = 0; k < len[r]; k++) hit_rec.push_back(start[r] + k);
            lines += (double)((start[r] + len[r] - 1) / 4 - start[r] / 4 + 1);
        }
        uint8_t* buf; uint32_t* d_hit;
        hipMalloc(&buf, bytes); hipMalloc(&d_hit, hits * 4);
        hipMemset(buf, 1, bytes);
        hipMemcpy(d_hit, hit_rec.data(), hits * 4, hipMemcpyHostToDevice);
        const uint32_t steps = 64;
        const uint64_t waves = (hits + 64ull * steps - 1) / (64ull * steps);
        const uint32_t blocks = (uint32_t)((waves + 3) / 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, buf, d_hit, hits, steps, out);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int k = 0; k < 5; k++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, buf, d_hit, hits, steps, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("  layout %d (%s): table %.2f GB  %.3f lines per run  %8.3f ms  %.2f TB/s of records  %.1f G hits/s\n", layout,
               layout == 0 ? "packed     " : layout == 1 ? "no-straddle" : "aligned    ", bytes / 1e9, lines / npos, best, hits * 32.0 / (best * 1e-3) / 1e12,
               hits / (best * 1e-3) / 1e9);
        hipFree(buf); hipFree(d_hit);
    }
    return 0;
}
