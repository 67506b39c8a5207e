// hit_shaped.hip -- why does the class filter fetch ~3.1 lines per 160-byte run on sparse-hit input when a bare walk of the same
// pieces fetches 2.0 (tools/micro/piece_order.hip)?  This tool reads runs of R records (32 bytes each) at random
// 32-byte-aligned offsets of a large buffer THE WAY THE FILTER DOES: a wave takes 64 consecutive HITS per step, lane L loads the
// record of hit g0 + L -- so one load instruction covers ~64 / R runs -- as two 16-byte loads (lo half, hi half of the record).
//   mode 0  both halves back to back (the filter)
//   mode 1  lo halves only (16 of the 32 bytes: the same lines)
//   mode 2  lo halves of step s, hi halves issued one step later (after the lo halves have arrived)
//   mode 3  mode 0 with the two loads marked non-temporal
//   mode 4  one lane per RUN loads the run's lines with 16-byte loads, lane L of the run's group taking bytes [16 L, +16) (line-shaped)
// Count TCC_EA0_RDREQ_sum / TCP_TCC_READ_REQ_sum per launch with rocprofv3 --pmc.  usage: hit_shaped [GB] [records per run] [runs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <random>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ run_off, uint32_t recs, uint64_t hits,
                                            uint32_t steps_per_wave, uint32_t* out) {
    const uint64_t wid = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    uint4 pend = {0, 0, 0, 0};
    const uint8_t* pend_p = nullptr;
    for (uint32_t s = 0; s < steps_per_wave; s++) {
        const uint64_t g = (wid * steps_per_wave + s) * 64 + (uint64_t)lane;
        if (g >= hits) break;
        const uint8_t* p = buf + run_off[g / recs] + 32u * (uint32_t)(g % recs);
        if (MODE == 0) {
            const uint4 a = *reinterpret_cast<const uint4*>(p), c = *reinterpret_cast<const uint4*>(p + 16);
            acc += a.x ^ a.w ^ c.x ^ c.w;
        } else if (MODE == 1) {
            const uint4 a = *reinterpret_cast<const uint4*>(p);
            acc += a.x ^ a.w;
        } else if (MODE == 2) {
            const uint4 a = *reinterpret_cast<const uint4*>(p);
            if (pend_p) { const uint4 c = *reinterpret_cast<const uint4*>(pend_p + 16); acc += c.x ^ c.w; }
            acc += a.x ^ a.w ^ pend.x;
            pend = a;
            pend_p = p;
        } else if (MODE == 3) {
            const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)), c = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + 16));
            acc += a.x ^ a.w ^ c.x ^ c.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const uint8_t* buf, const uint64_t* d_off, uint32_t recs, uint64_t nruns, uint32_t* out) {
    const uint64_t hits = nruns * recs;
    const uint32_t steps = 64;  // 4096 hits per wave, like the filter
    const uint64_t waves = (hits + 64ull * steps - 1) / (64ull * steps);
    const uint32_t blocks = (uint32_t)((waves + 3) / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(256), 0, 0, buf, d_off, recs, hits, steps, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(walk<MODE>, dim3(blocks), dim3(256), 0, 0, buf, d_off, recs, hits, steps, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    printf("  mode %d: %8.3f ms  %.2f TB/s of records (%.1f G hits/s)\n", MODE, ms, hits * 32.0 / (ms * 1e-3) / 1e12, hits / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 3.2;
    const uint32_t recs = argc > 2 ? (uint32_t)atoi(argv[2]) : 5;
    const uint64_t nruns = argc > 3 ? (uint64_t)atoll(argv[3]) : 17700000ull;
    const int only = argc > 4 ? atoi(argv[4]) : -1;
    const size_t bytes = (size_t)(gb * 1e9);
    std::mt19937_64 rng(777);
    std::vector<uint64_t> off(nruns);
    const uint64_t slots = (bytes - 32ull * recs - 4096) / 32;
    for (auto& o : off) o = (rng() % slots) * 32;
    uint8_t* buf; uint64_t* d_off; uint32_t* out;
    hipMalloc(&buf, bytes); hipMalloc(&d_off, nruns * 8); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    hipMemcpy(d_off, off.data(), nruns * 8, hipMemcpyHostToDevice);
    printf("%.1f GB table, %llu runs of %u records (%u bytes) at random 32-byte-aligned offsets, 64 hits per wave step\n", gb, (unsigned long long)nruns, recs, recs * 32);
    if (only < 0 || only == 0) run<0>(buf, d_off, recs, nruns, out);
    if (only < 0 || only == 1) run<1>(buf, d_off, recs, nruns, out);
    if (only < 0 || only == 2) run<2>(buf, d_off, recs, nruns, out);
    if (only < 0 || only == 3) run<3>(buf, d_off, recs, nruns, out);
    return 0;
}
