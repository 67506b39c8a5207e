// piece_order.hip -- does the ORDER in which the class filter walks the context table matter?  (round-4 question)
// The filter of a call reads ~3.4 M runs of ~2.5 KB (a query position's 13 buckets, 32-byte records) at what are, for the memory
// system, random places of a 34 GB table.  This tool reads the same number of equally sized pieces of a buffer of the same size
//   R  at uniformly random 32-byte-aligned offsets            (what the filter does today: query order = random key order)
//   S  at the same offsets, sorted ascending                   (what processing a call's positions in KEY order would give)
//   W  sorted inside windows of 4096 consecutive pieces        (a cheap partial order)
//   P<n>  partitioned into n equal table ranges, random inside  (round 6: one counting pass by the top bits of the run offset)
//   Q  pieces laid end to end                                  (the pure stream)
// with the filter's load shape (a wave takes 64 records = 2 KB per step, lane L loads bytes [32 L, 32 L + 32)), from a plain
// hipMalloc and from a range mapped in 1 GiB chunks like the engine's table arena.
// Build: hipcc --offload-arch=gfx950 -O3 -o piece_order piece_order.hip ; usage: piece_order [table GB] [piece bytes] [pieces]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ off, uint32_t npieces, uint32_t piece_bytes,
                                            uint32_t per_wave, uint32_t* out) {
    const uint32_t wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const uint32_t p0 = wid * per_wave;
    for (uint32_t p = p0; p < p0 + per_wave && p < npieces; p++) {
        const uint8_t* base = buf + off[p];
        for (uint32_t b = 0; b < piece_bytes; b += 2048) {
            const uint32_t o = b + 32u * (uint32_t)lane;
            if (o + 32 <= piece_bytes) {
                const uint4* q = reinterpret_cast<const uint4*>(base + o);
                const uint4 a = q[0], c = q[1];
                acc += a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static double run(const uint8_t* buf, const std::vector<uint64_t>& off, uint32_t piece_bytes, uint64_t* d_off, uint32_t* out, const char* tag) {
    const uint32_t n = (uint32_t)off.size();
    hipMemcpy(d_off, off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice);
    const uint32_t per_wave = 2;  // (the filter: 4096 hits = ~1.6 runs per wave)
    const uint32_t waves = (n + per_wave - 1) / per_wave, blocks = (waves + 3) / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, buf, d_off, n, piece_bytes, per_wave, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(walk, dim3(blocks), dim3(256), 0, 0, buf, d_off, n, piece_bytes, per_wave, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double tbs = (double)n * piece_bytes / (ms * 1e-3) / 1e12;
    printf("  %-28s %8.3f ms  %.2f TB/s\n", tag, ms, tbs);
    return tbs;
}

static uint8_t* map_chunks(size_t bytes, std::vector<hipMemGenericAllocationHandle_t>& hs) {
    const size_t CH = (size_t)1 << 30;
    const size_t total = (bytes + CH - 1) / CH * CH;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) return nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t o = 0; o < total; o += CH) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, CH, &prop, 0) != hipSuccess) return nullptr;
        if (hipMemMap((uint8_t*)va + o, CH, 0, h, 0) != hipSuccess) return nullptr;
        if (hipMemSetAccess((uint8_t*)va + o, CH, &acc, 1) != hipSuccess) return nullptr;
        hs.push_back(h);
    }
    return (uint8_t*)va;
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 34.0;
    const uint32_t piece = argc > 2 ? (uint32_t)atoi(argv[2]) : 2560;
    const uint32_t n = argc > 3 ? (uint32_t)atoi(argv[3]) : 3400000;
    const size_t bytes = (size_t)(gb * 1e9);
    std::mt19937_64 rng(12345);
    std::vector<uint64_t> R(n), S, W, Q(n);
    const uint64_t slots = (bytes - piece - 4096) / 32;
    for (uint32_t i = 0; i < n; i++) { R[i] = (rng() % slots) * 32; Q[i] = (uint64_t)i * piece % ((bytes - piece) / 32 * 32); }
    S = R;
    std::sort(S.begin(), S.end());
    W = R;
    for (size_t i = 0; i < W.size(); i += 4096) std::sort(W.begin() + i, W.begin() + std::min(W.size(), i + 4096));
    // P<n>: a PARTIAL key order -- the pieces partitioned into n equal ranges of the table (one counting pass over a call's position
    // records by the top bits of their run offset), random inside a range (round 6: would one cheap partition pass buy what S buys?)
    std::vector<std::vector<uint64_t>> P;
    const uint32_t parts[3] = {256, 4096, 65536};
    for (uint32_t np : parts) {
        std::vector<uint64_t> v = R;
        const uint64_t width = (bytes + np - 1) / np;
        std::stable_sort(v.begin(), v.end(), [width](uint64_t a, uint64_t b) { return a / width < b / width; });
        P.push_back(v);
    }
    uint64_t* d_off; uint32_t* out;
    hipMalloc(&d_off, (size_t)n * 8); hipMalloc(&out, 4);
    for (int vmm = 0; vmm < 2; vmm++) {
        uint8_t* buf = nullptr;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        if (vmm) buf = map_chunks(bytes, hs);
        else if (hipMalloc(&buf, bytes) != hipSuccess) buf = nullptr;
        if (!buf) { printf("allocation (%s) failed\n", vmm ? "mapped chunks" : "hipMalloc"); continue; }
        hipMemset(buf, 1, bytes);
        hipDeviceSynchronize();
        printf("%.1f GB table, %s; %u pieces of %u bytes (%.2f GB read per walk)\n", gb, vmm ? "1 GiB chunks mapped into a reserved range" : "plain hipMalloc", n, piece,
               (double)n * piece / 1e9);
        run(buf, R, piece, d_off, out, "R random order");
        run(buf, W, piece, d_off, out, "W sorted per 4096 pieces");
        run(buf, P[0], piece, d_off, out, "P256   partitioned into 256 table ranges");
        run(buf, P[1], piece, d_off, out, "P4096  partitioned into 4096 table ranges");
        run(buf, P[2], piece, d_off, out, "P65536 partitioned into 65536 table ranges");
        run(buf, S, piece, d_off, out, "S sorted (key order)");
        run(buf, Q, piece, d_off, out, "Q end to end (stream)");
        if (vmm) {
            const size_t CH = (size_t)1 << 30;
            for (size_t i = 0; i < hs.size(); i++) { hipMemUnmap(buf + i * CH, CH); hipMemRelease(hs[i]); }
            hipMemAddressFree(buf, hs.size() * CH);
        } else hipFree(buf);
    }
    return 0;
}
