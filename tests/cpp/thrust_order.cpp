// thrust_order.cpp -- test infrastructure (tests/test_gpu_thrust_order.py): the reference's ordering chain run by rocThrust ITSELF.
//
// The reference orders and de-duplicates the anchors of an iteration with thrust on the device: stable_sort(hspComp) ->
// unique_copy(hspEqual) -> stable_sort(hspCompLastz) (src/seed_filter.cu:776-782); the repeat masker: stable_sort(hspComp) ->
// unique_copy(hspEqual) -> stable_sort(hspDiagComp) -> unique_copy(hspDiagEqual) -> stable_sort(hspFinalComp)
// (repeat_masker_src/seed_filter.cu:819-831).  hspEqual is not transitive (containment on one diagonal), so WHICH pairs
// unique_copy compares decides the result: the oracle and the engine assume head flags on adjacent INPUT pairs (hazard H3).  This
// program makes no such assumption: it calls thrust::stable_sort / thrust::unique_copy of the rocThrust in this image on device
// vectors, with the reference's predicates restated below (src/seed_filter.cu:47-108, repeat_masker_src/seed_filter.cu:45-135),
// and writes what comes out.  rocThrust is the same algorithm family as the CUDA thrust the reference links (both implement
// unique_copy as a select-if over adjacent-pair head flags), not the same binary: a second route, not a pin.
//
// usage: thrust_order <in.bin> <out.bin>   in: u32 n, u32 rm, n x {u32 ref_start, query_start, len, i32 score}; out: u32 m, m records
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 thrust_order.cpp -o thrust_order
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>
#include <thrust/sort.h>
#include <thrust/unique.h>

#include <cstdint>
#include <cstdio>
#include <vector>

struct segmentPair { uint32_t ref_start, query_start, len; int score; };

// src/seed_filter.cu:47-52 (== hspDiagEqual of the repeat masker, :45-50): same diagonal, one interval inside the other
struct HspEqual {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        return ((x.ref_start - x.query_start) == (y.ref_start - y.query_start)) &&
               (((x.ref_start >= y.ref_start) && ((x.ref_start + x.len) <= (y.ref_start + y.len))) ||
                ((y.ref_start >= x.ref_start) && ((y.ref_start + y.len) <= (x.ref_start + x.len))));
    }
};
// src/seed_filter.cu:54-80: diagonal (wrapped u32), ref_start, len, score descending
struct HspComp {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        const uint32_t dx = x.ref_start - x.query_start, dy = y.ref_start - y.query_start;
        if (dx != dy) return dx < dy;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        return x.score > y.score;
    }
};
// src/seed_filter.cu:82-108: query_start, ref_start, len, score descending
struct HspCompLastz {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.len != y.len) return x.len < y.len;
        return x.score > y.score;
    }
};
// repeat masker :109-135 (its hspComp): query_start, len descending, ref_start, score descending
struct RmHspComp {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.len != y.len) return x.len > y.len;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        return x.score > y.score;
    }
};
// repeat masker :52-78 (hspDiagComp): diagonal (wrapped u32), ref_start, query_start, score descending -- no length key
struct RmDiagComp {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        const uint32_t dx = x.ref_start - x.query_start, dy = y.ref_start - y.query_start;
        if (dx != dy) return dx < dy;
        if (x.ref_start != y.ref_start) return x.ref_start < y.ref_start;
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        return x.score > y.score;
    }
};
// repeat masker :80-85 (its hspEqual): all four fields equal
struct RmHspEqual {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        return x.ref_start == y.ref_start && x.query_start == y.query_start && x.len == y.len && x.score == y.score;
    }
};
// repeat masker :87-107 (hspFinalComp): query_start, score descending, ref_start descending
struct RmFinalComp {
    __host__ __device__ bool operator()(segmentPair x, segmentPair y) const {
        if (x.query_start != y.query_start) return x.query_start < y.query_start;
        if (x.score != y.score) return x.score > y.score;
        return x.ref_start > y.ref_start;
    }
};

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    uint32_t hdr[2];
    if (!f || fread(hdr, 4, 2, f) != 2) return 2;
    const uint32_t n = hdr[0], rm = hdr[1];
    std::vector<segmentPair> h(n);
    if (n && fread(h.data(), sizeof(segmentPair), n, f) != n) return 2;
    fclose(f);
    thrust::device_vector<segmentPair> a(h.begin(), h.end()), b(n);
    size_t m = n;
    if (!rm) {
        thrust::stable_sort(a.begin(), a.begin() + m, HspComp());                                        // :776
        m = thrust::unique_copy(a.begin(), a.begin() + m, b.begin(), HspEqual()) - b.begin();            // :778
        thrust::stable_sort(b.begin(), b.begin() + m, HspCompLastz());                                   // :782
    } else {
        thrust::stable_sort(a.begin(), a.begin() + m, RmHspComp());                                      // rm :819
        m = thrust::unique_copy(a.begin(), a.begin() + m, b.begin(), RmHspEqual()) - b.begin();          // rm :821
        thrust::stable_sort(b.begin(), b.begin() + m, RmDiagComp());                                     // rm :825 (hspDiagComp)
        const size_t m2 = thrust::unique_copy(b.begin(), b.begin() + m, a.begin(), HspEqual()) - a.begin();  // rm :827 (hspDiagEqual)
        m = m2;
        thrust::stable_sort(a.begin(), a.begin() + m, RmFinalComp());                                    // rm :831
        thrust::copy(a.begin(), a.begin() + m, b.begin());
    }
    thrust::host_vector<segmentPair> r(b.begin(), b.begin() + m);
    FILE* o = fopen(argv[2], "wb");
    const uint32_t mm = (uint32_t)m;
    fwrite(&mm, 4, 1, o);
    if (m) fwrite(r.data(), sizeof(segmentPair), m, o);
    fclose(o);
    return 0;
}
