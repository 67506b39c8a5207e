// kmer_dev.h -- spaced-seed k-mer extraction from the encoded (one code byte per base) sequence.
// Restates GetKmerIndexAtPos (common/ntcoding.cpp:43-61) on codes: a window is valid iff all `span` codes are
// < 4 -- exactly the characters 'A','C','G','T' (anything else, incl. lower case, N and '&', is invalid there,
// ntcoding.cpp:10-19,49-51) -- and the key takes the care positions in shape order, first one most significant.
#pragma once
#include "kernels.h"

namespace sa {

__device__ __forceinline__ uint64_t load8u(const uint8_t* p) {  // unaligned 8-byte load (one global_load_dwordx2)
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ uint4 load16u(const uint8_t* p) {  // unaligned 16-byte load (one global_load_dwordx4)
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// low 2 bits of each of the 8 bytes of w -> 16 packed bits (byte 0 -> bits 1:0)
__device__ __forceinline__ uint32_t pack2(uint64_t w) {
    uint64_t x = w & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return (uint32_t)x;
}

// Reads the 32-byte window at seq+p (the caller guarantees p+span <= len; bytes past the span may be pad bytes,
// they are masked off).  Returns validity; key valid only if true.
__device__ __forceinline__ bool kmer_at(const uint8_t* __restrict__ seq, uint32_t p, const SeedShape& sh, uint32_t& key) {
    const uint8_t* w = seq + p;
    uint64_t w0 = load8u(w), w1 = load8u(w + 8), w2 = load8u(w + 16), w3 = load8u(w + 24);
    const int span = sh.span;
    // any code >= 4 inside the span?  codes are 0..7 -> test bit 2 of every byte
    auto mask_for = [span](int word) -> uint64_t {
        int nb = span - 8 * word;
        if (nb <= 0) return 0ull;
        if (nb >= 8) return 0x0404040404040404ull;
        return 0x0404040404040404ull & ((1ull << (8 * nb)) - 1ull);
    };
    uint64_t bad = (w0 & mask_for(0)) | (w1 & mask_for(1)) | (w2 & mask_for(2)) | (w3 & mask_for(3));
    uint64_t packed = (uint64_t)pack2(w0) | ((uint64_t)pack2(w1) << 16) | ((uint64_t)pack2(w2) << 32) |
                      ((uint64_t)pack2(w3) << 48);
    uint32_t k = 0;
    for (int j = 0; j < sh.weight; j++) k = (k << 2) | (uint32_t)((packed >> (2 * sh.pos[j])) & 3ull);
    key = k;
    return bad == 0;
}

// The k-mers of the FOUR positions p4 .. p4 + 3 (p4 a multiple of 4, seq dword aligned) from ONE set of aligned loads:
// byte-aligned 8-byte loads take the slow path of the texture addresser (tools/micro: 0.55 vs 0.39 ms for the same bytes in
// extend.hip 1c), so the 36 bytes are fetched as nine aligned dwords and the windows of the shifted positions are cut out
// with v_alignbyte_b32.  valid bit r of the result <=> position p4 + r holds a valid k-mer (key[r]).
__device__ __forceinline__ uint32_t kmer4_at(const uint8_t* __restrict__ seq, uint32_t p4, const SeedShape& sh, uint32_t key[4]) {
    const uint4* w = reinterpret_cast<const uint4*>(seq + p4);
    const uint4 a = w[0], b = w[1];
    const uint32_t c = *reinterpret_cast<const uint32_t*>(seq + p4 + 32);
    const uint32_t d[9] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c};
    const int span = sh.span;
    auto mask_for = [span](int word) -> uint64_t {
        int nb = span - 8 * word;
        if (nb <= 0) return 0ull;
        if (nb >= 8) return 0x0404040404040404ull;
        return 0x0404040404040404ull & ((1ull << (8 * nb)) - 1ull);
    };
    uint32_t valid = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = r == 0 ? d[j] : __builtin_amdgcn_alignbyte(d[j + 1], d[j], (uint32_t)r);
        const uint64_t w0 = ((uint64_t)e[1] << 32) | e[0], w1 = ((uint64_t)e[3] << 32) | e[2], w2 = ((uint64_t)e[5] << 32) | e[4],
                       w3 = ((uint64_t)e[7] << 32) | e[6];
        const uint64_t bad = (w0 & mask_for(0)) | (w1 & mask_for(1)) | (w2 & mask_for(2)) | (w3 & mask_for(3));
        const uint64_t packed = (uint64_t)pack2(w0) | ((uint64_t)pack2(w1) << 16) | ((uint64_t)pack2(w2) << 32) | ((uint64_t)pack2(w3) << 48);
        uint32_t k = 0;
        for (int j = 0; j < sh.weight; j++) k = (k << 2) | (uint32_t)((packed >> (2 * sh.pos[j])) & 3ull);
        key[r] = k;
        valid |= (bad == 0 ? 1u : 0u) << r;
    }
    return valid;
}

}  // namespace sa
