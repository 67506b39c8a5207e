// encode.hip -- ASCII -> 3-bit nucleotide codes (one byte per base), with optional reverse complement.
// Replaces compress_string (common/seed_filter_interface.cu:18-47), compress_string_rev_comp
// (src/seed_filter.cu:110-155) and rev_comp_string (repeat_masker_src/seed_filter.cu:137-167).
// HBM-bound streaming kernels: 16 B per lane per access on the forward stream, 256-entry LUT in LDS.
#include "kernels.h"

namespace sa {

__device__ __forceinline__ uint8_t code_of(uint8_t ch) {
    // common/parameters.h:4-13 ; seed_filter_interface.cu:27-43
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        case 'a': case 'c': case 'g': case 't': return 4;  // L (soft-masked)
        case 'n': case 'N': return 5;                      // N
        case '&': return 7;                                // E (record separator)
        default: return 6;                                 // X
    }
}
__device__ __forceinline__ uint8_t comp_of(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : c; }

__device__ __forceinline__ uint32_t map4(uint32_t w, const uint8_t* lut) {
    return (uint32_t)lut[w & 0xff] | ((uint32_t)lut[(w >> 8) & 0xff] << 8) | ((uint32_t)lut[(w >> 16) & 0xff] << 16) |
           ((uint32_t)lut[w >> 24] << 24);
}

// Forward encode.  `ascii` and `codes` are both 16-byte aligned (hipMalloc'ed staging / padded sequence buffer
// with a 64-byte front pad), so the body runs on uint4 and a scalar tail handles len % 16.
__global__ __launch_bounds__(256) void encode_kernel(const uint8_t* __restrict__ ascii, uint8_t* __restrict__ codes,
                                                     uint32_t len) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = code_of((uint8_t)threadIdx.x);
    __syncthreads();
    const uint32_t nvec = len / 16;
    const uint4* in4 = reinterpret_cast<const uint4*>(ascii);
    uint4* out4 = reinterpret_cast<uint4*>(codes);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
        uint4 v = in4[i];
        v.x = map4(v.x, lut); v.y = map4(v.y, lut); v.z = map4(v.z, lut); v.w = map4(v.w, lut);
        out4[i] = v;
    }
    if (blockIdx.x == 0) {
        for (uint32_t i = nvec * 16 + threadIdx.x; i < len; i += blockDim.x) codes[i] = lut[ascii[i]];
    }
}

// Reverse complement of an encoded sequence: out[len-1-i] = comp(in[i]).  Each lane produces one aligned dword of
// the output from four (reversed) input bytes; reads of a wave cover one contiguous 256-byte span.
__global__ __launch_bounds__(256) void rev_comp_codes_kernel(const uint8_t* __restrict__ codes,
                                                             uint8_t* __restrict__ rc, uint32_t len) {
    const uint32_t nw = (len + 3) / 4;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += gridDim.x * blockDim.x) {
        uint32_t o = w * 4;
        uint32_t packed = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t op = o + b;
            if (op < len) packed |= (uint32_t)comp_of(codes[len - 1 - op]) << (8 * b);
        }
        if (o + 4 <= len) *reinterpret_cast<uint32_t*>(rc + o) = packed;
        else for (int b = 0; o + b < len; b++) rc[o + b] = (uint8_t)(packed >> (8 * b));
    }
}

// out[i] = codes[i] << 3 : the extension kernel ORs this with the query codes to get the matrix index r*8+q
__global__ __launch_bounds__(256) void row_code_kernel(const uint8_t* __restrict__ codes, uint8_t* __restrict__ out,
                                                       uint32_t len) {
    const uint32_t nvec = len / 16;
    const uint4* in4 = reinterpret_cast<const uint4*>(codes);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
        uint4 v = in4[i];
        v.x = (v.x << 3) & 0x38383838u; v.y = (v.y << 3) & 0x38383838u;
        v.z = (v.z << 3) & 0x38383838u; v.w = (v.w << 3) & 0x38383838u;
        out4[i] = v;
    }
    if (blockIdx.x == 0)
        for (uint32_t i = nvec * 16 + threadIdx.x; i < len; i += blockDim.x) out[i] = (uint8_t)(codes[i] << 3);
}

// ---- packed copies for the packed X-drop filter (extend.hip 1b) ---------------------------------------------------
// 2 bits per base, codes >= 4 stored as 0; PHASE copy k holds bases [4j+k, 4j+k+4) in (logical) byte j, so that a window
// that starts (or ends) at ANY base position is byte aligned in the copy k = position & 3.
// Physical layout = OVERLAPPED 128-byte lines: line L holds the logical bytes [96 L, 96 L + 128) of the copy (logical
// byte jj = PACK2_BIAS + index of the 4-base group), i.e. the last 32 bytes of every line repeat the first 32 of the
// next.  Any 32-byte span that starts in the first 96 bytes of a line lies inside that line, so the filter fetches the
// left and right 16-byte windows of a hit (logical bytes [jj-16, jj+16)) from ONE line whatever the anchor position:
// 1.0 target lines per hit instead of 1 + 31/128 (storage x 4/3).  Physical byte of logical byte jj in the line chosen
// for logical byte jb (jb <= jj < jb + 32):  jj + 32 * (jb / 96).
__global__ __launch_bounds__(256) void pack2_phase_kernel(const uint8_t* __restrict__ codes, uint32_t len,
                                                          uint8_t* __restrict__ out, size_t copy_stride, uint32_t nphys) {
    const uint64_t total = (uint64_t)nphys * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = (uint32_t)(i / nphys), p = (uint32_t)(i % nphys);
        const int64_t jj = (int64_t)(p >> 7) * PACK2_PAYLOAD + (p & 127u);
        const int64_t j = jj - PACK2_BIAS;  // index of the 4-base group: bases [4j+k, 4j+k+4)
        uint32_t v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int64_t pos = j * 4 + k + b;
            const uint32_t c = (pos >= 0 && pos < (int64_t)len) ? codes[pos] : 0u;
            v |= (c < 4u ? c : 0u) << (2 * b);
        }
        out[k * copy_stride + p] = (uint8_t)v;
    }
}
// 4 bits per base (code & 7), first base of a byte in the low nibble.  Copy c = k + 2 s (k = base phase 0..1, s = byte shift
// 0..3) holds bases [2 (j + s) + k, 2 (j + s) + k + 2) in byte j: copy (k, 0) makes a window that starts at ANY base byte
// aligned, copy (k, s) additionally DWORD aligned for a window whose first byte o in copy (k, 0) has o & 3 == s (the
// context filter reads 56 bytes of query per hit; byte-aligned 16-byte loads take the slow path of the texture addresser,
// dword-aligned ones do not: 0.55 -> 0.39 ms per 52 M hits for its load stream alone).
__global__ __launch_bounds__(256) void pack4_phase_kernel(const uint8_t* __restrict__ codes, uint32_t len,
                                                          uint8_t* __restrict__ out, size_t copy_stride, uint32_t nbytes) {
    // Bytes BELOW 0 of a shifted copy hold real bases too (byte j of copy (k, s) starts at base 2 (j + s) + k, so bytes
    // -s-1 .. -1 carry bases 0 .. 2 s + k - 1, and byte -s-1 of a k = 1 copy carries base 0 in its high nibble): the left
    // windows of anchors near the block start read them, and a pad code there would cut a walk the reference continues.
    const uint32_t span = nbytes + PACK4_FRONT;  // bytes [-PACK4_FRONT, nbytes) of every copy
    const uint64_t total = (uint64_t)span * PACK4_COPIES;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(i / span);
        const int64_t j = (int64_t)(i % span) - PACK4_FRONT;
        const uint32_t k = c & 1u, sh = c >> 1;
        const int64_t p0 = (j + (int64_t)sh) * 2 + k, p1 = p0 + 1;
        const uint32_t c0 = (p0 >= 0 && p0 < (int64_t)len) ? (codes[p0] & 7u) : 7u;
        const uint32_t c1 = (p1 >= 0 && p1 < (int64_t)len) ? (codes[p1] & 7u) : 7u;
        out[(int64_t)c * (int64_t)copy_stride + j] = (uint8_t)(c0 | (c1 << 4));
    }
}

// ---- 2-bit copies of a QUERY strand for the class filter (extend.hip 1d) --------------------------------------------------
// Copy c = p + 4 s (p = base phase 0..3, s = byte shift 0..3): byte j holds bases [4 (j + s) + p, 4 (j + s) + p + 4), first base
// in the low bits, codes >= 4 stored as 0, zero beyond the block.  A 64-base window that starts at ANY base position `pos` is
// four DWORD-ALIGNED dwords of copy (pos & 3, (pos >> 2) & 3) at byte (pos >> 2) - ((pos >> 2) & 3): no byte-granular load
// and no funnel shifts in the filter.  Each thread produces one dword of one copy from the 19 codes it spans.
// (round 4: the class filter takes its windows out of copy 0 alone with funnel shifts -- extend.hip ONE_COPY -- so the engine builds
//  `copies` = 1 of them unless option cls_one_copy = 2 asks for the sixteen)
__global__ __launch_bounds__(256) void pack2_shifted_kernel(const uint8_t* __restrict__ codes, uint32_t len,
                                                            uint8_t* __restrict__ out, size_t copy_stride, uint32_t copies) {
    const uint32_t ndw = (uint32_t)(copy_stride / 4);
    const uint64_t total = (uint64_t)ndw * copies;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)(i / ndw), w = (uint32_t)(i % ndw);
        const uint32_t p = c & 3u, sh = c >> 2;
        const uint64_t b0 = ((uint64_t)w * 4 + sh) * 4 + p;  // first base of this dword
        uint32_t v = 0;
        if (b0 < len) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint64_t pos = b0 + k;
                const uint32_t cd = pos < len ? codes[pos] : 0u;
                v |= (cd < 4u ? cd : 0u) << (2 * k);
            }
        }
        *reinterpret_cast<uint32_t*>(out + (size_t)c * copy_stride + (size_t)w * 4) = v;
    }
}

// which of the 8 codes occur in a block: the class filter's score bounds only have to cover the codes that are there
__global__ __launch_bounds__(256) void code_presence_kernel(const uint8_t* __restrict__ codes, uint32_t len, uint32_t* __restrict__ mask) {
    uint32_t m = 0;
    const uint32_t nvec = len / 16;
    const uint4* in4 = reinterpret_cast<const uint4*>(codes);  // (the padded sequence buffers are 16-byte aligned)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += gridDim.x * blockDim.x) {
        const uint4 v = in4[i];
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int b = 0; b < 4; b++) m |= 1u << ((d[j] >> (8 * b)) & 7u);
    }
    if (blockIdx.x == 0)
        for (uint32_t i = nvec * 16 + threadIdx.x; i < len; i += blockDim.x) m |= 1u << (codes[i] & 7u);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m |= (uint32_t)__shfl_xor((int)m, off, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicOr(mask, m);
}

static inline int grid_for(uint64_t work_items, int block, int max_blocks = 256 * 8) {
    uint64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (uint64_t)max_blocks) g = max_blocks;
    return (int)g;
}

void launch_encode(const uint8_t* ascii, uint8_t* codes, uint32_t len, hipStream_t s) {
    if (len == 0) return;
    hipLaunchKernelGGL(encode_kernel, dim3(grid_for(len / 16 + 1, 256)), dim3(256), 0, s, ascii, codes, len);
}
void launch_rev_comp_codes(const uint8_t* codes, uint8_t* codes_rc, uint32_t len, hipStream_t s) {
    if (len == 0) return;
    hipLaunchKernelGGL(rev_comp_codes_kernel, dim3(grid_for((len + 3) / 4, 256)), dim3(256), 0, s, codes, codes_rc, len);
}
void launch_row_code(const uint8_t* codes, uint8_t* out, uint32_t len, hipStream_t s) {
    if (len == 0) return;
    hipLaunchKernelGGL(row_code_kernel, dim3(grid_for(len / 16 + 1, 256)), dim3(256), 0, s, codes, out, len);
}
uint32_t pack2_phys_bytes(uint32_t len) {  // physical bytes of one 2-bit copy: logical bytes up to len/4 + 64 + bias, in lines
    const uint64_t jjmax = (uint64_t)len / 4 + 64 + PACK2_BIAS;
    return (uint32_t)((jjmax / PACK2_PAYLOAD + 2) * 128);
}
void launch_pack2_phases(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t nphys, hipStream_t s) {
    hipLaunchKernelGGL(pack2_phase_kernel, dim3(grid_for((uint64_t)nphys * 4, 256)), dim3(256), 0, s, codes, len, out, copy_stride, nphys);
}
void launch_pack4_phases(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t nbytes, hipStream_t s) {
    hipLaunchKernelGGL(pack4_phase_kernel, dim3(grid_for((uint64_t)(nbytes + PACK4_FRONT) * PACK4_COPIES, 256)), dim3(256), 0, s, codes, len, out, copy_stride, nbytes);
}
size_t q2_copy_stride(uint32_t len) { return (((size_t)len / 4 + 1 + Q2_TAIL) + 127) & ~(size_t)127; }
void launch_pack2_shifted(const uint8_t* codes, uint32_t len, uint8_t* out, size_t copy_stride, uint32_t copies, hipStream_t s) {
    hipLaunchKernelGGL(pack2_shifted_kernel, dim3(grid_for((uint64_t)(copy_stride / 4) * copies, 256)), dim3(256), 0, s, codes, len, out, copy_stride, copies);
}
void launch_code_presence(const uint8_t* codes, uint32_t len, uint32_t* mask, hipStream_t s) {
    if (len == 0) return;
    hipLaunchKernelGGL(code_presence_kernel, dim3(grid_for(len / 16 + 1, 256)), dim3(256), 0, s, codes, len, mask);
}
void launch_encode_rev_comp(const uint8_t* ascii, uint8_t* codes, uint8_t* codes_rc, uint32_t len, hipStream_t s) {
    // two streaming passes: the second reads the freshly written codes (L2 / Infinity Cache resident)
    launch_encode(ascii, codes, len, s);
    launch_rev_comp_codes(codes, codes_rc, len, s);
}

}  // namespace sa
