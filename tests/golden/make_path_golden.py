#!/usr/bin/env python3
"""Generator of tests/golden/path_golden.json: the WHOLE hot path of the src/ binary executed from the reference's own files, end to end.

What runs -- every file as it lies under /root/reference, compiled with g++ into ONE program:
    src/seed_filter.cu                 InitializeProcessor, SendQueryWriteRequest, SeedAndFilter (the iteration plan over MAX_HITS, the
                                       thrust chain) and every kernel (compress_string_rev_comp, find_num_hits, find_hits, find_hsps, compress_output)
    common/seed_filter_interface.cu    InitializeInterface, SendRefWriteRequest (compress_string), the GPU token pool's globals
    common/seed_pos_table.cu           GenerateSeedPosTable, InclusivePrefixScan, SendSeedPosTable
    common/ntcoding.cpp                GenerateShapePos, GetKmerIndexAtPos, IsTransitionAtPos, RevComp          (unedited, no stand-in)
    src/seeder.cpp                     seeder_body::operator(): chunk loop, both strands, seed words, every g_SeedAndFilter call (unedited)
in the order src/main.cpp calls them (:296-297, :615, :621, :661, :377): g_InitializeInterface, g_InitializeProcessor, g_SendRefWriteRequest,
GenerateSeedPosTable, g_SendQueryWriteRequest, then seeder_body on every interval of the query block.  Recorded: what every g_SeedAndFilter call
returns (header element {len = HSPs, score = seed hits} + the HSPs in the order the reference leaves them).

What the image lacks and what stands in for it (this repository's code, written to a temporary directory together with the edited copies):
  * the CUDA runtime: malloc / memcpy / free behind cudaMalloc / cudaMemcpy / cudaFree, one device, totalGlobalMem taken from the case file
    (MAX_HITS = 4194304 x GB comes out of the reference's own arithmetic: 64 KiB -> 256 hits, so that calls split into iterations);
  * kernel launches: `k <<<grid, block>>> (args);` is rewritten (sed) to `launch(grid, block, [&]{ k(args); });` -- the SIMT emulation of
    make_find_hsps_golden.py re-done with fibers (one ucontext per CUDA thread of a block on ONE OS thread, resumed round-robin up to the
    next __syncthreads / __syncwarp / __shfl_up_sync: SeedAndFilter launches a block per seed word, std::barrier costs ~1 ms per block);
  * thrust: device_vector = std::vector, inclusive_scan = std::partial_sum, lower_bound / stable_sort = the std ones, unique_copy = head flags on
    adjacent INPUT pairs (hazard H3; tests/cpp/thrust_order.cpp runs rocThrust's own unique_copy against that reading on the GPU box);
  * TBB: tuple / get = std::tuple / std::get, parallel_for runs its body once over the whole range (serial);
  * DRAM's constructor (plain buffers);  the H1 edit of the other generators (count[4] indexed with codes 4..7 overruns the stack on a CPU).
A build with stand-ins does not pin the oracle.  What the vectors add: no stage of the path rests on this repository's reading
alone any more -- the restated orchestration that make_src_golden.py still carried around the kernels (scans, two-iteration plan, limit_pos,
the `num_iter--` case, per-iteration sort / unique / sort, concatenation) is here the reference's own text too.

usage: python tests/golden/make_path_golden.py   (needs /root/reference and g++)
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_find_hsps_golden import hoxd70  # noqa: E402
from make_rm_golden import pack_rows  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "path_golden.json")
S19 = "TTT0T00TT00T0T0TTTT"

PRELUDE = r'''#pragma once
// force-included in front of every translation unit: SIMT emulation + the CUDA runtime's names (this repository's code)
#include <ucontext.h>
#include <vector>
#include <memory>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cassert>
#include <cmath>
#include <math.h>            // `using std::log;` makes log(4.0f) the float overload as under nvcc (hazard H2)
#define __global__
#define __shared__ static     // function-local static == block-shared: blocks run one at a time
#define __host__
#define __device__
#define __restrict__
// One FIBER (ucontext) per CUDA thread of a block, all on the calling OS thread, resumed round-robin; a fiber runs until it reaches a
// synchronisation point whose other parties have not arrived yet.  (The std::thread / std::barrier emulation of the older generators
// spends ~1 ms per barrier of a 128-thread block: SeedAndFilter launches one block per seed word.)
struct dim3_ { unsigned x, y, z; };
static dim3_ threadIdx, blockIdx, blockDim, gridDim;  static const int warpSize = 32;   // threadIdx: set by the scheduler at every resume
struct SaFiber { ucontext_t ctx; char* stack; bool done; };
struct SaBar { unsigned arrived, gen; };
static ucontext_t sa_sched;  static std::vector<SaFiber> sa_fibers;  static unsigned sa_cur;
static SaBar sa_block_bar, sa_warp_bar[32];  static long long sa_slot[32][32];
static void (*sa_call)(void*);  static void* sa_arg;
static inline void sa_yield() { swapcontext(&sa_fibers[sa_cur].ctx, &sa_sched); }
static inline void sa_wait(SaBar& b, unsigned n) {
  const unsigned my = b.gen;
  if (++b.arrived == n) { b.arrived = 0; b.gen++; return; }
  while (b.gen == my) sa_yield(); }
static inline void __syncthreads() { sa_wait(sa_block_bar, blockDim.x); }
static inline void __syncwarp()    { sa_wait(sa_warp_bar[threadIdx.x / 32], 32); }
template <class T> static inline T __shfl_up_sync(unsigned, T v, int off) {   // lanes < off keep their own value
  const unsigned w = threadIdx.x / 32; const int lane = threadIdx.x % 32;
  sa_slot[w][lane] = (long long)v; sa_wait(sa_warp_bar[w], 32);
  T r = (lane >= off) ? (T)sa_slot[w][lane - off] : v; sa_wait(sa_warp_bar[w], 32); return r; }
static void sa_fiber_main() { sa_call(sa_arg); sa_fibers[sa_cur].done = true; }   // uc_link: back to the scheduler
template <class F> static void sa_tramp(void* p) { (*(F*)p)(); }
template <class F> static void launch(unsigned grid, unsigned block, F f) {       // block % 32 == 0
  static const size_t STACK = 256 << 10;
  gridDim = {grid,1,1}; blockDim = {block,1,1};
  while (sa_fibers.size() < block) { SaFiber fb; fb.stack = (char*)malloc(STACK); fb.done = true; sa_fibers.push_back(fb); }
  sa_call = sa_tramp<F>; sa_arg = (void*)&f;
  for (unsigned b = 0; b < grid; b++) {
    blockIdx = {b,0,0}; sa_block_bar = {0, 0}; for (auto& w : sa_warp_bar) w = {0, 0};
    for (unsigned t = 0; t < block; t++) { SaFiber& fb = sa_fibers[t]; getcontext(&fb.ctx); fb.ctx.uc_stack.ss_sp = fb.stack; fb.ctx.uc_stack.ss_size = STACK;
      fb.ctx.uc_link = &sa_sched; fb.done = false; makecontext(&fb.ctx, sa_fiber_main, 0); }
    unsigned live = block; unsigned long rounds = 0;
    while (live) {
      for (unsigned t = 0; t < block; t++) if (!sa_fibers[t].done) { sa_cur = t; threadIdx = {t,0,0}; swapcontext(&sa_sched, &sa_fibers[t].ctx); if (sa_fibers[t].done) live--; }
      if (++rounds > 100000000ul) { fprintf(stderr, "emulation: a block does not finish\n"); abort(); } } } }
// ---- CUDA runtime names ----
typedef int cudaError_t;  enum { cudaSuccess = 0 };  enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
struct cudaDeviceProp { size_t totalGlobalMem; };
extern size_t sa_fake_global_mem;      // harness: what cudaGetDeviceProperties reports
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
static inline cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
static inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp* p, int) { p->totalGlobalMem = sa_fake_global_mem; return cudaSuccess; }
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaDeviceReset() { return cudaSuccess; }
static inline const char* cudaGetErrorString(cudaError_t) { return ""; }
'''

FAKE_THRUST = r'''#pragma once
// stand-in for the thrust headers src/seed_filter.cu and common/seed_pos_table.cu include (this repository's code)
#include <algorithm>
#include <numeric>
#include <iterator>
#include <vector>
namespace thrust {
template <class T> using device_vector = std::vector<T>;
template <class T> static inline T* raw_pointer_cast(T* p) { return p; }
struct host_policy {}; static host_policy host;
template <class I, class O> static inline O inclusive_scan(I a, I b, O o) { return std::partial_sum(a, b, o); }
template <class I, class O> static inline O inclusive_scan(host_policy, I a, I b, O o) { return std::partial_sum(a, b, o); }
template <class I, class T> static inline I lower_bound(I a, I b, const T& v) { return std::lower_bound(a, b, v); }
template <class I> static inline typename std::iterator_traits<I>::difference_type distance(I a, I b) { return std::distance(a, b); }
template <class I, class C> static inline void stable_sort(I a, I b, C c) { std::stable_sort(a, b, c); }
// unique_copy on the device back ends: an element is kept when it differs from its predecessor IN THE INPUT (hazard H3)
template <class I, class O, class E> static inline O unique_copy(I a, I b, O o, E eq) {
  for (I i = a; i != b; ++i) if (i == a || !eq(*(i - 1), *i)) *o++ = *i;
  return o; }
}
'''

FAKE_TBB_FLOW = r'''#pragma once
// stand-in for tbb/flow_graph.h: the tuple names (std::tuple in oneTBB) and an empty node type
#include <tuple>
#include <cstddef>
namespace tbb { namespace flow { using std::tuple; using std::get; template <class In, class Out> struct multifunction_node { typedef int output_ports_type; }; } }
'''

FAKE_TBB_SORT = r'''#pragma once
// stand-in for tbb/parallel_sort.h as common/seed_pos_table.cu uses it: parallel_for over a blocked_range, run serially
#include <cstddef>
namespace tbb {
template <class T> struct blocked_range { T b, e; blocked_range(T b_, T e_, size_t) : b(b_), e(e_) {} T begin() const { return b; } T end() const { return e; } };
template <class R, class F> void parallel_for(const R& r, F f) { f(r); }
}
'''

HARNESS = r'''
#include <string>
#include "graph.h"
#include "ntcoding.h"
#include "seed_filter.h"
#include "seed_filter_interface.h"
#include "store.h"
// ---- what src/main.cpp owns (this repository's code) ----
size_t sa_fake_global_mem;
Configuration cfg;
DRAM *ref_DRAM, *query_DRAM, *query_rc_DRAM;
DRAM::DRAM() : size(0), seqSize(0), bufferPosition(0) { buffer = nullptr; }   // (common/DRAM.cpp needs TBB's allocator and 6 GB)
DRAM::~DRAM() {}
static FILE* g_out;
static SeedAndFilter_ptr g_real;
static std::vector<segmentPair> record(std::vector<uint64_t> seeds, bool rev, uint32_t buffer) {   // between the seeder and SeedAndFilter
  std::vector<segmentPair> r = g_real(seeds, rev, buffer);
  uint32_t h[6] = {rev ? 1u : 0u, buffer, (uint32_t)seeds.size(), (uint32_t)r.size() - 1u, r[0].len, (uint32_t)r[0].score};
  fwrite(h, 4, 6, g_out); fwrite(r.data() + 1, 16, r.size() - 1, g_out);
  return r;
}
// in : u64 fake_mem ; u32 t_len, t_start, q_len (block), q_start, chunk, transition, strand, step, shape_len, noentropy, n_intervals ; i32 xdrop, hspthresh ;
//      64 x i32 matrix ; shape ; target arena (t_start + t_len bytes) ; query arena (q_start + q_len) ; n_intervals x {start, end}
// out: per interval a marker {0xFFFFFFFF, k, 0, 0, 0, 0}; per call u32 rev, buffer, n_seeds, n_hsps, header.len, header.score + n_hsps x segmentPair
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  uint64_t mem; uint32_t hdr[11]; int par[2]; int mat[64];
  if (fread(&mem, 8, 1, f) != 1 || fread(hdr, 4, 11, f) != 11 || fread(par, 4, 2, f) != 2 || fread(mat, 4, 64, f) != 64) return 2;
  const uint32_t t_len = hdr[0], t_start = hdr[1], q_len = hdr[2], q_start = hdr[3];
  std::string shape(hdr[8], ' ');
  std::vector<char> t(t_start + t_len + 64, 'N'), fw(q_start + q_len + 64, 'N'), rc(q_start + q_len + 64, 'N');
  if (fread(&shape[0], 1, hdr[8], f) != hdr[8] || fread(t.data(), 1, t_start + t_len, f) != t_start + t_len ||
      fread(fw.data(), 1, q_start + q_len, f) != q_start + q_len) return 2;
  std::vector<uint32_t> iv(2 * hdr[10]);
  if (fread(iv.data(), 4, iv.size(), f) != iv.size()) return 2;
  fclose(f);
  sa_fake_global_mem = (size_t)mem;
  cfg.seed.shape = shape; cfg.seed.size = (int)shape.size(); cfg.seed.kmer_size = GenerateShapePos(shape);   // src/main.cpp:178-180
  cfg.seed.transition = hdr[5] != 0; cfg.wga_chunk_size = hdr[4]; cfg.step = hdr[7];
  cfg.strand = hdr[6] == 1 ? "plus" : hdr[6] == 2 ? "minus" : "both";
  cfg.xdrop = par[0]; cfg.hspthresh = par[1]; cfg.noentropy = hdr[9] != 0;
  for (int i = 0; i < 64; i++) cfg.sub_mat[i] = mat[i];
  ref_DRAM = new DRAM; query_DRAM = new DRAM; query_rc_DRAM = new DRAM;
  ref_DRAM->buffer = t.data(); query_DRAM->buffer = fw.data(); query_rc_DRAM->buffer = rc.data();
  RevComp(query_rc_DRAM->buffer, query_DRAM->buffer, q_start, q_start, q_len);   // src/main.cpp:377 (the whole block)
  cfg.num_gpu = g_InitializeInterface(1);                                                                                  // :296
  g_InitializeProcessor(cfg.seed.transition, cfg.wga_chunk_size, cfg.seed.size, cfg.sub_mat, cfg.xdrop, cfg.hspthresh, cfg.noentropy);   // :297
  g_SendRefWriteRequest(ref_DRAM->buffer, t_start, t_len);                                                                  // :615
  GenerateSeedPosTable(ref_DRAM->buffer, t_start, t_len, cfg.step, cfg.seed.size, cfg.seed.kmer_size);                     // :621
  g_SendQueryWriteRequest(q_start, q_len, 0);                                                                               // :661
  g_real = g_SeedAndFilter; g_SeedAndFilter = record;
  g_out = fopen(argv[2], "wb");
  seeder_body body;
  for (uint32_t k = 0; k < hdr[10]; k++) {
    seq_block b; b.r_index = 0; b.q_index = 0; b.r_start = t_start; b.q_start = q_start; b.r_len = t_len; b.q_len = q_len - (uint32_t)cfg.seed.size;   // :708
    seed_interval s; s.start = iv[2 * k]; s.end = iv[2 * k + 1]; s.num_invoked = k + 1; s.num_intervals = hdr[10]; s.buffer = 0;
    uint32_t mark[6] = {0xFFFFFFFFu, k, 0, 0, 0, 0}; fwrite(mark, 4, 6, g_out);
    body(seeder_input(seeder_payload(b, s), (size_t)0));
  }
  fclose(g_out);
  return 0;
}
'''

LAUNCH = re.compile(r"(\w+)\s*<<<\s*([^,>]+?)\s*,\s*([^>]+?)\s*>>>\s*\(([^;]*)\);")


def edited_copy(tmp, rel, h1=False):
    """the reference file with its kernel launches rewritten for the emulation (and the H1 edit), in the temp dir"""
    text = open(os.path.join(REF, rel)).read()
    text, n = LAUNCH.subn(r"launch(\2, \3, [&]{ \1(\4); });", text)
    if h1:
        for a, b in (("short count[4];", "short count[8] = {0};"), ("short count_del[4];", "short count_del[8] = {0};"),
                     ("\n    char r_chr;", "\n    char r_chr = 0;"), ("\n    char q_chr;", "\n    char q_chr = 1;")):
            assert a in text, a
            text = text.replace(a, b)
    path = os.path.join(tmp, os.path.basename(rel).replace(".cu", "_cu.cpp"))
    open(path, "w").write(text)
    return path, n


def mutate(rng, s, rate):
    s = s.copy()
    m = rng.random(s.size) < rate
    s[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
    return s


def design(seed, t_len, q_len, repeat_copies, div=0.06, with_junk=True, mask_every=0):
    """A target of a few '&'-joined records and a query made of diverged pieces of it (so that the path ends in real HSPs on both strands),
    a tandem repeat present in both (heavy buckets: calls that exceed a small MAX_HITS), soft-masked / N / IUPAC stretches."""
    from segalign_amd import synth
    rng = np.random.default_rng(seed)
    t = synth.random_dna(t_len, 9000 + seed).copy()
    unit = synth.random_dna(37, 9100 + seed)
    rep_at = t_len // 5
    for c in range(repeat_copies):
        t[rep_at + 37 * c: rep_at + 37 * (c + 1)] = mutate(rng, unit, 0.02)
    t[t_len // 2] = ord("&")
    t[(3 * t_len) // 4] = ord("&")
    q = synth.random_dna(q_len, 9200 + seed).copy()
    p = 60
    while p + 200 < q_len:
        n = min(int(rng.integers(150, 600)), q_len - p - 20)
        src = int(rng.integers(0, t_len - n))
        piece = mutate(rng, t[src:src + n], div)
        piece[piece == ord("&")] = ord("A")
        if rng.random() < 0.45:                      # a piece of the other strand
            comp = np.zeros(256, np.uint8); comp[:] = ord("N")
            for a, b in zip(b"ACGT", b"TGCA"):
                comp[a] = b
            piece = comp[piece[::-1]]
        q[p:p + n] = piece
        p += n + int(rng.integers(30, 160))
    for c in range(3):                               # three copies of the repeat unit in the query
        at = q_len // 3 + 300 * c
        q[at:at + 37] = mutate(rng, unit, 0.02)
    for at in range(400, q_len - 200, mask_every or q_len):   # 14of22 with transitions makes 15 seed words per position and the reference sizes its
        if mask_every:                                       # seed buffer for 13 (MAX_SEEDS, src/seed_filter.cu:838, asserted at :692): soft-mask
            q[at:at + 180] = np.frombuffer(bytes(q[at:at + 180]).lower(), dtype=np.uint8)   # enough of every chunk to stay below it
    if with_junk:
        q[100:160] = np.frombuffer(bytes(q[100:160]).lower(), dtype=np.uint8)
        t[200:260] = np.frombuffer(bytes(t[200:260]).lower(), dtype=np.uint8)
        q[q_len // 2: q_len // 2 + 12] = ord("N")
        t[t_len // 3] = ord("R")
        q[q_len - 300] = ord("&")
    return t, q


def check_design(shape, trans, strand, chunk, step, t_arena, t_start, t_len, q_arena, q_start, q_len, ivs, mem):
    """The reference's plan reads d_hit_num_vec[g][-1] when the first seed word already holds all the hits of a call (no hits at all included:
    src/seed_filter.cu:733-736, lower_bound == begin), and makes an iteration without seed words when one word has MAX_HITS hits or more: the
    cases must keep clear of both (the oracle's tables say so before the reference text is run)."""
    from oracle import oracle as O
    from segalign_amd import shard
    O.build(with_ref=False)
    k = O.generate_shape_pos(shape)
    index, _ = O.generate_seed_pos_table(t_arena.tobytes(), t_start, t_len, step, len(shape), k)
    counts = np.diff(np.concatenate([[0], index.astype(np.int64)]))
    max_hits = int(np.float32(4194304) * np.float32(mem / 1073741824.0))
    rc_block = O.rev_comp_ascii(q_arena.tobytes(), q_start, q_len)
    ql = q_len - len(shape)
    for (s, e) in ivs:
        for rev in (False, True):
            if not (strand & (2 if rev else 1)):
                continue
            for (a, b) in shard.chunks_of((s, e), chunk, ql, rev):
                seeds = O.make_seeds(rc_block, 0, a, b, len(shape), k, bool(trans)) if rev else O.make_seeds(q_arena.tobytes(), q_start, a, b, len(shape), k, bool(trans))
                if seeds.size == 0:
                    continue
                assert seeds.size <= (13 if trans else 1) * chunk, ("MAX_SEEDS", seeds.size)
                per = counts[(seeds >> np.uint64(32)).astype(np.int64)]
                assert per.sum() > per[0], ("a call whose hits all sit on its first seed word", s, e, rev, a, b, int(per.sum()))
                assert per.max() < max_hits, ("a seed word with MAX_HITS hits", int(per.max()), max_hits)
                # the plan of src/seed_filter.cu:718-745 walked on the scan: the LAST iteration takes whatever the planned ones left, and the
                # buffers hold MAX_HITS records (:878-885) -- heavy buckets against a small MAX_HITS overrun them (heap corruption here)
                scan = np.cumsum(per)
                num_hits = int(scan[-1])
                if num_hits >= max_hits:
                    num_iter, limit, start = num_hits // max_hits + 2, max_hits, 0
                    for i in range(num_iter - 1):
                        pos = int(np.searchsorted(scan, limit, side="left")) - 1
                        assert pos >= 0 and int(scan[pos]) - start <= max_hits and (i == 0 or int(scan[pos]) > start), ("plan", i, pos)
                        start = int(scan[pos])
                        limit = min(start + max_hits, num_hits)
                    assert num_hits - start <= max_hits, ("the last iteration overruns the buffers", num_hits - start, max_hits)


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    from segalign_amd import synth
    shape22 = re.search(r'"([T0]{22})"', open(os.path.join(ROOT, "tests", "test_gpu_edge_cases.py")).read()).group(1)
    tmp = tempfile.mkdtemp(prefix="sa_path_golden_")
    for d in ("thrust/iterator", "tbb"):
        os.makedirs(os.path.join(tmp, d))
    open(os.path.join(tmp, "prelude.h"), "w").write(PRELUDE)
    for h in ("binary_search.h", "device_vector.h", "execution_policy.h", "iterator/constant_iterator.h", "scan.h", "unique.h"):
        open(os.path.join(tmp, "thrust", h), "w").write(FAKE_THRUST)
    open(os.path.join(tmp, "tbb", "flow_graph.h"), "w").write(FAKE_TBB_FLOW)
    open(os.path.join(tmp, "tbb", "parallel_sort.h"), "w").write(FAKE_TBB_SORT)
    open(os.path.join(tmp, "harness.cpp"), "w").write(HARNESS)
    srcs, launches = [os.path.join(tmp, "harness.cpp")], 0
    for rel, h1 in (("src/seed_filter.cu", True), ("common/seed_filter_interface.cu", False), ("common/seed_pos_table.cu", False)):
        p, n = edited_copy(tmp, rel, h1)
        srcs.append(p)
        launches += n
    assert launches == 7, launches   # find_num_hits, find_hits, find_hsps x 2, compress_output, compress_string_rev_comp, compress_string
    srcs += [os.path.join(REF, "src", "seeder.cpp"), os.path.join(REF, "common", "ntcoding.cpp")]
    exe = os.path.join(tmp, "harness")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-w", "-include", os.path.join(tmp, "prelude.h"), "-I", tmp, "-I", os.path.join(REF, "src"),
                           "-I", os.path.join(REF, "common")] + srcs + ["-o", exe])
    cases = []
    GB = 1 << 30
    #        shape  trans strand chunk step  t_len  q_len t_start q_start mem      xdrop hspthresh noentropy copies intervals
    plan = ((S19,     1,    3,   900,  1,   9000,  5000,  0,      0,      GB,      910,  3000,     0,        30,    2700),
            (S19,     1,    3,   1200, 1,   9000,  5000,  0,      0,      1 << 16, 910,  3000,     0,        12,    2400),   # MAX_HITS 256: calls in several iterations
            (S19,     0,    3,   800,  1,   7000,  4000,  150,    90,     1 << 15, 910,  3000,     1,        8,     4000),   # MAX_HITS 128, arenas that start elsewhere, --noentropy
            (S19,     1,    1,   700,  2,   8000,  4200,  0,      0,      1 << 17, 500,  2200,     0,        20,    2100),   # step 2, plus strand, other thresholds
            (shape22, 1,    2,   1000, 1,   8000,  4000,  0,      0,      1 << 16, 910,  3000,     0,        12,    4000),   # 14of22, minus strand
            (S19,     1,    3,   5000, 1,   60000, 30000, 0,      0,      1 << 22, 910,  3000,     0,        25,    15000),  # six-chunk intervals, 65 k seed words per call, MAX_HITS 16384
            (S19,     1,    3,   25000, 1,  300000, 150000, 0,    0,      1 << 22, 910,  3000,     0,        60,    75000))  # 325 k seed words and ~50 k hits per call against MAX_HITS 16384
    only = [int(x) for x in os.environ.get("SA_PATH_ONLY", "").split(",") if x]   # regenerate these cases only, keep the others of the committed file
    if only:
        cases = json.load(open(OUT))["cases"]
    for ci, (shape, trans, strand, chunk, step, t_len, q_len, t_start, q_start, mem, xdrop, hspthresh, noentropy, copies, ivlen) in enumerate(plan):
        if only and ci not in only:
            continue
        ql = q_len - len(shape)
        ivs = [(s, min(s + ivlen, ql)) for s in range(0, ql, ivlen)]
        mat = hoxd70(xdrop)
        for attempt in range(40):   # the first design whose every call the reference's plan can run (check_design)
            t, q = design(100 * ci + attempt, t_len, q_len, copies, 0.06 if trans else 0.03, mask_every=1000 if len(shape) == 22 and trans else 0)
            t_arena = np.concatenate([synth.random_dna(t_start, 77), t]) if t_start else t
            q_arena = np.concatenate([synth.random_dna(q_start, 78), q]) if q_start else q
            try:
                check_design(shape, trans, strand, chunk, step, t_arena, t_start, t_len, q_arena, q_start, q_len, ivs, mem)
                break
            except AssertionError as e:
                print("case %d design %d: %s" % (ci, attempt, e.args[0][0]), flush=True)
        else:
            raise SystemExit("no design for case %d" % ci)
        inp, outp = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(inp, "wb") as f:
            f.write(struct.pack("<Q11I2i", mem, t_len, t_start, q_len, q_start, chunk, trans, strand, step, len(shape), noentropy, len(ivs), xdrop, hspthresh))
            f.write(mat.astype("<i4").tobytes())
            f.write(shape.encode())
            f.write(t_arena.tobytes())
            f.write(q_arena.tobytes())
            for a, b in ivs:
                f.write(struct.pack("<2I", a, b))
        subprocess.check_call([exe, inp, outp], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        raw = open(outp, "rb").read()
        off, calls, k = 0, [], -1
        while off < len(raw):
            h = struct.unpack_from("<6I", raw, off)
            off += 24
            if h[0] == 0xFFFFFFFF:
                k = h[1]
                continue
            assert h[3] == h[4], h                      # the header's len is the number of HSPs that follow
            calls.append(dict(interval=k, rev=h[0], buffer=h[1], n_seeds=h[2], n_hsps=h[3], num_hits=h[5], hsps=pack_rows(raw[off:off + 16 * h[3]])))
            off += 16 * h[3]
        max_hits = int(np.float32(4194304) * np.float32(mem / 1073741824.0))
        print("case %d: %d calls, %d HSPs, %d seed hits, MAX_HITS %d, calls above it: %d" %
              (ci, len(calls), sum(c["n_hsps"] for c in calls), sum(c["num_hits"] for c in calls), max_hits, sum(c["num_hits"] >= max_hits for c in calls)), flush=True)
        case = dict(shape=shape, transition=trans, strand=strand, chunk=chunk, step=step, t_start=t_start, q_start=q_start, t_len=t_len, q_len=q_len,
                    total_global_mem=mem, max_hits=max_hits, xdrop=xdrop, hspthresh=hspthresh, noentropy=noentropy, sub_mat=mat.tolist(),
                    target_arena=t_arena.tobytes().decode("ascii"), query_arena=q_arena.tobytes().decode("ascii"), intervals=ivs, calls=calls)
        if only and ci < len(cases):
            cases[ci] = case
        else:
            cases.append(case)
    json.dump(dict(note="every g_SeedAndFilter return of the reference's own files run end to end (tests/golden/make_path_golden.py: src/seed_filter.cu, "
                        "common/seed_filter_interface.cu, common/seed_pos_table.cu, common/ntcoding.cpp, src/seeder.cpp; CUDA runtime / thrust / TBB stood in for, "
                        "kernels under SIMT emulation).  Arenas are the DRAM buffers from 0 (t_start / q_start bases of another block in front); the query block "
                        "is q_len bases, the seeder is handed q_len - seed size; per call: interval, strand, seed words handed over, seed hits (header.score), HSPs "
                        "(rows of ref_start, query_start, len, score as u32 x 3 + i32; zlib + base64) in the reference's order.", cases=cases), open(OUT, "w"))
    print("wrote %s: %d cases" % (OUT, len(cases)))


if __name__ == "__main__":
    main()
