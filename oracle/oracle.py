"""ctypes door onto oracle/liboracle.so (the CPU restatement) and oracle/_ref/libntcoding_ref.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under segalign_amd/ imports this module (tests/test_no_oracle_in_product.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libntcoding_ref.so")

SEG_DTYPE = np.dtype([("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4")])


def build(with_ref=True):
    """Compile the checker: liboracle.so always; _ref only where /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if with_ref and os.path.isdir("/root/reference/common"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


class _ExtendParams(C.Structure):
    _fields_ = [("ref", C.c_void_p), ("query", C.c_void_p), ("ref_len", C.c_uint32), ("query_len", C.c_uint32),
                ("sub_mat", C.c_void_p), ("xdrop", C.c_int), ("hspthresh", C.c_int), ("noentropy", C.c_int),
                ("log4_is_float", C.c_int), ("entropy_ulps", C.c_int)]


class _SafStats(C.Structure):
    _fields_ = [("num_hits", C.c_uint64), ("num_survivors", C.c_uint64), ("num_examined", C.c_uint64),
                ("num_iter", C.c_uint32)]


class _SafParams(C.Structure):
    _fields_ = [("ext", _ExtendParams), ("index_table", C.c_void_p), ("pos_table", C.c_void_p),
                ("seed_size", C.c_uint32), ("max_hits", C.c_int64), ("num_threads", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build(with_ref=False)
        L = C.CDLL(_LIB)
        L.orc_kmer_index_at_pos.restype = C.c_uint32
        L.orc_kmer_index_at_pos.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
        L.orc_generate_shape_pos.argtypes = [C.c_char_p]
        L.orc_generate_seed_pos_table.restype = C.c_uint32
        L.orc_generate_seed_pos_table.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                  C.c_void_p, C.c_void_p]
        L.orc_make_seeds.restype = C.c_size_t
        L.orc_make_seeds.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                     C.c_void_p]
        L.orc_extend_hit.argtypes = [C.POINTER(_ExtendParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_extend_hit_tiled.argtypes = [C.POINTER(_ExtendParams), C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_seed_and_filter.restype = C.c_size_t
        L.orc_seed_and_filter.argtypes = [C.POINTER(_SafParams), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p),
                                          C.POINTER(_SafStats)]
        L.orc_seed_and_filter_rm.restype = C.c_size_t
        L.orc_seed_and_filter_rm.argtypes = [C.POINTER(_SafParams), C.c_void_p, C.c_size_t, C.c_int, C.c_uint32,
                                             C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(_SafStats)]
        L.orc_seed_and_filter_traced.restype = C.c_size_t
        L.orc_seed_and_filter_traced.argtypes = [C.POINTER(_SafParams), C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                                 C.POINTER(C.c_void_p), C.POINTER(_SafStats), C.c_void_p]
        L.orc_stage_trace_free.argtypes = [C.c_void_p]
        L.orc_order_hsps.restype = C.c_size_t
        L.orc_order_hsps.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_rm_coverage_intervals.restype = C.c_size_t
        L.orc_rm_coverage_intervals.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.orc_rm_plan.restype = C.c_size_t
        L.orc_rm_plan.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.POINTER(C.c_void_p)]
        L.orc_max_hits_for_mem.argtypes = [C.c_uint64]
        L.orc_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_encode_rev_comp.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_rev_comp_codes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_rev_comp_ascii.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
        L.orc_build_sub_mat.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def ref_lib():
    """The real reference ntcoding object, or None when it was never built (e.g. no /root/reference)."""
    if not os.path.exists(_REF):
        return None
    R = C.CDLL(_REF)
    R.ref_GetKmerIndexAtPos.restype = C.c_uint32
    R.ref_GetKmerIndexAtPos.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    R.ref_GenerateShapePos.argtypes = [C.c_char_p]
    R.ref_RevComp.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t]
    return R


# ---------------------------------------------------------------------------------------------------------
def build_sub_mat(xdrop=910, ambiguous="x", reward=0, penalty=0):
    m = np.zeros(64, dtype=np.int32)
    mode = {"x": 0, "n": 1, "iupac": 2}[ambiguous]
    lib().orc_build_sub_mat(m.ctypes.data, xdrop, mode, reward, penalty)
    return m


def encode(ascii_bytes):
    out = np.empty(len(ascii_bytes), dtype=np.uint8)
    lib().orc_encode(bytes(ascii_bytes), len(ascii_bytes), out.ctypes.data)
    return out


def encode_rev_comp(ascii_bytes):
    n = len(ascii_bytes)
    f = np.empty(n, dtype=np.uint8)
    r = np.empty(n, dtype=np.uint8)
    lib().orc_encode_rev_comp(bytes(ascii_bytes), n, f.ctypes.data, r.ctypes.data)
    return f, r


def rev_comp_codes(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    out = np.empty_like(codes)
    lib().orc_rev_comp_codes(codes.ctypes.data, codes.size, out.ctypes.data)
    return out


def rev_comp_ascii(src, start, length):
    dst = C.create_string_buffer(length)
    lib().orc_rev_comp_ascii(dst, bytes(src), 0, start, length)
    return dst.raw


def generate_shape_pos(shape):
    return lib().orc_generate_shape_pos(shape.encode())


def is_transition_at_pos(t):
    return lib().orc_is_transition_at_pos(t)


def kmer_index_at_pos(seq, pos, seed_size):
    return lib().orc_kmer_index_at_pos(bytes(seq), pos, seed_size)


def generate_seed_pos_table(ref, start_addr, ref_length, step, shape_size, kmer_size):
    nkeys = 1 << (2 * kmer_size)
    index = np.empty(nkeys, dtype=np.uint32)
    pos = np.empty(max(int(ref_length), 1), dtype=np.uint32)
    n = lib().orc_generate_seed_pos_table(bytes(ref), start_addr, ref_length, step, shape_size, kmer_size,
                                          index.ctypes.data, pos.ctypes.data)
    return index, pos[:n].copy()


def make_seeds(qbuf, q_block_start, i, e, seed_size, kmer_size, transition):
    out = np.empty(max((e - i) * (kmer_size + 1 if transition else 1), 1), dtype=np.uint64)
    n = lib().orc_make_seeds(bytes(qbuf), q_block_start, i, e, seed_size, kmer_size, int(transition), out.ctypes.data)
    return out[:n].copy()


def _ext_params(ref_codes, query_codes, sub_mat, xdrop, hspthresh, noentropy, log4_is_float=True, entropy_ulps=0):
    p = _ExtendParams()
    p.ref = ref_codes.ctypes.data
    p.query = query_codes.ctypes.data
    p.ref_len = ref_codes.size
    p.query_len = query_codes.size
    p.sub_mat = sub_mat.ctypes.data
    p.xdrop, p.hspthresh, p.noentropy, p.log4_is_float = xdrop, hspthresh, int(noentropy), int(log4_is_float)
    p.entropy_ulps = int(entropy_ulps)
    return p


def extend_hit(ref_codes, query_codes, sub_mat, ref_loc, query_loc, xdrop=910, hspthresh=3000, noentropy=False,
               tiled=0, log4_is_float=True):
    """Returns (passed, (ref_start, query_start, len, score), examined)."""
    ref_codes = np.ascontiguousarray(ref_codes, np.uint8)
    query_codes = np.ascontiguousarray(query_codes, np.uint8)
    sub_mat = np.ascontiguousarray(sub_mat, np.int32)
    p = _ext_params(ref_codes, query_codes, sub_mat, xdrop, hspthresh, noentropy, log4_is_float)
    out = np.zeros(1, dtype=SEG_DTYPE)
    ex = C.c_uint64(0)
    if tiled:
        ok = lib().orc_extend_hit_tiled(C.byref(p), ref_loc, query_loc, tiled, out.ctypes.data)
    else:
        ok = lib().orc_extend_hit(C.byref(p), ref_loc, query_loc, out.ctypes.data, C.addressof(ex))
    return bool(ok), tuple(int(x) for x in out[0]), ex.value


def extend_hits_pass(ref_codes, query_codes, sub_mat, pairs, xdrop=910, hspthresh=3000, noentropy=False, log4_is_float=True, entropy_ulps=0):
    """orc_extend_hit over many anchors (pairs[:, 0] = ref_loc, pairs[:, 1] = query_loc): bool array 'the hit passes' and the
    records.  One parameter block for the whole batch (the per-call wrapper above rebuilds it every time)."""
    ref_codes = np.ascontiguousarray(ref_codes, np.uint8)
    query_codes = np.ascontiguousarray(query_codes, np.uint8)
    sub_mat = np.ascontiguousarray(sub_mat, np.int32)
    p = _ext_params(ref_codes, query_codes, sub_mat, xdrop, hspthresh, noentropy, log4_is_float, entropy_ulps)
    pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
    ok = np.zeros(pairs.shape[0], dtype=bool)
    out = np.zeros(pairs.shape[0], dtype=SEG_DTYPE)
    ex = C.c_uint64(0)
    f = lib().orc_extend_hit
    base, step = out.ctypes.data, SEG_DTYPE.itemsize
    for i in range(pairs.shape[0]):
        ok[i] = bool(f(C.byref(p), int(pairs[i, 0]), int(pairs[i, 1]), base + i * step, C.addressof(ex)))
    return ok, out


def seed_and_filter(ref_codes, query_codes, index_table, pos_table, seeds, sub_mat, seed_size=19, xdrop=910,
                    hspthresh=3000, noentropy=False, max_hits=1 << 30, num_threads=0, log4_is_float=True,
                    rm=None):
    """Full SeedAndFilter restatement.  rm=None -> src/ variant; rm=(rev, ref_start, ref_end) -> repeat masker.
    Returns (segments structured array incl. header element 0, stats dict)."""
    ref_codes = np.ascontiguousarray(ref_codes, np.uint8)
    query_codes = np.ascontiguousarray(query_codes, np.uint8)
    index_table = np.ascontiguousarray(index_table, np.uint32)
    pos_table = np.ascontiguousarray(pos_table, np.uint32)
    seeds = np.ascontiguousarray(seeds, np.uint64)
    sub_mat = np.ascontiguousarray(sub_mat, np.int32)
    p = _SafParams()
    p.ext = _ext_params(ref_codes, query_codes, sub_mat, xdrop, hspthresh, noentropy, log4_is_float)
    p.index_table = index_table.ctypes.data
    p.pos_table = pos_table.ctypes.data if pos_table.size else None
    p.seed_size = seed_size
    p.max_hits = max_hits
    p.num_threads = num_threads if num_threads > 0 else (os.cpu_count() or 1)
    out = C.c_void_p()
    st = _SafStats()
    if rm is None:
        n = lib().orc_seed_and_filter(C.byref(p), seeds.ctypes.data, seeds.size, C.byref(out), C.byref(st))
    else:
        rev, rs, re_ = rm
        n = lib().orc_seed_and_filter_rm(C.byref(p), seeds.ctypes.data, seeds.size, int(rev), rs, re_, C.byref(out),
                                         C.byref(st))
    buf = (C.c_char * (n * SEG_DTYPE.itemsize)).from_address(out.value)
    segs = np.frombuffer(buf, dtype=SEG_DTYPE).copy()
    lib().orc_free(out)
    stats = dict(num_hits=st.num_hits, num_survivors=st.num_survivors, num_examined=st.num_examined,
                 num_iter=st.num_iter)
    return segs, stats


class _StageTrace(C.Structure):
    _fields_ = [("hits", C.c_void_p), ("ext", C.c_void_p), ("done", C.c_void_p), ("reduced", C.c_void_p),
                ("n_hits", C.c_size_t), ("n_ext", C.c_size_t), ("n_reduced", C.c_size_t),
                ("cap_hits", C.c_size_t), ("cap_ext", C.c_size_t), ("cap_reduced", C.c_size_t)]


def seed_and_filter_traced(ref_codes, query_codes, index_table, pos_table, seeds, sub_mat, seed_size=19, xdrop=910, hspthresh=3000,
                           noentropy=False, max_hits=1 << 30, rm=None):
    """seed_and_filter with its intermediate lists (orc_seed_and_filter_traced) -> (segments, dict(hits, ext, done, reduced)):
    the hit list as find_hits leaves it, records + done flags as find_hsps leaves them, the list compress_output writes."""
    ref_codes = np.ascontiguousarray(ref_codes, np.uint8)
    query_codes = np.ascontiguousarray(query_codes, np.uint8)
    index_table = np.ascontiguousarray(index_table, np.uint32)
    pos_table = np.ascontiguousarray(pos_table, np.uint32)
    seeds = np.ascontiguousarray(seeds, np.uint64)
    sub_mat = np.ascontiguousarray(sub_mat, np.int32)
    p = _SafParams()
    p.ext = _ext_params(ref_codes, query_codes, sub_mat, xdrop, hspthresh, noentropy, True)
    p.index_table = index_table.ctypes.data
    p.pos_table = pos_table.ctypes.data if pos_table.size else None
    p.seed_size = seed_size
    p.max_hits = max_hits
    p.num_threads = 1
    out = C.c_void_p()
    st = _SafStats()
    tr = _StageTrace()
    rev, rs, re_ = rm if rm is not None else (0, 0, 0)
    n = lib().orc_seed_and_filter_traced(C.byref(p), seeds.ctypes.data, seeds.size, int(rm is not None), int(rev), rs, re_, C.byref(out),
                                         C.byref(st), C.byref(tr))
    segs = np.frombuffer((C.c_char * (n * SEG_DTYPE.itemsize)).from_address(out.value), dtype=SEG_DTYPE).copy()
    lib().orc_free(out)

    def arr(ptr, k, dt):
        return np.frombuffer((C.c_char * (k * dt.itemsize)).from_address(ptr), dtype=dt).copy() if k else np.zeros(0, dtype=dt)
    res = dict(hits=arr(tr.hits, tr.n_hits, SEG_DTYPE), ext=arr(tr.ext, tr.n_ext, SEG_DTYPE),
               done=arr(tr.done, tr.n_ext, np.dtype(np.uint8)), reduced=arr(tr.reduced, tr.n_reduced, SEG_DTYPE))
    lib().orc_stage_trace_free(C.byref(tr))
    return segs, res


def order_hsps(records, rm=False):
    """the ordering chain alone (orc_order_hsps): stable sort -> adjacent-pair unique -> stable sort on one dedup scope"""
    h = np.ascontiguousarray(records, dtype=SEG_DTYPE)
    out = C.c_void_p()
    n = lib().orc_order_hsps(h.ctypes.data if h.size else None, h.size, int(bool(rm)), C.byref(out))
    return _take(n, out, SEG_DTYPE)


IVL_DTYPE = np.dtype([("query_start", "<u4"), ("len", "<u4")])
RM_TASK_DTYPE = np.dtype([("block_index", "<u4"), ("_pad", "<u4"), ("block_start", "<u8"), ("block_len", "<u4"),
                          ("start", "<u4"), ("end", "<u4"), ("ref_start", "<u4"), ("ref_end", "<u4"), ("_pad2", "<u4")])


def _take(n, out, dtype):
    if n == 0:
        lib().orc_free(out)
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(out.value)
    a = np.frombuffer(buf, dtype=dtype).copy()
    lib().orc_free(out)
    return a


def rm_coverage_intervals(hsps, block_len, M=1):
    """repeat_masker_src/seeder.cpp:153-188 (uint8_t counters, runs with count >= M)."""
    h = np.ascontiguousarray(hsps, dtype=SEG_DTYPE)
    out = C.c_void_p()
    n = lib().orc_rm_coverage_intervals(h.ctypes.data if h.size else None, h.size, block_len, M, C.byref(out))
    return _take(n, out, IVL_DTYPE)


def rm_plan(seq_len, seq_block_size=1000000000, lastz_interval_size=10000000, prop_neigh_interval=0.2, seed_size=19):
    """repeat_masker_src/main.cpp:316-436: one record per (block, interval) task."""
    out = C.c_void_p()
    n = lib().orc_rm_plan(seq_len, seq_block_size, lastz_interval_size, prop_neigh_interval, seed_size, C.byref(out))
    return _take(n, out, RM_TASK_DTYPE)


def max_hits_for_mem(total_global_mem):
    return lib().orc_max_hits_for_mem(total_global_mem)
