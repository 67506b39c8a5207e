"""ctypes binding of libsegalign_hip.so with the reference's own entry-point names.

This is plumbing for tests and bench.py: every function forwards 1:1 to the C-ABI of include/segalign_amd.h
(which in turn replaces, symbol by symbol, the engine boundary of gsneha26/SegAlign -- see INTEGRATION.md).
There is NO fallback: if the HIP library is missing or no GPU is present, calls fail loudly.

Reference symbol                       -> here
  g_InitializeInterface(num_gpu)        -> InitializeInterface        (common/seed_filter_interface.cu:49-80)
  g_InitializeProcessor(...)            -> InitializeProcessor        (src/seed_filter.cu:830-897)
  g_SendRefWriteRequest(seq,addr,len)   -> SendRefWriteRequest        (common/seed_filter_interface.cu:82-101)
  GenerateShapePos(shape)               -> GenerateShapePos           (common/ntcoding.cpp:21-37)
  GenerateSeedPosTable(...)             -> GenerateSeedPosTable       (common/seed_pos_table.cu:49-109)
  g_SendQueryWriteRequest(addr,len,buf) -> SendQueryWriteRequest      (src/seed_filter.cu:899-919)
  g_SeedAndFilter(seeds,rev,buf)        -> SeedAndFilter              (src/seed_filter.cu:682-828)
  g_ClearRef / g_ClearQuery / g_ShutdownProcessor
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsegalign_hip.so")

SEG_DTYPE = np.dtype([("ref_start", "<u4"), ("query_start", "<u4"), ("len", "<u4"), ("score", "<i4")])

# every symbol include/segalign_amd.h declares (tests check the library exports all of them)
C_ABI_SYMBOLS = [
    "sa_select_devices", "sa_initialize_interface", "sa_initialize_processor", "sa_shutdown_processor", "sa_send_ref_write_request",
    "sa_clear_ref", "sa_generate_shape_pos", "sa_generate_seed_pos_table", "sa_send_query_write_request",
    "sa_clear_query", "sa_seed_and_filter", "sa_seed_and_filter_range", "sa_free_segments",
    "sa_rm_send_query_write_request", "sa_rm_clear_query", "sa_rm_seed_and_filter", "sa_set_max_hits",
    "sa_get_max_hits", "sa_max_hits_for_mem", "sa_get_last_call_stats", "sa_set_count_examined",
    "sa_profile_enable", "sa_profile_reset", "sa_profile_num_entries", "sa_profile_get", "sa_profile_busy_ms", "sa_get_ref_len",
    "sa_get_num_index", "sa_get_index_table_size", "sa_copy_ref_codes", "sa_copy_index_table", "sa_copy_pos_table",
    "sa_copy_query_codes", "sa_get_query_len", "sa_device_make_seeds", "sa_version",
    "sa_rm_mask_interval", "sa_rm_coverage_intervals", "sa_free_intervals", "sa_get_filter_mode",
    "sa_seed_interval", "sa_seed_and_filter_chunks", "sa_max_chunks_per_call", "sa_get_chunks_per_call", "sa_extend_hits", "sa_order_hsps",
    "sa_get_lookup_mode", "sa_get_neighbourhood_entries",
    "sa_seed_calls", "sa_count_call_hits", "sa_count_chunk_hits", "sa_get_wga_chunk", "sa_release_arena", "sa_set_option", "sa_reset_option", "sa_get_option", "sa_option_count", "sa_option_name", "sa_get_audit",
]
IVL_DTYPE = np.dtype([("query_start", "<u4"), ("len", "<u4")])  # struct Segment, repeat_masker_src/graph.h:32-35
STRAND_PLUS, STRAND_MINUS, STRAND_BOTH = 1, 2, 3
PATH_LIST_REGROWN, PATH_DEDUP_FALLBACK, PATH_CHAIN_BUCKET_OVERFLOW, PATH_CHAIN_SLICED, PATH_HEAD_BITS_REGROWN, PATH_GENERAL_FALLBACK = 1, 2, 4, 8, 16, 32
PATH_KEY_ORDERED = 64


class CallStats(C.Structure):
    _fields_ = [("num_seeds", C.c_uint64), ("num_hits", C.c_uint64), ("num_survivors", C.c_uint64),
                ("num_anchors", C.c_uint64), ("num_examined", C.c_uint64), ("num_examined_filter", C.c_uint64),
                ("num_candidates", C.c_uint64), ("num_entropy", C.c_uint64), ("num_iter", C.c_uint32),
                ("device", C.c_int), ("lookup_path", C.c_int), ("path_flags", C.c_uint32), ("num_forwarded", C.c_uint64)]


_lib = None


def lib():
    """Load the HIP library.  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsegalign_hip.so is missing (%s). Build it with `python -m segalign_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.sa_initialize_interface.restype = C.c_int
    L.sa_initialize_interface.argtypes = [C.c_int]
    L.sa_select_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.sa_initialize_processor.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.sa_send_ref_write_request.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
    L.sa_generate_shape_pos.restype = C.c_int
    L.sa_generate_shape_pos.argtypes = [C.c_char_p]
    L.sa_generate_seed_pos_table.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.sa_send_query_write_request.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    L.sa_clear_query.argtypes = [C.c_uint32]
    L.sa_seed_and_filter.restype = C.c_size_t
    L.sa_seed_and_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.sa_seed_and_filter_range.restype = C.c_size_t
    L.sa_seed_and_filter_range.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.sa_free_segments.argtypes = [C.c_void_p]
    L.sa_seed_and_filter_chunks.restype = C.c_size_t
    L.sa_seed_and_filter_chunks.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.sa_seed_interval.restype = C.c_size_t
    L.sa_seed_interval.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(CallStats)]
    L.sa_rm_seed_and_filter.restype = C.c_size_t
    L.sa_rm_seed_and_filter.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.sa_rm_mask_interval.restype = C.c_size_t
    L.sa_rm_mask_interval.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p),
                                      C.c_void_p]
    L.sa_rm_coverage_intervals.restype = C.c_size_t
    L.sa_rm_coverage_intervals.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.sa_free_intervals.argtypes = [C.c_void_p]
    L.sa_set_max_hits.argtypes = [C.c_int64]
    L.sa_get_max_hits.restype = C.c_int64
    L.sa_max_hits_for_mem.restype = C.c_int
    L.sa_max_hits_for_mem.argtypes = [C.c_uint64]
    L.sa_get_last_call_stats.argtypes = [C.POINTER(CallStats)]
    L.sa_set_count_examined.argtypes = [C.c_int]
    L.sa_profile_enable.argtypes = [C.c_int]
    L.sa_profile_num_entries.restype = C.c_int
    L.sa_profile_get.restype = C.c_int
    L.sa_profile_get.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sa_profile_busy_ms.restype = C.c_double
    L.sa_profile_busy_ms.argtypes = [C.c_char_p]
    for f in ("sa_get_ref_len", "sa_get_num_index", "sa_get_index_table_size"):
        getattr(L, f).restype = C.c_uint32
    L.sa_get_query_len.restype = C.c_uint32
    L.sa_get_query_len.argtypes = [C.c_uint32]
    L.sa_copy_ref_codes.argtypes = [C.c_int, C.c_void_p]
    L.sa_copy_index_table.argtypes = [C.c_int, C.c_void_p]
    L.sa_copy_pos_table.argtypes = [C.c_int, C.c_void_p]
    L.sa_copy_query_codes.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    L.sa_device_make_seeds.restype = C.c_size_t
    L.sa_device_make_seeds.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_size_t]
    L.sa_order_hsps.restype = C.c_size_t
    L.sa_order_hsps.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.sa_extend_hits.restype = C.c_size_t
    L.sa_extend_hits.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.sa_get_neighbourhood_entries.restype = C.c_uint64
    L.sa_version.restype = C.c_char_p
    L.sa_seed_calls.restype = C.c_size_t
    L.sa_seed_calls.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(CallStats)]
    L.sa_count_call_hits.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p]
    L.sa_count_chunk_hits.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.sa_get_wga_chunk.restype = C.c_uint32
    L.sa_set_option.restype = C.c_int
    L.sa_set_option.argtypes = [C.c_char_p, C.c_int64]
    L.sa_reset_option.restype = C.c_int
    L.sa_reset_option.argtypes = [C.c_char_p]
    L.sa_get_option.restype = C.c_int64
    L.sa_get_option.argtypes = [C.c_char_p]
    L.sa_option_count.restype = C.c_int
    L.sa_option_name.restype = C.c_char_p
    L.sa_option_name.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.sa_get_audit.restype = C.c_size_t
    L.sa_get_audit.argtypes = [C.c_void_p, C.c_size_t]
    _lib = L
    return L


# ---- the reference surface --------------------------------------------------------------------------------------
def select_devices(ids):
    """One process per GPU: make the next InitializeInterface use exactly these HIP device ordinals."""
    arr = (C.c_int * len(ids))(*ids)
    lib().sa_select_devices(arr, len(ids))


def InitializeInterface(num_gpu=-1):
    return lib().sa_initialize_interface(num_gpu)


def InitializeProcessor(transition, wga_chunk, seed_size, sub_mat, xdrop, hspthresh, noentropy):
    m = np.ascontiguousarray(sub_mat, dtype=np.int32)
    assert m.size == 64
    lib().sa_initialize_processor(int(bool(transition)), wga_chunk, seed_size, m.ctypes.data, xdrop, hspthresh,
                                  int(bool(noentropy)))


def ShutdownProcessor():
    lib().sa_shutdown_processor()


def _as_u8(buf):
    """ASCII host buffer -> contiguous uint8 ndarray (kept alive by the caller for the duration of the call)."""
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf, dtype=np.uint8)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def SendRefWriteRequest(seq, addr, length):
    a = _as_u8(seq)
    lib().sa_send_ref_write_request(a.ctypes.data, addr, length)
    return a


def ClearRef():
    lib().sa_clear_ref()


def GenerateShapePos(shape):
    return lib().sa_generate_shape_pos(shape.encode())


def GenerateSeedPosTable(ref_str, start_addr, ref_length, step, shape_size, kmer_size):
    a = _as_u8(ref_str)
    lib().sa_generate_seed_pos_table(a.ctypes.data, start_addr, ref_length, step, shape_size, kmer_size)


def SendQueryWriteRequest(query_buffer, addr, length, buffer):
    a = _as_u8(query_buffer)
    lib().sa_send_query_write_request(a.ctypes.data, addr, length, buffer)


def ClearQuery(buffer):
    lib().sa_clear_query(buffer)


def _take(n, out):
    if n == 0 or not out.value:
        return np.zeros(0, dtype=SEG_DTYPE)
    buf = (C.c_char * (n * SEG_DTYPE.itemsize)).from_address(out.value)
    segs = np.frombuffer(buf, dtype=SEG_DTYPE).copy()
    lib().sa_free_segments(out)
    return segs


def SeedAndFilter(seed_offset_vector, rev, buffer):
    """-> structured array; element 0 is the header {len = #anchors, score = #hits} (seed_filter.cu:806-809)."""
    s = np.ascontiguousarray(seed_offset_vector, dtype=np.uint64)
    out = C.c_void_p()
    n = lib().sa_seed_and_filter(s.ctypes.data, s.size, int(bool(rev)), buffer, C.byref(out))
    return _take(n, out)


def SeedAndFilterRange(start, end, rev, buffer):
    """Additive entry: device-side seeding of query positions [start,end). Empty array when the
    chunk holds no valid seed (the reference would not call the engine then, seeder.cpp:76)."""
    out = C.c_void_p()
    n = lib().sa_seed_and_filter_range(start, end, int(bool(rev)), buffer, C.byref(out))
    return _take(n, out)


def SeedAndFilterChunks(start, end, rev, buffer):
    """Up to sa_max_chunks_per_call() consecutive chunks of one strand in one pass; returns one vector per chunk, each
    identical to SeedAndFilterRange of that chunk (empty array for a chunk without seeds)."""
    k = lib().sa_max_chunks_per_call()
    outs = (C.c_void_p * k)()
    counts = (C.c_size_t * k)()
    lib().sa_seed_and_filter_chunks(start, end, int(bool(rev)), buffer, outs, counts)
    res = []
    for c in range(k):
        res.append(_take(counts[c], C.c_void_p(outs[c])) if outs[c] else np.zeros(0, dtype=SEG_DTYPE))
    return res


class CallDesc(C.Structure):
    _fields_ = [("start", C.c_uint32), ("end", C.c_uint32), ("rev", C.c_int)]


class CallResult(C.Structure):
    _fields_ = [("hsps", C.c_void_p), ("num_hsps", C.c_size_t), ("num_hits", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32)]


def CountCallHits(calls, buffer=0, threads=4, per_chunk=False):
    """Seed hits of every call [(start, end, rev), ...], lookup only (no filtering, no extension): the weights a multi-GPU host
    deals the calls of a pass by.  per_chunk: the hits of every wga_chunk piece of every call instead, concatenated."""
    n = len(calls)
    descs = (CallDesc * max(n, 1))(*[CallDesc(int(a), int(b), int(bool(r))) for (a, b, r) in calls])
    hits = (C.c_uint64 * max(n, 1))()
    if not per_chunk:
        lib().sa_count_call_hits(descs, n, buffer, threads, hits)
        return [int(hits[i]) for i in range(n)]
    k = lib().sa_max_chunks_per_call()
    ch = (C.c_uint64 * max(n * k, 1))()
    lib().sa_count_chunk_hits(descs, n, buffer, threads, hits, ch)
    chunk = lib().sa_get_wga_chunk()
    out = []
    for i, (a, b, r) in enumerate(calls):
        kk = (int(b) - int(a) + chunk - 1) // chunk if b > a else 0
        out.extend(int(ch[i * k + c]) for c in range(kk))
    return out


def ReleaseArena():
    lib().sa_release_arena()


def SeedCalls(calls, buffer=0, threads=4, hits_out=None, devices_out=None):
    """calls: [(start, end, rev), ...], each up to sa_max_chunks_per_call() chunks of one strand; `threads` of them in flight on
    the engine's worker pool.  -> ([HSP array per call (chunks concatenated, headers removed)], summed stats dict).
    hits_out: a list that receives the seed hits of every call (sa_call_result.num_hits)."""
    n = len(calls)
    descs = (CallDesc * max(n, 1))(*[CallDesc(int(a), int(b), int(bool(r))) for (a, b, r) in calls])
    res = (CallResult * max(n, 1))()
    st = CallStats()
    lib().sa_seed_calls(descs, n, buffer, threads, res, C.byref(st))
    outs = []
    for i in range(n):
        if res[i].num_hsps:
            buf = (C.c_char * (res[i].num_hsps * SEG_DTYPE.itemsize)).from_address(res[i].hsps)
            outs.append(np.frombuffer(buf, dtype=SEG_DTYPE).copy())
        else:
            outs.append(np.zeros(0, dtype=SEG_DTYPE))
        lib().sa_free_segments(res[i].hsps)
    if hits_out is not None:
        hits_out.extend(int(res[i].num_hits) for i in range(n))
    if devices_out is not None:
        devices_out.extend(int(res[i].device) for i in range(n))
    return outs, {k: getattr(st, k) for k, _ in CallStats._fields_}


def OrderHsps(records, rm=False, path=1):
    """The ordering stage alone on `records` (SEG_DTYPE) as one dedup scope: sort -> adjacent-pair unique -> sort (sa_order_hsps)."""
    h = np.ascontiguousarray(records, dtype=SEG_DTYPE)
    out = C.c_void_p()
    n = lib().sa_order_hsps(h.ctypes.data if h.size else None, h.size, int(bool(rm)), int(path), C.byref(out))
    if not out.value:
        return np.zeros(0, dtype=SEG_DTYPE)
    res = np.frombuffer((C.c_char * (n * SEG_DTYPE.itemsize)).from_address(out.value), dtype=SEG_DTYPE).copy() if n else np.zeros(0, dtype=SEG_DTYPE)
    lib().sa_free_segments(out)
    return res


def ExtendHits(hits, rev, buffer):
    """Extension stage alone for anchors [(ref_loc, query_loc), ...]: passing records, unordered (header removed)."""
    h = np.ascontiguousarray(hits, dtype=np.uint32).reshape(-1, 2)
    out = C.c_void_p()
    n = lib().sa_extend_hits(h.ctypes.data, h.shape[0], int(bool(rev)), buffer, C.byref(out))
    return _take(n, out)[1:]


def SeedInterval(start, end, q_len, strands=STRAND_BOTH, buffer=0, threads=2):
    """seeder_body::operator() (src/seeder.cpp:12-127) for one query interval, chunk calls issued from C++ threads.
    Returns (plus-strand HSPs, minus-strand HSPs, summed statistics)."""
    fw, rc = C.c_void_p(), C.c_void_p()
    nf, nr = C.c_size_t(), C.c_size_t()
    st = CallStats()
    lib().sa_seed_interval(start, end, q_len, strands, buffer, threads, C.byref(fw), C.byref(nf), C.byref(rc), C.byref(nr),
                           C.byref(st))

    def take(ptr, n):
        if n == 0:
            lib().sa_free_segments(ptr)
            return np.zeros(0, dtype=SEG_DTYPE)
        buf = (C.c_char * (n * SEG_DTYPE.itemsize)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=SEG_DTYPE).copy()
        lib().sa_free_segments(ptr)
        return a

    return take(fw, nf.value), take(rc, nr.value), {f[0]: getattr(st, f[0]) for f in CallStats._fields_}


# ---- repeat masker ----------------------------------------------------------------------------------------------
def RmSendQueryWriteRequest():
    lib().sa_rm_send_query_write_request()


def RmClearQuery():
    lib().sa_rm_clear_query()


def RmSeedAndFilter(seed_offset_vector, rev, ref_start, ref_end):
    s = np.ascontiguousarray(seed_offset_vector, dtype=np.uint64)
    out = C.c_void_p()
    n = lib().sa_rm_seed_and_filter(s.ctypes.data, s.size, int(bool(rev)), ref_start, ref_end, C.byref(out))
    return _take(n, out)


def _take_intervals(n, out):
    if n == 0 or not out.value:
        return np.zeros(0, dtype=IVL_DTYPE)
    buf = (C.c_char * (n * IVL_DTYPE.itemsize)).from_address(out.value)
    iv = np.frombuffer(buf, dtype=IVL_DTYPE).copy()
    lib().sa_free_intervals(out)
    return iv


def RmMaskInterval(start_pos, end_pos, ref_start, ref_end, strands=STRAND_BOTH, M=1):
    """Device-side seeder_body::operator() of the repeat masker (repeat_masker_src/seeder.cpp:28-195) for one interval.
    Returns (intervals, dict(num_seeds, num_hits, num_hsps))."""
    out = C.c_void_p()
    tot = (C.c_uint64 * 3)()
    n = lib().sa_rm_mask_interval(start_pos, end_pos, ref_start, ref_end, strands, M, C.byref(out), tot)
    return _take_intervals(n, out), dict(num_seeds=tot[0], num_hits=tot[1], num_hsps=tot[2])


def RmCoverageIntervals(hsps, block_len, M=1):
    """repeat_masker_src/seeder.cpp:153-188 on the device for HSPs the host collected (headers removed)."""
    h = np.ascontiguousarray(hsps, dtype=SEG_DTYPE)
    out = C.c_void_p()
    n = lib().sa_rm_coverage_intervals(h.ctypes.data if h.size else None, h.size, block_len, M, C.byref(out))
    return _take_intervals(n, out)


# ---- knobs / introspection ---------------------------------------------------------------------------------------
def set_max_hits(v):
    lib().sa_set_max_hits(v)


def get_max_hits():
    return lib().sa_get_max_hits()


def max_hits_for_mem(total_global_mem):
    return lib().sa_max_hits_for_mem(total_global_mem)


def last_call_stats():
    st = CallStats()
    lib().sa_get_last_call_stats(C.byref(st))
    return {k: getattr(st, k) for k, _ in CallStats._fields_}


def lookup_mode():
    """0 general (seed words), 1 table-direct, 2 table-direct with target context (see sa_get_lookup_mode)."""
    return int(lib().sa_get_lookup_mode())


def neighbourhood_entries():
    return int(lib().sa_get_neighbourhood_entries())


def filter_mode():
    return int(lib().sa_get_filter_mode())


def set_count_examined(on):
    lib().sa_set_count_examined(int(bool(on)))


def profile_enable(on=True):
    lib().sa_profile_enable(int(bool(on)))


def profile_reset():
    lib().sa_profile_reset()


def profile_entries():
    """{kernel name: (total_ms, launches)} measured with HIP events on the engine's own streams."""
    out = {}
    L = lib()
    for i in range(L.sa_profile_num_entries()):
        name = C.create_string_buffer(64)
        ms = C.c_double()
        n = C.c_uint64()
        if L.sa_profile_get(i, name, 64, C.byref(ms), C.byref(n)) == 0:
            out[name.value.decode()] = (ms.value, n.value)
    return out


def profile_busy_ms(name):
    """ms during which at least one launch of the scope was running (union over the slots' streams)."""
    return float(lib().sa_profile_busy_ms(name.encode()))


def copy_ref_codes(dev=0):
    out = np.empty(lib().sa_get_ref_len(), dtype=np.uint8)
    lib().sa_copy_ref_codes(dev, out.ctypes.data)
    return out


def copy_index_table(dev=0):
    out = np.empty(lib().sa_get_index_table_size(), dtype=np.uint32)
    lib().sa_copy_index_table(dev, out.ctypes.data)
    return out


def copy_pos_table(dev=0):
    out = np.empty(lib().sa_get_num_index(), dtype=np.uint32)
    if out.size:
        lib().sa_copy_pos_table(dev, out.ctypes.data)
    return out


def copy_query_codes(buffer, rev, dev=0):
    out = np.empty(lib().sa_get_query_len(buffer), dtype=np.uint8)
    lib().sa_copy_query_codes(dev, buffer, int(bool(rev)), out.ctypes.data)
    return out


def device_make_seeds(start, end, rev, buffer, per=13):
    cap = max((end - start) * per, 1)
    out = np.empty(cap, dtype=np.uint64)
    n = lib().sa_device_make_seeds(start, end, int(bool(rev)), buffer, out.ctypes.data, cap)
    return out[:min(n, cap)].copy()


# ---- options (include/segalign_amd.h: one table, resolved at InitializeProcessor) -------------------------------
def set_option(name, value):
    if lib().sa_set_option(name.encode(), int(value)) != 0:
        raise KeyError("unknown engine option %r" % name)


def reset_option(name=None):
    lib().sa_reset_option(name.encode() if name else None)


def get_option(name):
    return int(lib().sa_get_option(name.encode()))


def options():
    """{name: test_only} of every option the engine knows."""
    out = {}
    for i in range(lib().sa_option_count()):
        t = C.c_int(0)
        out[lib().sa_option_name(i, C.byref(t)).decode()] = bool(t.value)
    return out


def get_audit(cap=1 << 22):
    """(ref_loc, query_loc) of the hits the filter levels rejected in this thread's last table-direct call (option audit_cap)."""
    buf = np.empty((cap, 2), dtype=np.uint32)
    n = lib().sa_get_audit(buf.ctypes.data, cap)
    return buf[:min(n, cap)].copy(), int(n)
