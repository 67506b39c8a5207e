"""Shared set-up for parity tests: drive the engine and the oracle through the same call sequence
(src/main.cpp:297-298,613-621,659-661 ; src/seeder.cpp:47-121) on the same synthetic block pair."""
import numpy as np

SHAPE_12OF19 = "TTT0T00TT00T0T0TTTT"  # src/main.cpp:160-163


class Case:
    """One target block x one query block, default parameters of the reference unless overridden."""

    def __init__(self, target, query, shape=SHAPE_12OF19, step=1, transition=True, xdrop=910, hspthresh=3000,
                 noentropy=False, chunk=250000, sub_mat=None):
        self.target = np.ascontiguousarray(target, np.uint8)
        self.query = np.ascontiguousarray(query, np.uint8)
        self.shape, self.step, self.transition = shape, step, transition
        self.xdrop, self.hspthresh, self.noentropy, self.chunk = xdrop, hspthresh, noentropy, chunk
        self.seed_size = len(shape)
        self.sub_mat = sub_mat

    # ---- oracle side ---------------------------------------------------------------------------------------
    def oracle_setup(self, O):
        self.O = O
        self.kmer_size = O.generate_shape_pos(self.shape)
        if self.sub_mat is None:
            self.sub_mat = O.build_sub_mat(self.xdrop)
        self.o_ref = O.encode(self.target.tobytes())
        self.o_q, self.o_qrc = O.encode_rev_comp(self.query.tobytes())
        self.o_index, self.o_pos = O.generate_seed_pos_table(self.target.tobytes(), 0, self.target.size, self.step,
                                                             self.seed_size, self.kmer_size)
        self.query_rc_ascii = np.frombuffer(O.rev_comp_ascii(self.query.tobytes(), 0, self.query.size), dtype=np.uint8)
        return self

    def host_seeds(self, start, end, rev):
        """seeder.cpp:57-74 / :94-109."""
        buf = self.query_rc_ascii if rev else self.query
        return self.O.make_seeds(buf.tobytes(), 0, start, end, self.seed_size, self.kmer_size, self.transition)

    def oracle_saf(self, seeds, rev, max_hits=1 << 30):
        q = self.o_qrc if rev else self.o_q
        return self.O.seed_and_filter(self.o_ref, q, self.o_index, self.o_pos, seeds, self.sub_mat,
                                      seed_size=self.seed_size, xdrop=self.xdrop, hspthresh=self.hspthresh,
                                      noentropy=self.noentropy, max_hits=max_hits)

    # ---- engine side (reference call order) ----------------------------------------------------------------
    def engine_setup(self, E, num_gpu=1):
        self.E = E
        E.InitializeInterface(num_gpu)
        k = E.GenerateShapePos(self.shape)
        if self.sub_mat is None:
            raise RuntimeError("call oracle_setup first or pass sub_mat")
        E.InitializeProcessor(self.transition, self.chunk, self.seed_size, self.sub_mat, self.xdrop, self.hspthresh,
                              self.noentropy)
        self._ref_keep = E.SendRefWriteRequest(self.target, 0, self.target.size)
        E.GenerateSeedPosTable(self._ref_keep, 0, self.target.size, self.step, self.seed_size, k)
        E.SendQueryWriteRequest(self.query, 0, self.query.size, 0)
        return self

    def chunks(self):
        """(start, end) chunk bounds over [0, len - seed_size) like src/main.cpp:383-393 + seeder.cpp:48-51."""
        end_pos = self.query.size - self.seed_size
        out = []
        for i in range(0, max(end_pos, 0), self.chunk):
            out.append((i, min(i + self.chunk, end_pos)))
        return out


def seg_equal(a, b):
    return a.shape == b.shape and bool(np.all(a == b))


def canonical_pos_table(index, pos):
    """Positions sorted inside every bucket (the engine leaves very large buckets in arrival order; bucket order is
    nondeterministic in the reference and cannot influence any result, hazard H7)."""
    import numpy as np
    index = np.asarray(index, dtype=np.int64)
    starts = np.concatenate([[0], index[:-1]])
    bucket_of = np.repeat(np.arange(index.size, dtype=np.int64), index - starts)
    order = np.lexsort((np.asarray(pos, dtype=np.int64), bucket_of))
    return np.asarray(pos)[order]


def check_seed_table_properties(E, target_size, seed_size=19):
    """Size-independent properties of the device-built seed position table (common/seed_pos_table.cu:49-109): offsets ascending and
    closed, every indexed position distinct, in range, never 0 (H6), ascending inside its bucket (buckets above 512 entries -- poly-A,
    microsatellites -- may keep their arrival order: the reference's order is atomic arrival order everywhere, hazard H7, and no
    consumer depends on it), and as many positions as there are windows of upper-case ACGT only."""
    import numpy as np
    index = E.copy_index_table()
    pos = E.copy_pos_table()
    assert index[-1] == pos.size and np.all(np.diff(index.astype(np.int64)) >= 0)
    assert pos.min() >= 1 and pos.max() <= target_size - seed_size
    assert np.unique(pos).size == pos.size
    starts = np.concatenate([[0], index[:-1].astype(np.int64)])
    desc = np.nonzero(np.diff(pos.astype(np.int64)) < 0)[0] + 1  # descents may only happen at bucket starts ...
    inside = desc[~np.isin(desc, starts)]
    if inside.size:                                               # ... or inside a bucket too large for the in-LDS sort
        b = np.searchsorted(index.astype(np.int64), inside, side="right")   # bucket that holds entry `inside`
        sizes = index.astype(np.int64)[b] - np.concatenate([[0], index.astype(np.int64)])[b]
        assert np.all(sizes > 512), (inside[:5], sizes[:5])
    codes = E.copy_ref_codes()
    bad = (codes >= 4).astype(np.int32)
    csum = np.concatenate([[0], np.cumsum(bad)])
    valid = (csum[seed_size:] - csum[:-seed_size]) == 0   # window starting at p = 0 .. len - seed_size
    assert int(valid[1:].sum()) == pos.size               # position 0 excluded
    return index, pos


def pos_tables_equal_by_bucket(index, pos_a, pos_b):
    """Two position tables under the same bucket ends hold the same positions bucket by bucket (order inside a bucket is free, hazard
    H7): only the buckets in which the two differ are sorted and compared -- at 100 Mbp that is the few hundred buckets too large for
    the device's in-LDS sort, not 80 M entries."""
    import numpy as np
    pos_a, pos_b = np.asarray(pos_a), np.asarray(pos_b)
    if pos_a.shape != pos_b.shape:
        return False
    diff = np.flatnonzero(pos_a != pos_b)
    if diff.size == 0:
        return True
    ends = np.asarray(index, dtype=np.int64)
    buckets = np.unique(np.searchsorted(ends, diff, side="right"))
    starts = np.concatenate([[0], ends])
    for b in buckets:
        lo, hi = int(starts[b]), int(ends[b])
        if not np.array_equal(np.sort(pos_a[lo:hi]), np.sort(pos_b[lo:hi])):
            return False
    return True
