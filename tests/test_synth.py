"""CPU: the synthetic block pairs the bench and the GPU tests run on (segalign_amd/synth.py) -- deterministic, and the "lumpy"
realistic-composition stand-in (BASELINE configs[1] substitute: the real ce11 / cb4 of the reference's README.md:69-78 cannot be
fetched here) really has what i.i.d. DNA lacks: a skewed k-mer spectrum, soft-masked repeats, N gaps, several records."""
import numpy as np

from segalign_amd import synth


def kmer_counts(seq, k=12):
    """occurrences of every k-mer of upper-case ACGT only (what a seed table would index, contiguous-seed simplification)"""
    code = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    c = code[seq]
    bad = (c == 255).astype(np.int32)
    csum = np.concatenate([[0], np.cumsum(bad)])
    ok = (csum[k:] - csum[:-k]) == 0
    keys = np.zeros(seq.size - k + 1, dtype=np.int64)
    for j in range(k):
        keys = keys * 4 + np.where(c[j:seq.size - k + 1 + j] == 255, 0, c[j:seq.size - k + 1 + j])
    return np.bincount(keys[ok], minlength=4 ** k)


def test_generators_are_deterministic():
    a = synth.make_pair(50000, 3, 4, sub_rate=0.08, mask_frac=0.2, records=3, indel_every=300, n_runs=2)
    b = synth.make_pair(50000, 3, 4, sub_rate=0.08, mask_frac=0.2, records=3, indel_every=300, n_runs=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    c = synth.make_realistic(600_000, 11, 12, records=3)
    d = synth.make_realistic(600_000, 11, 12, records=3)
    assert all(np.array_equal(x, y) for x, y in zip(c, d))
    e = synth.make_realistic(600_000, 13, 12, records=3)
    assert not np.array_equal(c[0], e[0])


def test_lumpy_stand_in_has_a_skewed_spectrum_masking_gaps_and_records():
    n = 2_000_000
    t, q = synth.make_realistic(n, 11, 12, records=4)
    u = synth.random_dna(n, 5)
    assert abs(t.size - n) < n // 50 and abs(q.size - n) < n // 5
    # records joined by '&' like the reference's block buffer (src/main.cpp:297-298), N gaps, soft masking
    assert int((t == ord("&")).sum()) == 3 and int((t == ord("N")).sum()) > 0
    lower = float(np.mean((t >= ord("a")) & (t <= ord("z"))))
    assert 0.03 < lower < 0.5, lower
    # AT-rich background (C. elegans: ~35 % GC)
    up = np.char.upper(t.view("S1")).view(np.uint8)
    gc = float(np.mean((up == ord("G")) | (up == ord("C"))))
    assert 0.30 < gc < 0.42, gc
    # the spectrum: the heaviest 12-mers of the unmasked sequence hold orders of magnitude more positions than uniform DNA's
    ct, cu = kmer_counts(t), kmer_counts(u)
    assert cu.max() < 12 and ct.max() > 20 * cu.max(), (ct.max(), cu.max())
    # sum of squares / positions = seed hits per indexed position of a self-comparison without transitions: what the engine's work
    # scales with (at 2 Mbp the repeat families have few copies yet: 1.8 x uniform DNA; at 100 Mbp the bench measures 3 x)
    per_pos = lambda c: float((c.astype(np.float64) ** 2).sum()) / float(c.sum())  # noqa: E731
    assert per_pos(ct) > 1.5 * per_pos(cu), (per_pos(ct), per_pos(cu))
    # the query is a diverged copy, not an independent sequence: it shares far more 12-mers with the target than random DNA does
    cq = kmer_counts(q)
    shared = float(np.minimum(ct, cq).sum()) / max(float(cq.sum()), 1.0)
    shared_u = float(np.minimum(ct, cu).sum()) / float(cu.sum())
    assert shared > 3 * shared_u, (shared, shared_u)
