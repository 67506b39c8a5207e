"""GPU: sa_rm_mask_interval groups the chunks of a strand into table-direct passes of up to 16 chunks
(repeat_masker_src/seeder.cpp:73-150 walks them one by one).  25 chunks per strand here: a full group, a remainder, and --
on the minus strand -- the chunk of the short last plus chunk, which overlaps its neighbour (:118-119) and must stay alone."""
import numpy as np
import pytest

from helpers import Case
from segalign_amd import synth
from test_gpu_rm_mask import as_list, model_mask_interval

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[20, None], ids=["chunks_per_call=20", "default grouping"])
def rm_case_small_chunks(oracle, engine, request):
    unit = synth.random_dna(500, 78)
    t = synth.random_dna(200000, 16)
    rng = np.random.default_rng(4)
    for i in range(120):  # a diverged repeat family, both orientations, everywhere in the block
        p = int(rng.integers(0, t.size - 600))
        cp = synth.mutate(unit, 700 + i, 0.06)
        t[p:p + cp.size] = cp if i % 2 else synth.reverse_complement(cp)
    t = synth.soft_mask(t, 6, 0.03)
    if request.param is not None:
        engine.set_option("chunks_per_call", request.param)  # (the grouping the cases below are laid out for)
    else:
        engine.reset_option("chunks_per_call")  # what the repeat-masker bench runs: the option's default
    c = Case(t, t, chunk=8000).oracle_setup(oracle).engine_setup(engine)
    engine.RmSendQueryWriteRequest()
    c.grouping = request.param
    yield c
    engine.RmClearQuery()
    engine.ShutdownProcessor()
    engine.reset_option("chunks_per_call")


@pytest.mark.parametrize("strands", [1, 2, 3])
def test_grouped_chunks_match_the_reference_loop(oracle, rm_case_small_chunks, strands):
    c, E, O = rm_case_small_chunks, rm_case_small_chunks.E, oracle
    L = c.target.size
    assert E.lookup_mode() == 2
    assert E.get_option("chunks_per_call") == (c.grouping or 40)  # (default 40: the repeat masker groups min(chunks_per_call, 20), api_rm.hip)
    for (s, e, ws, we, M) in ((0, L - 19, 0, L, 1),            # 25 chunks: 16 + 9 (plus), 1 + 16 + 8 (minus)
                              (3000, 195500, 0, L, 2),          # short last chunk -> overlapping minus chunk
                              (10000, 170000, 60000, 140000, 1),  # window inside the interval
                              (0, 136000, 0, L, 1)):            # exactly 17 chunks
        want, wt = model_mask_interval(c, O, s, e, ws, we, strands, M)
        got, gt = E.RmMaskInterval(s, e, ws, we, strands, M)
        assert as_list(got) == as_list(want), (s, e, ws, we, M, as_list(got)[:4], as_list(want)[:4])
        assert gt == wt
    assert len(as_list(got)) >= 3
