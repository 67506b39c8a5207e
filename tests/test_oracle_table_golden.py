"""CPU: the oracle's seed position table against tables built by the reference's own GenerateSeedPosTable text (common/
seed_pos_table.cu:49-109) linked with the real common/ntcoding.cpp, TBB and the upload stood in for (tests/golden/make_table_golden.py):
inclusive bucket ends, the step / offset rule (H6: position 0 never indexed for step 1; steps 2, 5, 19, 20), windows with soft-masked
bases, N, '&' or other letters skipped, heavy buckets, 12of19 and 14of22, a 40-base block.  A second route, not a pin."""
import numpy as np
import pytest

import table_golden as G

CASES = list(G.cases())


@pytest.mark.parametrize("c", CASES, ids=[G.case_id(c) for c in CASES])
def test_oracle_table_equals_the_reference_functions_table(oracle, c):
    k = oracle.generate_shape_pos(c["shape"])
    assert k == c["kmer_size"]
    index, pos = oracle.generate_seed_pos_table(c["target"].tobytes(), 0, c["target"].size, c["step"], len(c["shape"]), k)
    G.check_table(c, index, pos)
    if c["step"] == 1:
        assert pos.min() >= 1          # H6
