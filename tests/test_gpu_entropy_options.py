"""GPU: the entropy kernel under its two switches, against the oracle under the same switches, on the find_hsps golden set
(tests/golden/find_hsps_golden.json):  option log4_double (hazard H2: the divisor log(4.0) instead of (double)logf(4.0f)) and the
test knob entropy_ulps (hazard H13: the entropy factor moved by +-1 ulp before the truncating multiplies).  Equality under +-1 ulp
says more than equality at 0: the device's log() and glibc's would have to differ by more than the margin the golden hits have."""
import numpy as np
import pytest

from test_gpu_find_hsps_golden import as_set, setup
from test_oracle_find_hsps_golden import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("opts", [dict(log4_double=1), dict(entropy_ulps=1), dict(entropy_ulps=-1), dict(log4_double=1, entropy_ulps=-1)],
                         ids=["log4_double", "plus one ulp", "minus one ulp", "log4_double minus one ulp"])
def test_entropy_switches_match_the_oracle(oracle, engine, opts):
    E = engine
    checked = 0
    for c in CASES:
        if c["noentropy"]:
            continue
        E.reset_option(None)
        for k, v in opts.items():
            E.set_option(k, v)
        keep = setup(E, c)
        try:
            got = E.ExtendHits(c["hits_a"], False, 0)
            ok, recs = oracle.extend_hits_pass(c["ref_codes"], c["query_codes"], np.array(c["sub_mat"], dtype=np.int32), c["hits_a"], xdrop=c["xdrop"],
                                               hspthresh=c["hspthresh"], noentropy=False, log4_is_float=not opts.get("log4_double", 0),
                                               entropy_ulps=opts.get("entropy_ulps", 0))
            want = recs[ok]
            assert as_set(zip(got["ref_start"], got["query_start"], got["len"], got["score"])) == \
                   as_set(zip(want["ref_start"], want["query_start"], want["len"], want["score"]))
            checked += int(np.count_nonzero(ok))
        finally:
            del keep
            E.ShutdownProcessor()
            E.reset_option(None)
    assert checked > 1500
