"""GPU, multi-device (the reference's own model: ONE process, all visible devices, calls handed to whichever device has a
free token -- common/seed_filter_interface.cu:49-80, src/seed_filter.cu:699-706,798-803).  The driver's test box has one
GPU, so this module skips there with a reason; on a multi-GPU node it runs 8 host threads over every device and checks
every call against the oracle, both through the drop-in entry (host seed words) and the device-seeded entry."""
import threading

import numpy as np
import pytest

from helpers import Case, seg_equal
from segalign_amd import synth

pytestmark = pytest.mark.gpu


def device_count():
    import torch
    return torch.cuda.device_count()


def test_all_devices_token_pool_matches_oracle(oracle, engine):
    n = device_count()
    if n < 2:
        pytest.skip("needs >= 2 GPUs in this process (found %d): sa_initialize_interface(-1) over several devices" % n)
    E = engine
    t, q = synth.make_pair(300000, 13, 14, sub_rate=0.1, mask_frac=0.1, records=2, indel_every=600)
    c = Case(t, q, chunk=25000).oracle_setup(oracle)
    c.engine_setup(E, num_gpu=-1)  # every device gets target, tables and both query strands (seed_pos_table.cu:33-47)
    try:
        jobs = [(rev, s, e) for rev in (False, True) for (s, e) in c.chunks()]
        want = {j: c.oracle_saf(c.host_seeds(j[1], j[2], j[0]), j[0])[0] for j in jobs}
        got, got_dev, devices = {}, {}, set()
        lock = threading.Lock()

        def work(my):
            for j in my:
                a = E.SeedAndFilter(c.host_seeds(j[1], j[2], j[0]), j[0], 0)
                d1 = E.last_call_stats()["device"]
                b = E.SeedAndFilterRange(j[1], j[2], j[0], 0)
                d2 = E.last_call_stats()["device"]
                with lock:
                    got[j], got_dev[j] = a, b
                    devices.update((d1, d2))

        threads = [threading.Thread(target=work, args=(jobs[i::8],)) for i in range(8)]
        [th.start() for th in threads]
        [th.join() for th in threads]
        assert all(seg_equal(got[j], want[j]) and seg_equal(got_dev[j], want[j]) for j in jobs)
        assert len(devices) >= 2, devices  # the pool really spread the calls
        for d in range(n):  # replicated state is identical on every device
            assert np.array_equal(E.copy_index_table(d), c.o_index) and np.array_equal(E.copy_pos_table(d), c.o_pos)
    finally:
        E.ShutdownProcessor()
