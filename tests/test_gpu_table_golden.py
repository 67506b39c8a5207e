"""GPU: the device-built seed position table (table.hip: radix-partition build, two levels for 12of19, three for 14of22; every step)
against tables built by the reference's own GenerateSeedPosTable text (tests/golden/table_golden.json)."""
import numpy as np
import pytest

import table_golden as G
from helpers import canonical_pos_table

pytestmark = pytest.mark.gpu

CASES = list(G.cases())


@pytest.mark.parametrize("atomic", [0, 1], ids=["partition build", "atomic build"])
def test_device_table_equals_the_reference_functions_table(oracle, engine, atomic):
    E = engine
    try:
        for c in CASES:
            E.reset_option(None)
            E.set_option("table_atomic", atomic)
            E.InitializeInterface(1)
            k = E.GenerateShapePos(c["shape"])
            assert k == c["kmer_size"]
            E.InitializeProcessor(True, 250000, len(c["shape"]), oracle.build_sub_mat(910), 910, 3000, False)
            keep = E.SendRefWriteRequest(c["target"], 0, c["target"].size)
            E.GenerateSeedPosTable(keep, 0, c["target"].size, c["step"], len(c["shape"]), k)
            index, pos = E.copy_index_table(), E.copy_pos_table()
            G.check_table(c, index, canonical_pos_table(index, pos))
            E.ShutdownProcessor()
    finally:
        E.ShutdownProcessor()
        E.reset_option(None)
