"""GPU: the driver's contract with bench.py -- ONE JSON line on stdout with the agreed keys -- checked on the small plumbing
workload (BASELINE configs[0]) so that it stays cheap; the default workload prints the same structure."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "plumbing", "--steps", "3", "--warmup", "1",
                          "--partition", "hits", "--cpu-seconds", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                   ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[k], typ), (k, line.get(k))
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["vs_baseline"] is None and line["scaling"] == "strong" and line["unit"] == "Gbp/s"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["ms_per_step"] * 1e-3 * line["value"] * 1e9 - 1_000_000) < 0.05 * 1_000_000  # value = query bases / step time
    roof = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_symbol", "single_stream", "per_step", "seed_lookup"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "valu_issue") and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["single_stream"]["frac"] <= 1.0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["unit"] == "Gbp/s" and isinstance(cpu["sample"], str)
    assert line["config"]["partition_imbalance"]["by_hits"] >= 1.0
    # the order-independent checksum of a pass is part of the line (an N-GPU strong-scaling run must reproduce it)
    assert isinstance(line["config"]["hsp_checksum"], int)
