"""Error behaviour mirrors the reference: message on stderr + exit code (common/cuda_utils.h:4-37,
seed_filter_interface.cu:53-70).  CPU part: without a GPU the engine refuses to start -- there is no CPU fallback."""
import subprocess
import sys
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_py(code):
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n%s" % (ROOT, code)],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_no_gpu_means_exit_1_not_a_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = run_py("from segalign_amd import engine as E\nE.InitializeInterface(-1)\nprint('alive')")
    assert r.returncode == 1 and b"No GPU device found" in r.stderr and b"alive" not in r.stdout


def test_calls_before_initialisation_exit_1():
    r = run_py("from segalign_amd import engine as E\nimport numpy as np\nE.SeedAndFilter(np.zeros(1, dtype=np.uint64), False, 0)")
    assert r.returncode == 1 and b"before InitializeInterface" in r.stderr


@pytest.mark.gpu
def test_too_many_gpus_requested_exits_10():
    r = run_py("from segalign_amd import engine as E\nE.InitializeInterface(4096)")
    assert r.returncode == 10 and b"Requested GPUs greater than available GPUs" in r.stderr


@pytest.mark.gpu
def test_max_seeds_assert_like_the_reference():
    """src/seed_filter.cu:688-692: printf('MAX_SEEDS exceeded') + assert."""
    r = run_py("from segalign_amd import engine as E\nimport numpy as np\nE.InitializeInterface(1)\n"
               "E.GenerateShapePos('TTT0T00TT00T0T0TTTT')\n"
               "E.InitializeProcessor(True, 1000, 19, np.zeros(64, dtype=np.int32), 910, 3000, False)\n"
               "E.SeedAndFilter(np.zeros(13001, dtype=np.uint64), False, 0)")
    assert r.returncode != 0 and b"MAX_SEEDS exceeded" in r.stdout
