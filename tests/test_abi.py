"""CPU-only: the C-ABI library loads and exports every symbol include/segalign_amd.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "segalign_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_reference_boundary():
    syms = header_symbols()
    for s in ("sa_initialize_interface", "sa_initialize_processor", "sa_send_ref_write_request", "sa_clear_ref",
              "sa_generate_seed_pos_table", "sa_send_query_write_request", "sa_clear_query", "sa_seed_and_filter",
              "sa_shutdown_processor"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from segalign_amd.build import build_lib
    from segalign_amd import engine as E
    build_lib()
    L = E.lib()
    missing = [s for s in header_symbols() if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(E.C_ABI_SYMBOLS) == header_symbols()
    assert b"gfx950" in L.sa_version()


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "segalign_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower(), (dp, f)
    inc = open(os.path.join(ROOT, "include", "segalign_amd.h")).read()
    assert "oracle" not in inc.lower()


def test_missing_library_fails_loudly(monkeypatch):
    from segalign_amd import engine as E
    monkeypatch.setattr(E, "_lib", None)
    monkeypatch.setattr(E, "LIB_PATH", "/nonexistent/libsegalign_hip.so")
    with pytest.raises(RuntimeError):
        E.lib()


def test_library_leaves_the_hosts_environment_alone():
    """Four engine slots per device want more than the HIP runtime's default of four hardware queues (slots that share a queue
    serialise).  That is the HOST's choice (bench.py and the C++ hosts export GPU_MAX_HW_QUEUES=8 themselves):
    loading the library neither sets the variable nor changes a value the user has set."""
    import subprocess
    import sys
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from segalign_amd import engine as E; E.lib(); "
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; v = libc.getenv(b'GPU_MAX_HW_QUEUES'); "
            "print('unset' if v is None else v.decode())" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip() == "unset"
    env["GPU_MAX_HW_QUEUES"] = "2"
    assert subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip() == "2"
