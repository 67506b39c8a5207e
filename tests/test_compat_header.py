"""CPU-only: include/segalign_amd_compat.hpp defines the reference's own engine symbols and a host written against
them compiles and links with g++ against libsegalign_hip.so (nothing is executed: no GPU here)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(rm=False):
    """the miniature reference host against the src/ symbols, or (rm=True) against the repeat masker's symbols"""
    from segalign_amd.build import build_lib, LIB_DIR
    build_lib()
    src = os.path.join(ROOT, "tests", "cpp", "compat_driver.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "compat_driver_rm" if rm else "compat_driver")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("segalign_amd.h", "segalign_amd_compat.hpp")]
    if (not os.path.exists(exe)) or any(os.path.getmtime(p) > os.path.getmtime(exe) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++11", "-O2", "-pthread", "-Wall"] + (["-DCOMPAT_DRIVER_RM"] if rm else []) +
                              ["-I", os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L", LIB_DIR, "-lsegalign_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_compat_host_compiles_and_links():
    exe = build_driver()
    syms = subprocess.check_output(["nm", "-C", exe]).decode()
    for s in ("g_InitializeInterface", "g_InitializeProcessor", "g_SendRefWriteRequest", "g_SendQueryWriteRequest",
              "g_SeedAndFilter", "g_ClearRef", "g_ClearQuery", "g_ShutdownProcessor", "GenerateSeedPosTable"):
        assert s in syms, s


def test_repeat_masker_compat_host_compiles_and_links():
    """the repeat masker declares the same g_* names with other signatures (repeat_masker_src/seed_filter.h:4-14)"""
    exe = build_driver(rm=True)
    syms = subprocess.check_output(["nm", "-C", exe]).decode()
    for s in ("g_InitializeInterface", "g_InitializeProcessor", "g_SendRefWriteRequest", "g_SendQueryWriteRequest",
              "g_SeedAndFilter", "g_ClearRef", "g_ClearQuery", "g_ShutdownProcessor", "GenerateSeedPosTable",
              "sa_rm_seed_and_filter", "sa_rm_send_query_write_request", "sa_rm_clear_query"):
        assert s in syms, s
