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


def test_mixed_flavours_fail_at_link_time(tmp_path):
    """A translation unit of the repeat-masker binary that includes the header WITHOUT the _RM macro would see the src/ signatures
    for the same g_* names (silent type mismatch).  The header's flavour guard turns that into an undefined reference."""
    from segalign_amd.build import build_lib, LIB_DIR
    build_lib()
    inc = os.path.join(ROOT, "include")
    # (the globals of common/ntcoding.cpp:6-8 the compat GenerateSeedPosTable reads)
    (tmp_path / "defs_rm.cpp").write_text('int shape_pos[32];\nint shape_size;\nint transition_pos[32];\n#define SEGALIGN_AMD_COMPAT_DEFINE_RM\n'
                                          '#include "segalign_amd_compat.hpp"\nint main() { return 0; }\n')
    (tmp_path / "other_src_flavour.cpp").write_text('#include "segalign_amd_compat.hpp"\nint other() { return 1; }\n')
    (tmp_path / "other_rm_flavour.cpp").write_text('#define SEGALIGN_AMD_COMPAT_RM\n#include "segalign_amd_compat.hpp"\nint other() { return 1; }\n')
    base = ["g++", "-std=c++11", "-O2", "-pthread", "-I", inc]
    tail = ["-L", LIB_DIR, "-lsegalign_hip", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib"]
    ok = subprocess.run(base + [str(tmp_path / "defs_rm.cpp"), str(tmp_path / "other_rm_flavour.cpp"), "-o", str(tmp_path / "ok")] + tail,
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert ok.returncode == 0, ok.stdout.decode()
    bad = subprocess.run(base + [str(tmp_path / "defs_rm.cpp"), str(tmp_path / "other_src_flavour.cpp"), "-o", str(tmp_path / "bad")] + tail,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert bad.returncode != 0 and "segalign_amd_compat_flavour_src" in bad.stdout.decode(), bad.stdout.decode()
