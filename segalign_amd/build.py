"""Build libsegalign_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to compile."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsegalign_hip.so")
SOURCES = ["encode.hip", "scan.hip", "table.hip", "seeds.hip", "probe.hip", "join.hip", "extend.hip", "dedup.hip", "coverage.hip",
           "arena.hip", "options.hip", "profile.hip", "pool.hip", "front.hip", "core.hip", "api_setup.hip", "api_calls.hip", "api_rm.hip",
           "api_introspect.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_lib(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "segalign_amd.h"))
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(h, op) for h in headers):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, out.decode(errors="replace")))
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc build failed")
    if force or procs or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        subprocess.check_call(cmd)
    return LIB_PATH


HOST_SRC = os.path.join(HERE, "host", "segalign_host.cpp")
HOST_BIN = os.path.join(HERE, "bin", "segalign_host")
RM_HOST_SRC = os.path.join(HERE, "host", "segalign_rm_host.cpp")
RM_HOST_BIN = os.path.join(HERE, "bin", "segalign_rm_host")
HOST_COMMON = os.path.join(HERE, "host", "host_common.hpp")


def build_host(force=False):
    """The C++ host harness (FASTA -> .segments) on top of the C-ABI: plain g++ + zlib, links libsegalign_hip.so."""
    build_lib()
    os.makedirs(os.path.dirname(HOST_BIN), exist_ok=True)
    hdr = os.path.join(HERE, "..", "include", "segalign_amd.h")
    for src, dst in ((HOST_SRC, HOST_BIN), (RM_HOST_SRC, RM_HOST_BIN)):
        if force or _newer(src, dst) or _newer(hdr, dst) or _newer(HOST_COMMON, dst) or _newer(LIB_PATH, dst):
            subprocess.check_call(["g++", "-std=c++11", "-O2", "-pthread", "-Wall", "-I", os.path.join(HERE, "..", "include"), src,
                                   "-o", dst, "-L", LIB_DIR, "-lsegalign_hip", "-lz", "-Wl,-rpath," + LIB_DIR,
                                   "-Wl,-rpath,/opt/rocm/lib"])
    return HOST_BIN


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv))
