"""Builds the native pieces in-tree.

  libzsgpu.so      hipcc --offload-arch=gfx950 on csrc/zs_engine.hip (the product)
  libzsoracle.so   gcc on oracle/*.c (test infrastructure only; the product never loads it)

hipcc cross-compiles for gfx950 without a GPU.  The built .so files are
git-ignored but travel to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zlibstream_amd")
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libzsgpu.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libzsoracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_engine(force=False, verbose=False):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "zsgpu.h")]
    if not force and not _newer(LIB, srcs):
        return LIB
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build zlibstream_amd/libzsgpu.so")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-o", LIB, os.path.join(CSRC, "zs_engine.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=ROOT)
    return LIB


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("zs_oracle.c", "zs_inflate_oracle.c", "zs_oracle.h")]
    if not force and not _newer(ORACLE_LIB, srcs):
        return ORACLE_LIB
    subprocess.run(["make", "-C", ORACLE_DIR, "-s", "-B", "libzsoracle.so"], check=True)
    return ORACLE_LIB


def build_tools(force=False):
    """tools/stream_api_bench: the C++ mirror of the Stream API (include/zsgpu.hpp) driven like the reference's benchmarks;
    bench.py runs it for the `stream_api` figure."""
    exe = os.path.join(ROOT, "build", "stream_api_bench")
    srcs = [os.path.join(ROOT, "tools", "stream_api_bench.cpp"), os.path.join(ROOT, "include", "zsgpu.hpp"),
            os.path.join(ROOT, "include", "zsgpu.h"), LIB]
    if not force and not _newer(exe, srcs):
        return exe
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, srcs[0], "-L" + PKG, "-lzsgpu", "-Wl,-rpath," + PKG,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True, cwd=ROOT)
    return exe


if __name__ == "__main__":
    build_engine(force="--force" in sys.argv, verbose=True)
    build_oracle(force="--force" in sys.argv)
    build_tools(force="--force" in sys.argv)
    print("built", LIB, "and", ORACLE_LIB)
