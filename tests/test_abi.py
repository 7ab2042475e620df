"""The C-ABI library builds for gfx950, loads, and exports every symbol include/zsgpu.h declares.
No compute calls: this runs without a GPU."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zsgpu.h")).read()
    return sorted(set(re.findall(r"ZS_API\s+[\w\s\*]+?\b(zs_\w+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for s in ("zs_ctx_create", "zs_deflate_batch_device", "zs_deflate_batch", "zs_deflate_init", "zs_deflate", "zs_deflate_end",
              "zs_last_message", "zs_deflate_bound"):
        assert s in syms


def test_library_builds_and_exports_all_symbols():
    from zlibstream_amd import build, _native
    lib = build.build_engine()
    L = ctypes.CDLL(lib)
    for s in declared_symbols():
        assert hasattr(L, s), s
    assert sorted(_native.SYMBOLS) == declared_symbols()


def test_bound_is_pure_host_code():
    from zlibstream_amd import deflate_bound
    assert deflate_bound(0) >= 8
    assert deflate_bound(1 << 26) > (1 << 26)


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "zlibstream_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")) and f != "build.py":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "zs_oracle" not in text and "libzsoracle" not in text and "zso_" not in text, f


def test_engine_fails_loudly_without_gpu():
    import pytest
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    from zlibstream_amd import Engine
    with pytest.raises(RuntimeError):
        Engine(0)


def test_cpp_host_mirror_compiles():
    import subprocess
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")], check=True)


def test_geometry_tables_hold_their_invariants():
    """zs_core.h build_geometry on 20 000 random Write schedules -- fresh streams, runs behind a stop of the literal engine,
    runs that begin where a flush left the stream: the chunks tile the body, a segment's first chunk holds every loop-top at
    which its cluster's reads can fire, the clusters hold every Write end and window end once and in order, window bases and
    slide thresholds follow the window ends (tests/cpp/test_geometry.cpp; what the kernels read from these tables)."""
    exe = os.path.join(ROOT, "build", "test_geometry")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_geometry.cpp")], check=True)
    r = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("PASS"), r.stdout[-2000:]
