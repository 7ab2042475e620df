"""CPU walk-through of the GPU algorithm (tests/model/zs_model.cpp) against the oracle:
links -> matches -> chunk maps -> resolve -> symbols -> tail engine -> trees -> bits,
with the same ZS_HD code the kernels compile."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "zs_model")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", EXE, os.path.join(ROOT, "tests/model/zs_model.cpp"),
                    os.path.join(ROOT, "oracle/zs_oracle.c"), os.path.join(ROOT, "oracle/zs_inflate_oracle.c")], check=True)
    d = tmp_path_factory.mktemp("inputs")
    rng = np.random.default_rng(7)
    alice = open(os.path.join(ROOT, "tests/golden/corpus/alice29.txt"), "rb").read()
    files = {}

    def put(name, b):
        p = d / name
        p.write_bytes(b)
        files[name] = str(p)
    for n in (0, 1, 5, 261, 262, 263, 600, 65274, 65275, 65276, 65531, 65535, 65536, 65537, 65541, 65798, 98043, 98304, 98305):
        put("alice_%d" % n, (alice * 2)[:n])
        put("zeros_%d" % n, bytes(n))
        put("lowent_%d" % n, rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes())
    put("runs", np.repeat(rng.integers(0, 4, 30000, dtype=np.uint8), rng.integers(1, 40, 30000))[:200000].tobytes())
    for f in ("ptt5", "sum", "cp.html"):
        files[f] = os.path.join(ROOT, "tests/golden/corpus", f)
    return files


def _run(path, level, strategy=0, mode="chunk", wchunk=None, flush=0, env=None):
    cmd = [EXE, path, str(level), str(strategy), mode, str(wchunk or 0), str(flush)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0 and "PASS" in r.stdout, (cmd, r.stdout[-500:])


_pending = None


class batch:
    """The model runs inside the block go to the host's cores side by side (each is its own process); leaving the block waits for
    all of them and raises the first failure."""

    def __enter__(self):
        global _pending
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)))))
        self.futures = []
        _pending = self
        return self

    def __exit__(self, *exc):
        global _pending
        _pending = None
        try:
            if exc[0] is None:
                for f in self.futures:
                    f.result()
        finally:
            self.pool.shutdown(wait=True, cancel_futures=True)
        return False


def run(*args, **kwargs):
    if _pending is not None:
        _pending.futures.append(_pending.pool.submit(_run, *args, **kwargs))
    else:
        _run(*args, **kwargs)


def test_chunked_pipeline_matches_oracle(model):
    for name, path in model.items():
        for level in (4, 6, 9):
            if level == 9 and name in ("ptt5",):
                continue
            run(path, level)


def test_strategies_and_fast_levels(model):
    for name in ("alice_98304", "lowent_98305", "ptt5", "runs"):
        for strategy in (1, 2, 4):
            run(model[name], 6, strategy)
        for level in (1, 2, 3):
            run(model[name], level, 0, "seq")


def test_fast_levels_for_the_lanes_of_a_wave(model, tmp_path):
    """DeflateFast as zs_fast_vec_kernel does it (zs_fast_vec.h: 64 positions searched as if each were a loop-top, the
    parse's hops followed through the lanes, the inserted-position bitmap), with the kernels' own code on the CPU: levels 1-3,
    every strategy but Rle -- under HuffmanOnly no search happens, also not the one-candidate search behind an equal-bucket
    read (Deflate.Fast.cs:61-66) --, streams past the first and second window end."""
    rng = np.random.default_rng(11)
    extra = {"low150k": rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 150000).tobytes(), "zeros150k": bytes(150000)}
    files = dict(model)
    for k, v in extra.items():
        (tmp_path / k).write_bytes(v)
        files[k] = str(tmp_path / k)
    # (tools/fuzz_batch.py seed 208279: 65537 bytes over four symbols -- the tail engine takes over at the position the first
    # read inserted ahead, and what that insert finds as its bucket's head is the nearest *inserted* position)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_cases
    rng = np.random.default_rng(208279)
    rng.integers(0, 4)
    (tmp_path / "fuzz208279").write_bytes(fuzz_cases.deflate_batch_case(rng)[2][2])
    files["fuzz208279"] = str(tmp_path / "fuzz208279")
    with batch():
        for name in ("alice_98304", "zeros_98305", "lowent_98305", "alice_65537", "ptt5", "cp.html", "runs", "low150k", "zeros150k", "alice_600", "alice_5", "fuzz208279"):
            for level in (1, 2, 3):
                for strategy in (0, 2) if name in ("ptt5", "cp.html", "runs") else (0, 1, 2, 4):
                    run(files[name], level, strategy, "fvec")
                    # ... and as window-wide sweeps of a workgroup (zs_fast_sweep.h, zs_fast_sweep_kernel): the control flow of
                    # the kernel -- windows aligned to 64 positions, the guess of the inserted set, what a sweep makes final,
                    # events at a sweep's first loop-top, compressed links -- at the kernel's window and at small ones
                    run(files[name], level, strategy, "fsweep")
                    # ... and as rounds over the chunks of the stream (zs_fast_sweep.h "Rounds": every chunk of a round parsed from
                    # what the round before left -- entry loop-tops, the set from the planes of the chunks that own the positions,
                    # the cuts of equal-bucket events -- until a round changes nothing; zs_fast_commit_kernel's part behind it)
                    # (ZS_FR_RANGE: chunks a workgroup takes in turn, each reading what the ones before it have just left)
                    run(files[name], level, strategy, "frounds", env={"ZS_FR_CHUNK": str((1024, 4096, 10240)[(level + strategy) % 3]), "ZS_FR_RANGE": str((1, 3, 8)[(level + 2 * strategy) % 3]), **({"ZS_FR_RANGE_VARY": "1"} if level == 2 else {})})  # (VARY: another range every round, at most the given one)
    # several NoFlush Writes at the fast levels: a Write end is a read event like a window end (Stream.CopyTo's 81 920 bytes, 65 536,
    # 70 001; sizes whose ends fall where a loop-top may or may not slide the window -- 16 385-byte scanlines -- stay with the
    # literal engine, which the model then runs for the whole stream)
    # a body whose last match ends four bytes in front of the stream's end hands over at n - 4: the tail engine's restore inserts
    # the stream's last positions only where the parse did (tools/fuzz_streams.py seed 560345: 40 000 zeros in five Writes)
    (tmp_path / "zeros39996").write_bytes(bytes(39996))
    (tmp_path / "zeros40000").write_bytes(bytes(40000))
    with batch():
        for level in (1, 2, 3):
            for mode in ("fvec", "fsweep", "frounds"):
                run(str(tmp_path / "zeros39996"), level, 0, mode)
            run(str(tmp_path / "zeros40000"), level, 0, "fsweep", wchunk="6144,8192,1000,263,24401")
            run(str(tmp_path / "zeros40000"), level, 0, "frounds", wchunk="6144,8192,1000,263,24401")
    with batch():
        for name in ("alice_98304", "lowent_98305", "ptt5", "runs", "low150k"):
            for wchunk in (81920, 65536, 70001, 40000, 16385):
                run(files[name], 1 + wchunk % 3, 0, "fsweep", wchunk=wchunk)
                run(files[name], 1 + wchunk % 3, 0, "frounds", wchunk=wchunk, env={"ZS_FR_CHUNK": "2048", "ZS_FR_RANGE": "3"})
        for name in ("alice_98304", "zeros_98305", "lowent_98305", "runs", "ptt5", "fuzz208279"):
            for w, tile in ((64, 256), (256, 1024), (2048, 8192)):
                run(files[name], 1 + (w // 64) % 3, 0, "fsweep", env={"ZS_FS_W": str(w), "ZS_FS_TILE": str(tile)})


def test_rle_strategy_from_the_runs_of_equal_bytes(model, tmp_path):
    """CompressionStrategy.Rle (Deflate.Rle.cs:18-104) as the device does it (zs_rle.h): a position's part in the parse from where
    its run of equal bytes began -- symbols, block cuts (stored-block permission by the window base under the Rle refill
    threshold of 258) and bytes against the oracle; the hand-over to the literal engine at the first loop-top at or behind
    rle_body_end.  Text (runs of one), a bitmap, long and short runs, zeros, sizes around the window ends."""
    rng = np.random.default_rng(23)
    files = {"ptt5": model["ptt5"], "cp.html": model["cp.html"], "runs": model["runs"]}
    extra = {"zeros300k": bytes(300000),
             "long": np.repeat(rng.integers(0, 4, 20000, dtype=np.uint8), rng.integers(1, 700, 20000))[:1500000].tobytes(),
             "short": np.repeat(rng.integers(0, 3, 300000, dtype=np.uint8), rng.integers(1, 6, 300000))[:500001].tobytes()}
    for n in (4096 + 786, 65536, 65536 + 258, 65536 + 32768 - 258 + 600, 98304 + 600, 131072 + 522):
        extra["z%d" % n] = bytes(n)
        extra["r%d" % n] = np.repeat(rng.integers(0, 3, n, dtype=np.uint8), rng.integers(1, 400, n))[:n].tobytes()
    for k, v in extra.items():
        (tmp_path / k).write_bytes(v)
        files[k] = str(tmp_path / k)
    for name, path in files.items():
        for level in (1, 6, 9):
            run(path, level, 3, "rle")


def test_resumed_runs_in_the_chunked_form(model, tmp_path):
    """A stream flushed after its first F bytes (Deflate.cs:583-613), the rest one Write: the first Write on the literal engine,
    the run behind the flush laid out by build_geometry's GeoStart::at_read -- its first pass through the loop reads, with a
    window behind it -- and parsed in the chunked form on chains taken from that engine (prev[] below the flush, head[] for the
    first link of every bucket behind it: what zs_import_chains_kernel does on the device).  Symbols of both runs against the
    oracle's per-Write flush modes; flushes in the first bytes, around window ends and the slide threshold, Partial / Sync /
    Full (a FullFlush's forgotten heads are the case the data's own links get wrong: ZS_MODEL_NO_IMPORT shows it)."""
    rng = np.random.default_rng(31)
    alice = open(os.path.join(ROOT, "tests/golden/corpus/alice29.txt"), "rb").read()
    extra = {"alice300k": (alice * 3)[:300000], "low300k": rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 300000).tobytes(),
             "zeros200k": bytes(200000), "runs200k": np.repeat(rng.integers(0, 4, 30000, dtype=np.uint8), rng.integers(1, 40, 30000))[:200000].tobytes()}
    files = {}
    for k, v in extra.items():
        (tmp_path / k).write_bytes(v)
        files[k] = str(tmp_path / k)
    n_bulk = 0
    for name in files:
        for F in (3, 5, 262, 5000, 32768, 65273, 65274, 65275, 65536, 98304, 100000):
            for flush, level in ((2, 6), (3, 6), (1, 4), (3, 9)):
                if level == 9 and (name != "alice300k" or F not in (5000, 65536)):
                    continue
                if level == 4 and F not in (5, 65274, 100000):
                    continue
                r = subprocess.run([EXE, files[name], str(level), "0", "resume", str(F), str(flush)], capture_output=True, text=True)
                assert r.returncode == 0 and "PASS" in r.stdout, (name, F, flush, level, r.stdout[-400:])
                n_bulk += "mode=resume" in r.stdout
    # several NoFlush Writes behind the flush ("F,a,b,..": their sizes in turn): clusters of read events right behind the
    # run's first read, equal-bucket cuts among them
    for name in ("low300k", "alice300k", "runs200k"):
        for spec in ("5000,1000", "65536,16385", "100000,300,40000", "65274,100,263,5000", "3,70000", "32768,32768,200"):
            for flush in (2, 3):
                r = subprocess.run([EXE, files[name], "6", "0", "resume", spec, str(flush)], capture_output=True, text=True)
                assert r.returncode == 0 and "PASS" in r.stdout, (name, spec, flush, r.stdout[-400:])
                n_bulk += "mode=resume" in r.stdout
    assert n_bulk > 90  # (the chunked form really ran)
    # ... and the other way into a resumed run: no flush, the literal engine stops at the first loop-top at or behind F that no
    # read has touched (LitEngine::stop_abs), the chunked form goes on in the lazy parse's node it stands in -- a literal or
    # a match pending or not -- with the symbols of the block in progress in front of its own
    slots = set()
    for name in files:
        for F in (300, 33792, 40000, 65000, 65536, 70000, 100000, 131072, 150000):
            for level in (4, 6, 9):
                if level != 6 and (name not in ("alice300k", "low300k") or F not in (40000, 100000)):
                    continue
                r = subprocess.run([EXE, files[name], str(level), "0", "resume", str(F), "0"], capture_output=True, text=True)
                assert r.returncode == 0 and "PASS" in r.stdout, (name, F, level, r.stdout[-400:])
                if "slot=" in r.stdout:
                    slots.add(int(r.stdout.split("slot=")[1].split()[0]))
    assert len(slots) >= 3, slots  # (the run began in different nodes)
    # ... and the data's own links are not enough behind a FullFlush
    env = dict(os.environ, ZS_MODEL_NO_IMPORT="1")
    r = subprocess.run([EXE, files["alice300k"], "6", "0", "resume", "100000", "3"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "FAIL" in r.stdout


def test_multi_write_literal_engine(model):
    for name in ("alice_98304", "lowent_65537", "zeros_98305"):
        for w in (1, 100, 8192, 65536, 70000):
            for level in (1, 6):
                run(model[name], level, 0, "chunk", w)


def test_flush_modes_literal_engine_and_marker_accounting(model):
    """FlushMode Partial / Sync / Full (Deflate.cs:583-613): the literal engine closes a block at every Write end and
    the offsets stage replays Deflate.Compress's 512-byte chunk accounting for the markers and the extra empty blocks."""
    for name in ("alice_98304", "lowent_65537", "zeros_98305", "alice_5", "alice_0"):
        for flush in (1, 2, 3):
            for w in (0, 100, 4000, 8192, 70000):
                for level in (0, 1, 6):
                    run(model[name], level, 0, "chunk" if level >= 4 else "seq", w, flush)
            run(model[name], 6, 3, "chunk", 5000, flush)   # Rle
            run(model[name], 9, 4, "chunk", 3137, flush)   # Fixed


def test_regular_multi_write_takes_the_chunked_form(model):
    """NoFlush Writes whose sizes are multiples of 2048 (Stream.CopyTo's 81920 and the like): one read event per Write end,
    handled like the window-full refills -- including events whose loop-top shares its bucket with the next position
    (zeros, runs)."""
    def one(name, w, level):
        r = subprocess.run([EXE, model[name], str(level), "0", "chunk", str(w), "0"], capture_output=True, text=True)
        assert r.returncode == 0 and "PASS" in r.stdout, (name, w, level, r.stdout[-400:])
        if os.path.getsize(model[name]) > w + 600:
            assert "tail_from=0 " not in r.stdout + " ", (name, w, level)   # the bulk form really ran
    with batch() as b:
        for name in ("alice_98304", "alice_98305", "lowent_98043", "zeros_98305", "zeros_65541", "runs", "ptt5"):
            for w in (2048, 4096, 8192, 32768, 65536, 81920):
                for level in (4, 6, 9):
                    if level == 9 and name == "ptt5":
                        continue
                    b.futures.append(b.pool.submit(one, name, w, level))


def test_any_write_sizes_take_the_chunked_form(model, tmp_path):
    """NoFlush Writes of any size (ZlibOutputStream.cs:114-168: every Write end is a read event of Fill_window,
    Deflate.cs:967-1019): zs_core.h build_geometry cuts the parse at the clusters of read boundaries and the chunked form
    steps through a cluster's events per entry slot -- Write ends in the last 262 bytes of a window (whether the window
    slides there depends on the loop-top), Writes shorter than MIN_LOOKAHEAD mixed in, first Writes of a few bytes, scanline
    sized Writes; streams written a few bytes at a time throughout are left to the literal engine.  Bytes, symbols, blocks
    and event loop-tops against the oracle."""
    rng = np.random.default_rng(21)
    big = tmp_path / "low300k"
    big.write_bytes(rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 300000).tobytes())
    files = dict(model)
    files["low300k"] = str(big)
    bulk = 0
    for name in ("alice_98305", "lowent_98043", "zeros_98305", "runs", "low300k"):
        for spec in ("1000", "16385", "81921", "3000", "r1:263:3000", "r2:1:5000", "r3:100:600", "5000,3", "65530,4,1000", "40000,2,25534,5", "32766,2",
                     "100,100,100,5000", "263", "262"):
            for level in (4, 6, 9):
                if level != 6 and (name in ("runs", "zeros_98305") or spec not in ("1000", "r3:100:600", "5000,3", "32766,2")):
                    continue
                r = subprocess.run([EXE, files[name], str(level), "0", "chunk", spec, "0"], capture_output=True, text=True)
                assert r.returncode == 0 and "PASS" in r.stdout, (name, spec, level, r.stdout[-400:])
                bulk += "tail_from=0 " not in r.stdout + " "
    assert bulk > 70   # the chunked form really ran for most of them (263- and 262-byte Writes are one long cluster)


def test_level0_block_plan_equals_the_literal_engine(model, tmp_path):
    """zs_core.h plan_stored_blocks (what the device uses at level 0: the blocks of DeflateStored from the sizes alone)
    against the literal engine and the oracle's bytes: single Write, Writes on and off any grid, all flush modes, and
    sizes around the window / block / slide boundaries."""
    rng = np.random.default_rng(3)
    for name in ("alice_0", "alice_1", "alice_5", "alice_600", "alice_65274", "alice_65536", "alice_65537", "alice_98304", "zeros_98305"):
        run(model[name], 0, 0, "seq")
    big = tmp_path / "big0"
    big.write_bytes(rng.integers(0, 256, 700001, dtype=np.uint8).tobytes())
    run(str(big), 0, 0, "seq")
    for wchunk in (1, 7, 333, 32763, 32768, 65536, 81920, 100000):
        if wchunk < 333:
            run(model["alice_600"], 0, 0, "seq", wchunk)
        else:
            run(str(big), 0, 0, "seq", wchunk)
        for flush in (1, 2, 3):
            run(model["alice_98304"] if wchunk < 333 else str(big), 0, 0, "seq", max(wchunk, 100), flush)
    for s in (2, 4):  # strategies other than Rle do not change DeflateStored
        run(str(big), 0, s, "seq", 50000)


def test_incremental_runs_of_the_literal_engine(model, tmp_path):
    """The engine run Write by Write -- suspended where Deflate.Compress returns for more input (NeedMore under NoFlush,
    BlockDone after a flush), re-entered with the next Write, Finish as a last run without input -- gives the oracle's
    bytes: every engine (stored, fast, slow, Rle), every flush mode, Writes longer and shorter than MIN_LOOKAHEAD."""
    rng = np.random.default_rng(5)
    mixed = tmp_path / "mixed"
    alice = open(os.path.join(ROOT, "tests/golden/corpus/alice29.txt"), "rb").read()
    mixed.write_bytes(alice[:90000] + bytes(40000) + rng.integers(0, 256, 50000, dtype=np.uint8).tobytes() + alice[:70000])
    for level, strategy in ((0, 0), (1, 0), (3, 0), (4, 0), (6, 0), (9, 0), (6, 3), (6, 2)):  # level 0 + Rle overflows the reference's pending buffer on such data
        for wchunk in (100, 261, 5000, 32768, 65536, 100000):
            for flush in (0, 2) if wchunk in (261, 32768) else (0, 1, 2, 3):
                run(str(mixed), level, strategy, "inc", wchunk, flush)
    run(model["alice_600"], 6, 0, "inc", 1, 0)
    run(model["alice_600"], 6, 0, "inc", 7, 2)
    run(model["zeros_98305"], 6, 0, "inc", 4096, 1)
    run(model["alice_0"], 6, 0, "inc", 0, 0)


def test_tail_records_match_the_engine_search(model, tmp_path):
    """The searches of a stream's last loop-tops done ahead of the tail engine's parse (le_tail_record, zs_lit_engine.h;
    Longest_match Deflate.cs:1022-1100 with nice clipped to the lookahead and the length capped by it): tails around the
    window-slide points, periodic data, matches that run into the data end; levels 4-9, default / Filtered / Fixed.
    ZS_NO_TAIL_RECORDS=1 is the engine searching every position itself: both must be the oracle's symbols."""
    import random
    rng = random.Random(11)
    alice = open(os.path.join(ROOT, "tests/golden/corpus/alice29.txt"), "rb").read()
    npr = np.random.default_rng(5)
    sizes = [300, 520, 5000, 32768 + 261, 65274, 65275, 65535, 65536 + 200, 65536 + 262, 98304 - 100, 131072 - 261]

    def gen(kind, n):
        if kind == 0:
            o = rng.randrange(0, len(alice))
            return (alice * 3)[o:o + n]
        if kind == 1:
            pat = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
            return (pat * (n // len(pat) + 1))[:n]
        if kind == 2:
            return npr.integers(0, 4, n, dtype=np.uint8).tobytes()
        b = bytearray((alice * 3)[7:7 + n])  # the last few hundred bytes repeat an earlier stretch
        k = rng.randrange(10, 600)
        if n > 2 * k + 10:
            src = rng.randrange(0, n - 2 * k)
            b[n - k:] = b[src:src + k]
        return bytes(b)
    for it in range(24):
        n = rng.choice(sizes) + rng.randrange(-3, 4)
        p = tmp_path / ("tail_%d" % it)
        p.write_bytes(gen(it % 4, n))
        for level in (4, 6, 7, 9):
            run(str(p), level, rng.choice([0, 0, 1, 4]))
    os.environ["ZS_NO_TAIL_RECORDS"] = "1"
    try:
        run(str(tmp_path / "tail_3"), 6)
    finally:
        del os.environ["ZS_NO_TAIL_RECORDS"]
