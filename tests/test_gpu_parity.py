"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the oracle on the
same inputs -- bit-exact, every byte.  At BASELINE.json's full sizes (64 MiB) the oracle comparison
is replaced by size-independent properties (inflate round trip through an independent decoder,
Adler trailer) plus an oracle comparison of a sample, to keep the suite short."""
import hashlib
import io
import json
import os
import zlib

import numpy as np
import pytest

import oracle_binding
from zlibstream_amd import (CompressionLevel, CompressionStrategy, ZlibOptions, ZlibOutputStream, ZlibStreamException, datagen,
                            deflate_bound)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIG = json.load(open(os.path.join(GOLD, "oracle_digests.json")))
KAT = json.load(open(os.path.join(GOLD, "kat_sizes.json")))
CORPUS = sorted(os.listdir(oracle_binding.CORPUS))


def test_native_library_is_loaded(engine):
    maps = open("/proc/self/maps").read()
    assert "libzsgpu.so" in maps


@pytest.mark.parametrize("name", CORPUS)
def test_corpus_matches_committed_digests(engine, name):
    d = oracle_binding.corpus(name)
    levels = [4, 5, 6, 7, 8, 9] if len(d) < 600000 else [4, 6, 8]
    if len(d) < 200000:
        levels += [1, 2, 3]
    outs = engine.deflate_batch([d] * len(levels), level=6) if False else [engine.deflate_batch([d], level=l)[0] for l in levels]
    for lvl, z in zip(levels, outs):
        size, sha = DIG["oracle"]["%s:%d" % (name, lvl)]
        assert len(z) == size, (name, lvl)
        assert hashlib.sha256(z).hexdigest() == sha, (name, lvl)


@pytest.mark.parametrize("name", [n for n in CORPUS if n in oracle_binding.CRLF_FILES or n in ("cp.html", "sum", "ptt5")])
def test_published_sizes_on_device(engine, name):
    d = oracle_binding.corpus(name, canonical=True)
    z = engine.deflate_batch([d], level=6)[0]
    assert len(z) == KAT[name][2]  # benchmarks.md level-6 column
    assert zlib.decompress(z) == d


def _edge_inputs():
    rng = np.random.default_rng(11)
    alice = oracle_binding.corpus("alice29.txt") * 2
    out = {}
    for n in (0, 1, 2, 3, 4, 5, 6, 261, 262, 263, 520, 65274, 65275, 65276, 65531, 65535, 65536, 65537, 65541, 65798, 98043, 98304,
              98305, 98566, 131072):
        out["alice_%d" % n] = alice[:n]
        out["zeros_%d" % n] = bytes(n)
        out["lowent_%d" % n] = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes()
    out["random_100k"] = rng.integers(0, 256, 100000, dtype=np.uint8).tobytes()
    out["runs"] = np.repeat(rng.integers(0, 4, 60000, dtype=np.uint8), rng.integers(1, 40, 60000))[:400000].tobytes()
    out["period256"] = (bytes(range(256)) * 2000)[:500000]
    out["sparse_256"] = datagen.sparse(256, 256)
    return out


def test_edge_sizes_and_refill_quirks(engine, oracle):
    """Empty / tiny inputs, the 262-byte tail, window-slide boundaries (64 KiB, 96 KiB), zero runs that put
    equal-bucket positions on refill loop-tops, stale bytes past the end of input."""
    inputs = _edge_inputs()
    names = sorted(inputs)
    for lvl in (6, 4, 9):
        got = engine.deflate_batch([inputs[k] for k in names], level=lvl)
        for k, z in zip(names, got):
            assert z == oracle.compress(inputs[k], lvl), (k, lvl)


@pytest.mark.parametrize("strategy", [CompressionStrategy.Filtered, CompressionStrategy.HuffmanOnly, CompressionStrategy.Fixed])
def test_strategies(engine, oracle, strategy):
    for name in ("alice29.txt", "sum", "ptt5"):
        d = oracle_binding.corpus(name)[:200000]
        assert engine.deflate_batch([d], level=6, strategy=int(strategy))[0] == oracle.compress(d, 6, int(strategy))


def test_fast_levels_run_on_device(engine, oracle):
    for name in ("cp.html", "fields.c", "xargs.1"):
        d = oracle_binding.corpus(name)
        for lvl in (1, 2, 3):
            assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl)


def test_fast_levels_under_huffman_only_and_filtered(engine, oracle):
    """DeflateFast never calls Longest_match under HuffmanOnly (Deflate.Fast.cs:61-66) -- also not at the position a read
    inserted ahead when it shares its bucket with the loop-top (the one search that sees a single candidate): zeros, where
    every read is such a one, and streams past the first window end (found by tools/fuzz_batch.py)."""
    rng = np.random.default_rng(5)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 150000).tobytes()
    for d in (bytes(98304), bytes(150000), low, oracle_binding.corpus("alice29.txt")[:150000], bytes(65536), bytes(65537)):
        for lvl in (1, 2, 3):
            for strategy in (2, 1, 4):
                assert engine.deflate_batch([d], level=lvl, strategy=strategy)[0] == oracle.compress(d, lvl, strategy), (len(d), lvl, strategy)


def test_mul_hash_variant(engine, oracle):
    d = oracle_binding.corpus("alice29.txt")
    assert engine.deflate_batch([d], level=6, hash_variant=1)[0] == oracle.compress(d, 6, hash_variant=1)


def test_batch_of_ragged_buffers(engine, oracle):
    rng = np.random.default_rng(3)
    text = oracle_binding.corpus("lcet10.txt")
    bufs = []
    for i in range(96):
        n = int(rng.integers(0, 150000))
        o = int(rng.integers(0, len(text) - n))
        bufs.append(text[o:o + n] if i % 3 else datagen.sparse(64, max(1, n // 256), y0=i))
    got = engine.deflate_batch(bufs, level=6)
    for b, z in zip(bufs, got):
        assert z == oracle.compress(b, 6)


# ---- the reference's own tests against the device path (ZlibStreamTests.Roundtrip.cs:25-125) ----
@pytest.mark.parametrize("level", [CompressionLevel.NoCompression, CompressionLevel.Level1, CompressionLevel.Level2, CompressionLevel.Level3,
                                   CompressionLevel.Level4, CompressionLevel.Level5, CompressionLevel.Level6, CompressionLevel.Level7,
                                   CompressionLevel.BestCompression, CompressionLevel.DefaultCompression])
def test_encode_decode(engine, oracle, level):
    expected = oracle.dotnet_random(1, 2 * 4096 * 4)
    for strategy in CompressionStrategy:  # all five, like the reference's foreach over Enum.GetValues
        compressed = io.BytesIO()
        with ZlibOutputStream(compressed, ZlibOptions(CompressionLevel=level, CompressionStrategy=strategy), engine=engine) as deflate:
            deflate.write(expected)
        z = compressed.getvalue()
        assert zlib.decompress(z) == expected                      # independent decoder (SharpZipLib in the reference)
        assert z == oracle.compress(expected, int(level), int(strategy))


@pytest.mark.parametrize("level", [CompressionLevel.NoCompression, CompressionLevel.Level1, CompressionLevel.Level6,
                                   CompressionLevel.BestCompression])
def test_encode_decode_per_chunk(engine, oracle, level):
    count, chunk = 2 * 4096 * 4, 2 * 4096
    expected = oracle.dotnet_random(1, count)
    text = oracle_binding.corpus("alice29.txt")[:count * 3]
    for data in (expected, text):
        compressed = io.BytesIO()
        with ZlibOutputStream(compressed, level, engine=engine) as deflate:
            for i in range(0, len(data), chunk):
                deflate.write(data[i:i + chunk])
        z = compressed.getvalue()
        assert zlib.decompress(z) == data
        chunks = [min(chunk, len(data) - i) for i in range(0, len(data), chunk)]
        assert z == oracle.compress(data, int(level), chunks=chunks)   # Write boundaries are read events: bytes depend on them


def test_stored_level_and_rle_strategy(engine, oracle):
    """Level 0 (Deflate.Stored.cs) and CompressionStrategy.Rle (Deflate.Rle.cs) run on the device's literal engine."""
    for name in ("alice29.txt", "ptt5", "sum"):
        d = oracle_binding.corpus(name)[:300000]
        for level, strategy in ((0, 0), (0, 3), (6, 3), (1, 3), (0, 4)):
            if name == "ptt5" and (level, strategy) == (0, 3):
                continue
            assert engine.deflate_batch([d], level=level, strategy=strategy)[0] == oracle.compress(d, level, strategy), (name, level, strategy)
    z = engine.deflate_batch([bytes(200000)], level=0, strategy=3)[0]   # level 0 + Rle: static-tree blocks once the block start slid out
    assert z == oracle.compress(bytes(200000), 0, 3) and len(z) < 2000
    # level 0 + Rle on data whose first block spans > 32 KiB before the window slides: the reference overflows its
    # 32 KiB pending buffer (BlockCopy throws); the oracle reports it and the device path refuses too
    d = oracle_binding.corpus("ptt5")[:300000]
    with pytest.raises(RuntimeError):
        oracle.compress(d, 0, 3)
    with pytest.raises(ZlibStreamException):
        engine.deflate_batch([d], level=0, strategy=3)


def test_stream_api_errors_and_unsupported(engine):
    import ctypes
    L = engine._lib
    d = L.zs_deflate_init(engine._h, 6, 0, 12, 8, 0)  # windowBits 12 is not on the device path: loud, no CPU fallback
    assert d
    ai, ao, out = ctypes.c_int32(3), ctypes.c_int32(512), ctypes.create_string_buffer(512)
    assert L.zs_deflate(d, b"abc", ctypes.byref(ai), out, ctypes.byref(ao), 0, None, None, None) == -2
    assert b"windowBits" in L.zs_last_message(d)
    L.zs_deflate_end(d)
    with pytest.raises(ValueError):
        ZlibOutputStream(io.BytesIO(), 12, engine=engine)
    s = ZlibOutputStream(io.BytesIO(), CompressionLevel.Level6, engine=engine)
    s.write(b"")  # empty Write is a no-op (WriteCore)
    s.close()
    assert s.BaseStream.getvalue() == bytes.fromhex("789c030000000001")


def test_out_cap_too_small_is_buf_error(engine):
    import ctypes
    d = oracle_binding.corpus("sum")
    src = ctypes.create_string_buffer(d, len(d))
    dst = ctypes.create_string_buffer(100)
    rc, lens, status = engine._call_batch(engine._lib.zs_deflate_batch, [ctypes.addressof(src)], [len(d)], [ctypes.addressof(dst)],
                                          [100], 6, 0, 0)
    assert rc == -5 and status[0] == -5


def test_device_adler(engine):
    import torch
    d = oracle_binding.corpus("kennedy.xls")
    t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    import ctypes
    out = ctypes.c_uint32(0)
    assert engine._lib.zs_adler32_device(engine.handle, t.data_ptr(), len(d), 1, ctypes.byref(out), None) == 0
    assert out.value == zlib.adler32(d)


def test_device_resident_buffers_and_full_size_properties(engine, oracle):
    """BASELINE configs 2 and 3 at full size: 64 MiB english (L6) and 64 MiB sparse (L6), inputs resident in HBM."""
    import torch
    for name, data in (("english64", datagen.english(64 << 20)), ("sparse64", datagen.sparse(4096, 4096))):
        n = len(data)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        cap = deflate_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        out_len = engine.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=6,
                                              stream=torch.cuda.current_stream().cuda_stream)[0]
        z = d_out[:out_len].cpu().numpy().tobytes()
        assert zlib.decompress(z) == data, name                     # round trip through an independent decoder
        assert int.from_bytes(z[-4:], "big") == zlib.adler32(data)   # trailer
        assert z[:2] == b"\x78\x9c"
        # oracle on a bounded sample of the same workload (the first 4 MiB as its own stream)
        small = data[:4 << 20]
        assert engine.deflate_batch([small], level=6)[0] == oracle.compress(small, 6), name


def test_sparse_levels_4_and_9(engine, oracle):
    d = datagen.sparse(1024, 1024)
    for lvl in (4, 9):
        assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl)


# ---- inflate (SURVEY a16 / BASELINE config 5): any conformant decode is bit-exact by construction ----
def test_inflate_roundtrip_of_device_streams(engine):
    names = ["alice29.txt", "ptt5", "kennedy.xls", "sum", "xargs.1"]
    datas = [oracle_binding.corpus(n) for n in names] + [b"", b"a", bytes(70000), datagen.sparse(128, 128)]
    for lvl in (6, 9):
        zs = engine.deflate_batch(datas, level=lvl)
        outs = engine.inflate_batch(zs, [len(d) for d in datas])
        assert outs == datas


def test_inflate_foreign_streams(engine, oracle):
    """Streams made by another encoder (Python zlib): stored, fixed and dynamic blocks, all window sizes."""
    rng = np.random.default_rng(5)
    text = oracle_binding.corpus("lcet10.txt")
    cases = [zlib.compress(text, 0), zlib.compress(text, 1), zlib.compress(text, 9), zlib.compress(rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(), 6)]
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    cases.append(co.compress(text[:100000]) + co.flush())
    co = zlib.compressobj(6, zlib.DEFLATED, 9)
    cases.append(co.compress(text[:50000]) + co.flush())
    want = [zlib.decompress(z) for z in cases]
    got = engine.inflate_batch(cases, [len(w) for w in want])
    assert got == want
    for z, w in zip(cases, want):
        rc, out, msg = oracle.inflate(z, len(w))
        assert rc == 1 and out == w


def test_inflate_errors_match_reference_messages(engine, oracle):
    d = oracle_binding.corpus("fields.c")
    z = engine.deflate_batch([d], level=6)[0]
    bad_check = z[:-1] + bytes([z[-1] ^ 1])
    bad_hdr = b"\x78\x9d" + z[2:]
    bad_method = b"\x79\x9c" + z[2:]
    truncated = z[:len(z) // 2]
    for stream, cap in ((bad_check, len(d)), (bad_hdr, len(d)), (bad_method, len(d)), (truncated, len(d)), (z, len(d) - 10)):
        rc, _, msg = oracle.inflate(stream, cap)
        with pytest.raises(ZlibStreamException) as ei:
            engine.inflate_batch([stream], [cap])
        assert rc != 1
        assert str(ei.value) == "inflating: " + msg, (str(ei.value), msg)


def test_zlib_input_stream_mirror(engine):
    from zlibstream_amd import ZlibInputStream
    d = oracle_binding.corpus("asyoulik.txt")
    z = engine.deflate_batch([d], level=6)[0]
    s = ZlibInputStream(io.BytesIO(z), engine=engine)
    assert s.read(1000) == d[:1000]
    assert s.read() == d[1000:]
    assert s.TotalIn == len(z) and s.TotalOut == len(d) and s.Adler == zlib.adler32(d)
    assert s.read(10) == b""  # past the end of the stream


def test_zlib_input_stream_ends_with_the_stream_whatever_follows(engine):
    """Inflate.Decompress ends at the final block's Adler trailer (Inflate.cs:292-357) and ZlibInputStream.ReadCore stops
    there (ZlibInputStream.cs:133-186): bytes behind the stream in BaseStream are not the stream's.  zs_inflate looks for the
    end in what it has buffered as the input arrives: the payload is returned, TotalIn is the stream's length, and a long
    stream is delivered before BaseStream has been read to its end."""
    from zlibstream_amd import ZlibInputStream
    rng = np.random.default_rng(3)
    garbage = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    for d in (oracle_binding.corpus("asyoulik.txt"), datagen.english(8 << 20, 5), b"", b"x" * 70000):
        z = engine.deflate_batch([d], level=6)[0]
        for tail in (garbage, z, b"\0" * 5000):
            base = io.BytesIO(z + tail)
            s = ZlibInputStream(base, engine=engine)
            assert s.read() == d
            assert s.TotalIn == len(z) and s.TotalOut == len(d) and s.Adler == zlib.adler32(d)
            assert s.read(10) == b""
            if len(z) >= (1 << 20) and len(tail) >= (1 << 20):
                assert base.tell() < len(z) + len(tail)   # the end was found before BaseStream ran out


def test_zs_inflate_stream_protocol(engine):
    """zs_inflate under the ReadCore cadence (8 KiB input chunks, caller-sized output): several sizes incl. a stream
    that expands ~1000x (output buffer grown on the device side), a truncated stream (ZBUFERROR, as the managed engine
    reports when it cannot make progress) and a corrupted one (ZDATAERROR with the reference's message)."""
    from zlibstream_amd import ZlibInputStream
    for d in (b"", b"a", oracle_binding.corpus("fields.c"), bytes(5 << 20), datagen.english(3 << 20)):
        z = zlib.compress(d, 6)
        s = ZlibInputStream(io.BytesIO(z), engine=engine)
        got = bytearray()
        while True:
            part = s.read(70000)
            if not part:
                break
            got += part
        assert bytes(got) == d, len(d)
    z = zlib.compress(oracle_binding.corpus("alice29.txt"), 6)
    with pytest.raises(ZlibStreamException, match="inflating: buffer error"):
        ZlibInputStream(io.BytesIO(z[:len(z) // 2]), engine=engine).read()
    bad = bytearray(z)
    bad[-1] ^= 0x55
    with pytest.raises(ZlibStreamException, match="inflating: incorrect data check"):
        ZlibInputStream(io.BytesIO(bytes(bad)), engine=engine).read()


def test_inflate_large_device_resident(engine):
    import torch
    data = datagen.english(8 << 20)
    z = engine.deflate_batch([data], level=6)[0]
    d_in = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    d_out = torch.empty(len(data), dtype=torch.uint8, device="cuda")
    engine.set_profiling(True)
    n = engine.inflate_batch_device([d_in.data_ptr()], [len(z)], [d_out.data_ptr()], [len(data)],
                                    stream=torch.cuda.current_stream().cuda_stream)[0]
    stages = engine.stage_ms()
    engine.set_profiling(False)
    assert n == len(data) and d_out.cpu().numpy().tobytes() == data
    # the block-parallel path must have done it: a regular stream that silently falls back to the one-wave decoder is a bug
    assert stages.get("inf_decode", 0) > 0 and stages.get("inf_measure", 0) > 0, stages
    # many streams at once: enough candidates for the lane form of the measure pass
    zs = [engine.deflate_batch([datagen.english(6 << 20, 50 + i)], level=6)[0] for i in range(2)] * 32  # ~8000 candidates
    srcs = [torch.frombuffer(bytearray(x), dtype=torch.uint8).cuda() for x in zs]
    outs = [torch.empty(6 << 20, dtype=torch.uint8, device="cuda") for _ in zs]
    engine.set_profiling(True)
    lens = engine.inflate_batch_device([t.data_ptr() for t in srcs], [len(x) for x in zs], [o.data_ptr() for o in outs],
                                       [6 << 20] * len(zs), stream=torch.cuda.current_stream().cuda_stream)
    stages = engine.stage_ms()
    engine.set_profiling(False)
    assert stages.get("inf_decode", 0) > 0, stages
    for i, o in enumerate(outs):
        assert lens[i] == 6 << 20 and zlib.adler32(o.cpu().numpy().tobytes()) == int.from_bytes(zs[i][-4:], "big")


def test_inflate_block_parallel_path_on_foreign_large_streams(engine):
    """Large streams from another encoder through the block-parallel path: blocks the finder cannot see (stored, fixed)
    between dynamic ones, empty stored blocks from flushes (the next candidate is then not the block's end), tiny and huge
    blocks (zeros: 4 MiB of output per block; level 1: other block lengths), data that does not compress (stored only:
    the one-wave decoder takes it).  Bytes and lengths must be those of zlib."""
    rng = np.random.default_rng(11)
    text = datagen.english(3 << 20, 77)
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    zeros = bytes(24 << 20)
    cases = []
    for lvl in (1, 6, 9):
        cases.append(zlib.compress(text, lvl))
    cases.append(zlib.compress(zeros, 6))
    cases.append(zlib.compress(noise, 6))                       # stored blocks only
    cases.append(zlib.compress(text[:1 << 20] + noise + text[1 << 20:] + zeros[:1 << 20], 6))  # dynamic / stored / dynamic
    co = zlib.compressobj(6)
    parts = []
    for off in range(0, len(text), 300000):                     # sync / full flushes: empty stored blocks between the dynamic ones
        parts.append(co.compress(text[off:off + 300000]))
        parts.append(co.flush(zlib.Z_FULL_FLUSH if (off // 300000) % 2 else zlib.Z_SYNC_FLUSH))
    parts.append(co.flush())
    cases.append(b"".join(parts))
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)  # fixed blocks only: measured by the chain kernel, wave decoder
    cases.append(co.compress(text[:1 << 20]) + co.flush())
    co = zlib.compressobj(9, zlib.DEFLATED, 15, 9)
    cases.append(co.compress(datagen.sparse(1024, 1024)) + co.flush())
    want = [zlib.decompress(z) for z in cases]
    got = engine.inflate_batch(cases, [len(w) for w in want])
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_ == w_, (i, len(g_), len(w_))
    # the chain of a stream's blocks is found without a walk when the finder reported them all (zs_inf_chain_par_kernel:
    # the first three cases and the zeros) and by the walking kernel otherwise; forced to walk, the same bytes
    os.environ["ZS_INF_CHAIN_WALK"] = "1"
    try:
        assert engine.inflate_batch(cases, [len(w) for w in want]) == want
    finally:
        del os.environ["ZS_INF_CHAIN_WALK"]
    # a corrupted byte in the middle of a large stream is a data error here as it is in zlib (message classes are covered
    # by test_inflate_errors_match_reference_messages)
    bad = bytearray(cases[1])
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(ZlibStreamException):
        engine.inflate_batch([bytes(bad)], [len(want[1])])


@pytest.mark.parametrize("seed", [int(x) for x in os.environ.get("ZS_FUZZ_SEEDS", "2024,7").split(",")])
def test_inflate_fuzz_foreign_streams(engine, seed):
    """Randomised streams from another encoder (levels, strategies, window sizes, memLevels, flush points of every kind,
    data of every compressibility), large enough for the block-parallel path and small enough for the one-wave decoder,
    in one batch: bytes and lengths must be zlib's."""
    rng = np.random.default_rng(seed)
    text = datagen.english(2 << 20, 5)
    def piece(n):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            o = int(rng.integers(0, len(text) - n))
            return text[o:o + n]
        if kind == 1:
            return bytes(n)
        if kind == 2:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 3:
            period = bytes(rng.integers(0, 256, int(rng.integers(1, 40000)), dtype=np.uint8))
            return (period * (n // len(period) + 1))[:n]
        return rng.integers(0, 4, n, dtype=np.uint8).tobytes()  # low entropy: long codes are short, blocks are long
    cases, want = [], []
    for i in range(28):
        total = int(rng.integers(1000, 3 << 20)) if i % 4 else int(rng.integers(400000, 6 << 20))
        data = b"".join(piece(int(rng.integers(1, 400000))) for _ in range(1 + total // 200000))[:total]
        level = int(rng.integers(0, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 5))]
        co = zlib.compressobj(level, zlib.DEFLATED, int(rng.integers(9, 16)), int(rng.integers(1, 10)), strategy)
        parts, off = [], 0
        while off < len(data):
            n = int(rng.integers(1, 500000))
            parts.append(co.compress(data[off:off + n]))
            off += n
            mode = int(rng.integers(0, 6))
            if mode < 3 and off < len(data):
                parts.append(co.flush([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_BLOCK][mode]))
        parts.append(co.flush())
        cases.append(b"".join(parts))
        want.append(data)
    got = engine.inflate_batch(cases, [len(w) for w in want])
    for i, (g_, w_) in enumerate(zip(got, want)):
        assert g_ == w_, (i, len(g_), len(w_), len(cases[i]))


def test_cpp_host_mirror(tmp_path):
    """The C++ host-side mirror of ZlibOutputStream / ZlibInputStream (include/zsgpu.hpp) replays the reference's
    EncodeDecode / EncodeDecodePerChunk tests through the C ABI; bytes are checked against the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_host_mirror")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(root, "tests/cpp/test_host_mirror.cpp"),
                    os.path.join(root, "oracle/zs_oracle.c"), os.path.join(root, "oracle/zs_inflate_oracle.c"),
                    "-L" + os.path.join(root, "zlibstream_amd"), "-lzsgpu", "-Wl,-rpath," + os.path.join(root, "zlibstream_amd")],
                   check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rle_strategy_over_the_chip(engine, oracle, rate_floors):
    """CompressionStrategy.Rle (Deflate.Rle.cs:18-104) off the literal engine: a match never leaves the run of equal bytes it
    lies in, so a position's part in the parse follows from where its run began (zs_rle.h; kernels zs_rle.hip: a prefix maximum
    over the run starts, two passes over the positions).  Bytes against the oracle at levels 1 / 6 / 9 on text (runs of one),
    image rows, a bitmap, long and short runs, zeros, sizes around the window ends and the refill threshold (258 here, not
    262); streams of several Writes stay with the literal engine (same bytes); 64 MiB of image rows and of text at 5 GB/s
    and more (the literal engine: 0.5-2 MB/s)."""
    import time
    import torch
    rng = np.random.default_rng(17)
    cases = {
        "alice": oracle_binding.corpus("alice29.txt"), "ptt5": oracle_binding.corpus("ptt5"), "kennedy": oracle_binding.corpus("kennedy.xls"),
        "sparse": datagen.sparse(512, 300), "zeros": bytes(300000),
        "long runs": np.repeat(rng.integers(0, 4, 40000, dtype=np.uint8), rng.integers(1, 700, 40000))[:2000000].tobytes(),
        "short runs": np.repeat(rng.integers(0, 3, 400000, dtype=np.uint8), rng.integers(1, 6, 400000))[:700001].tobytes(),
    }
    for n in (4096 + 786, 5000, 65536, 65536 + 258, 65536 + 32768 - 258 + 600, 98304 + 600, 131072 + 522, 131072 + 786 + 258):
        cases["zeros %d" % n] = bytes(n)
        cases["runs %d" % n] = np.repeat(rng.integers(0, 3, n, dtype=np.uint8), rng.integers(1, 400, n))[:n].tobytes()
    for name, d in cases.items():
        for lvl in (1, 6, 9):
            assert engine.deflate_batch([d], level=lvl, strategy=3)[0] == oracle.compress(d, lvl, 3), (name, lvl)
    # a batch of them, and one written in several Writes (the literal engine)
    bufs = [cases[k] for k in ("alice", "sparse", "zeros", "short runs", "runs 5000")]
    assert engine.deflate_batch(bufs, level=6, strategy=3) == [oracle.compress(b, 6, 3) for b in bufs]
    d = cases["long runs"][:300000]
    ends = _write_ends(len(d), 70000, None)
    z, _ = _deflate_writes(engine, d, ends, 6, strategy=3)
    assert z == oracle.compress(d, 6, 3, chunks=[ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))])
    for name, data in (("sparse64", datagen.sparse(4096, 4096)), ("english64", datagen.english(64 << 20, datagen.GOLDEN))):
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        cap = deflate_bound(len(data))
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        engine.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=6, strategy=3)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = engine.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=6, strategy=3)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        z = d_out[:m].cpu().numpy().tobytes()
        assert zlib.decompress(z) == data
        assert z[:1 << 20] == oracle.compress(data[:8 << 20], 6, 3)[:1 << 20], name   # (the stream's first MiB: the same blocks)
        rate_floors.check(len(data) / dt >= 5e9, "%s under Rle: %.1f ms = %.2f GB/s" % (name, dt * 1e3, len(data) / dt / 1e9))


def test_fast_levels_single_stream_rate(engine, oracle, rate_floors):
    """DeflateFast on ONE text stream (Deflate.Fast.cs:20-128; the reference does 54.8 / 36.9 MB/s at levels 1 / 3 on its
    2018 laptop core, benchmarks.md:63,118; the oracle on the GPU box's host 79 / 50): as rounds over the stream's chunks
    zs_fast_sweep_kernel holds 289 / 437 MB/s on 8 MiB of text resident in HBM (one workgroup for the whole stream: 48 / 21;
    round 3's one-wave form: 9.1 / 4.0).  The floors below leave a third of margin for a busy box; the bytes of a 2 MiB prefix
    are the oracle's, and an 8 MiB text stream must not go through the speculative chunk runs first (zs_fast_probe_kernel:
    they never verify on text)."""
    import time
    import torch
    text = datagen.english(8 << 20, 77)
    d_in = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(text))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for lvl, floor in ((1, 190e6), (3, 290e6)):
        assert engine.deflate_batch([text[:2 << 20]], level=lvl)[0] == oracle.compress(text[:2 << 20], lvl), lvl
        engine.deflate_batch_device([d_in.data_ptr()], [len(text)], [d_out.data_ptr()], [cap], level=lvl)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = engine.deflate_batch_device([d_in.data_ptr()], [len(text)], [d_out.data_ptr()], [cap], level=lvl)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert zlib.decompress(d_out[:m].cpu().numpy().tobytes()) == text
        rate_floors.check(len(text) / dt >= floor, "level %d: %.1f ms = %.1f MB/s" % (lvl, dt * 1e3, len(text) / dt / 1e6))


@pytest.mark.gpu
def test_fast_levels_as_rounds_over_chunks_and_as_one_workgroup_per_stream(engine, oracle):
    """DeflateFast (Deflate.Fast.cs:20-128) in both forms of zs_fast_sweep_kernel: as rounds over the chunks of a stream
    (zs_fast_sweep.h "Rounds": few streams, or one much longer than the rest) and with one workgroup per stream (a batch of
    equals; ZS_FAST_NO_ROUNDS forces it).  Bytes against the oracle on text, zeros and few-symbol data (every read event an
    equal-bucket one: the cut of its chain travels from chunk to chunk), runs, a bitmap, a spreadsheet, sizes around the first
    window ends; chunk sizes from the smallest to what one staging of the tile covers; HuffmanOnly and Filtered; a batch that
    mixes long and short streams; a batch of equals."""
    rng = np.random.default_rng(41)
    alice = oracle_binding.corpus("alice29.txt")
    cases = {
        "text300k": (alice * 2)[:300000], "zeros200k": bytes(200000),
        "lowent": rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 180000).tobytes(),
        "runs": np.repeat(rng.integers(0, 4, 60000, dtype=np.uint8), rng.integers(1, 40, 60000))[:400000].tobytes(),
        "ptt5": oracle_binding.corpus("ptt5"), "kennedy": oracle_binding.corpus("kennedy.xls"),
        "english1m": datagen.english(1 << 20, 5),
    }
    for n in (262, 263, 2500, 39996, 65274, 65275, 65276, 65537, 98043, 98305):  # (39 996 zeros: the body hands over four bytes in front of the end)
        cases["alice_%d" % n] = (alice * 2)[:n]
        cases["zeros_%d" % n] = bytes(n)
    # (ZS_FR_RANGE: a workgroup takes that many consecutive chunks in turn, each reading what the ones before it have just left;
    # the engine lowers the ranges as fewer chunks change unless ZS_FR_RANGE_FIXED is set)
    envs = ({}, {"ZS_FR_CHUNK": "2048"}, {"ZS_FR_CHUNK": "10240"}, {"ZS_FAST_NO_ROUNDS": "1"}, {"ZS_FR_CHUNK": "2048", "ZS_FR_RANGE": "3"},
            {"ZS_FR_CHUNK": "4096", "ZS_FR_RANGE": "5", "ZS_FR_RANGE_FIXED": "1"})
    try:
        for ei, env in enumerate(envs):
            os.environ.update(env)
            for name, d in cases.items():
                for lvl in (1, 2, 3):
                    if ei and len(d) > 400000 and lvl == 2:
                        continue
                    assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl), (env, name, lvl)
            for strategy in (1, 2):
                for name in ("text300k", "zeros200k", "ptt5"):
                    d = cases[name]
                    assert engine.deflate_batch([d], level=1 + ei % 3, strategy=strategy)[0] == oracle.compress(d, 1 + ei % 3, strategy), (env, name, strategy)
            for k in env:
                del os.environ[k]
        # rounds that are cut short (they cannot be, by themselves: chunk r is final after round r) hand the batch to the other form
        os.environ["ZS_FR_MAX_ROUNDS"] = "3"
        try:
            for lvl in (1, 3):
                assert engine.deflate_batch([cases["english1m"]], level=lvl)[0] == oracle.compress(cases["english1m"], lvl), lvl
        finally:
            del os.environ["ZS_FR_MAX_ROUNDS"]
        # one long stream among short ones: rounds; twelve equals: one workgroup each
        mixed = [cases["kennedy"], cases["text300k"][:70000], b"", cases["zeros_65537"], cases["ptt5"], cases["alice_263"]]
        equals = [datagen.english(200000, 100 + i) for i in range(12)]
        for lvl in (1, 3):
            for batch in (mixed, equals):
                for z, d in zip(engine.deflate_batch(batch, level=lvl), batch):
                    assert z == oracle.compress(d, lvl), (lvl, len(d))
    finally:
        for env in envs:
            for k in env:
                os.environ.pop(k, None)


@pytest.mark.gpu
def test_fast_levels_large_streams_speculative_runs(engine, oracle):
    """Levels 1-3 on streams >= 1 MiB take the speculative chunk-run path (verified hand-over states, sequential
    fallback when a run does not verify): text, sparse rows, periodic data that resists re-synchronisation, zeros."""
    rng = np.random.default_rng(9)
    cases = {
        "english4m": datagen.english(4 << 20),
        "sparse4m": datagen.sparse(1024, 1024),
        "period": (bytes(range(256)) * 8192)[:(1 << 21) + 12345],
        "zeros": bytes((1 << 20) + 7),
        "lowent": rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), (1 << 20) + 999).tobytes(),
        "mixed": datagen.english(1 << 20) + datagen.sparse(512, 256) + oracle_binding.corpus("kennedy.xls"),
    }
    for name, d in cases.items():
        for lvl in (1, 2, 3):
            assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl), (name, lvl)


@pytest.mark.gpu
def test_speculative_runs_engine_on_image_like_data(engine, oracle):
    """Periodic data of 4 MiB and more at levels 1-3 goes through the speculative chunk runs, whose engine keeps its hot state in
    registers, takes runs of literals 64 at a time and reads its hash heads through an LDS cache (zs_lit_engine.h
    le_run_fast_hot): image rows with and without noise, rows of one period, long runs, tables, both hash variants, sizes
    around a chunk's end.  Byte-exact against the oracle, whichever path a stream ends up on."""
    rng = np.random.default_rng(41)
    wide = datagen.sparse(4096, 384)  # 6 MiB of config 3's rows: these verify
    for lvl in (1, 2, 3):
        assert engine.deflate_batch([wide], level=lvl)[0] == oracle.compress(wide, lvl), ("wide rows", lvl)
    assert engine.deflate_batch([wide[: (4 << 20) + 12345]], level=1)[0] == oracle.compress(wide[: (4 << 20) + 12345], 1)
    assert engine.deflate_batch([wide], level=2, hash_variant=1)[0] == oracle.compress(wide, 2, hash_variant=1)
    rows = np.frombuffer(datagen.sparse(2048, 640), dtype=np.uint8).copy()  # 5 MiB
    noisy = rows.copy()
    noisy[rng.integers(0, noisy.size, noisy.size // 200)] = rng.integers(0, 256, noisy.size // 200, dtype=np.uint8)
    filt = rows.reshape(640, 8192).copy()
    filt[:, 0] = rng.integers(0, 5, 640, dtype=np.uint8)  # a filter byte per scanline
    ramp = (np.arange(6 << 20, dtype=np.uint32) // 3 % 251).astype(np.uint8)
    cases = {
        "rows": rows.tobytes(),
        "rows + noise": noisy.tobytes(),
        "rows with filter bytes": filt.tobytes(),
        "rows, odd size": rows.tobytes()[: (4 << 20) + 262144 + 77],
        "rows, one chunk and a bit": rows.tobytes()[: (4 << 20) + 3],
        "ramp": ramp.tobytes(),
        "kennedy x 5": oracle_binding.corpus("kennedy.xls") * 5,
        "ptt5 x 9": oracle_binding.corpus("ptt5") * 9,
        "period 7": (bytes([1, 2, 3, 4, 5, 6, 7]) * (1 << 20))[: 5 << 20],
        "zeros + rows": bytes(3 << 20) + rows.tobytes()[: 2 << 20],
    }
    for name, d in cases.items():
        for lvl in (1, 2, 3):
            assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl), (name, lvl)
    d = cases["rows + noise"]
    assert engine.deflate_batch([d], level=1, hash_variant=1)[0] == oracle.compress(d, 1, hash_variant=1)
    # several such streams in one batch, and beside a text stream that takes the sweeps
    batch = [cases["rows"], datagen.english(1 << 20, 5), cases["rows, odd size"], cases["ptt5 x 9"]]
    assert engine.deflate_batch(batch, level=1) == [oracle.compress(b, 1) for b in batch]
    # below 4 MiB (from 64 KiB on, a few streams): planned for the sweeps and for the runs, the links decide -- data that is all
    # period takes the runs (and, where they do not verify, one run of the engine), anything else the sweeps
    small = {
        "zeros 1 MiB": bytes(1 << 20), "zeros 100 000": bytes(100000), "period 7": (bytes([1, 2, 3, 4, 5, 6, 7]) * 40000)[:270001],
        "rows 512 x 512": datagen.sparse(512, 512), "rows 4096 x 20": datagen.sparse(4096, 20), "rows 512 x 40": datagen.sparse(512, 40),
        "runs": np.repeat(rng.integers(0, 4, 3000, dtype=np.uint8), rng.integers(1, 600, 3000)).tobytes(),
        "text": datagen.english(300000, 8), "ptt5": oracle_binding.corpus("ptt5"), "just 64 KiB of zeros": bytes(65536), "65535 zeros": bytes(65535),
    }
    for name, d in small.items():
        for lvl in (1, 2, 3):
            assert engine.deflate_batch([d], level=lvl)[0] == oracle.compress(d, lvl), (name, lvl)
    for names in (("zeros 1 MiB", "rows 512 x 512", "period 7"), ("zeros 1 MiB", "text", "rows 4096 x 20"), ("rows 512 x 40", "65535 zeros", "runs")):
        batch = [small[k] for k in names]
        assert engine.deflate_batch(batch, level=1) == [oracle.compress(b, 1) for b in batch], names
    d = small["rows 512 x 512"]
    assert engine.deflate_batch([d], level=3, hash_variant=1)[0] == oracle.compress(d, 3, hash_variant=1)
    assert engine.deflate_batch([d], level=1, strategy=1)[0] == oracle.compress(d, 1, strategy=1)  # Filtered
    # (rows of 16 KiB -- config 3 -- and kennedy.xls verify; rows of 8 KiB, ptt5 and the ramp do not: their parses do not fall back
    # into step inside the warm-up, the batch is redone as rounds, which on such data settle one range a round)


def _fuzz_buffer(rng, i):
    """Inputs that stress different parts of the pipeline: long zero / byte runs (equal-bucket refills, one hash class
    getting every position), periodic data (many buckets a few times each), incompressible data (stored blocks, mid-stream
    alignment), text with planted long repeats, and mixtures with abrupt changes."""
    n = int(rng.integers(1, 3 << 20)) if i % 5 else int(rng.integers(1, 70000))
    kind = i % 8
    if kind == 0:
        return bytes(n)
    if kind == 1:
        return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 2:
        per = rng.integers(0, 256, int(rng.integers(1, 700)), dtype=np.uint8).tobytes()
        return (per * (n // len(per) + 1))[:n]
    if kind == 3:
        return np.repeat(rng.integers(0, 8, n // 4 + 1, dtype=np.uint8), rng.integers(1, 300, n // 4 + 1))[:n].tobytes()
    if kind == 4:
        return datagen.english(n, int(rng.integers(1, 1 << 30)))
    if kind == 5:
        t = bytearray(datagen.english(n, int(rng.integers(1, 1 << 30))))
        for _ in range(20):  # planted repeats at distances around the window size
            ln = int(rng.integers(3, 600))
            src = int(rng.integers(0, max(1, n - ln)))
            dst = src + int(rng.choice([1, 2, 255, 256, 4096, 32505, 32506, 32507, 32768, 40000]))
            if dst + ln <= n:
                t[dst:dst + ln] = t[src:src + ln]
        return bytes(t)
    if kind == 6:
        parts, left = [], n
        while left > 0:
            m = min(left, int(rng.integers(1, 200000)))
            parts.append(_fuzz_buffer(rng, int(rng.integers(0, 5)) * 8 + int(rng.integers(0, 5)))[:m])
            left -= len(parts[-1])
        return b"".join(parts)
    return datagen.sparse(int(rng.integers(8, 700)), max(1, n // 2800), y0=int(rng.integers(0, 255)))[:n]


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_mixed_inputs_all_slow_levels(engine, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    bufs = [_fuzz_buffer(rng, i) for i in range(24)]
    for lvl in ((6, 9) if seed == 1 else (4, 7)):
        got = engine.deflate_batch(bufs, level=lvl)
        for i, (b, z) in enumerate(zip(bufs, got)):
            assert z == oracle.compress(b, lvl), (seed, lvl, i, len(b))
    got = engine.deflate_batch(bufs[:8], level=5, strategy=int(CompressionStrategy.Filtered))
    for i, (b, z) in enumerate(zip(bufs[:8], got)):
        assert z == oracle.compress(b, 5, int(CompressionStrategy.Filtered)), (seed, i)


def test_streams_beyond_64_mib_bit_exact(engine, oracle):
    """Single streams larger than the benchmark size, every byte against the oracle (position / symbol / bit offsets
    that only grow past 2^26 .. 2^28): 256 MiB of sparse rows and 96 MiB of text."""
    import torch
    for name, data in (("sparse256", datagen.sparse(4096, 16384)), ("english96", datagen.english(96 << 20, 12345))):
        n = len(data)
        d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        cap = deflate_bound(n)
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        out_len = engine.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=6,
                                              stream=torch.cuda.current_stream().cuda_stream)[0]
        z = d_out[:out_len].cpu().numpy().tobytes()
        ref = oracle.compress(data, 6)
        assert len(z) == len(ref), name
        assert z == ref, name
        del d_in, d_out


@pytest.mark.parametrize("level,strategy", [(4, 0), (8, 0), (9, 0), (5, int(CompressionStrategy.Filtered)), (7, int(CompressionStrategy.Fixed))])
def test_other_levels_at_32_mib_bit_exact(engine, oracle, level, strategy):
    """The levels the headline run does not use, on a stream long enough for hundreds of refills and blocks."""
    data = datagen.english(32 << 20, 777 + level)
    assert engine.deflate_batch([data], level=level, strategy=strategy)[0] == oracle.compress(data, level, strategy)


@pytest.mark.gpu
@pytest.mark.parametrize("flush", [1, 2, 3])
def test_flush_modes_partial_sync_full(engine, oracle, flush):
    """ZlibOptions.FlushMode Partial / Sync / Full (FlushMode.cs; Deflate.cs:583-613, Trees.cs:658-680): every Write
    ends its block, then the marker; a flush that fills WriteCore's 512-byte chunk exactly re-enters the block function
    and adds an empty block.  Bytes against the oracle's literal loop, for all block functions and Write patterns."""
    alice = oracle_binding.corpus("alice29.txt")
    rng = np.random.default_rng(flush)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 150000).tobytes()
    cases = [(alice, [len(alice)]), (alice, [4000] * 37 + [481]), (alice * 2, [70000, 70000, 70000, 86962]),
             (low, [8192] * 18 + [2544]), (alice[:30000], [100] * 300), (b"hello", [5]), (bytes(98305), [65536, 32769])]
    for data, chunks in cases:
        assert sum(chunks) == len(data)
        for level, strategy in ((0, 0), (1, 0), (3, 0), (4, 0), (6, 0), (9, 0), (6, 3), (6, 2), (6, 4)):
            out = io.BytesIO()
            with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=CompressionStrategy(strategy),
                                                   FlushMode=flush), engine=engine) as s:
                o = 0
                for c in chunks:
                    s.write(data[o:o + c])
                    o += c
            z = out.getvalue()
            assert zlib.decompress(z) == data
            assert z == oracle.compress(data, level, strategy, chunks=chunks, flush=flush), (len(data), chunks[:3], level, strategy)


@pytest.mark.gpu
def test_writes_behind_a_flush_go_back_to_the_bulk_path(engine, oracle):
    """A stream that has flushed is incremental: its engine is kept suspended on the device.  A long Write behind the
    flush runs the literal engine for a window's length -- past the positions the flush left in the chains under hashes
    of bytes that were not there yet, and past what a FullFlush forgot -- and the bulk pipeline takes the parse over at a
    clean loop-top (zs_engine.hip RunOpts::resume): bytes against the oracle (Deflate.cs:583-613 for the flush, the
    WriteCore loop literally), and the time of 8 MiB Writes shows which path ran (the literal engine does 2 MB/s)."""
    import time
    text = datagen.english(20 << 20, datagen.GOLDEN)
    low = np.random.default_rng(9).choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 3 << 20).tobytes()
    cases = [(text, [5000, 8 << 20, 1000, 8 << 20], [2, 0, 0, 0], 6), (text, [300000, 8 << 20, 70001, 3 << 20], [3, 2, 0, 1], 6),
             (text[:6 << 20], [100, 3 << 20, 2 << 20], [1, 0, 2], 4), (low, [65536, 1 << 20, 1 << 20], [2, 2, 0], 9),
             (low, [4096, 2 << 20], [3, 0], 6), (bytes(3 << 20), [1000, 2 << 20], [2, 0], 6),
             (text[:5 << 20], [1000, 1 << 20, 16385, 1 << 20, 999, 1 << 20], [2, 0, 0, 0, 0, 0], 6)]
    for data, sizes, flushes, level in cases:
        chunks, fl, o = [], [], 0
        for c, f in zip(sizes, flushes):
            c = min(c, len(data) - o)
            if c > 0:
                chunks.append(c), fl.append(f)
                o += c
        if o < len(data):
            chunks.append(len(data) - o), fl.append(0)
        # (ZlibOptions.FlushMode is read by every WriteCore: set anew before every Write)
        out = io.BytesIO()
        t0 = time.perf_counter()
        s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), FlushMode=0), engine=engine)
        o = 0
        for c, f in zip(chunks, fl):
            s.Options.FlushMode = f
            s.write(data[o:o + c])
            o += c
        s.Options.FlushMode = 0
        s.close()
        dt = time.perf_counter() - t0
        z = out.getvalue()
        assert zlib.decompress(z) == data, (len(data), chunks[:4], fl[:4], level)
        assert z == oracle.compress_writes(data, level, 0, chunks, fl), (len(data), chunks[:4], fl[:4], level)
        if len(data) >= (16 << 20):
            assert dt < 6.0, "%.1f s: the Writes behind the flush did not leave the literal engine" % dt


def _flushed_stream(engine, data, chunks, fl, level, strategy=0):
    out = io.BytesIO()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=strategy, FlushMode=0), engine=engine)
    o = 0
    for c, f in zip(chunks, fl):
        s.Options.FlushMode = f
        s.write(data[o:o + c])
        o += c
    s.Options.FlushMode = 0
    s.close()
    return out.getvalue()


@pytest.mark.gpu
def test_the_bulk_path_goes_on_from_the_flush_itself(engine, oracle, monkeypatch):
    """Behind a flush the engine stands at a position with nothing read ahead and nothing pending: the bulk pipeline starts
    there (GeoStart::at_read -- the first pass through the loop reads, as a stream's first one does) on the chains the
    engine left (zs_import_chains_kernel: prev[] below the flush, head[] for the first member of a bucket behind it), which
    hold what the data alone does not tell -- the last positions in front of the flush went in under hashes of window bytes
    that were not theirs, or not at all (Deflate.Slow.cs:58,121-129), a FullFlush forgot the heads (Deflate.cs:596-604),
    equal-bucket reads cut chains.  Flushes at every kind of place: the first bytes of a stream, window ends and the slide
    threshold, one flush after the other; zeros, a small alphabet (equal buckets everywhere), random bytes, text; Partial,
    Sync and Full; and once more with the literal engine's warm-up in front (ZS_NO_FLUSH_RESUME), the route of a stream
    that is not at a flush."""
    text = datagen.english(6 << 20, datagen.GOLDEN)
    rng = np.random.default_rng(77)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 3 << 20).tobytes()
    rnd = rng.integers(0, 256, 2 << 20, dtype=np.uint8).tobytes()
    M = 1 << 20
    cases = [(text, [3, M], [2, 0], 6), (text, [1, M], [3, 0], 6), (text, [65536, M], [2, 0], 6), (text, [65273, M], [2, 0], 6),
             (text, [65274, M], [1, 0], 6), (text, [65275, M], [2, 0], 6), (text, [98304, M], [2, 0], 6), (text, [98303, M], [3, 0], 6),
             (text, [32768, M], [2, 0], 9), (text, [300000, 400000, 500000, 600000, 700000], [2, 2, 3, 1, 2], 6),
             (text, [M, 100, M, 5, M], [0, 2, 0, 3, 0], 6), (text, [500000, 300000, 300000, 300001], [2, 0, 0, 0], 4),
             (low, [70000, M, M], [2, 2, 0], 6), (low, [4096, 2 * M], [3, 0], 9), (low, [300000, 300000, 300000], [1, 3, 2], 8),
             (bytes(3 << 20), [100000, M, M], [2, 3, 0], 6), (bytes(2 << 20), [65536, M], [2, 0], 9),
             (rnd, [5000, M], [2, 0], 6), (rnd, [131072, 900000], [3, 2], 5), (text, [200000, 2 * M], [2, 0], 7)]
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("ZS_NO_FLUSH_RESUME", env)
        for data, sizes, flushes, level in (cases if not env else cases[::3]):
            chunks, fl, o = [], [], 0
            for c, f in zip(sizes, flushes):
                c = min(c, len(data) - o)
                if c > 0:
                    chunks.append(c), fl.append(f)
                    o += c
            data = data[:o]
            z = _flushed_stream(engine, data, chunks, fl, level)
            assert zlib.decompress(z) == data, (len(data), chunks, fl, level, env)
            assert z == oracle.compress_writes(data, level, 0, chunks, fl), (len(data), chunks, fl, level, env)


@pytest.mark.gpu
def test_short_runs_between_flushes_take_the_bulk_path(engine, oracle):
    """A stream that flushes every few KiB (a protocol's messages): every run behind a flush is the bulk pipeline's from
    1 KiB on (6 KiB until late in round 5).  Such a run may end before it has slid the window the engine before it left, or have that engine's positions
    in its own last window: the tail engine then takes what lies behind the data in the window from that engine's image, and
    its hash heads are that engine's under the run's own positions (zs_tail_kernel, StreamDesc::start_pos).  Random
    schedules over four kinds of data; flushes in the last 262 bytes of a window (the window slides before it is full,
    Deflate.cs:979) with short runs behind them; the time of a 4 MiB stream flushed every 64 KiB shows the path."""
    import time
    text = datagen.english(4 << 20, datagen.GOLDEN)
    rng = np.random.default_rng(2024)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 2 << 20).tobytes()
    rnd = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    runs = np.repeat(rng.integers(0, 256, 40000, dtype=np.uint8), rng.integers(1, 90, 40000))[:2 << 20].tobytes()
    cases = []
    for data, level in ((text, 6), (low, 9), (rnd, 5), (runs, 6), (text, 4), (bytes(1 << 20), 7), (low, 6), (text, 9)):
        sizes, fl, o = [], [], 0
        while o < min(len(data), 1536 << 10):
            c = int(rng.choice([6144, 7000, 8192, 20000, 32768, 50000, 65536, 100000, 150000, 300, 40]))
            c = min(c, len(data) - o)
            sizes.append(c), fl.append(int(rng.choice([0, 1, 2, 2, 3])))
            o += c
        cases.append((data[:o], sizes, fl, level))
    # flushes that leave the window to slide before it is full, short runs behind them
    for k in (1, 5, 100, 261, 262, 263):
        cases.append((text, [65536 - k, 7000, 9000, 32768 - 7000 - 9000 + k - 3, 8000, 50000], [2, 2, 1, 3, 2, 2], 6))
        cases.append((low, [98304 - k, 6500, 40000, 6200], [3, 2, 2, 0], 9))
    for data, sizes, fl, level in cases:
        data = data[:sum(sizes)]
        z = _flushed_stream(engine, data, sizes, fl, level)
        assert zlib.decompress(z) == data, (len(data), sizes[:8], fl[:8], level)
        assert z == oracle.compress_writes(data, level, 0, sizes, fl), (len(data), sizes[:8], fl[:8], level)
    sizes = [65536] * 64
    _flushed_stream(engine, text, sizes, [2] * 64, 6)
    t0 = time.perf_counter()
    z = _flushed_stream(engine, text, sizes, [2] * 64, 6)
    dt = time.perf_counter() - t0
    assert z == oracle.compress_writes(text, 6, 0, sizes, [2] * 64)
    assert dt < 1.0, "%.2f s for 64 flushed Writes of 64 KiB: the runs did not leave the literal engine (6 s)" % dt
    # late in round 5 the line moved from 6 KiB to 1 KiB (a resumed run's body may be shorter than a chunk): runs of 1 .. 5 KiB
    # behind every flush mode at slow and fast levels, byte-exact, and the literal engine's share of them by its counter
    for data, level in ((text, 6), (low, 9), (runs, 4), (text, 1), (low, 3), (rnd, 6), (bytes(1 << 20), 2)):
        sizes, fl, o = [], [], 0
        while o < 300000:
            c = int(rng.integers(1024, 5200))
            sizes.append(c), fl.append(int(rng.choice([1, 2, 2, 3])))
            o += c
        data = data[:o]
        before = engine.counter("lit_engine_bytes")
        z = _flushed_stream(engine, data, sizes, fl, level)
        lit = engine.counter("lit_engine_bytes") - before
        assert z == oracle.compress_writes(data, level, 0, sizes, fl), (level, sizes[:6], fl[:6])
        # (the first run -- nothing to resume from -- and what a schedule's odd spots leave; not run after run)
        assert lit <= 0.2 * len(data), "level %d: the literal engine took %d of %d bytes in runs of 1-5 KiB" % (level, lit, len(data))


@pytest.mark.gpu
def test_random_streams_that_once_differed(engine, oracle):
    """tools/fuzz_streams.py (random data kinds, Write sizes, flush modes, levels, strategies; 12 000 streams in 14 minutes)
    found two things, each about once in a thousand streams.  A flush whose run also carried NoFlush Writes that had been
    waiting could hand out its last byte with the caller's chunk exactly full; the caller's loop then comes back with nothing
    to write (ZlibOutputStream.cs:125-168), and that call ran the engine for the flush a second time: two more empty blocks.
    And a Write of one or two bytes behind a flush, where a read's two inserts share a bucket of stale hashes, leaves a forward
    pointer in prev[] that no later insert closes into a cycle (Deflate.cs:1009-1012 with lookahead < MIN_MATCH at the next
    loop-top): the chains cannot be handed to the bulk pipeline as links, the stream stays with the literal engine
    (zs_resume_check_kernel).  The seeds that showed them, whole."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cases
    # (504158: a stream in the batched cut rounds whose second round put records back -- and must put back the chunks' largest
    # match distances with them, which the repairs' scans go by)
    for seed in (2924, 3050, 3929, 4623, 4893, 5300, 6403, 7128, 8883, 11354, 11441, 11724, 504158):
        data, sizes, fl, level, strategy = fuzz_cases.make(np.random.default_rng(seed))
        z = _flushed_stream(engine, data, sizes, fl, level, strategy)
        assert zlib.decompress(z) == data, seed
        assert z == oracle.compress_writes(data, level, strategy, sizes, fl), seed


@pytest.mark.gpu
def test_inflate_batch_with_a_stream_whose_block_finder_overflows(engine):
    """tools/fuzz_batch.py seed 1305306: a batch whose first block-parallel stream (zlib's Z_HUFFMAN_ONLY at level 8) fills a
    chunk's candidate list to the brim.  Such a stream goes to the sequential decoder -- but its candidates were still handed
    to the measuring pass, unsorted and, behind the overflow, whatever the buffer held: a negative bit offset is a read in front
    of the input (a memory fault in this batch of three, not in any two of them; round 2's library has it too).  The measuring
    pass now takes no candidate of such a stream and checks every offset against the stream's length."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_cases
    rng = np.random.default_rng(1305306)
    rng.integers(0, 4)
    bufs, zs = fuzz_cases.inflate_batch_case(rng)
    for idx in ([1, 13, 15], list(range(len(bufs)))):
        outs = engine.inflate_batch([zs[i] for i in idx], [len(bufs[i]) for i in idx])
        assert all(o == bufs[i] for o, i in zip(outs, idx))


@pytest.mark.gpu
def test_flush_mode_single_write_takes_the_bulk_path(engine, oracle):
    """One Write under SyncFlush at level 6: the bulk pipeline runs (the tail engine closes the block, the offsets kernel
    adds the marker and the re-entered empty block); 8 MiB so that the sequential engine would be visible in the time."""
    d = datagen.english(8 << 20, datagen.GOLDEN)
    for flush in (1, 2, 3):
        out = io.BytesIO()
        with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel.Level6, FlushMode=flush), engine=engine) as s:
            s.write(d)
        z = out.getvalue()
        assert zlib.decompress(z) == d
        assert z == oracle.compress(d, 6, 0, chunks=[len(d)], flush=flush)


@pytest.mark.gpu
def test_multi_write_streams_on_the_bulk_path(engine, oracle):
    """Streams of several NoFlush Writes whose sizes are multiples of 2048 (Stream.CopyTo's 81920, 4 / 8 / 64 KiB buffers)
    have a regular read schedule -- the window-full events of a single Write plus one event per Write end, all on the
    chunk grid (zs_core.h build_read_events) -- and take the bulk pipeline; bytes against the oracle, which runs the
    reference's WriteCore loop literally.  8 MiB of text in ~2 MB/s of the literal engine would take seconds: the time
    bound shows which path ran."""
    import time
    text = datagen.english(8 << 20, datagen.GOLDEN)
    low = np.random.default_rng(9).choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 3 << 20).tobytes()
    zeros = bytes(1 << 20)
    for data, sizes in ((text, (81920, 4096, 65536, 2048, 1 << 20)), (low, (8192, 98304)), (zeros, (2048, 16384)),
                        (text[:200000], (4096,)), (text[:65536 + 300], (32768,)), (text[:3 * 81920], (81920,))):
        for size in sizes:
            for level in (6, 4, 9) if len(data) <= (3 << 20) else (6,):
                chunks = [min(size, len(data) - o) for o in range(0, len(data), size)]
                out = io.BytesIO()
                t0 = time.perf_counter()
                with ZlibOutputStream(out, CompressionLevel(level), engine=engine) as s:
                    o = 0
                    for c in chunks:
                        s.write(data[o:o + c])
                        o += c
                dt = time.perf_counter() - t0
                z = out.getvalue()
                assert z == oracle.compress(data, level, chunks=chunks), (len(data), size, level)
                if len(data) >= (8 << 20):
                    assert dt < 2.0, "8 MiB in %d-byte Writes took %.2f s: not the bulk path" % (size, dt)
    # Write sizes under a flush mode stay on the literal engine -- same bytes
    d = text[:300000]
    for size in (1000, 5000):
        chunks = [min(size, len(d) - o) for o in range(0, len(d), size)]
        out = io.BytesIO()
        with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel.Level6, FlushMode=2), engine=engine) as s:
            o = 0
            for c in chunks:
                s.write(d[o:o + c])
                o += c
        assert out.getvalue() == oracle.compress(d, 6, chunks=chunks, flush=2)


def _write_ends(n, spec, rng):
    ends, o = [], 0
    while o < n:
        if isinstance(spec, int):
            w = spec
        elif spec[0] == "r":
            w = int(rng.integers(spec[1], spec[2] + 1))
        else:
            w = spec[len(ends) % len(spec)]
        o = min(n, o + max(1, w))
        ends.append(o)
    return ends


def _deflate_writes(engine, data, ends, level, strategy=0):
    import ctypes
    import time
    import torch
    n = len(data)
    d_in = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
    cap = deflate_bound(n) + 4096
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    arr = (ctypes.c_int64 * len(ends))(*ends)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    olen = engine.deflate_writes_device(d_in.data_ptr(), n, arr, d_out.data_ptr(), cap, level=level, strategy=strategy)
    dt = time.perf_counter() - t0
    return d_out[:olen].cpu().numpy().tobytes(), dt


@pytest.mark.gpu
def test_several_writes_at_the_fast_levels_take_the_sweeps(engine, oracle):
    """Levels 1-3 (DeflateFast, Deflate.Fast.cs:20-128) on a stream written in several NoFlush Writes: a Write end is a read event
    like a window end (Fill_window reads, the position behind the loop-top is inserted ahead, Deflate.cs:967-1019), and the
    sweeps apply an event where a sweep -- or a chunk of the rounds -- starts.  Stream.CopyTo's 81 920 bytes, 65 536, 70 001,
    100 000, a first Write of 300 bytes then megabytes: byte-exact against the oracle's WriteCore loop, through the device entry
    and through the Stream classes, and at device speed (8 MiB in 81 920-byte Writes took the literal engine 4-16 s).  Sizes
    whose ends fall where a loop-top may or may not slide the window (16 385-byte scanlines) stay with the literal engine: exact,
    slow."""
    import time
    text = datagen.english(8 << 20, 31)
    ptt5 = oracle_binding.corpus("ptt5")
    for data, size in ((text, 81920), (text, 65536), (text[:3 << 20], 70001), (text[:3 << 20], 100000), (ptt5, 81920), (bytes(600000), 81920)):
        chunks = [size] * (len(data) // size) + ([len(data) % size] if len(data) % size else [])
        ends = list(np.cumsum(chunks))
        for lvl in (1, 2, 3):
            z, dt = _deflate_writes(engine, data, ends, lvl)
            assert z == oracle.compress(data, lvl, chunks=chunks), (len(data), size, lvl)
            if len(data) == 8 << 20:
                assert dt < 0.5, "%d-byte Writes at level %d: %.2f s -- the literal engine's pace" % (size, lvl, dt)
    # a short first Write, then long ones; through the Stream class (host memory, the WriteCore loop)
    data = text[:2 << 20]
    chunks = [300, 1 << 20, (1 << 20) - 300]
    out = io.BytesIO()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(1)), engine=engine)
    o = 0
    for c in chunks:
        s.write(data[o:o + c])
        o += c
    s.close()
    assert out.getvalue() == oracle.compress(data, 1, chunks=chunks)
    # scanline-sized Writes: the literal engine, the reference's bytes all the same
    small = text[:200000]
    chunks = [16385] * (len(small) // 16385) + [len(small) % 16385]
    z, _ = _deflate_writes(engine, small, list(np.cumsum(chunks)), 1)
    assert z == oracle.compress(small, 1, chunks=chunks)


@pytest.mark.gpu
def test_stream_input_sent_ahead_while_its_buffers_grow(engine, oracle):
    """ZlibOutputStream gathers NoFlush Writes in the context's pinned buffer and sends them to the device while the caller is
    still writing (zs_stream_api.inc).  Streams of growing sizes one after the other on one context: the pinned buffer and its
    device copy are outgrown in the middle of a stream, and a new device buffer may come back at the old one's address -- what
    had been sent must be sent again (tools/fuzz_streams.py seed 555024 behind its 22 predecessors: 2.8 MB of a 3 MB stream had
    been "sent" into a freed buffer).  Bytes against the oracle's WriteCore loop; a second context with the transfers turned
    off gives the same."""
    rng = np.random.default_rng(77)
    per = rng.integers(0, 256, 431, dtype=np.uint8).tobytes()
    for n in (40000, 1500000, 300000, 3000000, 1200000, 6500000, 2000000):
        data = (per * (n // len(per) + 1))[:n] if n % 3 == 0 else datagen.english(n, n)
        sizes, o = [], 0
        while o < n:
            c = min(int(rng.choice([32768, 65536, 65274, 98304])) - int(rng.integers(0, 300)), n - o)
            sizes.append(c)
            o += c
        for level in (4, 1):
            out = io.BytesIO()
            s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level)), engine=engine)
            o = 0
            for c in sizes:
                s.write(data[o:o + c])
                o += c
            s.close()
            assert out.getvalue() == oracle.compress(data, level, chunks=sizes), (n, level)


@pytest.mark.gpu
def test_rle_streams_written_in_several_writes_run_as_one(engine, oracle):
    """CompressionStrategy.Rle does not look at NoFlush Write ends (tests/test_oracle.py::test_rle_does_not_look_at_write_ends): a
    stream written in 1000-byte, scanline or CopyTo-sized Writes runs over the chip like a single Write's (zs_rle.hip) -- the
    oracle's bytes for that schedule, and at device speed; a schedule with a Write end just below a window end is the literal
    engine's, exact all the same."""
    rows = datagen.sparse(1024, 1024)  # 4 MiB
    text = datagen.english(2 << 20, 3)
    for data, size in ((rows, 1000), (rows, 16385), (text, 81920), (rows, 300)):
        chunks = [size] * (len(data) // size) + ([len(data) % size] if len(data) % size else [])
        ends = list(np.cumsum(chunks))
        safe = not any(E >= 65536 - 262 and E % 32768 >= 32768 - 262 for E in ends[:-1])
        for lvl in (1, 6):
            if not safe and len(data) > (1 << 20):
                data, chunks = data[:300000], None
                chunks = [size] * (len(data) // size) + ([len(data) % size] if len(data) % size else [])
                ends = list(np.cumsum(chunks))
            z, dt = _deflate_writes(engine, data, ends, lvl, strategy=3)
            assert z == oracle.compress(data, lvl, 3, chunks=chunks), (len(data), size, lvl)
            if safe and len(data) >= (2 << 20):
                assert dt < 0.2, "%d-byte Writes under Rle at level %d: %.2f s" % (size, lvl, dt)
    # a Write that ends 100 bytes below the first window end: the literal engine
    data = rows[:200000]
    chunks = [65436, len(data) - 65436]
    z, _ = _deflate_writes(engine, data, list(np.cumsum(chunks)), 6, strategy=3)
    assert z == oracle.compress(data, 6, 3, chunks=chunks)


@pytest.mark.gpu
def test_any_write_sizes_on_the_bulk_path(engine, oracle):
    """NoFlush Writes of any size -- 1000 bytes, a scanline of 16 385, Stream.CopyTo's 81 920 + 1, random sizes, Writes
    shorter than MIN_LOOKAHEAD mixed in -- take the bulk pipeline (zs_core.h build_geometry: segments cut at the clusters of
    read boundaries, a cluster's events stepped per entry slot; the resolve kernel cuts at every equal-bucket event in
    stream order), device-resident through zs_deflate_writes_device; every byte against the oracle's literal WriteCore
    loop (ZlibOutputStream.cs:114-168, Deflate.cs:967-1019).  The long-chain levels go through rounds."""
    rng = np.random.default_rng(5)
    alice = oracle_binding.corpus("alice29.txt")
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 1 << 20).tobytes()
    runs = np.repeat(rng.integers(0, 4, 60000, dtype=np.uint8), rng.integers(1, 40, 60000))[:500000].tobytes()
    for name, data in (("alice", (alice * 3)[:400000]), ("low", low), ("runs", runs), ("zeros", bytes(300000))):
        for spec in (1000, 16385, 81921, ("r", 263, 3000), ("r", 1, 5000), ("r", 100, 600), (5000, 3), (65530, 4, 1000), (100, 100, 100, 5000)):
            for level in (6, 4, 9):
                if level == 9 and (name in ("runs", "zeros") or spec not in (1000, (5000, 3), ("r", 100, 600))):
                    continue
                ends = _write_ends(len(data), spec, rng)
                z, _ = _deflate_writes(engine, data, ends, level)
                chunks = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
                assert z == oracle.compress(data, level, chunks=chunks), (name, spec, level)


@pytest.mark.gpu
def test_cut_rounds_with_more_slots_than_a_grid_dimension(engine, oracle):
    """The batched cut rounds launch one workgroup row per cut slot (one slot per data end and per parse segment); a grid's
    y dimension ends at 65 535.  24 MiB of zero pages and text in 300-byte Writes has ~84 000 data ends: with the rounds
    forced from the first cut on (ZS_FORCE_ROUNDS) the cuts in the slots behind 65 535 have to be repaired like the others
    (zs_cuts_repair_kernel takes the slots in turns) -- every byte against the oracle's WriteCore loop."""
    n = 24 << 20
    data = b"".join(datagen.english(8192, 900 + i) if i % 4 == 3 else bytes(8192) for i in range(n // 8192))
    ends = _write_ends(len(data), 300, None)
    assert len(ends) > 70000
    os.environ["ZS_FORCE_ROUNDS"] = "1"
    try:
        z, _ = _deflate_writes(engine, data, ends, 6)
    finally:
        del os.environ["ZS_FORCE_ROUNDS"]
    chunks = [ends[0]] + [ends[i] - ends[i - 1] for i in range(1, len(ends))]
    assert z == oracle.compress(data, 6, chunks=chunks)


@pytest.mark.gpu
def test_scanline_and_odd_sized_writes_at_64_mib_run_at_device_speed(engine, oracle, rate_floors):
    """The verdict's cases at BASELINE size: 64 MiB of text in 1000-byte and 81 921-byte Writes and a 4096 x 4096 RGBA image
    written one filtered scanline (16 385 bytes) per Write, level 6, resident in HBM: at least 1 GB/s (the literal engine
    does 2 MB/s), a round trip through an independent inflater, and the first 4 MiB -- written the same way -- byte for byte
    against the oracle."""
    text = datagen.english(64 << 20, datagen.GOLDEN)
    img = datagen.sparse(4096, 4096)
    rows = b"".join(b"\x01" + img[r * 16384:(r + 1) * 16384] for r in range(4096))  # a filter-type byte in front of every row
    for name, data, size in (("english64", text, 1000), ("english64", text, 81921), ("sparse64 rows", rows, 16385)):
        ends = _write_ends(len(data), size, None)
        z, dt = _deflate_writes(engine, data, ends, 6)   # (the first call sizes the workspace)
        z, dt = _deflate_writes(engine, data, ends, 6)
        assert zlib.decompress(z) == data, (name, size)
        rate_floors.check(len(data) / dt >= 1e9, "%s in %d-byte Writes: %.1f ms = %.2f GB/s" % (name, size, dt * 1e3, len(data) / dt / 1e9))
        part = data[:4 << 20]
        pe = _write_ends(len(part), size, None)
        zp, _ = _deflate_writes(engine, part, pe, 6)
        assert zp == oracle.compress(part, 6, chunks=[pe[0]] + [pe[i] - pe[i - 1] for i in range(1, len(pe))]), (name, size)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_stream_protocol_writes_and_flush_modes(engine, oracle, seed):
    """Random streams through the ZlibOutputStream mirror: data kind, level 0-9, strategy, Write sizes (on and off the
    chunk grid, tiny ones included) and FlushMode drawn at random; bytes against the oracle's literal WriteCore loop."""
    rng = np.random.default_rng(seed)
    alice = oracle_binding.corpus("alice29.txt")
    for case in range(14):
        n = int(rng.choice([0, 1, 300, 5000, 70000, 200000, 400000]))
        n += int(rng.integers(0, 3000)) if n > 300 else 0
        kind = int(rng.integers(0, 4))
        if kind == 0:
            data = (alice * 4)[:n]
        elif kind == 1:
            data = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), n).tobytes()
        elif kind == 2:
            data = bytes(n)
        else:
            data = np.repeat(rng.integers(0, 4, n // 8 + 1, dtype=np.uint8), rng.integers(1, 30, n // 8 + 1))[:n].tobytes()
            n = len(data)
        level = int(rng.integers(0, 10))
        strategy = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
        flush = int(rng.choice([0, 0, 0, 1, 2, 3]))
        unit = int(rng.choice([2048, 4096, 81920, 1000, 333, 65536, 7]))
        chunks, left = [], n
        while left > 0:
            c = unit if rng.random() < 0.8 else int(rng.integers(1, 2 * unit + 1))
            if unit == 7 and n > 5000:
                c = int(rng.integers(1, 4000))
            c = min(c, left)
            chunks.append(c)
            left -= c
        try:
            want = oracle.compress(data, level, strategy, chunks=chunks or None, flush=flush)
        except RuntimeError:
            continue  # level 0 + Rle overflow of the reference's pending buffer: covered elsewhere
        out = io.BytesIO()
        with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), CompressionStrategy=CompressionStrategy(strategy),
                                               FlushMode=flush), engine=engine) as s:
            o = 0
            for c in chunks:
                s.write(data[o:o + c])
                o += c
        assert out.getvalue() == want, (seed, case, n, kind, level, strategy, flush, unit, chunks[:4])


def test_tail_searches_done_ahead_many_odd_sized_streams(engine, oracle):
    """The tail kernel searches a stream's last loop-tops one position per thread before its engine parses them
    (le_tail_record; Longest_match Deflate.cs:1022-1100 under the end-of-stream rules): 160 streams whose sizes sit around
    the points where the window slides or fills, with periodic data and matches running into the data end, every byte
    against the oracle at levels 4-9 and under Filtered / Fixed."""
    rng = np.random.default_rng(4242)
    alice = open(os.path.join(os.path.dirname(__file__), "golden", "corpus", "alice29.txt"), "rb").read() * 3
    sizes = [263, 300, 520, 5000, 32768 + 261, 65274, 65275, 65535, 65536, 65536 + 200, 65536 + 262, 98304 - 100, 98304 + 5,
             131072 - 261, 131072 + 1]
    bufs = []
    for i in range(160):
        n = int(sizes[i % len(sizes)] + rng.integers(-3, 4))
        kind = i % 4
        if kind == 0:
            o = int(rng.integers(0, len(alice) // 3))
            b = alice[o:o + n]
        elif kind == 1:
            pat = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            b = (pat * (n // len(pat) + 1))[:n]
        elif kind == 2:
            b = rng.integers(0, 4, n, dtype=np.uint8).tobytes()
        else:
            t = bytearray(alice[11:11 + n])
            k = int(rng.integers(10, 600))
            if n > 2 * k + 10:
                src = int(rng.integers(0, n - 2 * k))
                t[n - k:] = t[src:src + k]
            b = bytes(t)
        bufs.append(b)
    for lvl, strat in ((4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (9, 0), (6, int(CompressionStrategy.Filtered)), (6, int(CompressionStrategy.Fixed))):
        got = engine.deflate_batch(bufs, level=lvl, strategy=strat)
        for i, (b, z) in enumerate(zip(bufs, got)):
            assert z == oracle.compress(b, lvl, strat), (lvl, strat, i, len(b))


def test_symbol_kernel_records_through_the_lds_ring_and_by_direct_loads(engine, oracle):
    """The symbol kernel (Tr_tally along the true path, Deflate.cs:910-948 / Deflate.Slow.cs:34-145) takes its match records
    from an LDS ring that two feeding waves fill ahead of each walking lane, and loads for itself where a long match jumps
    past the ring.  Text (short hops), long repeats at irregular distances (jumps of up to 258 positions, most of them
    beyond the ring's 64), streams that end inside a 128-byte line of records, many streams per workgroup -- against the
    oracle at levels 4-9 and under Filtered; and once more with the feeding waves switched off (ZS_K5_AHEAD=0), the same
    bytes through direct loads only.  HuffmanOnly (literals from the input, no records) takes the direct path as well."""
    rng = np.random.default_rng(555)
    alice = open(os.path.join(os.path.dirname(__file__), "golden", "corpus", "alice29.txt"), "rb").read()
    bufs = []
    for i in range(96):
        n = int(rng.integers(2400, 70000))
        t = bytearray(datagen.english(n, 100 + i)) if i % 3 else bytearray(alice[i * 97:i * 97 + n])
        n = len(t)
        if i % 2:  # copies of 40..600 bytes from somewhere earlier, every few hundred bytes
            q = 1000
            while q + 700 < n:
                k = int(rng.integers(40, 600))
                src = int(rng.integers(0, q - 600)) if q > 700 else 0
                t[q:q + k] = t[src:src + k]
                q += k + int(rng.integers(30, 900))
        bufs.append(bytes(t))
    bufs.append(datagen.english(3 << 20, 9))
    bufs.append(datagen.sparse(512, 512))
    cases = ((4, 0), (6, 0), (9, 0), (6, int(CompressionStrategy.Filtered)))
    want = {c: [oracle.compress(b, c[0], c[1]) for b in bufs] for c in cases}
    for c in cases:
        assert engine.deflate_batch(bufs, level=c[0], strategy=c[1]) == want[c], c
    os.environ["ZS_K5_AHEAD"] = "0"
    try:
        assert engine.deflate_batch(bufs, level=6) == want[(6, 0)]
    finally:
        del os.environ["ZS_K5_AHEAD"]
    ho = int(CompressionStrategy.HuffmanOnly)
    assert engine.deflate_batch(bufs[:8], level=6, strategy=ho) == [oracle.compress(b, 6, ho) for b in bufs[:8]]


def test_resolve_through_composed_segment_maps_and_row_by_row(engine, oracle):
    """The resolve kernel follows a long stream through the segment maps composed 16 at a time (zs_supmap_kernel) and goes
    back to the row-by-row walk when the path meets an equal-bucket refill (Deflate.cs:1010-1013 with both positions in one
    bucket: runs, periodic data).  Both ways, and the forced row-by-row way (ZS_NO_SUPMAP), give the oracle's bytes."""
    rng = np.random.default_rng(77)
    period = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
    bufs = [datagen.english(3 << 20, 21),                                   # no equal-bucket refill: the short way
            bytes(2 << 20),                                                # every refill is one
            (period * ((1 << 20) // 37 + 1))[:1 << 20] + datagen.english(2 << 20, 22),   # the short way fails part-way
            datagen.sparse(1024, 1024),
            datagen.english(700_000, 23) + bytes(300_000) + datagen.english(1_500_000, 24)]
    want = [oracle.compress(b, 6) for b in bufs]
    assert engine.deflate_batch(bufs, level=6) == want
    os.environ["ZS_NO_SUPMAP"] = "1"
    try:
        assert engine.deflate_batch(bufs, level=6) == want
    finally:
        del os.environ["ZS_NO_SUPMAP"]
    for lvl in (4, 9):
        assert engine.deflate_batch(bufs[:3], level=lvl) == [oracle.compress(b, lvl) for b in bufs[:3]]


def test_long_chain_levels_on_runs_and_zero_pages_go_through_rounds(engine, oracle):
    """Levels 8 and 9 (chains of 1024 / 4096) on data whose refills are equal-bucket ones with thousands of positions to walk
    again behind each: after kDeferBudget such cuts the resolve kernel gives the stream up and the batch is run again in
    rounds (zs_repair_kernel and zs_stalemaps_kernel over the chip between launches of the resolve kernel).  The bytes are
    the oracle's; a text stream in the same batch is carried along."""
    rng = np.random.default_rng(123)
    n = 3 << 20
    runs = np.repeat(rng.integers(0, 4, n // 8, dtype=np.uint8), rng.integers(1, 40, n // 8))[:n].tobytes()
    pages = b"".join(datagen.english(4096, 700 + i) if i % 3 else bytes(4096) for i in range(n // 4096))
    bufs = [runs, pages, datagen.english(1 << 20, 31)]
    for lvl in (8, 9):
        assert engine.deflate_batch(bufs, level=lvl) == [oracle.compress(b, lvl) for b in bufs], lvl
    # the same data at level 6 stays with the resolve kernel's own repairs
    want6 = [oracle.compress(b, 6) for b in bufs]
    assert engine.deflate_batch(bufs, level=6) == want6
    # ... and through the rounds all the same: given up after the budget (ZS_DEFER_ALL), or in rounds from the start
    # (ZS_FORCE_ROUNDS), with the edge inputs of the refill tests in the batch
    edge = _edge_inputs()
    names = ["runs", "zeros_98305", "lowent_98305", "lowent_131072", "alice_98566", "period256"]
    more = bufs + [edge[k] for k in names]
    want_more = want6 + [oracle.compress(edge[k], 6) for k in names]
    for var in ("ZS_DEFER_ALL", "ZS_FORCE_ROUNDS"):
        os.environ[var] = "1"
        try:
            assert engine.deflate_batch(more, level=6) == want_more, var
        finally:
            del os.environ[var]


@pytest.mark.gpu
def test_inflate_by_pieces_past_the_buffering_threshold():
    """zs_inflate's bounded form (zs_stream_api.inc: once ZS_INF_PIECE_BYTES -- 64 MiB in production -- have been buffered without the
    stream's end in sight, the complete blocks so far are decoded and leave the buffer): a text stream fed 8 KiB at a time
    through several thresholds with bytes behind its trailer, and 48 MiB of zeros whose pieces decode to a thousand times
    their size.  The threshold is read when the library loads: a process of its own."""
    import subprocess
    import sys
    code = r'''
import io, sys, zlib
sys.path.insert(0, %r)
from zlibstream_amd import ZlibInputStream, Engine, datagen
eng = Engine(0)
for d, trail in ((datagen.english(6 << 20, 5), b"BEHIND-THE-TRAILER" * 40), (bytes(48 << 20), b"xyz")):
    z = zlib.compress(d, 6)
    s = ZlibInputStream(io.BytesIO(z + trail), engine=eng)
    got = bytearray()
    while True:
        part = s.read(1 << 20)
        if not part:
            break
        got += part
    assert bytes(got) == d, (len(got), len(d))
    assert s.TotalIn == len(z) and s.TotalOut == len(d), (s.TotalIn, len(z), s.TotalOut)
    assert s.Adler == zlib.adler32(d)
print("pieces ok")
''' % oracle_binding.ROOT
    env = dict(os.environ, ZS_INF_PIECE_BYTES="20000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "pieces ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_runs_behind_a_flush_at_the_fast_levels_take_the_sweeps(engine, oracle, rate_floors):
    """Levels 1-3 (DeflateFast, Deflate.Fast.cs:20-128) behind a flush (Deflate.cs:583-613): until round 5 every run of such a
    stream was the one-wave literal engine's (0.5-2 MB/s -- ImageSharp's default is a fast level).  A run of one Write that begins
    where a flush left the engine now starts the sweeps' stream form at that position: the links below it are the suspended
    engine's prev[] (zs_import_chains_kernel: DeflateFast's chains hold inserted positions only, so they ARE the compressed
    links), the set below it "inserted" wherever the chains reach, the first read an event like any other.  Random schedules
    over four kinds of data, flushes around window ends, every flush mode; the literal engine's share by zs_ctx_counter."""
    import time
    text = datagen.english(3 << 20, datagen.GOLDEN + 5)
    rng = np.random.default_rng(77)
    low = rng.choice(np.array([0, 0, 0, 0, 1, 2, 255], dtype=np.uint8), 2 << 20).tobytes()
    rnd = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    runs = np.repeat(rng.integers(0, 256, 40000, dtype=np.uint8), rng.integers(1, 90, 40000))[:2 << 20].tobytes()
    img = datagen.sparse(512, 512)
    cases = []
    for data, level in ((text, 1), (text, 2), (text, 3), (low, 1), (rnd, 2), (runs, 3), (img, 1), (bytes(1 << 20), 2), (low, 3), (runs, 1)):
        sizes, fl, o = [], [], 0
        while o < min(len(data), 1536 << 10):
            c = int(rng.choice([6144, 7000, 8192, 20000, 32768, 50000, 65536, 100000, 150000, 262144, 300, 40]))
            c = min(c, len(data) - o)
            sizes.append(c), fl.append(int(rng.choice([1, 2, 2, 2, 3, 0])))
            o += c
        cases.append((data[:o], sizes, fl, level))
    for k in (1, 5, 100, 261, 262, 263):  # flushes that leave the window to slide before it is full, runs behind them
        cases.append((text, [65536 - k, 7000, 9000, 32768 - 7000 - 9000 + k - 3, 8000, 50000], [2, 2, 1, 3, 2, 2], 1))
        cases.append((low, [98304 - k, 6500, 40000, 6200], [3, 2, 2, 0], 3))
    for data, sizes, fl, level in cases:
        data = data[:sum(sizes)]
        z = _flushed_stream(engine, data, sizes, fl, level)
        assert zlib.decompress(z) == data, (len(data), sizes[:8], fl[:8], level)
        assert z == oracle.compress_writes(data, level, 0, sizes, fl), (len(data), sizes[:8], fl[:8], level)
    # a Sync flush behind every 256 KiB Write at level 1: the literal engine parses the runs' last 261 bytes and nothing else
    sizes = [262144] * 12
    before = engine.counter("lit_engine_bytes")
    _flushed_stream(engine, text, sizes, [2] * 12, 1)
    t0 = time.perf_counter()
    z = _flushed_stream(engine, text, sizes, [2] * 12, 1)
    dt = time.perf_counter() - t0
    assert z == oracle.compress_writes(text[:sum(sizes)], 1, 0, sizes, [2] * 12)
    lit = engine.counter("lit_engine_bytes") - before
    assert lit <= 4096, "%d bytes of two 3 MiB streams went through the literal engine beyond the runs' last 261" % lit
    rate_floors.check(sum(sizes) / dt >= 20e6, "level 1, a Sync flush behind every 256 KiB: %.1f ms = %.1f MB/s" % (dt * 1e3, sum(sizes) / dt / 1e6))


@pytest.mark.gpu
def test_inflate_device_pointers_at_odd_addresses(engine):
    """zs_inflate_batch_device takes the caller's pointers as they are: compressed streams that begin at any byte address (the
    measuring pass and the header check read the stream through aligned dwords -- SyncBits / HdrBits `skew` -- and the stream's
    last 128 bytes through a padded copy), outputs at any address (the resolve pass's 4-byte stores take the cells' phase into
    account).  Five streams in one device buffer at offsets 1, 2, 3, 5 and 7 past 16-byte boundaries, outputs likewise."""
    import torch
    datas = [datagen.english(1 << 20, 11), datagen.english((1 << 20) + 3, 12), bytes(700001), datagen.sparse(512, 300),
             np.random.default_rng(5).integers(0, 4, 900007, dtype=np.uint8).tobytes()]
    zs = [zlib.compress(d, 6) for d in datas]
    offs, total = [], 0
    for z, skew in zip(zs, (1, 2, 3, 5, 7)):
        total = (total + 15) // 16 * 16 + skew
        offs.append(total)
        total += len(z)
    zbuf = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    for z, o in zip(zs, offs):
        zbuf[o:o + len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    ooffs, ototal = [], 0
    for d, skew in zip(datas, (3, 1, 7, 2, 5)):
        ototal = (ototal + 15) // 16 * 16 + skew
        ooffs.append(ototal)
        ototal += len(d)
    obuf = torch.zeros(ototal + 64, dtype=torch.uint8, device="cuda")
    lens = engine.inflate_batch_device([zbuf.data_ptr() + o for o in offs], [len(z) for z in zs], [obuf.data_ptr() + o for o in ooffs],
                                       [len(d) for d in datas])
    host = obuf.cpu().numpy().tobytes()
    for i, d in enumerate(datas):
        assert lens[i] == len(d) and host[ooffs[i]:ooffs[i] + len(d)] == d, i
    # nothing written in front of or behind an output
    for i, d in enumerate(datas):
        assert host[ooffs[i] - 1] == 0 and host[ooffs[i] + len(d)] == 0, i


@pytest.mark.gpu
def test_inflate_small_streams_take_the_block_parallel_pass(engine, rate_floors):
    """Streams from 1 KiB of compressed bytes on go through the finder / measure / expand passes (until late in round 5: from
    256 KiB; below that one wave per stream at 4 MB/s -- 120 ms for a 600 KiB text stream): sizes around the line, kinds of
    data, one block and many, stored and fixed blocks, streams packed back to back in one device buffer (a reader that runs
    over its stream's end would see the next one's bytes), a batch of hundreds; and the times."""
    import time
    import torch
    rng = np.random.default_rng(77)
    cases = []
    for n in (1, 100, 1000, 2500, 4096, 20000, 65536, 100000, 262144, 600000):
        cases.append(datagen.english(n, 3 + n))
    cases += [bytes(5000), bytes(300000), datagen.sparse(128, 64), rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(),
              rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), rng.integers(0, 3, 50000, dtype=np.uint8).tobytes()]
    streams = []
    for i, d in enumerate(cases):
        lvl = (6, 1, 9, 0)[i % 4]
        c = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, (0, 0, 4, 2, 3)[i % 5])  # default, Fixed, HuffmanOnly, Rle among them
        z = c.compress(d[: len(d) // 2]) + (c.flush(zlib.Z_SYNC_FLUSH) if i % 3 == 0 else b"") + c.compress(d[len(d) // 2:]) + c.flush()
        streams.append(z)
    for z, d in zip(streams, cases):  # one at a time
        assert engine.inflate_batch([z], [len(d)])[0] == d, (len(d), len(z))
    assert engine.inflate_batch(streams, [len(d) for d in cases]) == cases  # all in one batch
    # packed back to back on the device
    zbuf = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).cuda()
    outs = [torch.zeros(len(d) + 1, dtype=torch.uint8, device="cuda") for d in cases]
    offs = np.cumsum([0] + [len(z) for z in streams])
    lens = engine.inflate_batch_device([zbuf.data_ptr() + int(o) for o in offs[:-1]], [len(z) for z in streams], [o.data_ptr() for o in outs], [len(d) for d in cases])
    for i, d in enumerate(cases):
        assert lens[i] == len(d) and outs[i][: len(d)].cpu().numpy().tobytes() == d and int(outs[i][len(d)]) == 0, i
    # a corrupted small stream still reports the reference's error
    bad = bytearray(streams[7])
    bad[len(bad) // 2] ^= 0x55
    with pytest.raises(Exception):
        engine.inflate_batch([bytes(bad)], [len(cases[7])])
    # the times: one 600 KiB text stream, and 256 streams of 64 KiB
    big = [datagen.english(64 << 10, 500 + i) for i in range(256)]
    zb = [zlib.compress(d, 6) for d in big]
    for zs_, ds_, what, floor in (([streams[9]], [cases[9]], "one 600 KB text stream", 100e6), (zb, big, "256 streams of 64 KiB", 2e9)):
        d_z = [torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda() for z in zs_]
        d_o = [torch.empty(len(d), dtype=torch.uint8, device="cuda") for d in ds_]
        a = ([z.data_ptr() for z in d_z], [len(z) for z in zs_], [o.data_ptr() for o in d_o], [len(d) for d in ds_])
        engine.inflate_batch_device(*a)
        torch.cuda.synchronize()
        t = time.perf_counter()
        got = engine.inflate_batch_device(*a)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert list(got) == [len(d) for d in ds_] and all(d_o[i].cpu().numpy().tobytes() == ds_[i] for i in range(0, len(ds_), 17))
        rate_floors.check(sum(len(d) for d in ds_) / dt >= floor, "inflate of %s: %.2f ms = %.1f MB/s" % (what, dt * 1e3, sum(len(d) for d in ds_) / dt / 1e6))


@pytest.mark.gpu
def test_inflate_streams_of_fixed_code_blocks(engine, rate_floors):
    """CompressionStrategy.Fixed / Z_FIXED: blocks without a header the finder could tell from data.  Their starts are found ahead
    of the chain's walk by waves that decode from guessed bits (zs_inf_fixed_scan_kernel), the walk measures what they missed with its
    64 lanes (inf_fixed_end): zlib's and this library's own fixed streams, several in a batch beside a dynamic one, sizes around a
    region's 64 KiB, a corrupted one; 16 MiB in tens of milliseconds (2.1 s when the walk went symbol by symbol)."""
    import time
    import torch
    rng = np.random.default_rng(5)
    datas = [datagen.english(n, 40 + i) for i, n in enumerate((70000, 300000, 1 << 20, (4 << 20) + 12345))]
    datas += [datagen.sparse(512, 700), rng.integers(0, 5, 900000, dtype=np.uint8).tobytes(), bytes(500000)]
    streams = []
    for d in datas:
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        streams.append(c.compress(d) + c.flush())
    for z, d in zip(streams, datas):
        assert engine.inflate_batch([z], [len(d)])[0] == d, len(d)
    own = engine.deflate_batch([datas[3], datas[4]], level=6, strategy=4)  # this library's CompressionStrategy.Fixed
    mixed_z = streams + own + [zlib.compress(datas[2], 6)]
    mixed_d = datas + [datas[3], datas[4], datas[2]]
    assert engine.inflate_batch(mixed_z, [len(d) for d in mixed_d]) == mixed_d
    bad = bytearray(streams[2])
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(Exception):
        engine.inflate_batch([bytes(bad)], [len(datas[2])])
    big = datagen.english(16 << 20, 7)
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    z = c.compress(big) + c.flush()
    d_z = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    out = torch.empty(len(big), dtype=torch.uint8, device="cuda")
    a = ([d_z.data_ptr()], [len(z)], [out.data_ptr()], [len(big)])
    engine.inflate_batch_device(*a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = engine.inflate_batch_device(*a)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    assert n == len(big) and out.cpu().numpy().tobytes() == big
    rate_floors.check(len(big) / dt >= 200e6, "inflate of 16 MiB under Z_FIXED: %.1f ms = %.1f MB/s" % (dt * 1e3, len(big) / dt / 1e6))
