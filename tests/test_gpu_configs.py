"""-m gpu: BASELINE.json configs 3, 4 and 5 at their stated sizes, determinism, the multi-context batch entry and the
limits of the C ABI.  Oracle comparisons are byte-exact; where the oracle would take minutes (full 64 MiB / 1 GiB
sizes) the size-independent properties stand in: round trip through an independent decoder, the Adler-32 trailer, and
`torch.equal` on the device for inflate."""
import ctypes
import hashlib
import threading
import zlib

import pytest

import oracle_binding
from zlibstream_amd import (Engine, ZlibStreamException, datagen, deflate_batch_multi, deflate_bound, device_count,
                            inflate_batch_multi)

pytestmark = pytest.mark.gpu


def _roundtrip_ok(z, data, header=None):
    assert zlib.decompress(z) == data
    assert int.from_bytes(z[-4:], "big") == zlib.adler32(data)
    if header:
        assert z[:2] == header


# ---------------------------------------------------------------- config 4: 1024 x 1 MiB (datagen.batch_buffer)
def test_config4_batch1024_first_128_bit_exact_all_1024_roundtrip(engine, oracle):
    """Buffers 0..127 byte-exact against the oracle; all 1024 (one device batch, inputs resident in HBM) through round trip +
    trailer.  Even buffers are english, odd ones sparse rows."""
    import torch
    bufs = [datagen.batch_buffer(i) for i in range(1024)]
    assert all(len(b) == 1 << 20 for b in bufs)
    d_ins = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_outs = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    lens = engine.deflate_batch_device([t.data_ptr() for t in d_ins], [len(b) for b in bufs], [t.data_ptr() for t in d_outs], caps, level=6,
                                       stream=torch.cuda.current_stream().cuda_stream)
    for i, b in enumerate(bufs):
        z = d_outs[i][:lens[i]].cpu().numpy().tobytes()
        _roundtrip_ok(z, b, b"\x78\x9c")
        if i < 128:
            assert z == oracle.compress(b, 6), "buffer %d differs from the oracle" % i


# ---------------------------------------------------------------- config 3: sparse64 at levels 1 and 9 (level 6 is in test_gpu_parity)
@pytest.mark.parametrize("level,header", [(1, b"\x78\x01"), (9, b"\x78\xda")])
def test_config3_sparse64_levels_1_and_9(engine, oracle, level, header):
    import torch
    data = datagen.sparse(4096, 4096)
    n = len(data)
    assert n == 64 << 20
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    m = engine.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=level,
                                    stream=torch.cuda.current_stream().cuda_stream)[0]
    _roundtrip_ok(d_out[:m].cpu().numpy().tobytes(), data, header)
    small = data[:4 << 20]
    assert engine.deflate_batch([small], level=level)[0] == oracle.compress(small, level)


def test_config2_english64_levels_1_and_9_sample_and_roundtrip(engine, oracle):
    data = datagen.english(64 << 20)
    for level in (9,):
        z = engine.deflate_batch([data], level=level)[0]
        _roundtrip_ok(z, data, b"\x78\xda")
    small = data[:4 << 20]
    for level in (1, 9):
        assert engine.deflate_batch([small], level=level)[0] == oracle.compress(small, level)


def test_config2_english64_at_the_fast_levels_bit_exact_and_at_rate(engine, oracle, rate_floors):
    """english64 under DeflateFast (levels 1-3, Deflate.Fast.cs:20-128), the whole 64 MiB against the oracle's bytes: one stream
    as rounds over 8191 chunks, 32 consecutive ones to a workgroup while most of them still change (zs_fast_sweep.h "Rounds").
    Measured 1.38 / 1.17 / 1.93 GB/s at levels 1 / 2 / 3 with the input resident in HBM (one workgroup for the stream: 48 / 21
    MB/s at levels 1 / 3); the floors leave a third of margin for a busy box."""
    import time
    import torch
    data = datagen.english(64 << 20)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(len(data))
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for level, floor in ((1, 900e6), (2, 780e6), (3, 1280e6)):
        engine.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=level)
        torch.cuda.synchronize()
        t = time.perf_counter()
        m = engine.deflate_batch_device([d_in.data_ptr()], [len(data)], [d_out.data_ptr()], [cap], level=level)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert d_out[:m].cpu().numpy().tobytes() == oracle.compress(data, level), level
        rate_floors.check(len(data) / dt >= floor, "level %d: %.1f ms = %.0f MB/s" % (level, dt * 1e3, len(data) / dt / 1e6))


# ---------------------------------------------------------------- config 5: 16 x 64 MiB level-6 streams -> 1 GiB
def test_config5_inflate_1gib_seeds_0_to_15_equal_on_device(engine):
    import torch
    size = 64 << 20
    cap = deflate_bound(size)
    d_in, d_z, z_len = [], [], []
    for i in range(16):
        t = torch.frombuffer(bytearray(datagen.english(size, (datagen.GOLDEN + i) & datagen.MASK)), dtype=torch.uint8).cuda()
        z = torch.empty(cap, dtype=torch.uint8, device="cuda")
        n = engine.deflate_batch_device([t.data_ptr()], [size], [z.data_ptr()], [cap], level=6)[0]
        d_in.append(t), d_z.append(z[:n].clone()), z_len.append(n)
        del z
    outs = [torch.zeros(size, dtype=torch.uint8, device="cuda") for _ in range(16)]
    lens = engine.inflate_batch_device([z.data_ptr() for z in d_z], z_len, [o.data_ptr() for o in outs], [size] * 16)
    assert lens == [size] * 16
    for i in range(16):
        assert torch.equal(outs[i], d_in[i]), "stream %d" % i
    # a flipped payload byte in one stream must surface as that stream's error, not as silence
    bad = d_z[3].clone()
    bad[z_len[3] // 2] ^= 0x10
    with pytest.raises(ZlibStreamException):
        engine.inflate_batch_device([bad.data_ptr()], [z_len[3]], [outs[3].data_ptr()], [size])


# ---------------------------------------------------------------- determinism (the link kernel relies on LDS lane order)
def test_determinism_repeats_and_concurrent_contexts():
    """english64 level 6: 3 repeats on one context, then 2 more contexts concurrently (2 runs each): one distinct output."""
    import torch
    data = datagen.english(64 << 20)
    n = len(data)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    cap = deflate_bound(n)
    res, errs = [], []

    def work(reps):
        try:
            eng = Engine(0)
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            for _ in range(reps):
                m = eng.deflate_batch_device([d_in.data_ptr()], [n], [d_out.data_ptr()], [cap], level=6)[0]
                res.append(hashlib.sha256(d_out[:m].cpu().numpy().tobytes()).hexdigest())
            eng.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    work(3)
    th = [threading.Thread(target=work, args=(2,)) for _ in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert len(res) == 7 and len(set(res)) == 1, res


# ---------------------------------------------------------------- multi-context batch entry (one GPU here, 8 on the node)
def test_batch_multi_two_contexts_matches_single_context_and_oracle(engine, oracle):
    assert device_count() >= 1
    bufs = [datagen.batch_buffer(i, 256 << 10) for i in range(24)] + [b"", b"x", oracle_binding.corpus("kennedy.xls"),
                                                                       datagen.english(3 << 20, 9)]
    e2 = Engine(0)
    try:
        got = deflate_batch_multi([engine, e2], bufs, level=6)
        assert got == engine.deflate_batch(bufs, level=6)
        for i in (0, 1, 24, 25, 26):
            assert got[i] == oracle.compress(bufs[i], 6)
        back = inflate_batch_multi([engine, e2], got, [len(b) for b in bufs])
        assert back == bufs
        # three contexts, fewer buffers than contexts, and a failing buffer that leaves the others delivered
        e3 = Engine(0)
        assert deflate_batch_multi([engine, e2, e3], bufs[:2], level=9) == engine.deflate_batch(bufs[:2], level=9)
        e3.close()
    finally:
        e2.close()


def test_batch_multi_over_device_pointers(engine, oracle):
    """zs_deflate_batch_multi_device / zs_inflate_batch_multi_device: the N-GPU data path without PCIe in it -- every buffer
    and its output already resident on the GPU of the context zs_partition gave it to.  On the one GPU of this box: two and
    three contexts, against the single-context call and the oracle; a part_of outside the contexts and the same context
    passed twice are refused (ZS_STREAM_ERROR) with nothing written."""
    import torch
    from zlibstream_amd import deflate_batch_multi_device, deflate_bound, inflate_batch_multi_device, shard
    bufs = [datagen.batch_buffer(i, 256 << 10) for i in range(20)] + [b"x", oracle_binding.corpus("kennedy.xls"), datagen.english(3 << 20, 9)]
    d_in = [torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda() for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    d_out = [torch.empty(c, dtype=torch.uint8, device="cuda") for c in caps]
    want = engine.deflate_batch(bufs, level=6)
    e2, e3 = Engine(0), Engine(0)
    try:
        for engines in ([engine, e2], [engine, e2, e3]):
            part = shard.part_of([len(b) for b in bufs], len(engines))
            assert set(part) == set(range(len(engines)))
            lens = deflate_batch_multi_device(engines, [t.data_ptr() for t in d_in], [len(b) for b in bufs], [t.data_ptr() for t in d_out], caps, part)
            torch.cuda.synchronize()
            got = [d_out[i][:lens[i]].cpu().numpy().tobytes() for i in range(len(bufs))]
            assert got == want
            for i in (0, 1, 20, 21, 22):
                assert got[i] == oracle.compress(bufs[i], 6)
            # ... and back: the streams stay where they are, partitioned by the decoded sizes
            d_z = [d_out[i][:lens[i]] for i in range(len(bufs))]
            d_back = [torch.empty(max(len(b), 1), dtype=torch.uint8, device="cuda") for b in bufs]
            blens = inflate_batch_multi_device(engines, [t.data_ptr() for t in d_z], lens, [t.data_ptr() for t in d_back], [len(b) for b in bufs], part)
            torch.cuda.synchronize()
            assert [d_back[i][:blens[i]].cpu().numpy().tobytes() for i in range(len(bufs))] == bufs
        ptrs = ([t.data_ptr() for t in d_in], [len(b) for b in bufs], [t.data_ptr() for t in d_out], caps)
        with pytest.raises(Exception):
            deflate_batch_multi_device([engine, e2], *ptrs, [2] * len(bufs))     # a part outside the contexts
        with pytest.raises(Exception):
            deflate_batch_multi_device([engine, engine], *ptrs, [0] * len(bufs))  # one context twice
    finally:
        e2.close()
        e3.close()


def test_a_batch_larger_than_the_device_runs_in_sub_batches(engine, oracle):
    """A host batch whose staging and workspace (~19 bytes per input byte) exceed the device's memory -- 24 GiB of input: 300
    zero buffers and 84 text buffers of 64 MiB, the same two host buffers passed again and again -- is split into
    sub-batches by zs_deflate_batch itself (the streams are independent); the bytes are those of the buffers alone, which are
    the oracle's (a 4 MiB prefix compared here, the whole stream by its inflation)."""
    import zlib
    size = 64 << 20
    text, zeros = datagen.english(size, 77), bytes(size)
    one = engine.deflate_batch([text, zeros], level=6)
    assert zlib.decompress(one[0]) == text and zlib.decompress(one[1]) == zeros
    assert engine.deflate_batch([text[:4 << 20]], level=6)[0] == oracle.compress(text[:4 << 20], 6)
    kt, kz = ctypes.create_string_buffer(text, size), ctypes.create_string_buffer(zeros, size)
    n = 384
    is_text = [i % 32 < 7 for i in range(n)]  # 84 text streams among the 384
    caps = [deflate_bound(size) if t else 1 << 20 for t in is_text]
    outs = [ctypes.create_string_buffer(c) for c in caps]
    rc, lens, status = engine._call_batch(engine._lib.zs_deflate_batch, [ctypes.addressof(kt if t else kz) for t in is_text], [size] * n,
                                          [ctypes.addressof(o) for o in outs], caps, 6, 0, 0)
    assert rc == 0 and all(x == 0 for x in status), (rc, engine.last_error())
    for i in range(n):
        assert outs[i].raw[:lens[i]] == (one[0] if is_text[i] else one[1]), i


def test_batch_reports_every_stream_when_one_fails(engine, oracle):
    """One undersized output in a batch: that stream is ZBUFERROR, the others are delivered with their lengths."""
    bufs = [oracle_binding.corpus("sum"), oracle_binding.corpus("kennedy.xls"), oracle_binding.corpus("cp.html")]
    keep = [ctypes.create_string_buffer(b, len(b)) for b in bufs]
    caps = [deflate_bound(len(bufs[0])), 100, deflate_bound(len(bufs[2]))]
    outs = [ctypes.create_string_buffer(c) for c in caps]
    rc, lens, status = engine._call_batch(engine._lib.zs_deflate_batch, [ctypes.addressof(k) for k in keep], [len(b) for b in bufs],
                                          [ctypes.addressof(o) for o in outs], caps, 6, 0, 0)
    assert rc == -5 and status == [0, -5, 0]
    assert outs[0].raw[:lens[0]] == oracle.compress(bufs[0], 6)
    assert outs[2].raw[:lens[2]] == oracle.compress(bufs[2], 6)


# ---------------------------------------------------------------- Huft_build's "incomplete" rule (InfTree.cs:364)
class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, value, nbits):  # LSB-first fields
        self.v |= value << self.n
        self.n += nbits

    def code(self, code, nbits):  # Huffman codes go out MSB first
        for i in range(nbits - 1, -1, -1):
            self.put((code >> i) & 1, 1)

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def _single_code_stream(lit_len, dist_len, payload):
    """zlib stream with one dynamic block whose code lengths are: literal 'A' and END_BLOCK `lit_len` bits each... see below.
    lit_len == 1: 'A' and END_BLOCK both 1 bit (complete).  lit_len == 0: only END_BLOCK, `dist_len` bits (incomplete).
    The distance tree is one code of `dist_len` bits (incomplete; accepted by the reference only when dist_len == 1)."""
    b = _Bits()
    b.put(1, 1), b.put(2, 2)          # BFINAL, dynamic
    b.put(0, 5), b.put(0, 5), b.put(14, 4)  # HLIT 257, HDIST 1, HCLEN 18
    bl = {0: 1, 1: 2, 2: 3, 18: 3}    # bit-length code: 0 -> '0', 1 -> '10', 2 -> '110', 18 -> '111'
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    for s in order[:18]:
        b.put(bl.get(s, 0), 3)
    codes = {0: (0, 1), 1: (2, 2), 2: (6, 3), 18: (7, 3)}

    def zeros(k):
        while k:
            r = min(k, 138)
            assert r >= 11
            b.code(*codes[18]), b.put(r - 11, 7)
            k -= r
    if lit_len == 1:
        zeros(65), b.code(*codes[1]), zeros(190), b.code(*codes[1])   # 'A' = 65 and END_BLOCK = 256: 1 bit each
    else:
        zeros(138), zeros(118), b.code(*codes[dist_len])              # END_BLOCK alone
    b.code(*codes[dist_len])                                          # the single distance code
    if lit_len == 1:
        for ch in payload:
            assert ch == 65
            b.put(0, 1)
        b.put(1, 1)  # END_BLOCK
    else:
        b.put(0, dist_len)  # END_BLOCK = all-zero code
    raw = b.bytes()
    return b"\x78\x9c" + raw + zlib.adler32(payload).to_bytes(4, "big")


@pytest.mark.parametrize("lit_len,dist_len,payload,want", [
    (1, 1, b"AAA", None),                                   # single 1-bit distance code: accepted
    (1, 2, b"AAA", "incomplete distance tree"),             # single 2-bit distance code: rejected by Huft_build
    (0, 1, b"", None),                                      # END_BLOCK alone, 1 bit: accepted
    (0, 2, b"", "incomplete literal/length tree"),
])
def test_inflate_incomplete_single_code_trees_follow_huft_build(engine, oracle, lit_len, dist_len, payload, want):
    z = _single_code_stream(lit_len, dist_len, payload)
    rc, out, msg = oracle.inflate(z, 16)
    if want is None:
        assert rc == 1 and out == payload
        assert engine.inflate_batch([z], [16]) == [payload]
    else:
        assert rc == -3 and msg == want
        with pytest.raises(ZlibStreamException) as ei:
            engine.inflate_batch([z], [16])
        assert str(ei.value) == "inflating: " + want


# ---------------------------------------------------------------- no length limit on a stream
def test_zs_deflate_takes_streams_beyond_2_gib(engine):
    """A run of the device pipeline indexes its input with 32-bit positions, a stream does not: past 1 GiB of buffered
    NoFlush input the stream turns incremental (bounded host memory) and goes on.  34 Writes of 64 MiB at level 0 (the
    reference handles unbounded streams; round 1 truncated the length): every byte comes back, TotalIn is 64-bit."""
    lib = engine._lib
    z = lib.zs_deflate_init(engine.handle, 0, 0, 15, 8, 0)
    assert z
    try:
        piece = bytes(64 << 20)
        src = ctypes.create_string_buffer(piece, len(piece))
        cap = 4 << 20
        out = ctypes.create_string_buffer(cap)
        adler, tin, tout = ctypes.c_uint32(1), ctypes.c_int64(0), ctypes.c_int64(0)
        d = zlib.decompressobj()
        decoded = 0

        def call(n_in, flush):
            nonlocal decoded
            avail_in = ctypes.c_int32(n_in)
            while True:
                avail_out = ctypes.c_int32(cap)
                off = len(piece) - avail_in.value if n_in else 0
                rc = lib.zs_deflate(z, ctypes.addressof(src) + off, ctypes.byref(avail_in), ctypes.addressof(out), ctypes.byref(avail_out), flush,
                                    ctypes.byref(adler), ctypes.byref(tin), ctypes.byref(tout))
                assert rc in (0, 1), (rc, lib.zs_last_message(z))
                got = cap - avail_out.value
                if got:
                    chunk = d.decompress(out.raw[:got])
                    assert chunk.count(0) == len(chunk)
                    decoded += len(chunk)
                if rc == 1 or not (avail_in.value > 0 or avail_out.value == 0):
                    return rc
        n_writes = 34
        for _ in range(n_writes):
            assert call(len(piece), 0) == 0
        assert call(0, 4) == 1
        assert tin.value == n_writes * len(piece) > (1 << 31)
        assert decoded == n_writes * len(piece) and d.eof
    finally:
        lib.zs_deflate_end(z)


# ---------------------------------------------------------------- incremental streams: flushes deliver, memory stays bounded
@pytest.mark.parametrize("level,flush", [(6, 2), (1, 2), (6, 1), (9, 3), (0, 2)])
def test_flush_makes_the_data_readable_before_finish(engine, oracle, level, flush):
    """Deflate.cs:583-613: after a Write under Partial / Sync / Full flush the reader can decode everything written so far
    from the bytes delivered so far.  The stream's final bytes are the oracle's for the same Writes and mode."""
    import io
    from zlibstream_amd import CompressionLevel, FlushMode, ZlibInputStream, ZlibOptions, ZlibOutputStream
    text = datagen.english(700000, 31)
    pieces = [text[:300000], text[300000:300007], text[300007:520000], text[520000:]]

    class Pipe(io.RawIOBase):  # what the writer has delivered so far, read by the reader as far as it goes
        def __init__(self):
            super().__init__()
            self.data, self.rpos = bytearray(), 0

        def write(self, b):
            self.data += bytes(b)
            return len(b)

        def read(self, n=-1):
            k = len(self.data) - self.rpos if n < 0 else min(n, len(self.data) - self.rpos)
            r = bytes(self.data[self.rpos:self.rpos + k])
            self.rpos += k
            return r

    out = Pipe()
    s = ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level), FlushMode=FlushMode(flush)), engine=engine)
    # the repo's own reader on the other end (ZlibInputStream.ReadCore, ZlibInputStream.cs:133-186): behind every flushed
    # Write it reads, without the writer's Finish, exactly what that Write brought (zs_inflate decodes the complete blocks
    # of what has arrived at the call that comes without input; Inflate.cs:103-357)
    r = ZlibInputStream(out, engine=engine)
    done = b""
    for p in pieces:
        s.write(p)
        done += p
        d = zlib.decompressobj()
        assert d.decompress(bytes(out.data)) == done, "the data written so far is not readable after the flush"
        got = bytearray(len(p))
        assert r.readinto(got) == len(p) and bytes(got) == p, "ZlibInputStream does not deliver what the flush made readable"
    s.close()
    z = bytes(out.data)
    assert zlib.decompress(z) == text
    assert z == oracle.compress(text, level, 0, chunks=[len(p) for p in pieces], flush=flush)
    assert r.read(100) == b"" and r.TotalOut == len(text)  # the trailer behind Finish: the stream's end, nothing more
    r.close()


def test_noflush_stream_that_outgrows_the_buffer_becomes_incremental(oracle):
    """A NoFlush stream is buffered for the bulk pipeline only up to a limit (1 GiB; lowered here through the environment):
    beyond it the stream turns incremental -- the buffered part through the bulk pipeline as a run that is not the end, the
    rest continued from the suspended engine -- with bounded host memory and no length limit.  Bytes: the oracle's."""
    import io
    import os
    import subprocess
    import sys
    code = r"""
import io, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle_binding
from zlibstream_amd import CompressionLevel, Engine, ZlibOptions, ZlibOutputStream, datagen
eng = Engine(0)
oracle = oracle_binding.Oracle()
text = datagen.english(1500000, 77)
for level, wsize in ((6, 81920), (6, 1000), (4, 65536), (1, 50000), (0, 81920)):
    chunks = [min(wsize, len(text) - o) for o in range(0, len(text), wsize)]
    out = io.BytesIO()
    with ZlibOutputStream(out, ZlibOptions(CompressionLevel=CompressionLevel(level)), engine=eng) as s:
        o = 0
        early = 0
        for c in chunks:
            s.write(text[o:o + c]); o += c
            early = max(early, len(out.getvalue()))
    assert early > 1000, "nothing was delivered before Finish"
    assert out.getvalue() == oracle.compress(text, level, 0, chunks=chunks), (level, wsize)
print("OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZS_INC_SWITCH_BYTES="300000")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_zlib_output_stream_without_a_level_inflates(engine):
    """ZlibStream.cs:18-29: options without a CompressionLevel put the stream in inflate mode -- what is written to a
    ZlibOutputStream is inflated into its BaseStream (ZlibOutputStream.cs:125-168 with compress == false)."""
    import io
    from zlibstream_amd import ZlibOptions, ZlibOutputStream
    data = oracle_binding.corpus("alice29.txt") + datagen.sparse(64, 64)
    z = engine.deflate_batch([data], level=6)[0]
    out = io.BytesIO()
    with ZlibOutputStream(out, ZlibOptions(), engine=engine) as s:
        for o in range(0, len(z), 10000):
            s.write(z[o:o + 10000])
    assert out.getvalue() == data
    bad = z[:-1] + bytes([z[-1] ^ 1])
    with pytest.raises(ZlibStreamException) as ei:
        with ZlibOutputStream(io.BytesIO(), ZlibOptions(), engine=engine) as s:
            s.write(bad)
    assert str(ei.value) == "inflating: incorrect data check"


# ---------------------------------------------------------------- PNG scanline filters feeding the deflate path
def _png_filter_reference(img, row_bytes, height, bpp, ftype):
    """PNG specification 9.2 in numpy: -> height * (row_bytes + 1) bytes."""
    import numpy as np
    a = np.frombuffer(img, dtype=np.uint8).reshape(height, row_bytes).astype(np.int32)
    left = np.zeros_like(a)
    left[:, bpp:] = a[:, :-bpp]
    up = np.zeros_like(a)
    up[1:] = a[:-1]
    ul = np.zeros_like(a)
    ul[1:, bpp:] = a[:-1, :-bpp]
    p = left + up - ul
    pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    cands = [a, a - left, a - up, a - ((left + up) >> 1), a - paeth]
    cands = [(c & 0xFF).astype(np.uint8) for c in cands]
    out = np.empty((height, row_bytes + 1), dtype=np.uint8)
    for y in range(height):
        if ftype == 5:
            sums = [int(np.abs(c[y].view(np.int8).astype(np.int32)).sum()) for c in cands]
            f = sums.index(min(sums))
        else:
            f = ftype
        out[y, 0] = f
        out[y, 1:] = cands[f][y]
    return out.tobytes()


@pytest.mark.parametrize("ftype", [0, 1, 2, 3, 4, 5])
def test_png_filter_kernel_and_the_deflate_of_its_rows(engine, oracle, ftype):
    """The scanline filters on the device against a numpy restatement of the PNG specification, and the filtered rows
    through the deflate path: the oracle's bytes for the same rows (one Write), and an independent inflate gives them back."""
    import numpy as np
    import torch
    from zlibstream_amd import png_filter_device
    rng = np.random.default_rng(4)
    w, h = 333, 97
    grad = (np.add.outer(np.arange(h), np.arange(w * 4)) % 251).astype(np.uint8)
    noisy = (grad + rng.integers(0, 3, grad.shape, dtype=np.uint8)).astype(np.uint8)
    for img, row_bytes, height, bpp in ((datagen.sparse(512, 256), 2048, 256, 4), (noisy.tobytes(), w * 4, h, 4), (noisy.tobytes(), w * 4, h, 3),
                                        (bytes(rng.integers(0, 256, 77 * 5, dtype=np.uint8)), 77, 5, 1)):
        d_img = torch.frombuffer(bytearray(img), dtype=torch.uint8).cuda()
        d_out = torch.zeros(height * (row_bytes + 1), dtype=torch.uint8, device="cuda")
        png_filter_device(engine, d_img.data_ptr(), row_bytes, height, bpp, ftype, d_out.data_ptr())
        got = d_out.cpu().numpy().tobytes()
        assert got == _png_filter_reference(img, row_bytes, height, bpp, ftype), (ftype, row_bytes, height, bpp)
        cap = deflate_bound(len(got))
        d_z = torch.empty(cap, dtype=torch.uint8, device="cuda")
        n = engine.deflate_batch_device([d_out.data_ptr()], [len(got)], [d_z.data_ptr()], [cap], level=6)[0]
        z = d_z[:n].cpu().numpy().tobytes()
        assert z == oracle.compress(got, 6)
        assert zlib.decompress(z) == got


def test_z_stream_adler_field_after_every_write(engine):
    """ZLibStream.Adler is updated as ReadBuffer copies the caller's bytes (ZlibStream.cs:197-222): zs_deflate reports it
    after every call.  Pieces of every size class of the host routine (scalar tail, 8-byte steps, 32-byte AVX2 blocks, the
    4 KiB reduction boundary) against zlib.adler32 of what has been written so far."""
    lib = engine._lib
    import numpy as np
    data = np.random.default_rng(9).integers(0, 256, 3_000_000, dtype=np.uint8).tobytes()
    z = lib.zs_deflate_init(engine.handle, 1, 0, 15, 8, 0)
    assert z
    try:
        src = ctypes.create_string_buffer(data, len(data))
        out = ctypes.create_string_buffer(1 << 16)
        adler, tin, tout = ctypes.c_uint32(1), ctypes.c_int64(0), ctypes.c_int64(0)
        off = 0
        for n in (0, 1, 7, 8, 9, 31, 32, 33, 63, 64, 65, 100, 4095, 4096, 4097, 5551, 5552, 5553, 65536, 1_000_003, 1_234_567):
            avail_in, avail_out = ctypes.c_int32(n), ctypes.c_int32(1 << 16)
            rc = lib.zs_deflate(z, ctypes.addressof(src) + off, ctypes.byref(avail_in), ctypes.addressof(out), ctypes.byref(avail_out), 0,
                                ctypes.byref(adler), ctypes.byref(tin), ctypes.byref(tout))
            assert rc == 0 and avail_in.value == 0
            off += n
            assert tin.value == off and adler.value == zlib.adler32(data[:off]), (n, off)
    finally:
        lib.zs_deflate_end(z)
