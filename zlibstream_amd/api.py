"""Python mirror of the reference's Stream-level API for the deflate path.

Names, argument meaning and error behaviour follow src/ZlibStream/
ZlibOutputStream.cs, ZlibOptions.cs, CompressionLevel.cs,
CompressionStrategy.cs, FlushMode.cs, ZlibStreamException.cs and
ThrowHelper.cs:21-23 of the reference, so that the parity tests read like the
reference's own (tests/ZlibStream.Tests/ZlibStreamTests.Roundtrip.cs).
"""
import ctypes
import enum
import io

from . import _native


class CompressionLevel(enum.IntEnum):  # CompressionLevel.cs
    DefaultCompression = -1
    Level0 = 0
    NoCompression = 0
    Level1 = 1
    BestSpeed = 1
    Level2 = 2
    Level3 = 3
    Level4 = 4
    Level5 = 5
    Level6 = 6
    Level7 = 7
    Level8 = 8
    Level9 = 9
    BestCompression = 9


class CompressionStrategy(enum.IntEnum):  # CompressionStrategy.cs
    DefaultStrategy = 0
    Filtered = 1
    HuffmanOnly = 2
    Rle = 3
    Fixed = 4


class FlushMode(enum.IntEnum):  # FlushMode.cs
    NoFlush = 0
    PartialFlush = 1
    SyncFlush = 2
    FullFlush = 3
    Finish = 4


class CompressionState(enum.IntEnum):  # CompressionState.cs
    ZVERSIONERROR = -6
    ZBUFERROR = -5
    ZMEMERROR = -4
    ZDATAERROR = -3
    ZSTREAMERROR = -2
    ZERRNO = -1
    ZOK = 0
    ZSTREAMEND = 1
    ZNEEDDICT = 2


class ZlibStreamException(Exception):  # ZlibStreamException.cs
    pass


class ZlibOptions:  # ZlibOptions.cs
    def __init__(self, CompressionLevel=None, CompressionStrategy=CompressionStrategy.DefaultStrategy,
                 FlushMode=FlushMode.NoFlush):
        self.CompressionLevel = CompressionLevel
        self.CompressionStrategy = CompressionStrategy
        self.FlushMode = FlushMode


def deflate_bound(n):
    return int(_native.lib().zs_deflate_bound(int(n)))


class Engine:
    """One zs_ctx: a GPU plus its reusable workspace."""

    def __init__(self, device=0):
        self._lib = _native.lib()
        h = ctypes.c_void_p()
        rc = self._lib.zs_ctx_create(int(device), ctypes.byref(h))
        if rc != 0 or not h:
            raise RuntimeError("zs_ctx_create(device=%d) failed with %d: no usable MI355X / HIP device. "
                               "There is no CPU fallback." % (device, rc))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def last_error(self):
        return (self._lib.zs_ctx_last_error(self._h) or b"").decode()

    def set_profiling(self, on):
        self._lib.zs_ctx_set_profiling(self._h, 1 if on else 0)

    def counter(self, name):
        """zs_ctx_counter: "fast_rounds", "fast_fallbacks", "round_runs", "cut_rounds", "lit_fallbacks", "lit_engine_bytes"."""
        return int(self._lib.zs_ctx_counter(self._h, name.encode()))

    def stage_ms(self):
        n = self._lib.zs_ctx_stage_count(self._h)
        return {self._lib.zs_ctx_stage_name(self._h, i).decode(): self._lib.zs_ctx_stage_ms(self._h, i) for i in range(n)}

    def _call_batch(self, fn, in_ptrs, in_lens, out_ptrs, out_caps, level, strategy, hash_variant, extra=()):
        n = len(in_ptrs)
        VP = ctypes.c_void_p * n
        I64 = ctypes.c_int64 * n
        I32 = ctypes.c_int * n
        out_len = I64()
        status = I32()
        rc = fn(self._h, n, VP(*in_ptrs), I64(*in_lens), VP(*out_ptrs), I64(*out_caps), out_len, status, int(level),
                int(strategy), int(hash_variant), *extra)
        return rc, list(out_len), list(status)

    class DeviceBatch:
        """The four argument arrays of zs_deflate_batch_device as C arrays, made once: a C# or C++ caller hands the library
        arrays it already has, while turning Python lists of thousands of streams into ctypes arrays costs a millisecond or
        two per call -- as much as the device takes for a tenth of such a batch."""

        def __init__(self, in_ptrs, in_lens, out_ptrs, out_caps):
            n = self.n = len(in_ptrs)
            self.in_ptrs = (ctypes.c_void_p * n)(*in_ptrs)
            self.in_lens = (ctypes.c_int64 * n)(*in_lens)
            self.out_ptrs = (ctypes.c_void_p * n)(*out_ptrs)
            self.out_caps = (ctypes.c_int64 * n)(*out_caps)
            self.out_len = (ctypes.c_int64 * n)()
            self.status = (ctypes.c_int * n)()

    def deflate_device_batch(self, batch, level=6, strategy=0, hash_variant=0, stream=None):
        """zs_deflate_batch_device on a DeviceBatch; the output lengths are left in batch.out_len (a C array)."""
        rc = self._lib.zs_deflate_batch_device(self._h, batch.n, batch.in_ptrs, batch.in_lens, batch.out_ptrs, batch.out_caps,
                                               batch.out_len, batch.status, int(level), int(strategy), int(hash_variant),
                                               ctypes.c_void_p(stream or 0))
        if rc != 0:
            raise ZlibStreamException("deflating: " + self.last_error())
        return batch.out_len

    def deflate_batch_device(self, in_ptrs, in_lens, out_ptrs, out_caps, level=6, strategy=0, hash_variant=0, stream=None):
        """Device-resident buffers (raw device pointers as ints).  Returns the output lengths."""
        rc, lens, status = self._call_batch(self._lib.zs_deflate_batch_device, in_ptrs, in_lens, out_ptrs, out_caps, level,
                                            strategy, hash_variant, (ctypes.c_void_p(stream or 0),))
        if rc != 0:
            raise ZlibStreamException("deflating: " + self.last_error())
        return lens

    def deflate_writes_device(self, in_ptr, in_len, write_ends, out_ptr, out_cap, level=6, strategy=0, hash_variant=0, stream=None):
        """One device-resident stream written in several NoFlush Writes (zs_deflate_writes_device): `write_ends` are the
        cumulative Write ends (a sequence of ints or a ctypes int64 array).  Returns the output length."""
        if not isinstance(write_ends, ctypes.Array):
            write_ends = (ctypes.c_int64 * len(write_ends))(*write_ends)
        olen = ctypes.c_int64(0)
        rc = self._lib.zs_deflate_writes_device(self._h, ctypes.c_void_p(in_ptr), int(in_len), write_ends, len(write_ends),
                                                ctypes.c_void_p(out_ptr), int(out_cap), ctypes.byref(olen), int(level), int(strategy),
                                                int(hash_variant), ctypes.c_void_p(stream or 0))
        if rc != 0:
            raise ZlibStreamException("deflating: " + self.last_error())
        return olen.value

    def deflate_batch(self, buffers, level=6, strategy=0, hash_variant=0):
        """Host buffers (bytes-like) -> list of zlib streams (bytes)."""
        bufs = [bytes(b) for b in buffers]
        n = len(bufs)
        if n == 0:
            return []
        keep = [ctypes.create_string_buffer(b, len(b)) if len(b) else ctypes.create_string_buffer(1) for b in bufs]
        caps = [deflate_bound(len(b)) for b in bufs]
        outs = [ctypes.create_string_buffer(c) for c in caps]
        rc, lens, status = self._call_batch(self._lib.zs_deflate_batch, [ctypes.addressof(k) for k in keep],
                                            [len(b) for b in bufs], [ctypes.addressof(o) for o in outs], caps, level,
                                            strategy, hash_variant)
        if rc != 0:
            raise ZlibStreamException("deflating: " + self.last_error())
        return [outs[i].raw[:lens[i]] for i in range(n)]


    def _call_inflate(self, fn, in_ptrs, in_lens, out_ptrs, out_caps, extra=()):
        n = len(in_ptrs)
        VP = ctypes.c_void_p * n
        I64 = ctypes.c_int64 * n
        I32 = ctypes.c_int * n
        out_len = I64()
        status = I32()
        rc = fn(self._h, n, VP(*in_ptrs), I64(*in_lens), VP(*out_ptrs), I64(*out_caps), out_len, status, *extra)
        return rc, list(out_len), list(status)

    def inflate_batch_device(self, in_ptrs, in_lens, out_ptrs, out_caps, stream=None):
        rc, lens, status = self._call_inflate(self._lib.zs_inflate_batch_device, in_ptrs, in_lens, out_ptrs, out_caps,
                                              (ctypes.c_void_p(stream or 0),))
        if rc != 0:
            raise ZlibStreamException("inflating: " + self.last_error())  # ThrowHelper.cs:21-23
        return lens

    def inflate_batch(self, streams, out_sizes):
        """Host buffers: zlib streams -> decoded bytes; out_sizes[i] is the capacity for stream i."""
        zs = [bytes(z) for z in streams]
        n = len(zs)
        if n == 0:
            return []
        keep = [ctypes.create_string_buffer(z, len(z)) if len(z) else ctypes.create_string_buffer(1) for z in zs]
        outs = [ctypes.create_string_buffer(max(int(c), 1)) for c in out_sizes]
        rc, lens, status = self._call_inflate(self._lib.zs_inflate_batch, [ctypes.addressof(k) for k in keep], [len(z) for z in zs],
                                              [ctypes.addressof(o) for o in outs], [int(c) for c in out_sizes])
        if rc != 0:
            raise ZlibStreamException("inflating: " + self.last_error())
        return [outs[i].raw[:lens[i]] for i in range(n)]


def deflate_batch_multi(engines, buffers, level=6, strategy=0, hash_variant=0):
    """Host buffers sharded over several engines (one per GPU) by zs_deflate_batch_multi -> zlib streams in input order."""
    lib = _native.lib()
    bufs = [bytes(b) for b in buffers]
    n = len(bufs)
    if n == 0:
        return []
    keep = [ctypes.create_string_buffer(b, len(b)) if len(b) else ctypes.create_string_buffer(1) for b in bufs]
    caps = [deflate_bound(len(b)) for b in bufs]
    outs = [ctypes.create_string_buffer(c) for c in caps]
    VP, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    out_len, status = I64(), I32()
    ctxs = (ctypes.c_void_p * len(engines))(*[e.handle for e in engines])
    rc = lib.zs_deflate_batch_multi(ctxs, len(engines), n, VP(*[ctypes.addressof(k) for k in keep]), I64(*[len(b) for b in bufs]),
                                    VP(*[ctypes.addressof(o) for o in outs]), I64(*caps), out_len, status, int(level), int(strategy),
                                    int(hash_variant))
    if rc != 0:
        bad = [e.last_error() for e in engines if e.last_error()]
        raise ZlibStreamException("deflating: " + (bad[0] if bad else "error %d" % rc))
    return [outs[i].raw[:out_len[i]] for i in range(n)]


def inflate_batch_multi(engines, streams, out_sizes):
    """zlib streams sharded over several engines by zs_inflate_batch_multi -> decoded bytes in input order."""
    lib = _native.lib()
    zs = [bytes(z) for z in streams]
    n = len(zs)
    if n == 0:
        return []
    keep = [ctypes.create_string_buffer(z, len(z)) if len(z) else ctypes.create_string_buffer(1) for z in zs]
    outs = [ctypes.create_string_buffer(max(int(c), 1)) for c in out_sizes]
    VP, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    out_len, status = I64(), I32()
    ctxs = (ctypes.c_void_p * len(engines))(*[e.handle for e in engines])
    rc = lib.zs_inflate_batch_multi(ctxs, len(engines), n, VP(*[ctypes.addressof(k) for k in keep]), I64(*[len(z) for z in zs]),
                                    VP(*[ctypes.addressof(o) for o in outs]), I64(*[int(c) for c in out_sizes]), out_len, status)
    if rc != 0:
        bad = [e.last_error() for e in engines if e.last_error()]
        raise ZlibStreamException("inflating: " + (bad[0] if bad else "error %d" % rc))
    return [outs[i].raw[:out_len[i]] for i in range(n)]


def _multi_device_call(fn_name, engines, in_ptrs, in_lens, out_ptrs, out_caps, part_of, extra=()):
    lib = _native.lib()
    n = len(in_ptrs)
    VP, I64, I32 = ctypes.c_void_p * max(n, 1), ctypes.c_int64 * max(n, 1), ctypes.c_int * max(n, 1)
    out_len, status = I64(), I32()
    ctxs = (ctypes.c_void_p * len(engines))(*[e.handle for e in engines])
    rc = getattr(lib, fn_name)(ctxs, len(engines), n, VP(*[int(p) for p in in_ptrs]), I64(*[int(x) for x in in_lens]), VP(*[int(p) for p in out_ptrs]),
                               I64(*[int(x) for x in out_caps]), out_len, status, I32(*[int(x) for x in part_of]), *extra)
    return rc, list(out_len)[:n], list(status)[:n]


def deflate_batch_multi_device(engines, in_ptrs, in_lens, out_ptrs, out_caps, part_of, level=6, strategy=0, hash_variant=0):
    """Device-resident buffers sharded over several engines (zs_deflate_batch_multi_device): buffer i and its output live on
    the GPU of engines[part_of[i]] (shard.partition / zs_partition gives a balanced part_of).  -> compressed lengths."""
    rc, lens, _ = _multi_device_call("zs_deflate_batch_multi_device", engines, in_ptrs, in_lens, out_ptrs, out_caps, part_of,
                                     (int(level), int(strategy), int(hash_variant)))
    if rc != 0:
        bad = [e.last_error() for e in engines if e.last_error()]
        raise ZlibStreamException("deflating: " + (bad[0] if bad else "error %d" % rc))
    return lens


def inflate_batch_multi_device(engines, in_ptrs, in_lens, out_ptrs, out_caps, part_of):
    """Device-resident zlib streams sharded over several engines (zs_inflate_batch_multi_device).  -> decoded lengths."""
    rc, lens, _ = _multi_device_call("zs_inflate_batch_multi_device", engines, in_ptrs, in_lens, out_ptrs, out_caps, part_of)
    if rc != 0:
        bad = [e.last_error() for e in engines if e.last_error()]
        raise ZlibStreamException("inflating: " + (bad[0] if bad else "error %d" % rc))
    return lens


def device_count():
    return int(_native.lib().zs_device_count())


def png_filter_device(engine, pixels_ptr, row_bytes, height, bpp, filter_type, out_ptr, stream=None):
    """PNG scanline filtering of a device-resident image into a device buffer of height * (row_bytes + 1) bytes."""
    rc = _native.lib().zs_png_filter_device(engine.handle, ctypes.c_void_p(pixels_ptr), int(row_bytes), int(height), int(bpp),
                                            int(filter_type), ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream or 0))
    if rc != 0:
        raise ValueError("zs_png_filter_device rejected the arguments (%d)" % rc)


_default_engine = None


def default_engine():
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def compress(data, level=6, strategy=0, engine=None):
    """`using (var s = new ZlibOutputStream(ms, level)) s.Write(data)` in one call."""
    return (engine or default_engine()).deflate_batch([data], level, strategy)[0]


class ZlibInputStream(io.RawIOBase):
    """ZlibInputStream.cs: a read-only stream that inflates `base_stream`.

    Same loop as ReadCore (ZlibInputStream.cs:133-186): 8 KiB chunks of BaseStream go to `Inflate(flush)` while the
    caller's buffer has room and the state is ZOK.  The device engine (zs_inflate) takes the chunks in; a whole stream
    is decoded on the GPU at the call that completes it, and a call that arrives without input (BaseStream has nothing more
    for now: a reader behind a writer's flush) decodes the complete blocks of what has arrived and serves them.
    """

    BUFFER_SIZE = 8192

    def __init__(self, base_stream, engine=None):
        super().__init__()
        self.BaseStream = base_stream
        self._engine = engine or default_engine()
        self._lib = _native.lib()
        self._z = self._lib.zs_inflate_init(self._engine.handle, 15)
        if not self._z:
            raise ValueError("zs_inflate_init")
        self._chunk = b""
        self._chunk_pos = 0
        self._no_more_input = False
        self.TotalIn = 0
        self.TotalOut = 0
        self.Adler = 1

    def readable(self):
        return True

    def close(self):
        if getattr(self, "_z", None):
            self._lib.zs_inflate_end(self._z)
            self._z = None
        super().close()

    def __del__(self):
        try:
            if getattr(self, "_z", None):
                self._lib.zs_inflate_end(self._z)
                self._z = None
        except Exception:
            pass

    def readinto(self, b):
        view = memoryview(b).cast("B")
        if len(view) == 0:
            return 0
        out = (ctypes.c_uint8 * len(view))()
        avail_out = ctypes.c_int32(len(view))
        adler, tin, tout = ctypes.c_uint32(self.Adler), ctypes.c_int64(self.TotalIn), ctypes.c_int64(self.TotalOut)
        out_index = 0
        while True:
            if self._chunk_pos == len(self._chunk) and not self._no_more_input:
                self._chunk = self.BaseStream.read(self.BUFFER_SIZE) or b""
                self._chunk_pos = 0
            n_in = len(self._chunk) - self._chunk_pos
            src = (ctypes.c_uint8 * max(1, n_in)).from_buffer_copy(self._chunk[self._chunk_pos:] or b"\0")
            avail_in = ctypes.c_int32(n_in)
            before_out = avail_out.value
            state = self._lib.zs_inflate(self._z, ctypes.cast(src, ctypes.c_void_p), ctypes.byref(avail_in),
                                         ctypes.c_void_p(ctypes.addressof(out) + out_index), ctypes.byref(avail_out), 0,
                                         ctypes.byref(adler), ctypes.byref(tin), ctypes.byref(tout))
            self._chunk_pos += n_in - avail_in.value
            out_index += before_out - avail_out.value
            if state not in (0, 1):
                msg = (self._lib.zs_inflate_message(self._z) or b"").decode()
                raise ZlibStreamException("inflating: " + msg)  # ThrowHelper.cs:21-23
            if not (avail_out.value > 0 and state == 0):
                break
        self.Adler, self.TotalIn, self.TotalOut = adler.value, tin.value, tout.value
        n = len(view) - avail_out.value
        view[:n] = bytes(out)[:n]
        return n


class ZlibOutputStream(io.RawIOBase):
    """ZlibOutputStream.cs: a write-only stream that deflates into `base_stream`.

    Same loop structure as WriteCore / Finish (ZlibOutputStream.cs:125-168,
    213-256): 512-byte chunk buffer, `Deflate(flush)` until the input is
    consumed and the chunk buffer was not filled completely.
    """

    BUFFER_SIZE = 512

    def __init__(self, base_stream, level_or_options=CompressionLevel.DefaultCompression, engine=None, hash_variant=0):
        super().__init__()
        if isinstance(level_or_options, ZlibOptions):
            self.Options = level_or_options
        else:
            self.Options = ZlibOptions(CompressionLevel=level_or_options)
        self.BaseStream = base_stream
        self._engine = engine or default_engine()
        self._lib = _native.lib()
        # ZlibStream.cs:18-29: a null level means inflate mode -- the stream then inflates what is written to it
        self._compress = self.Options.CompressionLevel is not None
        if self._compress:
            level = int(self.Options.CompressionLevel)
            strategy = int(self.Options.CompressionStrategy)
            if level < -1 or level > 9:
                raise ValueError("level")  # ArgumentOutOfRangeException (Deflate.cs:273-276)
            if strategy < 0 or strategy > 4:
                raise ValueError("strategy")
            self._z = self._lib.zs_deflate_init(self._engine.handle, level, strategy, 15, 8, hash_variant)
            if not self._z:
                raise ValueError("zs_deflate_init rejected the arguments")
        else:
            self._z = self._lib.zs_inflate_init(self._engine.handle, 15)
            if not self._z:
                raise ValueError("zs_inflate_init")
        self._chunk = ctypes.create_string_buffer(self.BUFFER_SIZE)
        self._finished = False
        self._total_in = ctypes.c_int64(0)
        self._total_out = ctypes.c_int64(0)
        self._adler = ctypes.c_uint32(1)

    @property
    def TotalIn(self):
        return self._total_in.value

    @property
    def TotalOut(self):
        return self._total_out.value

    def writable(self):
        return True

    def readable(self):
        return False

    def seekable(self):
        return False

    def _deflate_loop(self, data, flush, until_end):
        buf = ctypes.create_string_buffer(bytes(data), len(data)) if len(data) else None
        avail_in = ctypes.c_int32(len(data))
        consumed = 0
        while True:
            avail_out = ctypes.c_int32(self.BUFFER_SIZE)
            next_in = ctypes.c_void_p(ctypes.addressof(buf) + consumed) if buf is not None else ctypes.c_void_p(0)
            before = avail_in.value
            fn = self._lib.zs_deflate if self._compress else self._lib.zs_inflate
            state = fn(self._z, next_in, ctypes.byref(avail_in), ctypes.addressof(self._chunk), ctypes.byref(avail_out), int(flush),
                       ctypes.byref(self._adler), ctypes.byref(self._total_in), ctypes.byref(self._total_out))
            consumed += before - avail_in.value
            if state not in (CompressionState.ZOK, CompressionState.ZSTREAMEND):
                msg = (self._lib.zs_last_message if self._compress else self._lib.zs_inflate_message)(self._z)
                raise ZlibStreamException(("deflating: " if self._compress else "inflating: ") + (msg.decode() if msg else ""))  # ThrowHelper.cs:21-23
            got = self.BUFFER_SIZE - avail_out.value
            if got:
                self.BaseStream.write(self._chunk.raw[:got])
            if not self._compress and avail_in.value == 0 and avail_out.value == 0 and not until_end:
                break  # ZlibOutputStream.cs:155-158
            if state == CompressionState.ZSTREAMEND:
                break
            if not (avail_in.value > 0 or avail_out.value == 0):
                break

    def write(self, b):
        if self._finished:
            raise ValueError("write to finished stream")
        b = bytes(b)
        if len(b) == 0:
            return 0  # WriteCore returns immediately on an empty span
        self._deflate_loop(b, self.Options.FlushMode, False)
        return len(b)

    def WriteByte(self, value):
        self.write(bytes([value]))

    def Finish(self):
        if not self._finished:
            self._deflate_loop(b"", FlushMode.Finish, True)
            self._finished = True
            if hasattr(self.BaseStream, "flush"):
                self.BaseStream.flush()

    def close(self):
        if not self.closed:
            try:
                if getattr(self, "_z", None):
                    self.Finish()
            finally:
                if getattr(self, "_z", None):
                    (self._lib.zs_deflate_end if self._compress else self._lib.zs_inflate_end)(self._z)
                    self._z = None
                super().close()
