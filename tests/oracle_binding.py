"""ctypes binding of oracle/libzsoracle.so -- the CHECKER.  Only tests (and
bench.py's cpu_baseline leg / smoke()) may touch the oracle."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORPUS = os.path.join(ROOT, "tests", "golden", "corpus")
# .gitattributes of the reference normalises text files to LF; the published sizes are for the canonical CRLF files
CRLF_FILES = {"alice29.txt", "lcet10.txt", "plrabn12.txt"}


def corpus(name, canonical=False):
    d = open(os.path.join(CORPUS, name), "rb").read()
    if canonical and name in CRLF_FILES:
        d = d.replace(b"\n", b"\r\n")
    return d


class Oracle:
    def __init__(self):
        path = os.path.join(ROOT, "oracle", "libzsoracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("zs_oracle.c", "zs_inflate_oracle.c", "zs_oracle.h")]
        if not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "libzsoracle.so"], check=True)
        L = ctypes.CDLL(path)
        L.zso_compress_stream.restype = ctypes.c_size_t
        L.zso_compress_stream.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                          ctypes.c_size_t, ctypes.c_void_p]
        L.zso_adler32.restype = ctypes.c_uint32
        L.zso_adler32.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
        L.zso_hash_u32.restype = ctypes.c_uint32
        L.zso_hash_u32.argtypes = [ctypes.c_uint32, ctypes.c_int]
        L.zso_inflate_oneshot.restype = ctypes.c_int
        L.zso_inflate_oneshot.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                                          ctypes.POINTER(ctypes.c_char_p)]
        L.zso_dotnet_random_bytes.restype = None
        L.zso_dotnet_random_bytes.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        self.L = L

    def compress(self, data, level=6, strategy=0, chunks=None, hash_variant=0, flush=0):
        data = bytes(data)
        cap = len(data) + len(data) // 8 + 1024 + 48 * len(chunks or [])
        out = ctypes.create_string_buffer(cap)
        if chunks:
            arr = (ctypes.c_size_t * len(chunks))(*chunks)
            n = self.L.zso_compress_stream(data, len(data), arr, len(chunks), level, strategy, flush, hash_variant, out, cap, None)
        else:
            n = self.L.zso_compress_stream(data, len(data), None, 0, level, strategy, flush, hash_variant, out, cap, None)
        if n == ctypes.c_size_t(-1).value:
            raise RuntimeError("oracle deflate failed")
        return out.raw[:n]

    def compress_writes(self, data, level, strategy, chunks, flushes, hash_variant=0):
        """One Write per chunk with ZlibOptions.FlushMode set to flushes[i] before it, then Finish."""
        data = bytes(data)
        cap = len(data) + len(data) // 8 + 1024 + 64 * len(chunks)
        out = ctypes.create_string_buffer(cap)
        arr = (ctypes.c_size_t * len(chunks))(*chunks)
        fl = (ctypes.c_int * len(chunks))(*flushes)
        self.L.zso_compress_stream_modes.restype = ctypes.c_size_t
        self.L.zso_compress_stream_modes.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        n = self.L.zso_compress_stream_modes(data, len(data), arr, len(chunks), level, strategy, 0, fl, hash_variant, out, cap, None)
        if n == ctypes.c_size_t(-1).value:
            raise RuntimeError("oracle deflate failed")
        return out.raw[:n]

    def adler32(self, data, seed=1):
        return self.L.zso_adler32(seed, bytes(data), len(data))

    def inflate(self, z, out_cap):
        out = ctypes.create_string_buffer(max(out_cap, 1))
        olen, used = ctypes.c_size_t(0), ctypes.c_size_t(0)
        msg = ctypes.c_char_p()
        rc = self.L.zso_inflate_oneshot(bytes(z), len(z), out, out_cap, ctypes.byref(olen), ctypes.byref(used), ctypes.byref(msg))
        return rc, out.raw[:olen.value], (msg.value.decode() if msg.value else None)

    def dotnet_random(self, seed, n):
        buf = ctypes.create_string_buffer(max(n, 1))
        self.L.zso_dotnet_random_bytes(seed, buf, n)
        return buf.raw[:n]
