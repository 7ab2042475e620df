"""ctypes binding of include/zsgpu.h.  Loading fails loudly: there is no CPU
fallback for the compression path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZS_DEV=1 ZS_LIB=<path>: another build of the same library (the A/B tools under tools/ select a variant this way instead of
# overwriting the product library); without the explicit development flag the in-tree build is the one that loads
LIB_PATH = (os.environ.get("ZS_LIB") if os.environ.get("ZS_DEV") == "1" else None) or os.path.join(_HERE, "libzsgpu.so")

SYMBOLS = [
    "zs_ctx_create", "zs_ctx_destroy", "zs_ctx_last_error", "zs_deflate_bound", "zs_deflate_batch_device",
    "zs_deflate_batch", "zs_ctx_counter", "zs_ctx_set_profiling", "zs_ctx_stage_count", "zs_ctx_stage_name", "zs_ctx_stage_ms",
    "zs_deflate_init", "zs_deflate", "zs_deflate_end", "zs_last_message", "zs_adler32_device",
    "zs_inflate_batch_device", "zs_inflate_batch", "zs_inflate_init", "zs_inflate", "zs_inflate_end", "zs_inflate_message", "zs_inflate_surplus",
    "zs_device_count", "zs_partition", "zs_deflate_batch_multi", "zs_inflate_batch_multi", "zs_png_filter_device", "zs_deflate_writes_device", "zs_deflate_batch_multi_device", "zs_inflate_batch_multi_device",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; the loader shares one HIP runtime between
    # torch and this library only when torch's copy is mapped first (same SONAME).
    import sys
    if "torch" not in sys.modules and os.environ.get("ZS_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "zlibstream_amd/libzsgpu.so is missing: build it with `python -m zlibstream_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    P = ctypes.POINTER
    L.zs_ctx_create.restype = i32
    L.zs_ctx_create.argtypes = [i32, P(vp)]
    L.zs_ctx_destroy.restype = None
    L.zs_ctx_destroy.argtypes = [vp]
    L.zs_ctx_last_error.restype = ctypes.c_char_p
    L.zs_ctx_last_error.argtypes = [vp]
    L.zs_deflate_bound.restype = i64
    L.zs_deflate_bound.argtypes = [i64]
    batch_args = [vp, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32), i32, i32, i32]
    L.zs_deflate_batch_device.restype = i32
    L.zs_deflate_batch_device.argtypes = batch_args + [vp]
    L.zs_deflate_batch.restype = i32
    L.zs_deflate_batch.argtypes = batch_args
    if hasattr(L, "zs_deflate_writes_device"):  # (an older build selected with ZS_LIB for an A/B run may lack it)
        L.zs_deflate_writes_device.restype = i32
        L.zs_deflate_writes_device.argtypes = [vp, vp, i64, P(i64), i64, vp, i64, P(i64), i32, i32, i32, vp]
    inf_args = [vp, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32)]
    L.zs_inflate_batch_device.restype = i32
    L.zs_inflate_batch_device.argtypes = inf_args + [vp]
    L.zs_inflate_batch.restype = i32
    L.zs_inflate_batch.argtypes = inf_args
    L.zs_ctx_counter.restype = i64
    L.zs_ctx_counter.argtypes = [vp, ctypes.c_char_p]
    L.zs_ctx_set_profiling.restype = None
    L.zs_ctx_set_profiling.argtypes = [vp, i32]
    L.zs_ctx_stage_count.restype = i32
    L.zs_ctx_stage_count.argtypes = [vp]
    L.zs_ctx_stage_name.restype = ctypes.c_char_p
    L.zs_ctx_stage_name.argtypes = [vp, i32]
    L.zs_ctx_stage_ms.restype = ctypes.c_double
    L.zs_ctx_stage_ms.argtypes = [vp, i32]
    L.zs_deflate_init.restype = vp
    L.zs_deflate_init.argtypes = [vp, i32, i32, i32, i32, i32]
    L.zs_deflate.restype = i32
    L.zs_deflate.argtypes = [vp, vp, P(ctypes.c_int32), vp, P(ctypes.c_int32), i32, P(ctypes.c_uint32), P(i64), P(i64)]
    L.zs_deflate_end.restype = None
    L.zs_deflate_end.argtypes = [vp]
    L.zs_last_message.restype = ctypes.c_char_p
    L.zs_last_message.argtypes = [vp]
    L.zs_inflate_init.restype = vp
    L.zs_inflate_init.argtypes = [vp, i32]
    L.zs_inflate.restype = i32
    L.zs_inflate.argtypes = [vp, vp, P(ctypes.c_int32), vp, P(ctypes.c_int32), i32, P(ctypes.c_uint32), P(i64), P(i64)]
    L.zs_inflate_end.restype = None
    L.zs_inflate_end.argtypes = [vp]
    L.zs_inflate_message.restype = ctypes.c_char_p
    L.zs_inflate_message.argtypes = [vp]
    L.zs_inflate_surplus.restype = i64
    L.zs_inflate_surplus.argtypes = [vp, P(vp)]
    L.zs_adler32_device.restype = i32
    L.zs_adler32_device.argtypes = [vp, vp, i64, ctypes.c_uint32, P(ctypes.c_uint32), vp]
    L.zs_device_count.restype = i32
    L.zs_device_count.argtypes = []
    L.zs_partition.restype = i32
    L.zs_partition.argtypes = [P(i64), i32, i32, P(i32)]
    if hasattr(L, "zs_deflate_batch_multi_device"):
        L.zs_deflate_batch_multi_device.restype = i32
        L.zs_deflate_batch_multi_device.argtypes = [P(vp), i32, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32), P(i32), i32, i32, i32]
    L.zs_deflate_batch_multi.restype = i32
    L.zs_deflate_batch_multi.argtypes = [P(vp), i32, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32), i32, i32, i32]
    L.zs_inflate_batch_multi_device.restype = i32
    L.zs_inflate_batch_multi_device.argtypes = [P(vp), i32, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32), P(i32)]
    L.zs_inflate_batch_multi.restype = i32
    L.zs_inflate_batch_multi.argtypes = [P(vp), i32, i32, P(vp), P(i64), P(vp), P(i64), P(i64), P(i32)]
    L.zs_png_filter_device.restype = i32
    L.zs_png_filter_device.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp]
    _lib = L
    return L
