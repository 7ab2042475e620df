"""zlibstream_amd -- MI355X (gfx950) deflate engine behind the ZlibStream API.

Host-side mirror of the reference's public surface (ZlibOutputStream,
ZlibOptions, CompressionLevel, CompressionStrategy, FlushMode,
ZlibStreamException) over the C ABI in include/zsgpu.h.  The compression path
has no CPU fallback: it needs libzsgpu.so and an MI355X.
"""
from .api import (CompressionLevel, CompressionState, CompressionStrategy, Engine, FlushMode, ZlibInputStream, ZlibOptions,  # noqa: F401
                  ZlibOutputStream, ZlibStreamException, compress, deflate_batch_multi, deflate_batch_multi_device, deflate_bound,
                  device_count, inflate_batch_multi, inflate_batch_multi_device, png_filter_device)
