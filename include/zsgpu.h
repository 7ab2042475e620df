/*
 * zsgpu.h -- C ABI of the MI355X (gfx950) deflate engine for SixLabors/ZlibStream.
 *
 * The reference has no native boundary today (it is 100 % managed C#).  The
 * seam this library replaces is the zlib-style pair inside the reference's
 * internal z_stream facade:
 *
 *     CompressionState ZLibStream.Deflate(FlushMode)   src/ZlibStream/ZlibStream.cs:164-167
 *       -> Deflate.Compress(ZLibStream, FlushMode)     src/ZlibStream/Deflate.cs:436-636
 *     CompressionState ZLibStream.Inflate(FlushMode)   src/ZlibStream/ZlibStream.cs:119-122
 *       -> Inflate.Decompress(ZLibStream, FlushMode)   src/ZlibStream/Inflate.cs:103-357
 *
 * i.e. ZlibOutputStream / ZlibInputStream keep their public surface and their
 * WriteCore / Finish / ReadCore loops (ZlibOutputStream.cs:125-168, 213-256,
 * ZlibInputStream.cs:133-186); the engine object behind them is this library.
 * INTEGRATION.md shows the P/Invoke stub.
 *
 * Conventions: plain C, no C++ or torch types; return values are the
 * reference's CompressionState codes (CompressionState.cs); all pointers are
 * used only for the duration of the call unless stated otherwise (the
 * reference pins caller spans only inside WriteCore/ReadCore).
 *
 * There is NO CPU fallback: every entry point that compresses fails with
 * ZS_STREAM_ERROR and a message when no gfx950 device is usable.
 */
#ifndef ZSGPU_H
#define ZSGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define ZS_API __declspec(dllexport)
#else
#define ZS_API __attribute__((visibility("default")))
#endif

/* CompressionState.cs */
enum {
    ZS_VERSION_ERROR = -6,
    ZS_BUF_ERROR = -5,
    ZS_MEM_ERROR = -4,
    ZS_DATA_ERROR = -3,
    ZS_STREAM_ERROR = -2,
    ZS_ERRNO = -1,
    ZS_OK = 0,
    ZS_STREAM_END = 1,
    ZS_NEED_DICT = 2
};
/* FlushMode.cs */
enum { ZS_NO_FLUSH = 0, ZS_PARTIAL_FLUSH = 1, ZS_SYNC_FLUSH = 2, ZS_FULL_FLUSH = 3, ZS_FINISH = 4 };
/* CompressionStrategy.cs */
enum { ZS_DEFAULT_STRATEGY = 0, ZS_FILTERED = 1, ZS_HUFFMAN_ONLY = 2, ZS_RLE = 3, ZS_FIXED = 4 };
/* Deflate.Intrinsics.cs:295-307: which UpdateHash the managed build would take */
enum { ZS_HASH_CRC32C = 0, ZS_HASH_MUL = 1 };

/* ------------------------------------------------------------------ */
/* Engine context: one per GPU (device ordinal as seen by HIP).  Owns the HIP
 * stream-ordered workspace that is reused across calls.  Not thread-safe;
 * distinct contexts are independent (as distinct reference Deflate instances
 * are, Deflate.Buffers.cs). */
typedef struct zs_ctx zs_ctx;

ZS_API int zs_ctx_create(int device, zs_ctx **out);
ZS_API void zs_ctx_destroy(zs_ctx *ctx);
ZS_API const char *zs_ctx_last_error(const zs_ctx *ctx);

/* Upper bound of the zlib stream produced for n input bytes. */
ZS_API int64_t zs_deflate_bound(int64_t n);

/* ------------------------------------------------------------------ */
/* Throughput entry points: n independent buffers, each compressed exactly as
 *     using (var s = new ZlibOutputStream(dst, level)) s.Write(buf, 0, len);
 * does (one Write of the whole buffer, then Dispose -> Finish;
 * ZlibOutputStream.cs:114-168,186-256; DeflateCorpusBenchmark.cs:86-100).
 *
 * _device: in[i] / out[i] are DEVICE pointers on ctx's GPU (inputs already
 * resident in HBM); `hip_stream` is a hipStream_t (NULL = the context's own
 * stream).  out_len[i] is a HOST array; the call returns after the results
 * are known (it synchronises the stream once).
 * Returns ZS_OK, or the first failing stream's code; per-stream codes are in
 * status[i] when status != NULL (ZS_BUF_ERROR when out_cap[i] is too small). */
ZS_API int zs_deflate_batch_device(zs_ctx *ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                                   const int64_t *out_cap, int64_t *out_len, int *status, int level, int strategy,
                                   int hash_variant, void *hip_stream);

/* One stream written in several NoFlush Writes, resident in HBM:
 *     using (var s = new ZlibOutputStream(dst, level)) foreach (var w in writes) s.Write(w);
 * (ZlibOutputStream.cs:114-168: every Write is a call of Deflate(NoFlush), and every Write end a read event of
 * Fill_window, Deflate.cs:967-1019, which changes the bytes).  write_ends: the n_writes cumulative Write ends (HOST
 * array, increasing, the last one = in_len).  What zs_deflate does at Finish for the Writes it has buffered, without
 * the per-call protocol: the entry an encoder that has its image on the device (zs_png_filter_device) writes rows with. */
ZS_API int zs_deflate_writes_device(zs_ctx *ctx, const void *in, int64_t in_len, const int64_t *write_ends, int64_t n_writes,
                                    void *out, int64_t out_cap, int64_t *out_len, int level, int strategy, int hash_variant,
                                    void *hip_stream);

/* Host-pointer form: copies in over PCIe, runs the device path, copies out. */
ZS_API int zs_deflate_batch(zs_ctx *ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                            const int64_t *out_cap, int64_t *out_len, int *status, int level, int strategy,
                            int hash_variant);

/* Inflate (Inflate.Decompress, Inflate.cs:103-357; InflateBlocks.cs; InfCodes.cs; InfTree.cs): n independent
 * zlib streams, each decoded completely (what `new ZlibInputStream(src).Read(...)` to the end of the stream
 * yields, ZlibInputStream.cs:133-186).  out_cap[i] must hold the whole output.  status[i] is ZS_STREAM_END on
 * success; ZS_DATA_ERROR / ZS_BUF_ERROR / ZS_NEED_DICT with the reference's message in zs_ctx_last_error
 * otherwise (e.g. "incorrect data check", Inflate.cs:339).  Returns ZS_OK when every stream ended cleanly. */
ZS_API int zs_inflate_batch_device(zs_ctx *ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                                   const int64_t *out_cap, int64_t *out_len, int *status, void *hip_stream);
ZS_API int zs_inflate_batch(zs_ctx *ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                            const int64_t *out_cap, int64_t *out_len, int *status);

/* ------------------------------------------------------------------ */
/* Multi-GPU batch entry points (SURVEY.md 8(b) "zs_deflate_batch(..., device_mask)", 8(e)): the n independent
 * buffers are partitioned over n_ctx contexts -- normally one per GPU of the node, zs_ctx_create(0 .. count-1) --
 * by size (longest-processing-time, zs_partition), each context's share runs on its own host thread through the
 * host-pointer batch call above, and results land in input order.  No collective and no peer traffic: a zlib
 * stream cannot be split bit-exactly, so the buffer is the unit (DeflateCorpusBenchmark.cs:86-100 compresses
 * independent buffers the same way, one after the other).  Contexts must be distinct; several may name the same
 * device.  Returns ZS_OK or the first failing context's code; per-buffer codes in status[i]. */
ZS_API int zs_device_count(void);
/* part_of[i] = context index of buffer i; deterministic (ties: earlier buffer first, lower context first). */
ZS_API int zs_partition(const int64_t *sizes, int n, int n_parts, int *part_of);
ZS_API int zs_deflate_batch_multi(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len,
                                  void *const *out, const int64_t *out_cap, int64_t *out_len, int *status, int level,
                                  int strategy, int hash_variant);
/* Device-pointer form: in[i] / out[i] live on the GPU of context part_of[i] (the caller places the buffers, e.g. by
 * zs_partition over the sizes, and keeps them resident): no PCIe traffic, each context's share runs on its own host
 * thread through zs_deflate_batch_device.  out_len / status are HOST arrays. */
ZS_API int zs_deflate_batch_multi_device(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len,
                                         void *const *out, const int64_t *out_cap, int64_t *out_len, int *status,
                                         const int *part_of, int level, int strategy, int hash_variant);
ZS_API int zs_inflate_batch_multi(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len,
                                  void *const *out, const int64_t *out_cap, int64_t *out_len, int *status);
/* ... over device pointers, as zs_deflate_batch_multi_device: stream i and its output live on the GPU of ctxs[part_of[i]]
 * (zs_partition over the decoded sizes gives a balanced part_of). */
ZS_API int zs_inflate_batch_multi_device(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len,
                                         void *const *out, const int64_t *out_cap, int64_t *out_len, int *status, const int *part_of);

/* ------------------------------------------------------------------ */
/* PNG scanline filtering on the device (the caller path of the sparse case -- the reference exists
 * for ImageSharp's PNG encoder, readme.md:16-19, which filters every scanline and writes the rows to ZlibOutputStream).
 * pixels: height rows of row_bytes bytes (device pointer); bpp: bytes per complete pixel, 1..8 (PNG specification 9.2);
 * filter: 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth, 5 adaptive (per row the filter with the smallest sum of absolute
 * values, first one on ties).  out (device pointer): height * (row_bytes + 1) bytes, every row preceded by its filter
 * type -- the IDAT payload before compression, ready for zs_deflate_batch_device.  With hip_stream == NULL the call
 * returns when the rows are written; otherwise it is ordered on that stream. */
ZS_API int zs_png_filter_device(zs_ctx *ctx, const void *pixels, int64_t row_bytes, int64_t height, int bpp, int filter,
                                void *out, void *hip_stream);

/* Stage timing of the last *_batch_device call, measured with hipEvents on
 * the stream the kernels ran on.  Enable before the call. */
/* Counters of a context for tests and measurements (-1: no such counter): "fast_rounds" -- rounds the last call's DeflateFast took
 * over its chunks (0: one workgroup per stream); "fast_fallbacks", "round_runs", "cut_rounds", "lit_fallbacks" -- batches that took
 * one of the slower paths since the context was made; "lit_engine_bytes" -- input bytes the one-wave literal engine parsed
 * beyond the streams' last 261. */
ZS_API int64_t zs_ctx_counter(const zs_ctx *ctx, const char *name);
ZS_API void zs_ctx_set_profiling(zs_ctx *ctx, int enable);
ZS_API int zs_ctx_stage_count(const zs_ctx *ctx);
ZS_API const char *zs_ctx_stage_name(const zs_ctx *ctx, int stage);
ZS_API double zs_ctx_stage_ms(const zs_ctx *ctx, int stage);

/* ------------------------------------------------------------------ */
/* z_stream-shaped streaming interface: what ZLibStream.Deflate(flush) binds to.
 *
 * zs_deflate_init  <- Deflate..ctor (Deflate.cs:228-310): level -1..9,
 *   strategy 0..4, window_bits +-9..15 (negative = no zlib header/trailer),
 *   mem_level 1..9.  Argument errors return NULL (the reference throws
 *   ArgumentOutOfRangeException).
 * zs_deflate       <- Deflate.Compress (Deflate.cs:436-636).  The cursor fields
 *   of ZLibStream (ZlibStream.cs:34-94) are passed explicitly: *avail_in /
 *   *avail_out are decremented, *total_in / *total_out advanced, *adler
 *   updated.  Input is copied during the call.  NoFlush Writes are buffered
 *   and a stream that never flushes is compressed by the bulk pipeline when
 *   ZS_FINISH arrives, after which output is handed out avail_out bytes at a
 *   time exactly like Flush_pending (Deflate.cs:828-854).
 *   ZS_PARTIAL_FLUSH / ZS_SYNC_FLUSH / ZS_FULL_FLUSH (Deflate.cs:583-613) run the
 *   engine at that call and deliver everything up to and including the flush
 *   marker, so the reader can decode what has been written so far; from then on
 *   the stream is incremental (the engine is kept suspended in device memory,
 *   consumed input is dropped); at levels 4-9 the runs behind a flush are the
 *   bulk pipeline's again, started at the flush on the suspended engine's hash
 *   chains, so NoFlush Writes behind a flush wait for the next flush or Finish
 *   like those in front of the first.  A NoFlush stream with more than 1 GiB buffered
 *   becomes incremental too: a stream has no length limit (a single call takes
 *   up to 2 GiB - 1 KiB; 2 GiB - 65 KiB on a stream that has become incremental, whose run keeps 64 KiB of history).  The bytes are the reference's for a caller that runs
 *   ZlibOutputStream.WriteCore's loop (ZlibOutputStream.cs:125-168: a fresh
 *   output chunk of the same size for every call -- the size is taken from the
 *   first call): block end + Tr_align / empty stored block after every flushed
 *   Write, FullFlush forgetting the hash heads, and the extra empty blocks of
 *   flushes that fill the chunk exactly.
 * zs_deflate_end   <- Deflate.Dispose.
 * zs_last_message  <- ZLibStream.Message. */
typedef struct zs_deflate_stream zs_deflate_stream;

ZS_API zs_deflate_stream *zs_deflate_init(zs_ctx *ctx, int level, int strategy, int window_bits, int mem_level,
                                          int hash_variant);
ZS_API int zs_deflate(zs_deflate_stream *s, const uint8_t *next_in, int32_t *avail_in, uint8_t *next_out,
                      int32_t *avail_out, int flush, uint32_t *adler, int64_t *total_in, int64_t *total_out);
ZS_API void zs_deflate_end(zs_deflate_stream *s);
ZS_API const char *zs_last_message(const zs_deflate_stream *s);

/* zs_inflate_init <- Inflate..ctor (Inflate.cs:76-96); only window_bits 15 (zlib-wrapped) runs on the device: NULL otherwise.
 * zs_inflate      <- Inflate.Decompress (Inflate.cs:103-357) as ZLibStream.Inflate(FlushMode) calls it
 *   (ZlibStream.cs:119-122), driven by ZlibInputStream.ReadCore (ZlibInputStream.cs:133-186).  Same cursor convention as
 *   zs_deflate.  Calls that bring input return ZS_OK after taking it.  A whole stream is decoded by the block-parallel
 *   decoder at the call whose input completes it -- the end (final block + Adler-32 trailer, Inflate.cs:292-357) is looked
 *   for in what has been buffered each time the buffered bytes have quadrupled, from 1 MiB on.  A call with *avail_in == 0
 *   (the reader has run out of input for now: ZlibInputStream.ReadCore behind a writer's flush) decodes what the bytes so
 *   far hold in complete blocks -- a flush ends on a block boundary (Deflate.cs:583-613), so everything the writer flushed
 *   is delivered -- and the stream goes on piece by piece from there with the next input; so does a stream that is fed 64
 *   MiB without a call for output (bounded host memory for a stream of any length).  ZS_STREAM_END comes with the last
 *   byte.  total_in is the stream's length with its trailer; bytes behind the trailer that the call which met the end
 *   brought are left to the caller (*avail_in), as the managed engine leaves them; bytes behind it from earlier calls
 *   (the end is only looked for now and then) are kept: zs_inflate_surplus.  A call without input that has nothing new to
 *   give is ZS_BUF_ERROR (Inflate.Decompress's "no progress"); corrupt data gives ZS_DATA_ERROR with the reference's
 *   message (zs_inflate_message).
 *   Malformed streams: incomplete code sets are accepted exactly where Huft_build accepts them (a single code of length
 *   1, InfTree.cs:364); one deliberate difference -- a match distance that reaches before the first output byte is
 *   ZS_DATA_ERROR "invalid distance code" here, the managed engine copies from its zeroed window instead
 *   (InfCodes.cs:241,659). */
typedef struct zs_inflate_stream zs_inflate_stream;
ZS_API zs_inflate_stream *zs_inflate_init(zs_ctx *ctx, int window_bits);
ZS_API int zs_inflate(zs_inflate_stream *s, const uint8_t *next_in, int32_t *avail_in, uint8_t *next_out, int32_t *avail_out,
                      int flush, uint32_t *adler, int64_t *total_in, int64_t *total_out);
ZS_API void zs_inflate_end(zs_inflate_stream *s);
ZS_API const char *zs_inflate_message(const zs_inflate_stream *s);
/* Bytes fed behind the stream's trailer by calls before the one that met the stream's end (no counterpart in the reference,
 * whose engine stops at the trailer byte for byte): count, *p -> the bytes (owned by the stream object). */
ZS_API int64_t zs_inflate_surplus(const zs_inflate_stream *s, const uint8_t **p);

/* Adler32.Calculate (Adler32.cs:61-78) on the GPU, for a device-resident
 * buffer; result returned to the host. */
ZS_API int zs_adler32_device(zs_ctx *ctx, const void *d_buf, int64_t len, uint32_t seed, uint32_t *out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
