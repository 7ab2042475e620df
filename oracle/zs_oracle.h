/*
 * zs_oracle.h -- CPU restatement of SixLabors/ZlibStream's deflate/inflate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under zlibstream_amd/ (the product) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / reported CPU baseline.
 *
 * The reference is managed C# and cannot be built in this image (no dotnet /
 * mono), so this is a literal restatement in C of the algorithm in
 *   src/ZlibStream/Deflate.cs, Deflate.Slow.cs, Deflate.Fast.cs,
 *   Deflate.Stored.cs, Deflate.Rle.cs, Deflate.Intrinsics.cs (semantics),
 *   Deflate.Buffers.cs (sizes / overlay), Trees.cs, Trees.Static.cs,
 *   Adler32.cs (scalar), ZlibStream.cs (ReadBuffer), ZlibOutputStream.cs
 *   (caller protocol),
 * with fresh zero-initialised work buffers (the reference rents them from
 * ArrayPool without clearing, Deflate.Buffers.cs:115-135).
 *
 * Parity pin: the 36 compressed sizes published in the reference's
 * benchmarks.md (tests/test_oracle_kat.py) -- see oracle/README.md.
 */
#ifndef ZS_ORACLE_H
#define ZS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CompressionState (CompressionState.cs) */
enum {
    ZSO_VERSION_ERROR = -6,
    ZSO_BUF_ERROR = -5,
    ZSO_MEM_ERROR = -4,
    ZSO_DATA_ERROR = -3,
    ZSO_STREAM_ERROR = -2,
    ZSO_ERRNO = -1,
    ZSO_OK = 0,
    ZSO_STREAM_END = 1,
    ZSO_NEED_DICT = 2
};

/* FlushMode (FlushMode.cs) */
enum { ZSO_NO_FLUSH = 0, ZSO_PARTIAL_FLUSH = 1, ZSO_SYNC_FLUSH = 2, ZSO_FULL_FLUSH = 3, ZSO_FINISH = 4 };

/* CompressionStrategy (CompressionStrategy.cs) */
enum { ZSO_DEFAULT_STRATEGY = 0, ZSO_FILTERED = 1, ZSO_HUFFMAN_ONLY = 2, ZSO_RLE = 3, ZSO_FIXED = 4 };

/* Hash variant (Deflate.Intrinsics.cs:295-307): the published numbers are
 * produced by the SSE4.2 CRC32C path. */
enum { ZSO_HASH_CRC32C = 0, ZSO_HASH_MUL = 1 };

typedef struct zso_deflate zso_deflate;

/* Optional instrumentation: never changes the produced bytes. */
typedef struct zso_trace {
    /* every tallied symbol: dist==0 -> literal lc, else match (dist, lc+3) */
    void (*on_symbol)(void *user, int dist, int lc, int64_t abs_pos);
    /* every Fill_window read: loop-top absolute strStart, bytes read, whether
     * the strStart+1 pre-insert ran (Deflate.cs:1010-1013) */
    void (*on_read)(void *user, int64_t abs_strstart, int nread, int preinsert, int64_t abs_base);
    /* every window slide (Deflate.cs:980-989) */
    void (*on_slide)(void *user, int64_t abs_strstart, int64_t new_abs_base);
    /* every block: type 0 stored / 1 static / 2 dynamic */
    void (*on_block)(void *user, int type, int nsyms, int64_t abs_start, int stored_len, int eof,
                     int64_t out_bit_start);
    /* every Longest_match call (result is the returned length) */
    void (*on_match)(void *user, int64_t abs_strstart, int prev_length, int result, int64_t abs_match_start);
    void *user;
} zso_trace;

/* Deflate..ctor (Deflate.cs:228-310).  level -1 => 6.  window_bits < 0 => raw
 * deflate (no zlib header/trailer).  Returns NULL on argument errors (the
 * reference throws ArgumentOutOfRangeException). */
zso_deflate *zso_deflate_new(int level, int strategy, int window_bits, int mem_level, int hash_variant);
void zso_deflate_free(zso_deflate *s);
void zso_deflate_set_trace(zso_deflate *s, const zso_trace *t);

/* Deflate.Compress (Deflate.cs:436-636) with the z_stream cursor passed
 * explicitly: consumes from next_in/avail_in, produces into
 * next_out/avail_out (both updated). */
int zso_deflate_call(zso_deflate *s, const uint8_t **next_in, int *avail_in, uint8_t **next_out, int *avail_out,
                     int flush);
const char *zso_deflate_message(const zso_deflate *s);
uint32_t zso_deflate_adler(const zso_deflate *s);
int64_t zso_deflate_total_in(const zso_deflate *s);
int64_t zso_deflate_total_out(const zso_deflate *s);

/* The ZlibOutputStream protocol (ZlibOutputStream.cs:125-168, 213-256):
 * one Write per chunk (chunk_lens[i] bytes, empty chunks are skipped like
 * WriteCore does), 512-byte output buffer, then Finish.  n_chunks == 0 or
 * chunk_lens == NULL means a single Write of the whole buffer.
 * Returns the number of bytes produced, or (size_t)-1 on error / overflow. */
size_t zso_compress_stream(const uint8_t *in, size_t n, const size_t *chunk_lens, size_t n_chunks, int level,
                           int strategy, int flush_mode, int hash_variant, uint8_t *out, size_t out_cap,
                           const zso_trace *trace);

/* ... with ZlibOptions.FlushMode set anew before every Write: flush_modes[i] (NULL: flush_mode_all for every Write) */
size_t zso_compress_stream_modes(const uint8_t *in, size_t n, const size_t *chunk_lens, size_t n_chunks, int level,
                                 int strategy, int flush_mode_all, const int *flush_modes, int hash_variant, uint8_t *out, size_t out_cap,
                                 const zso_trace *trace);

size_t zso_compress_bound(size_t n);

/* Adler32.Calculate (Adler32.cs:61-78, scalar :270-326) */
uint32_t zso_adler32(uint32_t adler, const uint8_t *buf, size_t len);

/* UpdateHash (Deflate.Intrinsics.cs:295-307) on one little-endian u32 */
uint32_t zso_hash_u32(uint32_t v, int hash_variant);

/* Inflate (Inflate.cs, InflateBlocks.cs, InfCodes.cs, InfTree.cs): any
 * RFC 1950/1951 conformant decode produces the same bytes; this one keeps the
 * reference's error classes and messages for the conditions listed in
 * oracle/README.md.  Returns ZSO_STREAM_END on success and sets *out_len and
 * *in_used; on failure returns the error code and *msg points to a static
 * string. */
int zso_inflate_oneshot(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
                        size_t *in_used, const char **msg);

/* .NET System.Random(seed).NextBytes, used by the reference tests
 * (ZlibStreamTests.Roundtrip.cs:169-175) to build their input buffer. */
void zso_dotnet_random_bytes(int seed, uint8_t *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif
