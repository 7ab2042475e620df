// zs_inflate.hip -- stream-parallel inflate (BASELINE config 5).
//
// RFC 1950/1951 decoding is deterministic, so any conformant decoder returns the bytes the
// reference's Inflate.cs / InflateBlocks.cs / InfCodes.cs / InfTree.cs return; what is kept
// from the reference is the error taxonomy (Inflate.cs:134,142,166,243,339;
// InflateBlocks.cs:237,278,394,569; InfCodes.cs:294,349; InfTree.cs:377-427).
//
// One wave per stream.  A deflate stream is bit-serial and every match may reference the
// 32 KiB before it, so a single stream offers little parallelism without speculation;
// this first version runs the symbol decode wave-uniformly (every lane executes the same
// scalar code on the same bits) and uses the 64 lanes for what is parallel: match copies,
// table fills and the coalesced flush of the 64 KiB LDS output ring to HBM.
#include <hip/hip_runtime.h>

#include "zs_device.h"

namespace zs {

enum InfMsg {
    kInfOk = 0, kInfBadMethod, kInfBadWindow, kInfBadHeaderCheck, kInfNeedDict, kInfBadBlockType, kInfBadStoredLen,
    kInfTooManySyms, kInfBadRepeat, kInfOverBl, kInfIncompleteBl, kInfOverLit, kInfIncompleteLit, kInfOverDist, kInfIncompleteDist,
    kInfEmptyDist, kInfBadLitCode, kInfBadDistCode, kInfTruncated, kInfOutputFull, kInfBadCheck, kInfMsgCount
};

struct InfDesc {
    const uint8_t *in;
    uint8_t *out;
    int64_t in_len, out_cap;
    // A stream decoded piece by piece (zs_inflate, zs_stream_api.inc: Inflate.Decompress hands out what has arrived,
    // Inflate.cs:103-357): partial != 0 -- decode the blocks that are complete in [start_bit, 8 in_len) and stop in front of the
    // first that is not (no error: InfState::good_bits / out_len say where); partial == 2: no zlib header, a block
    // begins at that bit (partial == 2: a piece behind the stream's first, also when that bit is 0); hist: the last
    // hist_have <= 32 768 bytes the stream produced before this piece.
    const uint8_t *hist;
    int64_t start_bit;
    int32_t hist_have, partial;
};
struct InfState {
    int64_t out_len, in_used;
    uint32_t adler_stored;
    int32_t status;  // CompressionState
    int32_t msg;     // InfMsg
    int32_t pad_;
    int64_t good_bits;  // partial: the bit behind the last complete block (behind the trailer when the stream ended)
};

constexpr int kInfRing = 65536;
constexpr int kInfLitBits = 10, kInfDistBits = 9;
constexpr uint16_t kInfEsc = 0xFFFF;

struct InfTables {
    uint16_t lit[1 << kInfLitBits];    // sym << 4 | len, or kInfEsc
    uint16_t dist[1 << kInfDistBits];
    uint16_t lcount[16], dcount[16];   // canonical description for codes longer than the primary index
    uint16_t lsym[288], dsym[32];
};

constexpr int kInfInBuf = 512;   // LDS window over the compressed bytes [ibase, ibase + kInfInBuf): small, so that 32 decoder waves fit a CU
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;  // (an attribute on the cast's type itself is ignored: the load would be assumed aligned)
struct InfBits {
    const uint8_t *in;
    int64_t n, pos;  // next byte to load
    uint64_t buf;
    int cnt;
    bool bad;  // a read ran past the end of the input
    uint8_t *ibuf;
    int64_t ibase;
    // every lane calls this (wave-uniform): slide the LDS window so that [pos, pos + 64) is resident
    __device__ void stage() {
        if (pos - ibase < kInfInBuf - 64 && ibase >= 0) return;
        const int lane = threadIdx.x & 63;
        ibase = pos & ~(int64_t)15;
        for (int o = lane * 16; o < kInfInBuf; o += 64 * 16) {
            int64_t a = ibase + o;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a + 16 <= n && ((uintptr_t)in & 15) == 0) {
                v = *(const uint4 *)(in + a);
            } else {
                uint32_t t[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; k++)
                    if (a + k < n) t[k >> 2] |= (uint32_t)in[a + k] << (8 * (k & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            *(uint4 *)(ibuf + o) = v;
        }
        __syncthreads();
    }
    __device__ void fill() {  // top the 64-bit buffer up to >= 56 bits while input lasts
        stage();
        if (cnt < 56 && pos + 8 <= n) {
            // one unaligned 8-byte LDS read; bits above the whole bytes taken are re-ORed identically next time
            buf |= *(const u64_unaligned *)(ibuf + (pos - ibase)) << cnt;
            const int adv = (63 - cnt) >> 3;
            pos += adv;
            cnt += adv * 8;
        } else {
            while (cnt <= 56 && pos < n) {
                buf |= (uint64_t)ibuf[pos - ibase] << cnt;
                pos++;
                cnt += 8;
            }
        }
    }
    __device__ uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    __device__ void drop(int k) {
        if (k > cnt) bad = true, k = cnt;
        buf >>= k;
        cnt -= k;
    }
    __device__ uint32_t take(int k) {
        uint32_t v = peek(k);
        drop(k);
        return v;
    }
};

// canonical tables from code lengths; returns 0 complete, > 0 incomplete, < 0 oversubscribed.
// One wave, its lanes on the symbols and then on the table entries (round 5; until then every lane ran the reference's serial
// loops over all symbols -- InfTree.cs:125-365 -- with the same values: ~0.1 ms per block header, more than the decode of the
// block's symbols by 64 lanes took):
//   counts and ranks   lane l has symbols l, l + 64, ...; a ballot per code length gives the symbols of that length in the
//                      chunk and, below the lane, the symbol's rank among them -- its place in symtab (by length, then symbol);
//   primary table      lane l has entries l, l + 64, ...: an entry's index is the first pbits bits of a code; canonical codes
//                      ascend with their length, so its length is the number of lengths whose codes end at or below it and
//                      its symbol follows from where that length's codes begin (left-aligned 15-bit values).
__device__ int inf_build(const uint8_t *lens, int n, uint16_t *primary, int pbits, uint16_t *count, uint16_t *symtab) {
    const int lane = threadIdx.x & 63;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    int cnt[16];
#pragma unroll
    for (int L = 0; L < 16; L++) cnt[L] = 0;
    int myl[5], myr[5];  // n <= 320: at most five symbols per lane
#pragma unroll
    for (int c = 0; c < 5; c++) {
        myl[c] = 16, myr[c] = 0;
        if (c * 64 < n) {  // (uniform)
            const int i = c * 64 + lane;
            const int l = i < n ? (int)lens[i] : 16;
            int r = 0;
#pragma unroll
            for (int L = 0; L < 16; L++) {
                const uint64_t m = __ballot(l == L);
                r = l == L ? cnt[L] + (int)__builtin_popcountll(m & below) : r;
                cnt[L] += (int)__builtin_popcountll(m);
            }
            myl[c] = l, myr[c] = r;
        }
    }
#pragma unroll
    for (int L = 0; L < 16; L++)
        if (lane == L) count[L] = (uint16_t)cnt[L];
    int left = 1;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= cnt[len];
        if (left < 0) return left;
    }
    int offs[16];  // symbols shorter than len
    uint32_t end[16];  // left-aligned 15-bit value behind the last code of length len
    offs[0] = 0, offs[1] = 0, end[0] = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        if (len < 15) offs[len + 1] = offs[len] + cnt[len];
        end[len] = end[len - 1] + ((uint32_t)cnt[len] << (15 - len));
    }
#pragma unroll
    for (int c = 0; c < 5; c++) {
        int o = 0;
#pragma unroll
        for (int L = 1; L <= 15; L++) o = myl[c] == L ? offs[L] : o;
        if (myl[c] >= 1 && myl[c] <= 15) symtab[o + myr[c]] = (uint16_t)(c * 64 + lane);
    }
    __syncthreads();
    uint32_t endp = end[10];  // pbits is 7 (bit-length code), 9 (distances) or 10 (literals / lengths)
    endp = pbits == 7 ? end[7] : pbits == 9 ? end[9] : endp;
    for (int x = lane; x < (1 << pbits); x += 64) {
        const uint32_t rv = (__brev((uint32_t)x) >> (32 - pbits)) << (15 - pbits);  // the entry's bits, first bit on top
        uint16_t e = kInfEsc;  // a longer code, or none
        if (rv < endp) {
            int len = 1;
#pragma unroll
            for (int L = 1; L <= 9; L++) len += (L < pbits && rv >= end[L]) ? 1 : 0;
            uint32_t lo = 0;
            int id = 0;
#pragma unroll
            for (int L = 1; L <= 9; L++) {
                lo = len == L + 1 ? end[L] : lo;
                id = len == L + 1 ? offs[L + 1] : id;
            }
            e = (uint16_t)((symtab[id + (int)((rv - lo) >> (15 - len))] << 4) | len);
        }
        primary[x] = e;
    }
    __syncthreads();
    return left;
}

// symbol for codes longer than the primary index (puff-style canonical walk, MSB-first code growth)
__device__ int inf_slow(InfBits &b, const uint16_t *count, const uint16_t *symtab, int &len_out) {
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1);
        bits >>= 1;
        int c = count[len];
        if (code - c < first) {
            len_out = len;
            return symtab[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    len_out = 0;
    return -1;
}

__global__ __launch_bounds__(64) void zs_inflate_kernel(const InfDesc *descs, InfState *states) {
    // ~70 KiB of LDS: dynamic allocation, carved by hand
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *ring = smem;
    InfTables &T = *(InfTables *)(smem + kInfRing);
    uint8_t *lens = smem + kInfRing + sizeof(InfTables);
    uint8_t *ll = lens + 320;
    uint8_t *ibuf = ll + 320 + 64;  // 16-byte aligned (kInfRing and sizeof(InfTables) are multiples of 16)
    const int lane = threadIdx.x;
    const InfDesc d = descs[blockIdx.x];
    InfState &st = states[blockIdx.x];
    InfBits b{d.in, d.in_len, 0, 0, 0, false, ibuf, -1};
    // a piece of a stream: the history lies in the ring below the piece's first byte, which is ring (and output) position
    // pos0 = 32 768 whatever there is of it (the flushes below go by 4-byte steps from there)
    const int64_t pos0 = d.partial ? kWSize : 0, valid_from = pos0 - d.hist_have;
    uint8_t *const outp = d.out - pos0;
    int64_t pos = pos0, flushed = pos0;
    int64_t good_bits = d.start_bit, good_out = pos0;
    if (d.partial && d.hist_have > 0) {
        for (int i = lane; i < d.hist_have; i += 64) ring[(valid_from + i) & (kInfRing - 1)] = d.hist[i];
        __syncthreads();
    }
    enum { ZS_OK_ = 0, ZS_END_ = 1, ZS_NEED_DICT_ = 2, ZS_DATA_ = -3, ZS_BUF_ = -5 };
    int status = ZS_OK_, msg = kInfOk;
#define INF_FAIL(code, m) \
    do {                  \
        status = (code);  \
        msg = (m);        \
        goto done;        \
    } while (0)
    auto flush_to = [&](int64_t upto) {  // ring -> HBM, coalesced
        for (int64_t o = flushed + lane * 4; o + 3 < upto; o += 256)
            *(uint32_t *)(outp + o) = *(const uint32_t *)(ring + (o & (kInfRing - 1)));  // flushed and ring offsets are multiples of 4 apart
        int64_t tail = flushed + ((upto - flushed) & ~3LL);
        if (lane < (int)(upto - tail)) outp[tail + lane] = ring[(tail + lane) & (kInfRing - 1)];
        __syncthreads();
        flushed = upto;
    };
    // zlib header (Inflate.cs:120-170, 243)
    if (d.partial == 2) {  // a piece behind the stream's first: no header, a block begins at start_bit
        b.pos = d.start_bit >> 3;
        b.fill();
        b.drop((int)(d.start_bit & 7));
    } else {
    b.fill();
    if (d.in_len < 2) INF_FAIL(ZS_BUF_, kInfTruncated);
    {
        unsigned cmf = b.take(8), flg = b.take(8);
        if ((cmf & 0x0F) != 8) INF_FAIL(ZS_DATA_, kInfBadMethod);
        if ((cmf >> 4) + 8 > 15) INF_FAIL(ZS_DATA_, kInfBadWindow);
        if (((cmf << 8) + flg) % 31 != 0) INF_FAIL(ZS_DATA_, kInfBadHeaderCheck);
        if (flg & 0x20) INF_FAIL(ZS_NEED_DICT_, kInfNeedDict);
    }
    }
    for (int last = 0; !last;) {
        good_bits = 8 * b.pos - b.cnt, good_out = pos;  // a block begins here: everything in front of it is complete
        b.fill();
        if (b.cnt < 3) INF_FAIL(ZS_BUF_, kInfTruncated);
        last = (int)b.take(1);
        const unsigned type = b.take(2);
        if (type == 0) {  // stored (InflateBlocks.cs:244-330)
            b.drop(b.cnt & 7);
            b.fill();
            if (b.cnt < 32) INF_FAIL(ZS_BUF_, kInfTruncated);
            unsigned len = b.take(16), nlen = b.take(16);
            if (len != (~nlen & 0xFFFF)) INF_FAIL(ZS_DATA_, kInfBadStoredLen);
            // give the bytes still in the bit buffer back to the byte cursor
            int64_t src = b.pos - (b.cnt >> 3);
            if (src + len > d.in_len) INF_FAIL(ZS_BUF_, kInfTruncated);
            if (pos - pos0 + len > d.out_cap) INF_FAIL(ZS_BUF_, kInfOutputFull);
            for (unsigned done = 0; done < len;) {
                unsigned room = (unsigned)(flushed + kInfRing / 2 + kInfRing / 4 - pos);
                unsigned n = len - done < room ? len - done : room;
                for (unsigned i = lane; i < n; i += 64) ring[(pos + i) & (kInfRing - 1)] = d.in[src + done + i];
                __syncthreads();
                pos += n;
                done += n;
                if (pos - flushed >= kInfRing / 2) flush_to(flushed + kInfRing / 2);
            }
            b.pos = src + len;
            b.buf = 0;
            b.cnt = 0;
            b.ibase = -1;
            continue;
        }
        if (type == 3) INF_FAIL(ZS_DATA_, kInfBadBlockType);
        if (type == 1) {  // fixed codes
            for (int i = lane; i < 288; i += 64) lens[i] = (uint8_t)static_llen(i);
            __syncthreads();
            inf_build(lens, 288, T.lit, kInfLitBits, T.lcount, T.lsym);
            for (int i = lane; i < 32; i += 64) lens[i] = 5;
            __syncthreads();
            inf_build(lens, 30, T.dist, kInfDistBits, T.dcount, T.dsym);
        } else {  // dynamic codes (InflateBlocks.cs:380-610)
            b.fill();
            if (b.cnt < 14) INF_FAIL(ZS_BUF_, kInfTruncated);
            const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
            if (nlen > 286 || ndist > 30) INF_FAIL(ZS_DATA_, kInfTooManySyms);
            for (int i = lane; i < 320; i += 64) lens[i] = 0;
            __syncthreads();
            for (int i = 0; i < ncode; i++) {
                b.fill();
                if (b.cnt < 3) INF_FAIL(ZS_BUF_, kInfTruncated);
                unsigned v = b.take(3);
                if (lane == 0) lens[bl_order(i)] = (uint8_t)v;
            }
            __syncthreads();
            int r = inf_build(lens, 19, T.lit, 7, T.lcount, T.lsym);  // bit-length code: <= 7 bits, fits the primary table
            if (r < 0) INF_FAIL(ZS_DATA_, kInfOverBl);
            // an incomplete code is accepted only when it is a single code of length 1 (Huft_build: `y != 0 && g != 1`,
            // InfTree.cs:364; Inflate_trees_bits :378-382)
            if (r > 0 && !(T.lcount[1] == 1 && 19 - T.lcount[0] == 1)) INF_FAIL(ZS_DATA_, kInfIncompleteBl);
            __syncthreads();
            uint8_t prev = 0;
            int idx = 0;
            while (idx < nlen + ndist) {
                if (b.bad) INF_FAIL(ZS_BUF_, kInfTruncated);
                b.fill();
                uint16_t e = T.lit[b.peek(7)];
                if (e == kInfEsc || (int)(e & 15) > b.cnt) INF_FAIL(b.cnt < 7 && b.pos >= b.n ? ZS_BUF_ : ZS_DATA_, kInfBadRepeat);
                b.drop(e & 15);
                int sym = e >> 4;
                if (sym < 16) {
                    if (lane == 0) ll[idx] = (uint8_t)sym;
                    prev = (uint8_t)sym;
                    idx++;
                } else {
                    int rep;
                    uint8_t val = 0;
                    if (sym == 16) {
                        if (idx == 0) INF_FAIL(ZS_DATA_, kInfBadRepeat);
                        val = prev;
                        rep = 3 + (int)b.take(2);
                    } else if (sym == 17) {
                        rep = 3 + (int)b.take(3);
                    } else {
                        rep = 11 + (int)b.take(7);
                    }
                    if (idx + rep > nlen + ndist) INF_FAIL(ZS_DATA_, kInfBadRepeat);
                    if (lane < rep) ll[idx + lane] = val;
                    if (lane + 64 < rep) ll[idx + lane + 64] = val;
                    if (lane + 128 < rep) ll[idx + lane + 128] = val;
                    prev = val;
                    idx += rep;
                }
            }
            __syncthreads();
            for (int i = lane; i < 320; i += 64) lens[i] = i < nlen + ndist ? ll[i] : 0;
            __syncthreads();
            r = inf_build(lens, nlen, T.lit, kInfLitBits, T.lcount, T.lsym);
            if (r < 0) INF_FAIL(ZS_DATA_, kInfOverLit);
            if (r > 0 && !(T.lcount[1] == 1 && nlen - T.lcount[0] == 1)) INF_FAIL(ZS_DATA_, kInfIncompleteLit);  // InfTree.cs:364,397-410
            r = inf_build(lens + nlen, ndist, T.dist, kInfDistBits, T.dcount, T.dsym);
            if (r < 0) INF_FAIL(ZS_DATA_, kInfOverDist);
            if (r > 0) {  // InfTree.cs:364,413-431
                if (ndist - T.dcount[0] == 0) {
                    if (nlen > 257) INF_FAIL(ZS_DATA_, kInfEmptyDist);
                } else if (!(T.dcount[1] == 1 && ndist - T.dcount[0] == 1)) {
                    INF_FAIL(ZS_DATA_, kInfIncompleteDist);
                }
            }
        }
        __syncthreads();
        // ---- symbols (InfCodes.cs:106-386) ----
        for (;;) {
            if (b.bad) INF_FAIL(ZS_BUF_, kInfTruncated);
            b.fill();
            int sym, clen;
            {
                uint16_t e = T.lit[b.peek(kInfLitBits)];
                if (e != kInfEsc) sym = e >> 4, clen = e & 15;
                else sym = inf_slow(b, T.lcount, T.lsym, clen);
            }
            if (sym < 0 || clen > b.cnt) {
                // the bits ran out inside a code (the window is zero-padded past the end): not corrupt data, a truncated stream
                const bool trunc = b.pos >= b.n && b.cnt < 15;
                INF_FAIL(trunc ? ZS_BUF_ : ZS_DATA_, trunc || sym >= 0 ? kInfTruncated : kInfBadLitCode);
            }
            b.drop(clen);
            if (sym < 256) {
                if (pos - pos0 >= d.out_cap) INF_FAIL(ZS_BUF_, kInfOutputFull);
                if (lane == 0) ring[pos & (kInfRing - 1)] = (uint8_t)sym;
                pos++;
            } else if (sym == 256) {
                break;
            } else {
                sym -= 257;
                if (sym >= 29) INF_FAIL(ZS_DATA_, kInfBadLitCode);
                const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)b.take(extra_lbits(sym));
                b.fill();
                int ds, dl;
                {
                    uint16_t e = T.dist[b.peek(kInfDistBits)];
                    if (e != kInfEsc) ds = e >> 4, dl = e & 15;
                    else ds = inf_slow(b, T.dcount, T.dsym, dl);
                }
                if (ds < 0 || ds >= 30 || dl > b.cnt) {
                    const bool trunc = b.pos >= b.n && b.cnt < 15;
                    INF_FAIL(trunc ? ZS_BUF_ : ZS_DATA_, trunc ? kInfTruncated : kInfBadDistCode);
                }
                b.drop(dl);
                const int dist = base_dist(ds) + 1 + (int)b.take(extra_dbits(ds));
                if (b.bad) INF_FAIL(ZS_BUF_, kInfTruncated);  // extra bits past the end of the input
                if (dist > pos - valid_from || dist > kWSize) INF_FAIL(ZS_DATA_, kInfBadDistCode);
                if (pos - pos0 + mlen > d.out_cap) INF_FAIL(ZS_BUF_, kInfOutputFull);
                // one wave: DS operations execute in program order, so lane 0's literal stores are
                // visible to every lane here without a barrier
                for (int i = lane; i < mlen; i += 64) {
                    int srcoff = dist >= mlen ? i : i % dist;
                    ring[(pos + i) & (kInfRing - 1)] = ring[(pos - dist + srcoff) & (kInfRing - 1)];
                }
                pos += mlen;
            }
            if (pos - flushed >= kInfRing / 2 + 1024) flush_to(flushed + kInfRing / 2);
        }
    }
    // Adler-32 trailer (Inflate.cs:300-345): compared on the host against the device-computed checksum
    b.drop(b.cnt & 7);
    b.fill();
    if (b.cnt < 32) INF_FAIL(ZS_BUF_, kInfTruncated);
    {
        uint32_t a = b.take(8);
        a = (a << 8) | b.take(8);
        a = (a << 8) | b.take(8);
        a = (a << 8) | b.take(8);
        if (lane == 0) st.adler_stored = a;
    }
    status = b.bad ? (int)ZS_BUF_ : (int)ZS_END_;
    if (b.bad) msg = kInfTruncated;
    else good_bits = 8 * b.pos - b.cnt, good_out = pos;
done:
    __syncthreads();
    // a piece whose input ends inside a block (or inside the trailer): not an error, the piece ends in front of that block
    // a piece stops in front of the first block that is not complete -- or does not fit the output any more, as long as a
    // block before it did (the caller takes the piece and comes again: a stream of any length in bounded memory)
    if (d.partial && status == ZS_BUF_ && (msg != kInfOutputFull || good_out > pos0)) status = ZS_OK_, msg = kInfOk;
    if (pos > flushed) {
        if (pos - flushed > kInfRing / 2) flush_to(flushed + kInfRing / 2);
        flush_to(pos);
    }
    if (lane == 0) {
        st.out_len = d.partial ? good_out - pos0 : pos;
        st.good_bits = good_bits;
        st.in_used = b.pos - (b.cnt >> 3);
        st.status = status;
        st.msg = msg;
    }
#undef INF_FAIL
}

constexpr int kInfLds = kInfRing + (int)sizeof(InfTables) + 320 + 320 + 64 + kInfInBuf + 64;
static_assert(sizeof(InfTables) % 16 == 0, "LDS carve alignment");

}  // namespace zs
