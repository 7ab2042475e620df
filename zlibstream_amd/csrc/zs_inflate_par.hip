// zs_inflate_par.hip -- block-parallel inflate (BASELINE config 5).
//
// A deflate stream is bit-serial and every block may reference the 32 KiB before it, so
// one stream decoded front to back keeps one wave busy (zs_inflate.hip).  This path finds
// parallelism inside a stream the way pugz / rapidgzip do on CPUs:
//
//   F  find     every bit offset at which a *dynamic-Huffman block header* validates
//               completely (BTYPE, HLIT/HDIST ranges, a complete bit-length code, complete
//               literal/length and distance codes with an end-of-block code) is a candidate
//               block start; a random position passes with negligible probability.
//   D1 measure  one wave per candidate decodes its block without output: end bit offset,
//               output size, BFINAL, validity -- its 64 lanes on 64 subsequences of the block's
//               bits (Huffman streams self-synchronise; exits are chained to a proven prefix),
//               which also leaves a checkpoint (bit, output position) per subsequence.
//   C  chain    one wave per stream walks from the real first block (bit 16) through the
//               candidates (end of block i = start of block i+1); blocks the finder cannot
//               see (stored / fixed) are measured on the spot.  Output offsets follow.
//   D2 decode   one wave per block, one lane per subsequence, decodes again into 16-bit cells:
//               a literal byte, a marker 0x8000 | i for "byte i of the 32 KiB window before this
//               block", or a marker for "the cell d positions before this lane's subsequence"
//               (copies of markers stay markers), so subsequences and blocks decode
//               independently; a flatten pass removes the second kind.
//   W  windows  per stream, block by block: the resolved last 32 KiB after each block.
//   R  resolve  every cell of every block -> byte, in parallel.
//
// Any irregularity (no chain, overflow of a fixed-size table, an undecodable block) makes the
// host fall back to the sequential kernel for that stream, so the result is always that of
// a conformant decode (Inflate.cs / InflateBlocks.cs / InfCodes.cs / InfTree.cs).
#include <hip/hip_runtime.h>

#include "zs_inflate.hip"

namespace zs {
// 16 bytes stored / loaded at 2-byte alignment (cells are 16-bit): the alignment is part of the type, so the compiler emits
// global_store_dwordx4 without assuming what does not hold (gfx9 global memory takes unaligned accesses)
struct __attribute__((packed, aligned(2))) uint4_a2 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ void store_u4_a2(uint16_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    uint4_a2 v{a, b, c, d};
    *(uint4_a2 *)p = v;
}

constexpr int kFindChunk = 4096;     // input bytes per finder workgroup
constexpr int kFindMaxCand = 32;     // candidates kept per chunk (12 until late in round 5: blocks of ~100 symbols -- memLevel 1 -- are 22 to a chunk)
constexpr int kFindMaxSurv = 512;    // prefilter survivors per chunk (32768 bit offsets; ~0.5 % survive on random data)
constexpr int kParMaxBlocks = 1 << 16;

struct ParStream {
    const uint8_t *in;
    uint8_t *out;
    int64_t in_len, out_cap;
    int32_t chunk_off, nchunks;   // finder chunks
    int32_t cand_off, max_cand;   // flattened candidate list
    int32_t blk_off, max_blk;     // chain blocks
    int64_t cell_off;             // u16 cells of this stream's output
    int32_t fx_off, fx_regions;   // fixed-code blocks found ahead of the walk (zs_inf_fixed_scan_kernel): the stream's regions in the table
};
struct ParCand {
    int64_t bit;       // block header bit offset
    int64_t end_bit;   // bit offset after the block's EOB
    int64_t out_bytes;
    int32_t bfinal, ok;
    int32_t tab, pad_;  // >= 0: measured by the lane kernel, with tables and checkpoints in tabs[tab] (see zs_inf_decode_lane_kernel)
    int64_t tok_off;    // the candidate's slab in the token array (zs_inflate_tok.hip), and its room in tokens
    int32_t tok_cap, pad2_;
};
struct ParBlock {
    int64_t bit, out_off, out_bytes;
    int32_t tab, pad_;  // >= 0: decoded by sub-blocks from tabs[tab]; -1: by the wave decoder
};
struct ParState {
    int32_t ncand, nblk;
    int32_t ok;        // 1: block-parallel path valid for this stream
    int32_t win_off;   // index of the stream's first window in `windows` (set by the host once the block counts are known)
    int64_t out_len, end_bit;
    int32_t lane_blocks, pad_;  // 1: the chain has blocks with checkpoints but no tokens (the lane decoder and its flatten pass have work)
    int64_t tok_base, tok_need;  // the stream's slabs in the token array: where they begin, how many tokens they hold (zs_inflate_tok.hip)
};

// ---- shared block decoder (wave-uniform) ----
// MODE 0: measure only.  MODE 1: write 16-bit cells at o16[0 ..) (block-relative).
struct BlockOut {
    int64_t out_bytes, end_bit;
    int bfinal, err;
};

__device__ __forceinline__ void inf_seek(InfBits &b, int64_t bit) {
    b.pos = bit >> 3;
    b.buf = 0;
    b.cnt = 0;
    b.bad = false;
    b.ibase = -1;
    b.fill();
    b.drop((int)(bit & 7));
}
__device__ __forceinline__ int64_t inf_tell(const InfBits &b) { return b.pos * 8 - b.cnt; }

__device__ __forceinline__ uint16_t cell_load(const uint16_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // L2-coherent read of the wave's own earlier stores
}

// dynamic block header after BTYPE (wave-uniform): HLIT / HDIST / HCLEN, the bit-length code, the code lengths and both
// decode tables (InflateBlocks.cs:237-420, InfTree.cs:377-427).  Returns 0, or 1 for anything a conformant decoder rejects.
__device__ int inf_dyn_tables(InfBits &b, InfTables &T, uint8_t *lens, uint8_t *ll) {
    const int lane = threadIdx.x & 63;
    b.fill();
    if (b.cnt < 14) {
        return 1;
    }
    const int nlen = (int)b.take(5) + 257, ndist = (int)b.take(5) + 1, ncode = (int)b.take(4) + 4;
    if (nlen > 286 || ndist > 30) {
        return 1;
    }
    for (int i = lane; i < 320; i += 64) lens[i] = 0;
    __syncthreads();
    for (int i = 0; i < ncode; i++) {
        b.fill();
        unsigned v = b.take(3);
        if (lane == 0) lens[bl_order(i)] = (uint8_t)v;
    }
    __syncthreads();
    if (b.bad || inf_build(lens, 19, T.lit, 7, T.lcount, T.lsym) != 0) {
        return 1;
    }
    __syncthreads();
    uint8_t prev = 0;
    int idx = 0;
    while (idx < nlen + ndist) {
        b.fill();
        uint16_t e = T.lit[b.peek(7)];
        if (b.bad || e == kInfEsc || (int)(e & 15) > b.cnt) {
            return 1;
        }
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
                if (idx == 0) {
                    return 1;
                }
                val = prev;
                rep = 3 + (int)b.take(2);
            } else if (sym == 17) {
                rep = 3 + (int)b.take(3);
            } else {
                rep = 11 + (int)b.take(7);
            }
            if (idx + rep > nlen + ndist) {
                return 1;
            }
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
    int q = inf_build(lens, nlen, T.lit, kInfLitBits, T.lcount, T.lsym);
    if (q < 0 || (q > 0 && !(T.lcount[1] == 1 && nlen - T.lcount[0] == 1)) || lens[256] == 0) {
        return 1;
    }
    q = inf_build(lens + nlen, ndist, T.dist, kInfDistBits, T.dcount, T.dsym);
    if (q < 0 || (q > 0 && ndist - T.dcount[0] >= 1 && !(T.dcount[1] == 1 && ndist - T.dcount[0] == 1))) {
        return 1;
    }
    return 0;
}

constexpr int kParMaxSyms = 1 << 20;  // no encoder emits blocks this long; bounds the work a false candidate can cause

// abs_off: output position of the block's first byte in the stream (a match may not reach before 0)
template <int MODE>
__device__ BlockOut inf_block(InfBits &b, InfTables &T, uint8_t *lens, uint8_t *ll, uint16_t *o16, int64_t max_out, int64_t abs_off = 0) {
    const int lane = threadIdx.x & 63;
    BlockOut r{0, 0, 0, 0};
    b.fill();
    if (b.cnt < 3) {
        r.err = 1;
        return r;
    }
    r.bfinal = (int)b.take(1);
    const unsigned type = b.take(2);
    int64_t pos = 0;
    if (type == 3) {
        r.err = 1;
        return r;
    }
    if (type == 0) {
        b.drop(b.cnt & 7);
        b.fill();
        if (b.cnt < 32) {
            r.err = 1;
            return r;
        }
        unsigned len = b.take(16), nlen = b.take(16);
        int64_t src = b.pos - (b.cnt >> 3);
        if (len != (~nlen & 0xFFFF) || src + len > b.n || (int64_t)len > max_out) {
            r.err = 1;
            return r;
        }
        if (MODE == 1)
            for (unsigned i = lane; i < len; i += 64) o16[i] = b.in[src + i];
        b.pos = src + len;
        b.buf = 0, b.cnt = 0, b.ibase = -1;
        r.out_bytes = len;
        r.end_bit = b.pos * 8;
        return r;
    }
    if (type == 1) {
        for (int i = lane; i < 288; i += 64) lens[i] = (uint8_t)static_llen(i);
        __syncthreads();
        inf_build(lens, 288, T.lit, kInfLitBits, T.lcount, T.lsym);
        for (int i = lane; i < 32; i += 64) lens[i] = 5;
        __syncthreads();
        inf_build(lens, 30, T.dist, kInfDistBits, T.dcount, T.dsym);
    } else if (inf_dyn_tables(b, T, lens, ll)) {
        r.err = 1;
        return r;
    }
    __syncthreads();
    for (int nsym = 0;; nsym++) {
        if (b.bad || nsym > kParMaxSyms) {
            r.err = 1;
            return r;
        }
        b.fill();
        int sym, clen;
        {
            uint16_t e = T.lit[b.peek(kInfLitBits)];
            if (e != kInfEsc) sym = e >> 4, clen = e & 15;
            else sym = inf_slow(b, T.lcount, T.lsym, clen);
        }
        if (sym < 0 || clen > b.cnt) {
            r.err = 1;
            return r;
        }
        b.drop(clen);
        if (sym < 256) {
            if (pos >= max_out) {
                r.err = 2;
                return r;
            }
            if (MODE == 1 && lane == 0) o16[pos] = (uint16_t)sym;
            pos++;
        } else if (sym == 256) {
            break;
        } else {
            sym -= 257;
            if (sym >= 29) {
                r.err = 1;
                return r;
            }
            const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)b.take(extra_lbits(sym));
            b.fill();
            int ds, dl;
            {
                uint16_t e = T.dist[b.peek(kInfDistBits)];
                if (e != kInfEsc) ds = e >> 4, dl = e & 15;
                else ds = inf_slow(b, T.dcount, T.dsym, dl);
            }
            if (ds < 0 || ds >= 30 || dl > b.cnt) {
                r.err = 1;
                return r;
            }
            b.drop(dl);
            const int dist = base_dist(ds) + 1 + (int)b.take(extra_dbits(ds));
            if (pos + mlen > max_out) {
                r.err = 2;
                return r;
            }
            if (MODE == 1) {
                if (abs_off + pos - dist < 0) {  // before the start of the stream: "invalid distance" (InfCodes.cs:294)
                    r.err = 1;
                    return r;
                }
                // the wave's earlier cell stores must have reached L2 (vmcnt) before they are read back past L1 (sc1 loads)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int i = lane; i < mlen; i += 64) {
                    const int srcoff = dist >= mlen ? i : i % dist;
                    const int64_t sp = pos - dist + srcoff;  // block-relative source position
                    // before the block: byte (32768 + sp) of the window that precedes the block
                    const uint16_t v = sp >= 0 ? cell_load(o16 + sp) : (uint16_t)(0x8000u | (uint32_t)(kWSize + sp));
                    o16[pos + i] = v;
                }
            }
            pos += mlen;
        }
    }
    r.out_bytes = pos;
    r.end_bit = inf_tell(b);
    return r;
}

struct ParLds {
    InfTables T;
    uint8_t lens[320], ll[320 + 64];
    uint8_t ibuf[kInfInBuf + 64];
};

// ------------------------------------------------------------------ F
// 256 threads per 4 KiB of input; each thread tests the 128 bit offsets of its 16 bytes (prefilter), the survivors get
// the full header check, one lane each.
//
// The check is written for a wave whose lanes are all somewhere else in their headers: no tables in memory, no loops
// whose trip count differs from lane to lane except the one over the code-length symbols.  The bit-length code (<= 19
// symbols, <= 7 bits, complete) is decoded from three packed registers: lim (byte l = the left-aligned code value at
// which lengths > l begin, i.e. the Kraft sum of the lengths <= l in units of 2^-7), base (byte l = number of symbols
// shorter than l) and the symbols sorted by (length, symbol), 6 bits each.  With rv = the next 7 bits, first bit on top,
// the code length is 1 + #{l < 7 : rv >= lim[l]} and the symbol is entry base[l] + ((rv - lim[l - 1]) >> (7 - l)).
// Runs (symbols 16-18) are accounted in closed form.
struct LaneBits {
    const __attribute__((address_space(1))) uint8_t *in;  // the stream's input (global memory: no flat loads)
    int64_t n, pos;
    uint64_t buf;
    int cnt;
    bool bad;
    __device__ void fill() {
        if (cnt > 40) return;  // a symbol needs at most 15 + 5 + 15 + 13 bits; every load is on the decode's dependency chain
        if (pos + 8 <= n) {
            buf |= *(const __attribute__((address_space(1))) u64_unaligned *)(in + pos) << cnt;
            const int adv = (63 - cnt) >> 3;
            pos += adv;
            cnt += adv * 8;
        } else {
            while (cnt <= 56 && pos < n) {
                buf |= (uint64_t)in[pos] << cnt;
                pos++;
                cnt += 8;
            }
        }
    }
    __device__ void seek(int64_t bit) {
        pos = bit >> 3, buf = 0, cnt = 0, bad = false;
        fill();
        drop((int)(bit & 7));
    }
    __device__ int64_t tell() const { return pos * 8 - cnt; }
    __device__ uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    __device__ void drop(int k) {
        if (k > cnt) bad = true, k = cnt;
        buf >>= k;
        cnt -= k;
    }
    __device__ uint32_t take(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
};
template <typename Bits>
__device__ int lane_slow(const Bits &b, const uint16_t *count, const uint16_t *symtab, int &len_out) {
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int len = 1; len <= 15; len++) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int c = count[len];
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
// The header check's reader (round 5).  The plain reader loads 8 bytes whenever its lane runs low, and waits for them.  Here a
// lane asks for its next 128 bytes at once (eight 16-byte loads, one wait), parks them in its column of an LDS window and tops
// its buffer up from there; the rare header that is longer asks again.  The window is moved back where it would reach behind the
// caller's buffer; the position is compared with the stream's length before a header is accepted.  (Measured: 1.96 -> 1.8 ms
// per GiB of output.  The kernel's time is not the loads: half of the prefilter's survivors -- 16 per 4 KiB of input -- run
// through their ~100 code-length symbols before the literal/length code turns out incomplete; a quick pass capped at 24
// symbols with a packed second pass for the rest was slower, 2.6 ms.)
constexpr int kHdrWin = 32;  // dwords
typedef uint32_t hb_u32x4 __attribute__((ext_vector_type(4)));
typedef hb_u32x4 hb_u32x4_a4 __attribute__((aligned(4)));
struct HdrBits {
    const __attribute__((address_space(1))) uint32_t *base;  // the stream's first byte lies in base[0]
    uint32_t *win;   // LDS, the lane's column: win[j * 64] = dword wbase + j
    int d_last, skew, dw, wbase;
    uint64_t buf;
    int cnt;
    __device__ __forceinline__ void load() {
        int q = dw;
        q = q > d_last - (kHdrWin - 1) ? d_last - (kHdrWin - 1) : q;  // (streams here are longer than a window: kParMinInput)
        q = q < 0 ? 0 : q;
        wbase = q;
        const __attribute__((address_space(1))) uint32_t *p = base + q;
        hb_u32x4 v[kHdrWin / 4];
#pragma unroll
        for (int i = 0; i < kHdrWin / 4; i++) v[i] = *(const __attribute__((address_space(1))) hb_u32x4_a4 *)(p + 4 * i);
#pragma unroll
        for (int i = 0; i < kHdrWin / 4; i++) win[(4 * i) * 64] = v[i][0], win[(4 * i + 1) * 64] = v[i][1], win[(4 * i + 2) * 64] = v[i][2], win[(4 * i + 3) * 64] = v[i][3];
    }
    __device__ __forceinline__ void seek(const __attribute__((address_space(1))) uint8_t *in, int64_t n, int64_t bit, uint32_t *col) {
        const uintptr_t a = (uintptr_t)in;
        base = (const __attribute__((address_space(1))) uint32_t *)(a & ~(uintptr_t)3);
        skew = (int)(a & 3) * 8;
        d_last = (int)(((int64_t)(a & 3) + n - 1) >> 2);
        win = col;
        const int64_t sb = bit + skew;
        dw = (int)(sb >> 5);
        load();
        buf = 0, cnt = 0;
        fill();
        drop((int)(sb & 31));
        fill();
    }
    __device__ __forceinline__ void fill() {  // more than 32 bits in the buffer behind it
        if (cnt <= 32) {
            if (dw - wbase >= kHdrWin) load();  // (a header longer than the window)
            const int j = dw - wbase;
            buf |= (uint64_t)(dw <= d_last ? win[(j < 0 ? 0 : j) * 64] : 0u) << cnt;
            cnt += 32;
            dw++;
        }
    }
    __device__ __forceinline__ int64_t tell() const { return (int64_t)dw * 32 - cnt - skew; }
    __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    __device__ __forceinline__ void drop(int k) {
        buf >>= k;
        cnt -= k;
    }
    __device__ __forceinline__ uint32_t take(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
};
__device__ bool find_check_header(const __attribute__((address_space(1))) uint8_t *in, int64_t n, int64_t bit, uint32_t *lds_col) {
    if (bit + 17 > n * 8) return false;
    HdrBits b;
    b.seek(in, n, bit, lds_col);
    const uint32_t h = b.take(17);
    if (((h >> 1) & 3) != 2) return false;
    const int nlen = (int)((h >> 3) & 31) + 257, ndist = (int)((h >> 8) & 31) + 1, ncode = (int)((h >> 13) & 15) + 4;
    if (nlen > 286 || ndist > 30) return false;
    // (HCLEN + 4) 3-bit lengths in bl_order; the fields beyond count as 0
    b.fill();
    const int nb = 3 * ncode;
    uint32_t g0 = b.peek(30);
    b.drop(nb < 30 ? nb : 30);
    b.fill();
    uint32_t g1 = b.peek(27);
    b.drop(nb > 30 ? nb - 30 : 0);
    if (b.tell() > n * 8) return false;
    g0 = nb < 30 ? g0 & ((1u << nb) - 1u) : g0;
    g1 = nb <= 30 ? 0u : g1 & ((1u << (nb - 30)) - 1u);
    uint64_t bl = 0;  // 3 bits per symbol, in symbol order
#pragma unroll
    for (int i = 0; i < 19; i++) {
        const uint32_t v = i < 10 ? (g0 >> (3 * i)) & 7u : (g1 >> (3 * (i - 10))) & 7u;
        bl |= (uint64_t)v << (3 * bl_order(i));
    }
    // symbols per length: bit-sliced compare of all 19 fields with l, then a population count
    const uint64_t ones = 0x0249249249249249ull;  // bit 0 of each of the 19 fields
    const uint64_t s0 = bl & ones, s1 = (bl >> 1) & ones, s2 = (bl >> 2) & ones;
    uint64_t lim = 0, base = 0, offs = 0;
    {
        int cum = 0, nsym = 0;
#pragma unroll
        for (int l = 1; l <= 7; l++) {
            const uint64_t eq = ((l & 1) ? s0 : ~s0) & ((l & 2) ? s1 : ~s1) & ((l & 4) ? s2 : ~s2) & ones;
            const int c = __builtin_popcountll(eq);
            base |= (uint64_t)nsym << (8 * l);
            nsym += c;
            cum += c << (7 - l);
            lim |= (uint64_t)(cum > 255 ? 255 : cum) << (8 * l);
        }
        if (cum != 128) return false;  // the bit-length code must be complete
        offs = base;
    }
    uint64_t tab_lo = 0, tab_hi = 0;  // sorted symbols, 6 bits each: entries 0..9 and 10..18
#pragma unroll
    for (int sy = 0; sy < 19; sy++) {
        const int l = (int)((bl >> (3 * sy)) & 7);
        const int pos = (int)((offs >> (8 * l)) & 0xFF);
        offs += l ? 1ull << (8 * l) : 0ull;
        const uint64_t e = l ? (uint64_t)sy << (6 * (pos < 10 ? pos : pos - 10)) : 0ull;
        tab_lo |= pos < 10 ? e : 0ull;
        tab_hi |= pos < 10 ? 0ull : e;
    }
    const uint32_t lim_lo = (uint32_t)lim, lim_hi = (uint32_t)(lim >> 32);
    // decode nlen + ndist code lengths, accumulate Kraft sums of both alphabets (units of 2^-15)
    const int total = nlen + ndist;
    int idx = 0, prev = 0, nz_dist = 0, eob_len = 0;
    uint32_t klit = 0, kdist = 0;
    while (idx < total) {
        if (klit > 32768u || kdist > 32768u) return false;  // oversubscribed already: random data dies here within a few symbols
        b.fill();
        const uint32_t rv = __brev(b.peek(7)) >> 25;
        const int l = 1 + (int)(rv >= ((lim_lo >> 8) & 0xFF)) + (int)(rv >= ((lim_lo >> 16) & 0xFF)) + (int)(rv >= (lim_lo >> 24)) +
                      (int)(rv >= (lim_hi & 0xFF)) + (int)(rv >= ((lim_hi >> 8) & 0xFF)) + (int)(rv >= ((lim_hi >> 16) & 0xFF));
        const uint32_t fa = (uint32_t)(lim >> (8 * (l - 1))) & 0xFF;
        const int pos = (int)((base >> (8 * l)) & 0xFF) + (int)((rv - fa) >> (7 - l));
        const int sym = (int)(((pos < 10 ? tab_lo : tab_hi) >> (6 * (pos < 10 ? pos : pos - 10))) & 63);
        b.drop(l);
        if (sym == 16 && idx == 0) return false;
        const int eb = sym < 16 ? 0 : sym == 16 ? 2 : sym == 17 ? 3 : 7;
        const int rep = (sym < 16 ? 1 : sym == 18 ? 11 : 3) + (int)b.take(eb);
        const int val = sym < 16 ? sym : sym == 16 ? prev : 0;
        if (idx + rep > total) return false;
        // positions [idx, idx + rep): those below nlen belong to the literal/length code
        int nl = (idx + rep < nlen ? idx + rep : nlen) - idx;
        nl = nl < 0 ? 0 : nl;
        const int nd = rep - nl;
        if (val) {
            const uint32_t wgt = 32768u >> val;
            klit += (uint32_t)nl * wgt;
            kdist += (uint32_t)nd * wgt;
            nz_dist += nd;
            if (idx <= 256 && idx + rep > 256) eob_len = val;
        }
        prev = val;
        idx += rep;
    }
    if (!eob_len || b.tell() > n * 8) return false;         // (nothing behind the stream's last bit is a header)
    if (klit != 32768u) return false;                       // an encoder's literal/length code is complete
    if (!(kdist == 32768u || nz_dist <= 1)) return false;  // distance code: complete, or at most one code
    return true;
}

__global__ __launch_bounds__(256) void zs_inf_prefilter_kernel(const ParStream *ps, const uint2 *work, int32_t *surv_g, int32_t *surv_cnt) {
    __shared__ int nsurv;
    __shared__ int32_t surv[kFindMaxSurv];  // bit offsets (chunk-relative) that passed the register prefilter
    __shared__ uint8_t kraft9[512];  // entry i: the Kraft weights (128 >> v, 0 for v = 0) of the three 3-bit fields of i
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    const int chunk = (int)w.y;
    if (threadIdx.x == 0) nsurv = 0;
    for (int i = threadIdx.x; i < 512; i += 256)
        kraft9[i] = (uint8_t)(((0x80u >> (i & 7)) & 0x7Fu) + ((0x80u >> ((i >> 3) & 7)) & 0x7Fu) + ((0x80u >> (i >> 6)) & 0x7Fu));
    __syncthreads();
    // the thread's 16 bytes plus the 16 that follow, as four 64-bit words (zero past the end of the stream)
    const int64_t byte0 = (int64_t)chunk * kFindChunk + (int64_t)threadIdx.x * 16;
    // 32 bytes as eight 32-bit words (64-bit shifts are quarter rate; v_alignbit_b32 is full rate)
    uint32_t bw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (byte0 + 32 <= s.in_len && (((uintptr_t)(s.in + byte0)) & 15) == 0) {
        const uint4 *p16 = (const uint4 *)(s.in + byte0);
        const uint4 u = p16[0], v = p16[1];
        bw[0] = u.x, bw[1] = u.y, bw[2] = u.z, bw[3] = u.w, bw[4] = v.x, bw[5] = v.y, bw[6] = v.z, bw[7] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const int64_t a = byte0 + k;
            if (a < s.in_len) bw[k >> 2] |= (uint32_t)s.in[a] << (8 * (k & 3));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {  // compile-time word index: bw[] stays in registers
        const uint32_t c0 = bw[j], c1 = bw[j + 1], c2 = bw[j + 2], c3 = bw[j + 3];
        // First test, the word's 32 bit offsets at once (bit r of sh(k) is stream bit r + k): BTYPE == 2 is bit 1 clear and
        // bit 2 set; HLIT > 29 is bits 4..7 all set, HDIST > 29 bits 9..12.  About a fifth of all offsets pass.
        auto sh = [&](int k) { return __builtin_amdgcn_alignbit(c1, c0, k); };
        uint32_t m = ~sh(1) & sh(2);
        m &= ~(sh(4) & sh(5) & sh(6) & sh(7));
        m &= ~(sh(9) & sh(10) & sh(11) & sh(12));
        const int64_t base = byte0 * 8 + j * 32;
        if (base < 16) m &= ~((1u << (16 - base)) - 1u);         // the zlib header is no block
        const int64_t last = s.in_len * 8 - 17 - base;            // last offset with 17 header bits inside the stream
        if (last < 31) m = last < 0 ? 0u : m & ((2u << last) - 1u);
        // Second test, one passing offset per trip (a wave goes round as often as its fullest lane needs): the bit-length code
        // -- (HCLEN + 4) 3-bit fields from bit 17 -- must be complete (Kraft sum 1)
        while (m) {
            const int r = __builtin_ctz(m);
            m &= m - 1;
            // 96 bits starting at bit offset r of (c0, c1, c2, c3)
            const uint32_t x0 = __builtin_amdgcn_alignbit(c1, c0, r), x1 = __builtin_amdgcn_alignbit(c2, c1, r),
                           x2 = __builtin_amdgcn_alignbit(c3, c2, r);
            const int ncode = (int)((x0 >> 13) & 15) + 4;
            uint32_t g0 = (x0 >> 17) | (x1 << 15);  // bits 17..48: fields 0..9 (and two bits of field 10)
            uint32_t g1 = (x1 >> 15) | (x2 << 17);  // bits 47..78: fields 10..18
            // fields at and beyond HCLEN + 4 count as length 0
            const int nb = 3 * ncode;
            g0 = nb < 30 ? g0 & ((1u << nb) - 1u) : g0;
            g1 = nb <= 30 ? 0u : g1 & ((1u << (nb - 30)) - 1u);
            // units of 2^-7: a length v > 0 adds 128 >> v.  Three fields a lookup (a 512-entry table in LDS; field by field the sum was
            // 95 of the trip's ~110 vector instructions, and a fifth of all bit offsets take this trip)
            const int kraft = (int)kraft9[g0 & 511u] + (int)kraft9[(g0 >> 9) & 511u] + (int)kraft9[(g0 >> 18) & 511u] + (int)kraft9[(g0 >> 27) & 7u] +
                              (int)kraft9[g1 & 511u] + (int)kraft9[(g1 >> 9) & 511u] + (int)kraft9[g1 >> 18];
            if (kraft != 128) continue;
            int at = atomicAdd(&nsurv, 1);
            if (at < kFindMaxSurv) surv[at] = (int32_t)(base + r - (int64_t)chunk * kFindChunk * 8);
        }
    }
    __syncthreads();
    // the survivors go out for the header-check kernel (a count above kFindMaxSurv flags the overflow)
    const int ns = nsurv < kFindMaxSurv ? nsurv : kFindMaxSurv;
    int32_t *dst = surv_g + (int64_t)blockIdx.x * kFindMaxSurv;
    for (int i = threadIdx.x; i < ns; i += 256) dst[i] = surv[i];
    if (threadIdx.x == 0) surv_cnt[blockIdx.x] = nsurv;
}

// Full header check of a chunk's survivors: one lane each, one wave per chunk.
// (a chunk has ~30 survivors: two chunks share a wave, 32 lanes each)
__global__ __launch_bounds__(64) void zs_inf_check_kernel(const ParStream *ps, const uint2 *work, int nwork, const int32_t *surv_g,
                                                          const int32_t *surv_cnt, int64_t *cand_bits, int32_t *cand_cnt) {
    __shared__ int64_t found[2][kFindMaxCand];
    __shared__ int nfound[2];
    __shared__ uint32_t hwin[kHdrWin * 64];  // the lanes' header bytes (HdrBits)
    const int half = threadIdx.x >> 5, hl = threadIdx.x & 31;
    const int wi = blockIdx.x * 2 + half;
    const bool live = wi < nwork;
    const uint2 w = work[live ? wi : 0];
    const ParStream s = ps[w.x];
    const int chunk = (int)w.y;
    const int nsurv = live ? surv_cnt[wi] : 0;
    const int32_t *surv = surv_g + (int64_t)wi * kFindMaxSurv;
    if (hl == 0) nfound[half] = 0;
    __syncthreads();
    const int ns = nsurv < kFindMaxSurv ? nsurv : kFindMaxSurv;
    for (int i = hl; i < ns; i += 32) {
        const int64_t bit = (int64_t)chunk * kFindChunk * 8 + surv[i];
        if (find_check_header((const __attribute__((address_space(1))) uint8_t *)(uintptr_t)s.in, s.in_len, bit, hwin + threadIdx.x)) {
            int at = atomicAdd(&nfound[half], 1);
            if (at < kFindMaxCand) found[half][at] = bit;
        }
    }
    __syncthreads();
    if (hl == 0 && live) {
        int nf = nfound[half];
        if (nsurv > kFindMaxSurv) nf = kFindMaxCand + 1;  // survivor list overflow: give the stream to the sequential decoder
        int n = nf < kFindMaxCand ? nf : kFindMaxCand;
        int64_t *fd = found[half];
        for (int i = 1; i < n; i++) {  // insertion sort: a handful of entries
            int64_t v = fd[i];
            int j = i - 1;
            while (j >= 0 && fd[j] > v) fd[j + 1] = fd[j], j--;
            fd[j + 1] = v;
        }
        const int64_t base = ((int64_t)s.chunk_off + chunk) * kFindMaxCand;
        for (int i = 0; i < n; i++) cand_bits[base + i] = fd[i];
        cand_cnt[s.chunk_off + chunk] = nf;  // > kFindMaxCand flags an overflow
    }
}

// flatten the per-chunk lists into one ordered list per stream: one workgroup per stream, 256 chunks per step
// (a workgroup-wide prefix sum of the counts places every chunk's candidates)
__global__ __launch_bounds__(256) void zs_inf_flatten_kernel(const ParStream *ps, ParState *st, const int64_t *cand_bits, const int32_t *cand_cnt, ParCand *cands,
                                                             int nstreams) {
    __shared__ int sc[256];
    __shared__ int s_ok;
    const int si = blockIdx.x, tid = threadIdx.x;
    if (si >= nstreams) return;
    const ParStream s = ps[si];
    if (tid == 0) s_ok = 1;
    __syncthreads();
    int base = 0;
    for (int c0 = 0; c0 < s.nchunks; c0 += 256) {
        const int c = c0 + tid;
        int k = c < s.nchunks ? cand_cnt[s.chunk_off + c] : 0;
        if (k > kFindMaxCand) s_ok = 0, k = kFindMaxCand;  // a chunk's list overflowed
        sc[tid] = k;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int t = tid >= d ? sc[tid - d] : 0;
            __syncthreads();
            sc[tid] += t;
            __syncthreads();
        }
        const int at = base + sc[tid] - k;
        for (int i = 0; i < k; i++) {
            if (at + i >= s.max_cand) {
                s_ok = 0;
                break;
            }
            ParCand &d = cands[s.cand_off + at + i];
            d.bit = cand_bits[((int64_t)s.chunk_off + c) * kFindMaxCand + i];
            d.end_bit = 0, d.out_bytes = 0, d.bfinal = 0, d.ok = 0, d.tab = -1, d.pad_ = 0;
            d.tok_off = 0, d.tok_cap = 0, d.pad2_ = 0;
        }
        base += sc[255];
        __syncthreads();
    }
    if (tid == 0) {
        st[si].ncand = base < s.max_cand ? base : s.max_cand;
        st[si].ok = s_ok;
        st[si].nblk = 0;
        st[si].out_len = 0;
        st[si].end_bit = 0;
        st[si].lane_blocks = 0, st[si].pad_ = 0;
        st[si].tok_base = 0, st[si].tok_need = 0;
    }
}

// ------------------------------------------------------------------ D1, wave form: one wave per candidate (fastest chain per block;
// used while every candidate of the batch gets a wave at once)
__global__ __launch_bounds__(64) void zs_inf_measure_kernel(const ParStream *ps, const ParState *st, const uint2 *work, ParCand *cands) {
    __shared__ ParLds L;
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    if ((int)w.y >= st[w.x].ncand) return;
    ParCand &c = cands[s.cand_off + w.y];
    if (!st[w.x].ok || c.bit < 16 || c.bit + 17 > s.in_len * 8) {  // (as in zs_inf_measure_sync_kernel)
        if (threadIdx.x == 0) c.ok = 0;
        return;
    }
    InfBits b{s.in, s.in_len, 0, 0, 0, false, L.ibuf, -1};
    inf_seek(b, c.bit);
    BlockOut r = inf_block<0>(b, L.T, L.lens, L.ll, nullptr, (int64_t)1 << 40);
    if (threadIdx.x == 0) {
        c.end_bit = r.end_bit;
        c.out_bytes = r.out_bytes;
        c.bfinal = r.bfinal;
        c.ok = r.err == 0;
    }
}

// ------------------------------------------------------------------ D1, self-synchronising form
// Measuring a candidate is a decode without output, and a block's decode is one dependency chain of ~16 Ki symbols.  A
// Huffman bit stream decoded from a wrong bit offset falls into step with the true symbol boundaries after a few dozen
// symbols, so the chain can be cut: the block's bits are divided into 64 subsequences of S bits (S from the distance to
// the next candidate), lane j decodes subsequence j from its nominal first bit until it crosses into subsequence j + 1
// and reports the symbol boundary it arrived at (its exit).  Lane 0 starts at the true first symbol; every other lane
// whose entry differs from its predecessor's exit decodes again from that exit, all of them at once, until the chain of
// exits = entries is unbroken from lane 0 on -- each pass extends the proven prefix by at least one lane, so the result
// is that of the sequential decode however badly a subsequence synchronises (typically two or three passes in all).
// One wave per candidate, the block's tables in LDS shared by its lanes.  The entry of every subsequence (bit position,
// output position) is a checkpoint for the decode pass, which decodes the subsequences independently the same way.
// Only dynamic blocks come out of the finder; whatever this kernel does not accept (ok = 0) the chain kernel measures
// with the wave decoder, and every accepted size is checked again by the decode pass and the Adler-32.
#ifndef ZS_SUB_MIN
#define ZS_SUB_MIN 1024
#endif
constexpr int kSubMinBits = ZS_SUB_MIN, kSubMaxBits = 16384;  // subsequence length: a multiple of 64 bits in this range
constexpr int kCkMax = 256;  // checkpoints kept per block; a block with more subsequences goes to the wave decoder
struct LaneTabs {
    // the block's decode tables exactly as InfTables lays them out (copied with 16-byte stores), kept for the decode pass
    uint16_t lit[1 << kInfLitBits];
    uint16_t dist[1 << kInfDistBits];
    uint16_t lcount[16], dcount[16];
    uint16_t lsym[288], dsym[32];
    int32_t nsub, pad_;
    uint32_t ck_bit[kCkMax + 1];  // checkpoint k: bit position relative to the block header, at the first symbol of subsequence k
    uint32_t ck_out[kCkMax + 1];  //               block-relative output position; entry nsub = the block's end
};
static_assert(sizeof(InfTables) % 16 == 0 && sizeof(LaneTabs) % 16 == 0 && offsetof(LaneTabs, nsub) == sizeof(InfTables),
              "LaneTabs begins with an InfTables image copied with 16-byte stores");
// One lane decodes (without output) from bit `entry` to the first symbol boundary at or after `gend`, or to END_BLOCK.
// flags: 0 = crossed gend, 1 = END_BLOCK (exit_bit is the bit after it), 2 = not decodable from here.
__device__ __forceinline__ void sub_measure(const __attribute__((address_space(1))) uint8_t *in, int64_t n, const InfTables &T, int64_t entry,
                                            int64_t gend, int64_t &exit_bit, int &nout, int &nsym, int &flags) {
    LaneBits b{in, n, 0, 0, 0, false};
    b.seek(entry);
    int out = 0, ns = 0, fl = 0;
    int64_t cur = entry;
    while (cur < gend) {
        b.fill();
        int sym, clen;
        {
            const uint16_t e = T.lit[b.peek(kInfLitBits)];
            if (e != kInfEsc) sym = e >> 4, clen = e & 15;
            else sym = lane_slow(b, T.lcount, T.lsym, clen);
        }
        if (sym < 0 || clen > b.cnt) {
            fl = 2;
            break;
        }
        b.drop(clen);
        if (sym < 256) {
            out++;
        } else if (sym == 256) {
            fl = 1;
            cur = b.tell();
            break;
        } else {
            sym -= 257;
            if (sym >= 29) {
                fl = 2;
                break;
            }
            const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)b.take(extra_lbits(sym));
            b.fill();
            int ds, dl;
            {
                const uint16_t e = T.dist[b.peek(kInfDistBits)];
                if (e != kInfEsc) ds = e >> 4, dl = e & 15;
                else ds = lane_slow(b, T.dcount, T.dsym, dl);
            }
            if (ds < 0 || ds >= 30 || dl > b.cnt) {
                fl = 2;
                break;
            }
            b.drop(dl);
            (void)b.take(extra_dbits(ds));
            out += mlen;
        }
        if (b.bad) {
            fl = 2;
            break;
        }
        ns++;
        cur = b.tell();
    }
    exit_bit = cur, nout = out, nsym = ns, flags = fl;
}
__global__ __launch_bounds__(64) void zs_inf_measure_sync_kernel(const ParStream *ps, const ParState *st, const uint2 *work, ParCand *cands,
                                                                 LaneTabs *tabs) {
    __shared__ __attribute__((aligned(16))) ParLds L;
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    const int ncand = st[w.x].ncand;
    if ((int)w.y >= ncand) return;
    ParCand &c = cands[s.cand_off + w.y];
    const int lane = threadIdx.x;
    const int64_t cbit = c.bit, nbits = s.in_len * 8;
    if (!st[w.x].ok || cbit < 16 || cbit + 17 > nbits) {  // (not a header offset of this stream: nothing is read through it)
        if (lane == 0) c.ok = 0;
        return;
    }
    // the block most likely ends where the next candidate begins (candidate bits are final since the flatten pass)
    int64_t hint = (int)w.y + 1 < ncand ? cands[s.cand_off + w.y + 1].bit : nbits;
    if (hint <= cbit || hint > nbits) hint = nbits;
    InfBits hb{s.in, s.in_len, 0, 0, 0, false, L.ibuf, -1};
    inf_seek(hb, cbit);
    hb.fill();
    int bfinal = 0;
    bool ok = hb.cnt >= 3;
    if (ok) {
        bfinal = (int)hb.take(1);
        ok = hb.take(2) == 2;
    }
    ok = ok && inf_dyn_tables(hb, L.T, L.lens, L.ll) == 0;
    __syncthreads();
    if (!ok) {
        if (lane == 0) c.ok = 0;
        return;
    }
    const int64_t b0 = inf_tell(hb);  // first symbol of the block
    int S = kSubMinBits;
    if (hint > b0) {
        const int64_t per = ((hint - b0 + 63) / 64 + 63) & ~(int64_t)63;
        S = per < kSubMinBits ? kSubMinBits : per > kSubMaxBits ? kSubMaxBits : (int)per;
    }
    const __attribute__((address_space(1))) uint8_t *gin = (const __attribute__((address_space(1))) uint8_t *)(uintptr_t)s.in;
    LaneTabs &T = tabs[blockIdx.x];
    int64_t entry0 = b0, out_base = 0, total_syms = 0, end_bit = 0;
    int nck = 0, result = 0;  // result: 1 = END_BLOCK reached on the proven chain, 2 = not decodable
    bool store = true, hint_ok = true;
    for (int round = 0; result == 0; round++) {
        const int64_t g = b0 + ((int64_t)round * 64 + lane) * S, gend = g + S;
        int64_t entry = lane == 0 ? entry0 : g, exit_bit = -1;
        int nout = 0, nsym = 0, flags = 0, nvalid = 0, lf = 0;
        bool spec = false;  // this lane has decoded its subsequence (from `entry`)
        bool run = lane == 0 || (g < nbits && (g < hint || !hint_ok));
        for (;;) {
            if (run) {
                sub_measure(gin, s.in_len, L.T, entry, gend, exit_bit, nout, nsym, flags);
                spec = true;
            }
            const int64_t pe = __shfl_up(exit_bit, 1);
            const int pf = __shfl_up(flags, 1);
            const bool pspec = __shfl_up((int)spec, 1) != 0;
            const bool link = lane == 0 || (pspec && pf == 0 && spec && entry == pe);
            const uint64_t m = __ballot(link);
            nvalid = m == ~0ull ? 64 : (int)__builtin_ctzll(~m);  // lanes [0, nvalid) are proven
            lf = __shfl(flags, nvalid - 1);
            if (nvalid == 64 || lf != 0) break;
            // lane nvalid decodes from a proven exit; the lanes behind it whose entry no longer fits their predecessor's exit
            // go again too (their predecessor's exit is usually right already: that is the self-synchronisation)
            run = lane >= nvalid && pspec && pf == 0 && (!spec || entry != pe);
            if (run) entry = pe;
            // the chain has walked past the hint: it was not the block's end, so everyone behind speculates as well
            if (__shfl((int)spec, nvalid) == 0) hint_ok = false;
            if (!hint_ok && !spec && !run && lane > nvalid && g < nbits) run = true;
        }
        // checkpoints of the proven lanes
        const bool valid = lane < nvalid;
        int incl = valid ? nout : 0;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        int sy = valid ? nsym : 0;
        for (int d = 32; d; d >>= 1) sy += __shfl_xor(sy, d);
        if (store && nck + nvalid <= kCkMax) {
            if (valid) {
                T.ck_bit[nck + lane] = (uint32_t)(entry - cbit);
                T.ck_out[nck + lane] = (uint32_t)(out_base + incl - nout);
            }
        } else {
            store = false;
        }
        nck += nvalid;
        out_base += __shfl(incl, 63);
        total_syms += sy;
        if (lf == 1) {
            end_bit = __shfl(exit_bit, nvalid - 1);
            result = 1;
        } else if (lf == 2 || total_syms > kParMaxSyms || out_base >= ((int64_t)1 << 31)) {
            result = 2;
        } else {
            entry0 = __shfl(exit_bit, 63);
            if (entry0 - cbit >= ((int64_t)1 << 32)) result = 2;
        }
    }
    if (result != 1) {
        if (lane == 0) c.ok = 0;
        return;
    }
    if (store && end_bit - cbit < ((int64_t)1 << 32)) {
        // the decode pass takes the block by subsequences: tables and checkpoints
        const uint4 *src = (const uint4 *)&L.T;
        uint4 *dst = (uint4 *)&T;
        for (int i = lane; i < (int)(sizeof(InfTables) / 16); i += 64) dst[i] = src[i];
        if (lane == 0) {
            T.ck_bit[nck] = (uint32_t)(end_bit - cbit);
            T.ck_out[nck] = (uint32_t)out_base;
            T.nsub = nck;
            c.tab = (int32_t)blockIdx.x;
        }
    }
    if (lane == 0) {
        c.end_bit = end_bit;
        c.out_bytes = out_base;
        c.bfinal = bfinal;
        c.ok = 1;
    }
}

// ------------------------------------------------------------------ C
// The chain of a stream's blocks without the walk, for the stream whose blocks the finder has all reported (dynamic blocks:
// everything this library's own deflate writes at levels >= 1 on text): every measured candidate looks up the candidate
// that starts where it ends (binary search over the sorted header offsets, in LDS), the candidates reachable from the one
// at bit 16 are marked by pointer doubling (15 rounds for <= 16 Ki candidates), and since a block starts after the one
// before it, the chain's order is the candidates' order: block numbers and output offsets are prefix sums over the marked
// ones.  Anything else -- a block the finder did not report, a measure that failed, too many candidates -- leaves
// nblk = -1 and zs_inf_chain_kernel walks the stream as before.
constexpr int kChainParMax = 16384;
constexpr int kChainParLds = kChainParMax * 9 + 64;
__global__ __launch_bounds__(1024) void zs_inf_chain_par_kernel(const ParStream *ps, ParState *st, const ParCand *cands, ParBlock *blocks,
                                                                int lane_decode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *bits = (uint32_t *)smem;              // header bit offsets; later the second jump array
    int32_t *jump = (int32_t *)(bits + kChainParMax);
    uint8_t *reach = (uint8_t *)(jump + kChainParMax);
    __shared__ int sh_fail;
    __shared__ int64_t w_bytes[16];
    __shared__ int w_cnt[16];
    const ParStream s = ps[blockIdx.x];
    ParState &ss = st[blockIdx.x];
    if (!ss.ok) return;
    const int n = ss.ncand, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ParCand *cd = cands + s.cand_off;
    bool can = n >= 1 && n <= kChainParMax && s.in_len >= 6 && s.in_len < ((int64_t)1 << 28);
    // zlib header (Inflate.cs:120-170): anything unusual is the walking kernel's to classify
    if (can && ((s.in[0] & 0x0F) != 8 || (s.in[0] >> 4) > 7 || (((unsigned)s.in[0] << 8) + s.in[1]) % 31 != 0 || (s.in[1] & 0x20))) can = false;
    if (!can) {
        if (tid == 0) ss.nblk = -1;
        return;
    }
    if (tid == 0) sh_fail = 0;
    for (int i = tid; i < n; i += 1024) bits[i] = (uint32_t)cd[i].bit, reach[i] = 0;
    __syncthreads();
    // successor of every candidate: -1 none, itself for a final block
    int bad = 0;
    for (int i = tid; i < n; i += 1024) {
        const ParCand q = cd[i];
        if (i + 1 < n && bits[i] >= bits[i + 1]) bad = 1;  // the offsets must ascend
        int nx = -1;
        if (q.ok && q.bfinal) nx = i;
        else if (q.ok) {
            const uint32_t e = (uint32_t)q.end_bit;
            int lo = i + 1, hi = n;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (bits[mid] < e) lo = mid + 1;
                else hi = mid;
            }
            if (lo < n && bits[lo] == e && cd[lo].ok) nx = lo;
        }
        jump[i] = nx;
    }
    if (tid == 0 && !(bits[0] == 16u && cd[0].ok)) bad = 1;  // the first block starts behind the 2-byte header
    if (bad) sh_fail = 1;
    __syncthreads();
    if (sh_fail) {
        if (tid == 0) ss.nblk = -1;
        return;
    }
    // reachable from candidate 0: round k marks what lies 2^k steps behind a marked candidate, then the jumps double
    int32_t *ja = jump, *jb = (int32_t *)bits;
    if (tid == 0) reach[0] = 1;
    __syncthreads();
    for (int r = 0; r < 15; r++) {
        for (int i = tid; i < n; i += 1024)
            if (reach[i] && ja[i] >= 0) reach[ja[i]] = 1;
        for (int i = tid; i < n; i += 1024) {
            const int j = ja[i];
            jb[i] = j < 0 ? -1 : (ja[j] >= 0 ? ja[j] : j);  // a candidate without a successor keeps its -1: a dead end
        }
        __syncthreads();
        int32_t *t = ja;
        ja = jb, jb = t;
    }
    // block numbers and output offsets: prefix sums over the marked candidates, 16 consecutive ones per thread
    const int per = (n + 1023) / 1024, i0 = tid * per, i1 = i0 + per < n ? i0 + per : n;
    int cnt = 0, dead = 0;
    int64_t bytes = 0;
    for (int i = i0; i < i1; i++)
        if (reach[i]) {
            cnt++;
            bytes += cd[i].out_bytes;
            if (ja[i] < 0) dead = 1;
        }
    int pc = cnt;
    int64_t pb = bytes;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int c2 = __shfl_up(pc, o);
        const int64_t b2 = __shfl_up(pb, o);
        if (lane >= o) pc += c2, pb += b2;
    }
    if (lane == 63) w_cnt[wave] = pc, w_bytes[wave] = pb;
    if (dead) sh_fail = 1;
    __syncthreads();
    int base_c = 0, tot_c = 0;
    int64_t base_b = 0, tot_b = 0;
    for (int k = 0; k < 16; k++) {
        if (k < wave) base_c += w_cnt[k], base_b += w_bytes[k];
        tot_c += w_cnt[k], tot_b += w_bytes[k];
    }
    if (sh_fail || tot_c < 1 || tot_c > s.max_blk || tot_b > s.out_cap) {
        if (tid == 0) ss.nblk = -1;
        return;
    }
    int nb = base_c + pc - cnt;
    int64_t out = base_b + pb - bytes;
    ParBlock *bl = blocks + s.blk_off;
    for (int i = i0; i < i1; i++)
        if (reach[i]) {
            const ParCand q = cd[i];
            bl[nb] = {q.bit, out, q.out_bytes, lane_decode ? q.tab : -1, 0};
            if (lane_decode && q.tab >= 0 && !(q.tab & 0x40000000)) ss.lane_blocks = 1;
            nb++, out += q.out_bytes;
            if (q.bfinal) ss.end_bit = q.end_bit;  // exactly one marked candidate is final: the chain ends there
        }
    if (tid == 0) ss.nblk = tot_c, ss.out_len = tot_b;
}

#ifndef ZS_NO_FIXED_END
constexpr bool kNoFixedEnd = false;
#else
constexpr bool kNoFixedEnd = true;
#endif
constexpr int64_t kWalkMinInput = 256 * 1024;  // below it the walking kernel measures stored blocks only (zs_inf_chain_kernel)

// ---- where a fixed-code block ends (InfTree.cs:43-85 / Inflate_trees_fixed: 7-9 bit literal/length codes, 5-bit distances) ----
// The finder cannot tell a fixed block's three header bits from data, so a stream of them (CompressionStrategy.Fixed,
// Z_FIXED; the short blocks of a small memLevel) is walked block by block, and a block's end is known only once its symbols
// have been decoded: by one wave symbol after symbol that was 8 MB/s.  Here the wave's 64 lanes decode 64 consecutive
// stretches of the block's bits at once, each from a guessed start; a Huffman decoder that starts in the middle of a symbol
// falls into step within a few symbols, so most lanes leave their stretch where the true parse leaves it.  Lane 0's start is
// true; every lane then takes its predecessor's exit for its start and decodes again if that is not where it began -- until
// nothing changes (at most 64 times; four to seven on text: a run of 8-bit literal codes keeps a decoder that is out of step out
// of step, only the matches' odd lengths bring it back).  Nothing is written: the block's end, its output bytes.  16 MiB of text
// under Z_FIXED: 2.1 s -> 0.19 s, three quarters of it this walk.
struct FxBits {
    const uint8_t *in;
    int64_t n;  // bytes
    // the bits from `pos` on, at least 50 of them (zeros behind the input)
    __device__ __forceinline__ uint64_t peek(int64_t pos) const {
        typedef const __attribute__((address_space(1))) uint32_t __attribute__((aligned(1))) *g_u32u;
        const int64_t by = pos >> 3;
        uint64_t v = 0;
        if (by + 8 <= n) {
            const g_u32u q = (g_u32u)(in + by);
            v = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
        } else {
            for (int k = 0; k < 8; k++)
                if (by + k < n) v |= (uint64_t)in[by + k] << (8 * k);
        }
        return v >> (pos & 7);
    }
};
// A lane's window of 128 bits of the stream (two loads for ~9 symbols: a load per symbol made the pass 0.5 us a symbol).
struct FxWin {
    uint64_t lo, hi;
    int64_t at;  // bit position of lo's bit 0 (a multiple of 8); < 0: nothing loaded
    __device__ __forceinline__ void load(const FxBits &B, int64_t pos) {
        typedef const __attribute__((address_space(1))) uint32_t __attribute__((aligned(1))) *g_u32u;
        const int64_t by = pos >> 3;
        at = by << 3;
        if (by + 16 <= B.n) {
            const g_u32u q = (g_u32u)(B.in + by);
            lo = (uint64_t)q[0] | ((uint64_t)q[1] << 32), hi = (uint64_t)q[2] | ((uint64_t)q[3] << 32);
        } else {
            lo = hi = 0;
            for (int k = 0; k < 16; k++)
                if (by + k < B.n) (k < 8 ? lo : hi) |= (uint64_t)B.in[by + k] << (8 * (k & 7));
        }
    }
    // 32 bits from `pos` on (pos - at <= 96)
    __device__ __forceinline__ uint32_t bits(int64_t pos) const {
        const int c = (int)(pos - at);
        const uint64_t v = c < 64 ? ((lo >> c) | (c ? hi << (64 - c) : 0ull)) : (hi >> (c - 64));
        return (uint32_t)v;
    }
};
// one symbol at bit `pos`: 0 -- decoded (pos and the output count move on), 1 -- it was the end-of-block code, 2 -- not a symbol
// of the fixed code, or the input ends inside it
__device__ __forceinline__ int fx_symbol(const FxBits &B, FxWin &W, int64_t &pos, int64_t &outb, int64_t nbits) {
    if (pos + 7 > nbits) return 2;
    if (W.at < 0 || pos - W.at > 96) W.load(B, pos);
    uint32_t w = W.bits(pos);
    const uint32_t r = __builtin_bitreverse32(w & 0x1FFu) >> 23;  // the next nine bits, first bit on top
    int clen, sym;
    if ((r >> 2) < 24u) clen = 7, sym = 256 + (int)(r >> 2);
    else if ((r >> 1) < 192u) clen = 8, sym = (int)(r >> 1) - 48;
    else if ((r >> 1) < 200u) clen = 8, sym = 280 + (int)(r >> 1) - 192;
    else clen = 9, sym = 144 + (int)r - 400;
    w >>= clen;
    int used = clen;
    if (sym <= 256) {
        pos += used;
        if (pos > nbits) return 2;
        if (sym == 256) return 1;
        outb += 1;
        return 0;
    }
    sym -= 257;
    if (sym >= 29) return 2;
    const int xl = extra_lbits(sym);
    const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)(w & ((1u << xl) - 1u));
    w >>= xl, used += xl;  // (<= 14 bits so far, the distance code is in the 32)
    const uint32_t ds = __builtin_bitreverse32(w & 31u) >> 27;
    if (ds >= 30u) return 2;
    used += 5 + extra_dbits((int)ds);
    pos += used;
    if (pos > nbits) return 2;
    outb += mlen;
    return 0;
}
__device__ BlockOut inf_fixed_end(const uint8_t *in, int64_t n_bytes, int64_t hdr_bit) {
    const int lane = (int)(threadIdx.x & 63);
    const FxBits B{in, n_bytes};
    const int64_t nbits = n_bytes * 8;
    BlockOut r{0, 0, 0, 0};
    if (hdr_bit + 3 > nbits) {
        r.err = 1;
        return r;
    }
    r.bfinal = (int)(B.peek(hdr_bit) & 1u);
    int64_t base = hdr_bit + 3, total = 0;
    int S = 256;  // bits per lane and round: short at first (a block of a hundred symbols), then 4096
    for (;;) {
        const int64_t my_end = base + (int64_t)(lane + 1) * S;
        int64_t start = base + (int64_t)lane * S, done_for = -1, e = 0, ob = 0;
        int flag = 0;
        [[maybe_unused]] int fx_its = 0;
        if (lane > 0) {
            // a running start: from 128 bits in front of the stretch the decode is in step by the time it reaches it, nine times in
            // ten -- the first boundary in the stretch is then where the predecessor will leave, and one pass is all it takes
            int64_t p = start - 128 > base ? start - 128 : base, dummy = 0;
            FxWin W;
            W.at = -1;
            while (p < start)
                if (fx_symbol(B, W, p, dummy, nbits)) {
                    p = start;
                    break;
                }
            start = p;
        }
        for (int it = 0; it < 66; it++) {
            fx_its++;
            if (start != done_for) {
                int64_t p = start;
                ob = 0, flag = 0;
                FxWin W;
                W.at = -1;
                while (p < my_end) {
                    const int f = fx_symbol(B, W, p, ob, nbits);
                    if (f) {
                        flag = f;
                        break;
                    }
                }
                e = p, done_for = start;
            }
            // the predecessor's exit is where this lane's stretch truly begins -- unless the predecessor stopped (the block's end, or
            // bits that are no symbols: a lane behind it has nothing to decode)
            const int64_t pe = __shfl_up(e, 1);
            const int pf = __shfl_up(flag, 1);
            const int64_t ns = (lane == 0 || pf != 0) ? start : pe;
            const bool moved = ns != start;
            start = ns;
            if (!__ballot(moved)) break;
        }
#ifdef ZS_FX_DEBUG
        if (lane == 0 && hdr_bit < 3000000) printf("FX block at bit %lld: round S=%d took %d iterations\n", (long long)hdr_bit, S, fx_its);
#endif
        // the lanes up to the first one that stopped have decoded the true parse
        const uint64_t fm = __ballot(flag != 0);
        const int lstar = fm ? (int)__builtin_ctzll(fm) : 63;
        int64_t sum = lane <= lstar ? ob : 0;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        total += sum;
        const int fstar = __shfl(flag, lstar);
        const int64_t estar = __shfl(e, lstar);
        if (fstar == 2) {
            r.err = 1;
            return r;
        }
        if (fstar == 1) {
            r.end_bit = estar, r.out_bytes = total;
            return r;
        }
        base = estar;  // (lstar == 63: the last lane's exit)
        S = S < 4096 ? S * 4 : 4096;
    }
}
// ---- fixed-code blocks found ahead of the walk.  The walk over a stream of fixed blocks is one chain -- a block's start is the
// block before's end -- but what the walk needs of a block (its end, its output bytes) is a function of its start alone.  So one
// wave per 64 KiB of the stream decodes from a guessed bit (as if a block began there) to the next end-of-block code -- a true
// block's end once the decoder is in step -- and goes on block by block while the headers say "fixed", writing down
// (start, end, bytes) for every block it believes in.  The walk then looks its block's start up in its region's entries: found, the
// block costs a search; not found (the region's wave was not in step by its first end-of-block, a block of another kind in
// between), the walk measures the block itself (inf_fixed_end).  An entry is looked up by a true start only, and for a true start it
// holds what inf_fixed_end would give: entries behind guesses that went wrong are never read.
constexpr int64_t kFxRegionBits = 8 * 65536;
constexpr int kFxEntries = 16;  // per region (a fixed block of 16 Ki symbols is ~25 KB: two or three to a region)
struct FxEntry {
    int64_t start, end, out_bytes;
    int32_t bfinal, pad_;
};
__global__ __launch_bounds__(64) void zs_inf_fixed_scan_kernel(const ParStream *ps, const ParState *st, FxEntry *tab) {
    const uint2 w = make_uint2(blockIdx.y, blockIdx.x);  // (stream, region)
    const ParStream s = ps[w.x];
    if ((int)w.y >= s.fx_regions) return;
    const int lane = (int)(threadIdx.x & 63);
    FxEntry *out = tab + ((int64_t)s.fx_off + w.y) * kFxEntries;
    if (lane < kFxEntries) out[lane].start = -1;
    if (!st[w.x].ok || st[w.x].nblk >= 1) return;  // (no chain to make, or zs_inf_chain_par_kernel has made it)
    // (a stream of fixed blocks begins with one: a stream whose blocks do not chain for another reason -- the empty stored blocks of
    // flush markers -- is not decoded as fixed code on a guess, 3.5 ms per call for nothing)
    if (s.in_len < 3 || ((s.in[2] >> 1) & 3u) != 1u) return;
    const int64_t nbits = s.in_len * 8, r_lo = (int64_t)w.y * kFxRegionBits, r_hi = r_lo + kFxRegionBits;
    // The guess lies a region's length in front of the region (two or three blocks): a decoder that begins in the middle of a
    // symbol may take bits for an end-of-block code before it is in step -- what follows is then no header -- but the decode
    // from any position ends at a true end-of-block once it is in step, and everything behind that is true.  `pseudo`: pos is a
    // position symbols are decoded from on trust, not a header.
    int64_t pos = w.y == 0 ? 16 : (r_lo - kFxRegionBits > 16 ? r_lo - kFxRegionBits : 19);
    bool pseudo = w.y != 0;
    int n = 0;
    for (int budget = 96; budget > 0 && n < kFxEntries && pos < r_hi && pos + 3 <= nbits; budget--) {
        if (!pseudo) {
            const int64_t by = pos >> 3;
            const uint32_t two = (uint32_t)s.in[by] | (by + 1 < s.in_len ? (uint32_t)s.in[by + 1] << 8 : 0u);
            if (((two >> ((pos & 7) + 1)) & 3u) != 1u) {  // a stored or dynamic block, or bits that are no header: on as a guess
                pseudo = true, pos += 3;
                continue;
            }
        }
        const BlockOut r = inf_fixed_end(s.in, s.in_len, pseudo ? pos - 3 : pos);
        if (r.err) {
            pseudo = true, pos += 997;
            continue;
        }
        if (!pseudo && pos >= r_lo) {
            if (lane == 0) out[n] = FxEntry{pos, r.end_bit, r.out_bytes, r.bfinal, 0};
            n++;
        }
        pos = r.end_bit, pseudo = false;
    }
}

__global__ __launch_bounds__(64) void zs_inf_chain_kernel(const ParStream *ps, ParState *st, const ParCand *cands, ParBlock *blocks,
                                                          int lane_decode, int tried, const FxEntry *fxtab) {
    __shared__ ParLds L;
    const ParStream s = ps[blockIdx.x];
    ParState &ss = st[blockIdx.x];
    if (!ss.ok) return;
    if ((tried & 1) && ss.nblk >= 1) return;  // zs_inf_chain_par_kernel has done it
    const bool probing = (tried & 2) != 0;  // zs_inflate asking whether the stream's end has arrived (nothing is decoded)
    const ParCand *cd = cands + s.cand_off;
    ParBlock *bl = blocks + s.blk_off;
    const int ncand = ss.ncand;
    int64_t cur = 16, out = 0;
    int nb = 0, ok = 1, ci = 0;
    // zlib header (Inflate.cs:120-170): anything unusual goes to the sequential decoder, which reports it
    if (s.in_len < 6 || (s.in[0] & 0x0F) != 8 || (s.in[0] >> 4) > 7 || (((unsigned)s.in[0] << 8) + s.in[1]) % 31 != 0 || (s.in[1] & 0x20)) ok = 0;
    InfBits b{s.in, s.in_len, 0, 0, 0, false, L.ibuf, -1};
    // The walk is one dependency chain over the stream's blocks; its candidates come 64 at a time into the lanes'
    // registers (lane l holds candidate cbase + l) and are read with wave-uniform shuffles, so that a step is not a chain
    // of dependent loads.
    int cbase = -(1 << 30);
    int64_t m_bit = 0, m_end = 0, m_out = 0;
    int m_ok = 0, m_bfinal = 0, m_tab = -1;
    auto at = [&](int i) {  // make candidate i (wave-uniform, < ncand) resident; returns its lane
        if (i < cbase || i >= cbase + 64) {
            cbase = i;
            const int idx = i + (int)threadIdx.x;
            if (idx < ncand) {
                const ParCand q = cd[idx];
                m_bit = q.bit, m_end = q.end_bit, m_out = q.out_bytes, m_ok = q.ok, m_bfinal = q.bfinal, m_tab = q.tab;
            }
        }
        return i - cbase;
    };
    // the lane index is wave-uniform: v_readlane, not a permute through the LDS
    auto rl = [](int v, int k) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(k)); };
    auto rl64 = [&](int64_t v, int k) {
        return (int64_t)(((uint64_t)(uint32_t)rl((int)(v >> 32), k) << 32) | (uint32_t)rl((int)(uint32_t)v, k));
    };
    while (ok) {
        while (ci < ncand) {
            const int kk = at(ci);  // (first: it may reload the registers that are read)
            if (rl64(m_bit, kk) >= cur) break;
            ci++;
        }
        int64_t end, nbytes;
        int bfinal, tab = -1;
        const int k = ci < ncand ? at(ci) : 0;
        if (ci < ncand && rl64(m_bit, k) == cur && rl(m_ok, k)) {
            end = rl64(m_end, k), nbytes = rl64(m_out, k), bfinal = rl(m_bfinal, k);
            tab = lane_decode ? rl(m_tab, k) : -1;
        } else if (probing && ci < ncand && rl64(m_bit, k) == cur) {
            // the measuring pass has been through this block and did not reach its end: the input ends inside it (a whole
            // block decoded again by this one wave, to find the same, cost the caller's loop 3-6 ms per look)
            ok = 0;
            break;
        } else {
            // a block the finder does not report (stored / fixed codes): measure it here.  (Not a compressed block of a short
            // stream: measured here symbol by symbol and then decoded again by one wave it takes twice what the one-wave
            // decoder takes for the whole stream -- 60 KB of fixed-code text 21 ms against 13; the stream is left to that one.)
            inf_seek(b, cur);
            if (!probing && s.in_len < kWalkMinInput && cur + 3 <= s.in_len * 8) {
                const int64_t by = cur >> 3;
                const uint32_t two = (uint32_t)s.in[by] | (by + 1 < s.in_len ? (uint32_t)s.in[by + 1] << 8 : 0u);
                const uint32_t btype = (two >> ((cur & 7) + 1)) & 3u;
                if (btype >= 2u || (btype == 1u && (kNoFixedEnd || s.in_len < 48 * 1024))) {  // (a fixed block of a stream of several: inf_fixed_end below)
                    ok = 0;
                    break;
                }
            }
            BlockOut r;
            {
                const int64_t by = cur >> 3;
                const uint32_t two = by < s.in_len ? ((uint32_t)s.in[by] | (by + 1 < s.in_len ? (uint32_t)s.in[by + 1] << 8 : 0u)) : 0u;
                if (((two >> ((cur & 7) + 1)) & 3u) == 1u && !kNoFixedEnd) {
                    // a fixed-code block: found ahead of the walk (zs_inf_fixed_scan_kernel), or 64 lanes on it now
                    bool have = false;
                    const int64_t reg = cur / kFxRegionBits;
                    if (fxtab && reg < s.fx_regions) {
                        const FxEntry *fe = fxtab + ((int64_t)s.fx_off + reg) * kFxEntries;
                        const int lane = (int)(threadIdx.x & 63);
                        const uint64_t hit = __ballot(lane < kFxEntries && fe[lane < kFxEntries ? lane : 0].start == cur);
                        if (hit) {
                            const FxEntry e = fe[__builtin_ctzll(hit)];
                            r = BlockOut{e.out_bytes, e.end, e.bfinal, 0};
                            have = true;
                        }
                    }
                    if (!have) r = inf_fixed_end(s.in, s.in_len, cur);
                }
                else r = inf_block<0>(b, L.T, L.lens, L.ll, nullptr, (int64_t)1 << 40);
            }
            if (r.err) {
                ok = 0;
                break;
            }
            end = r.end_bit, nbytes = r.out_bytes, bfinal = r.bfinal;
        }
        if (nb >= s.max_blk || out + nbytes > s.out_cap) {
            ok = 0;
            break;
        }
        if (threadIdx.x == 0) {
            bl[nb] = {cur, out, nbytes, tab, 0};
            if (tab >= 0 && !(tab & 0x40000000)) ss.lane_blocks = 1;
        }
        nb++;
        out += nbytes;
        cur = end;
        if (bfinal) break;
    }
    if (threadIdx.x == 0) {
        ss.ok = ok;
        ss.nblk = nb;
        ss.out_len = out;
        ss.end_bit = cur;
    }
}

// ------------------------------------------------------------------ D2
__global__ __launch_bounds__(64) void zs_inf_decode_kernel(const ParStream *ps, const ParState *st, const uint2 *work, const ParBlock *blocks,
                                                           uint16_t *cells, int32_t *fail) {
    __shared__ ParLds L;
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    if (!st[w.x].ok || (int)w.y >= st[w.x].nblk) return;
    const ParBlock k = blocks[s.blk_off + w.y];
    if (k.tab >= 0) return;  // decoded by sub-blocks (zs_inf_decode_lane_kernel)
    InfBits b{s.in, s.in_len, 0, 0, 0, false, L.ibuf, -1};
    inf_seek(b, k.bit);
    BlockOut r = inf_block<1>(b, L.T, L.lens, L.ll, cells + s.cell_off + k.out_off, k.out_bytes, k.out_off);
    if (threadIdx.x == 0 && (r.err || r.out_bytes != k.out_bytes)) fail[w.x] = 1;
}

// ------------------------------------------------------------------ D2, lane form
// The wave decoder spends ~130 instructions of a whole wave on every symbol, and a CU issues about one per cycle: the
// pass is bound by instruction issue with 63 of 64 lanes doing nothing useful.  Here every lane decodes something of
// its own: the measure pass left a checkpoint at the first symbol of every subsequence, so a block falls into
// sub-blocks that decode independently -- kDecBlocks blocks per workgroup, kDecSubLanes lanes per block taking its
// sub-blocks in turn, the literal/length tables in LDS (copied from the measure pass), the other tables in their HBM slab.
// Cells: a byte; 0x8000 | i = byte i of the 32 KiB before the *block* (as the wave decoder writes them); and, new,
// 0x100 + (d - 1) = the cell d positions before the start of this lane's *sub-block* (d <= 32512; a source further
// back inside the block fails the stream over to the sequential decoder -- zlib's MAX_DIST is 32506).  Copies of
// markers stay markers; zs_inf_cellflat_kernel then follows the sub-block markers, so that the window and resolve
// passes see the wave decoder's two kinds only.
constexpr int kSubMarkBase = 0x100, kSubMarkMax = 0x8000 - kSubMarkBase;  // 32512 distances
#ifndef ZS_DEC_SUBLANES
#define ZS_DEC_SUBLANES 64
#endif
constexpr int kDecSubLanes = ZS_DEC_SUBLANES;      // lanes per block
constexpr int kDecBlocks = 64 / kDecSubLanes;      // blocks per workgroup
constexpr int kDecCopy = 8;    // cells of a match copied per round trip
__global__ __launch_bounds__(64) void zs_inf_decode_lane_kernel(const ParStream *ps, const ParState *st, const uint2 *work, int nwork,
                                                                const ParBlock *blocks, const LaneTabs *tabs, uint16_t *cells,
                                                                int32_t *fail) {
    // the block's decode tables, all of them: a code longer than the primary index is rare for a lane but not for a wave,
    // and its canonical walk is up to 15 dependent lookups
    __shared__ __attribute__((aligned(16))) InfTables tab_s[kDecBlocks];
    const int grp = threadIdx.x / kDecSubLanes, sub0 = threadIdx.x % kDecSubLanes;
    const int wi = blockIdx.x * kDecBlocks + grp;
    bool live = wi < nwork;
    uint2 w = make_uint2(0, 0);
    if (live) w = work[wi];
    ParStream s = ps[w.x];
    live = live && st[w.x].ok && (int)w.y < st[w.x].nblk;
    ParBlock k = {0, 0, 0, -1, 0};
    if (live) k = blocks[s.blk_off + w.y];
    live = live && k.tab >= 0 && !(k.tab & 0x40000000);  // (kTabTok: the block has tokens, zs_inf_expand_kernel's)
    const LaneTabs *T = tabs + (live ? k.tab : 0);
    if (live) {
        const uint4 *src = (const uint4 *)T;  // LaneTabs starts with an InfTables image: 16-byte aligned
        uint4 *dst = (uint4 *)&tab_s[grp];
        for (int i = sub0; i < (int)(sizeof(InfTables) / 16); i += kDecSubLanes) dst[i] = src[i];
    }
    __syncthreads();
    if (!live) return;
    const InfTables &L = tab_s[grp];
    const uint16_t *lit = L.lit;
    uint16_t *o16 = cells + s.cell_off + k.out_off;
    const int nsub = T->nsub;
    bool bad = false;
    for (int sub = sub0; sub < nsub && !bad; sub += kDecSubLanes) {
        const int64_t bit0 = k.bit + T->ck_bit[sub];
        const int S = (int)T->ck_out[sub], E = (int)T->ck_out[sub + 1];  // block-relative, < 2^31 (measure pass)
        const int64_t bit1 = k.bit + T->ck_bit[sub + 1];  // the next sub-block's first symbol, or the bit after END_BLOCK
        LaneBits b{(const __attribute__((address_space(1))) uint8_t *)(uintptr_t)s.in, s.in_len, 0, 0, 0, false};
        b.seek(bit0);
        int pos = S;
        bool eob = false, fin = false;
        // A lane is either decoding its next symbol or copying the match it decoded, kDecCopy cells per trip: a long match
        // keeps its own lane busy for several trips while the other lanes decode on (a wave pays for a trip once, however
        // many of its lanes take it; as a loop inside the match it ran as often as the wave's longest match needed).
        int cp_left = 0, cp_sp0 = 0, cp_off = 0, cp_dist = 1;
        const int sub_mark = kSubMarkBase - 1 + S;  // sub-block marker of source position sp: sub_mark - sp
        // Literals wait in two registers, up to 8 cells for positions [pos - lb_n, pos), and leave as one 16-byte store:
        // a 2-byte store per literal had the L2 evict sectors it had a few bytes of and write them to HBM again and again
        // (14.8 GB written per GiB of output, PMC).  A store of fewer than 8 waiting cells overshoots with zeros, like the
        // match stores below: cells the lane itself writes next.  They are flushed before a match, which may read them.
        uint64_t lb_lo = 0, lb_hi = 0;
        int lb_n = 0;
        auto flush_lits = [&]() {
            if (lb_n == 0) return;
            const int p0 = pos - lb_n;
            if (p0 + 8 <= E) {
                store_u4_a2(o16 + p0, (uint32_t)lb_lo, (uint32_t)(lb_lo >> 32), (uint32_t)lb_hi, (uint32_t)(lb_hi >> 32));
            } else {
                for (int u = 0; u < lb_n; u++) o16[p0 + u] = (uint16_t)((u < 4 ? lb_lo >> (16 * u) : lb_hi >> (16 * (u - 4))) & 0xFFFFu);
            }
            lb_lo = lb_hi = 0, lb_n = 0;
        };
        while (!bad && (!fin || cp_left > 0)) {
            if (cp_left == 0) {
                if (b.tell() >= bit1) {
                    flush_lits();
                    fin = true;
                    continue;
                }
                b.fill();
                int sym, clen;
                {
                    const uint16_t e = lit[b.peek(kInfLitBits)];
                    if (e != kInfEsc) sym = e >> 4, clen = e & 15;
                    else sym = lane_slow(b, L.lcount, L.lsym, clen);
                }
                if (sym < 0 || clen > b.cnt) {
                    bad = true;
                    break;
                }
                b.drop(clen);
                if (sym < 256) {
                    if (pos >= E) {
                        bad = true;
                        break;
                    }
                    const uint64_t v = (uint64_t)(uint32_t)sym << (16 * (lb_n & 3));
                    lb_lo |= lb_n < 4 ? v : 0, lb_hi |= lb_n < 4 ? 0 : v;
                    lb_n++, pos++;
                    if (lb_n == 8) flush_lits();
                    continue;
                }
                flush_lits();
                if (sym == 256) {
                    eob = true;
                    fin = true;
                    continue;
                }
                sym -= 257;
                if (sym >= 29) {
                    bad = true;
                    break;
                }
                const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)b.take(extra_lbits(sym));
                b.fill();
                int ds, dl;
                {
                    const uint16_t e = L.dist[b.peek(kInfDistBits)];
                    if (e != kInfEsc) ds = e >> 4, dl = e & 15;
                    else ds = lane_slow(b, L.dcount, L.dsym, dl);
                }
                if (ds < 0 || ds >= 30 || dl > b.cnt) {
                    bad = true;
                    break;
                }
                b.drop(dl);
                const int dist = base_dist(ds) + 1 + (int)b.take(extra_dbits(ds));
                const int sp0 = pos - dist;  // block-relative source of the first byte
                // a source cell inside the block but further before S than a sub-block marker can say fails the stream over
                const int span = dist < mlen ? dist : mlen, lo = sp0 < 0 ? 0 : sp0;  // source cells sp0 .. sp0 + span - 1
                if (b.bad || pos + mlen > E || k.out_off + sp0 < 0 || (lo < S && lo < sp0 + span && S - lo > kSubMarkMax)) {
                    bad = true;
                    break;
                }
                cp_left = mlen, cp_sp0 = sp0, cp_off = 0, cp_dist = dist;
                // sources are older than this match (offset i mod dist); the lane's own cells are read back from L2 once
                // its stores have landed
                if (sp0 + span > S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (cp_left > 0) {
                uint32_t v[kDecCopy];
                const int first = cp_sp0 + cp_off;
                if (cp_off + kDecCopy <= cp_dist && first >= S) {
                    // 8 consecutive cells of the lane's own sub-block: one 16-byte load past L1 (what lies beyond the match's
                    // last source cell is not used)
                    u32x4 r;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(o16 + first) : "memory");
#pragma unroll
                    for (int u = 0; u < kDecCopy; u++) v[u] = (r[u >> 1] >> (16 * (u & 1))) & 0xFFFFu;
                    cp_off = cp_off + kDecCopy == cp_dist ? 0 : cp_off + kDecCopy;
                } else {
                    // cell by cell (a source that wraps inside the trip, or straddles S).  Before the block: marker of byte
                    // 32768 + sp of the window (0x8000 | (32768 + sp) is sp's low 16 bits); before the sub-block: its
                    // distance from S; else the cell itself, read past L1 -- the loads are issued together and waited for once
                    uint32_t ld[kDecCopy];
                    bool own[kDecCopy];
#pragma unroll
                    for (int u = 0; u < kDecCopy; u++) {
                        const int sp = cp_sp0 + cp_off;
                        v[u] = sp < 0 ? (uint32_t)sp : (uint32_t)(sub_mark - sp);
                        own[u] = u < cp_left && sp >= S;
                        ld[u] = 0;
                        if (own[u]) asm volatile("global_load_ushort %0, %1, off sc1" : "=v"(ld[u]) : "v"(o16 + sp) : "memory");
                        cp_off = cp_off + 1 == cp_dist ? 0 : cp_off + 1;
                    }
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7])
                                 :
                                 : "memory");
#pragma unroll
                    for (int u = 0; u < kDecCopy; u++) v[u] = own[u] ? ld[u] : v[u];
                }
                // One 16-byte store (at 2-byte alignment) for the trip's 8 cells while that stays inside the lane's own
                // sub-block: the cells beyond the match's end are the lane's to write anyway, and it writes them (again)
                // before any of its later matches can read them.  The pass is bound by the count of vector-memory
                // instructions, whatever their width or their active lanes.
                if (pos + kDecCopy <= E) {
                    static_assert(kDecCopy == 8, "one 16-byte store per trip");
                    store_u4_a2(o16 + pos, (v[0] & 0xFFFFu) | (v[1] << 16), (v[2] & 0xFFFFu) | (v[3] << 16),
                                (v[4] & 0xFFFFu) | (v[5] << 16), (v[6] & 0xFFFFu) | (v[7] << 16));
                } else {
#pragma unroll
                    for (int u = 0; u < kDecCopy; u++)
                        if (u < cp_left) o16[pos + u] = (uint16_t)v[u];
                }
                const int n = cp_left < kDecCopy ? cp_left : kDecCopy;
                pos += n;
                cp_left -= n;
            }
        }
        // a sub-block ends where the next checkpoint says (the last one at END_BLOCK)
        const bool last = sub + 1 == nsub;
        if (bad || b.bad || pos != E || eob != last || b.tell() != bit1) bad = true;
    }
    if (bad) fail[w.x] = 1;
}

// Follow the sub-block markers of the lane decoder: every such cell becomes a byte or a block marker.  In place: a cell
// read while another thread rewrites it holds either form, and both say the same thing.
__global__ __launch_bounds__(256) void zs_inf_cellflat_kernel(const ParStream *ps, const ParState *st, const uint2 *work,
                                                              const ParBlock *blocks, const LaneTabs *tabs, uint16_t *cells) {
    __shared__ uint32_t ck[kCkMax + 1];
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    if (!st[w.x].ok || (int)w.y >= st[w.x].nblk) return;
    const ParBlock k = blocks[s.blk_off + w.y];
    if (k.tab < 0 || (k.tab & 0x40000000)) return;
    const LaneTabs &T = tabs[k.tab];
    const int nsub = T.nsub;
    for (int i = threadIdx.x; i <= nsub; i += 256) ck[i] = T.ck_out[i];
    __syncthreads();
    uint16_t *cl = cells + s.cell_off + k.out_off;
    auto sub_index = [&](uint32_t p) {  // the sub-block that holds block-relative position p
        int lo = 0, hi = nsub - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ck[mid] <= p) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    };
    // (every hop of a chase asks for a sub-block's start: eight dependent LDS reads as a bisection; a table with the
    // sub-block of every 512th position leaves a step or two)
    constexpr int kHintShift = 9, kHintMax = 512;
    __shared__ uint16_t hint[kHintMax];
    const bool hinted = (k.out_bytes >> kHintShift) < kHintMax;
    if (hinted)
        for (int g = threadIdx.x; g <= (int)(k.out_bytes >> kHintShift); g += 256) hint[g] = (uint16_t)sub_index((uint32_t)g << kHintShift);
    __syncthreads();
    auto sub_start = [&](uint32_t p) {
        if (!hinted) return ck[sub_index(p)];
        int lo = hint[p >> kHintShift];
        while (lo + 1 < nsub && ck[lo + 1] <= p) lo++;
        return ck[lo];
    };
    // (One cell per lane and trip.  Round 3 tried four -- one 8-byte load, the pass being bound by the count of its 2-byte
    // loads: with a lane chasing its own four cells one after the other the decode stage was 0.7 ms per GiB slower, with
    // the markers of 1024 cells queued in LDS and chased one per lane 0.4 ms slower.  The chases -- dependent loads through
    // L2, a binary search over the sub-block starts in front of each -- are what the pass takes its time for, and this form
    // starts every one of them as early as it can.)
    for (int64_t i = threadIdx.x; i < k.out_bytes; i += 256) {
        uint16_t c = cl[i];
        if (c < kSubMarkBase || c >= 0x8000) continue;
        uint32_t p = (uint32_t)i;
        do {
            const uint32_t S = sub_start(p), d = (uint32_t)(c - kSubMarkBase + 1);
            if (d > S) {  // not something the lane decoder writes: its stream has failed over already
                c = 0;
                break;
            }
            p = S - d;
            c = cell_load(cl + p);
        } while (c >= kSubMarkBase && c < 0x8000);
        cl[i] = c;
    }
}

// ------------------------------------------------------------------ W
// win[k] = the 32 KiB of resolved output that end with block k.  Copies of markers are markers, so on text a byte's chain
// of sources runs back through block after block: the windows are the one place where that chain is cut, and the pass is
// serial along a stream -- a step has to be short.  It is bound by instruction issue (16 waves on one CU), so the LDS
// image of a window is laid out such that a cell *is* its own gather address: a 256-byte identity table at offset 0 (a
// literal cell b reads b), the window at offset 0x8000 (a marker 0x8000 | i reads window byte i); two such 64 KiB images,
// the previous window and the one being built.  A thread owns runs of 4 consecutive window bytes: cells requested a block
// ahead (two aligned 8-byte loads and a funnel shift), four byte gathers, one LDS dword and one HBM dword written.
//
// The serial pass is a composition of maps ("byte i of this window is a literal, or byte j of the window before"), and
// maps compose associatively.  So a stream's blocks are cut into groups and the pass runs in three launches:
//   A  zs_inf_window_kernel over (stream, group): group 0 as above, from the zero window before the stream; every other
//      group in *map mode*: the same steps over 16-bit entries, starting from the identity map, which leaves the group's
//      map M_g: byte i of the window after the group's last block = a literal, or byte j of the window before the group;
//   B  zs_inf_winchain_kernel, one workgroup per stream: the window before group g = M_(g-1) applied to the window before
//      group g - 1 (32 Ki gathers per group);
//   C  zs_inf_window_kernel again over the groups >= 1, now in byte mode from their real entry windows.
// Twice the work, 1 / groups of the length of the chain: a lone 64 MiB stream no longer waits 4.3 ms for one workgroup.
constexpr int kWinImage = 0x10000;
constexpr int kWinLds = 2 * kWinImage;
constexpr int kWinMapImage = 2 * (256 + kWSize);   // map mode: 16-bit entries, identity part + window part
static_assert(2 * kWinMapImage <= 160 * 1024 - 1024, "two map images fit the LDS");
constexpr int kWinMapLds = 2 * kWinMapImage;
struct WinGroup {
    int32_t stream, first, count;  // blocks [first, first + count) of the stream
    int32_t mode;                  // 0: bytes from the zero window; 1: map from the identity; 2: bytes from entries[slot]
    int32_t slot, pad_;            // index of the group's map / entry window
};
template <bool MAP>
__device__ __forceinline__ void inf_window_pass(uint8_t *wl, const ParStream &s, const ParState &ss, const WinGroup &g, const ParBlock *blocks,
                                                const uint16_t *cells, uint8_t *windows, const uint8_t *entry, uint16_t *map_out) {
    const uint16_t *cl = cells + s.cell_off;
    uint8_t *win = windows + (int64_t)ss.win_off * kWSize;
    uint16_t *wl16 = (uint16_t *)wl;
    if (MAP) {
        // entry i of the window part = "byte i of the window before the group"; literal cells read themselves
        for (int i = threadIdx.x; i < kWSize; i += 1024) wl16[256 + i] = (uint16_t)(0x8000u | i);
        if (threadIdx.x < 256) wl16[threadIdx.x] = (uint16_t)threadIdx.x, wl16[kWinMapImage / 2 + threadIdx.x] = (uint16_t)threadIdx.x;
    } else {
        for (int i = threadIdx.x; i < kWSize / 4; i += 1024) ((uint32_t *)(wl + 0x8000))[i] = entry ? ((const uint32_t *)entry)[i] : 0u;
        if (threadIdx.x < 256) wl[threadIdx.x] = (uint8_t)threadIdx.x, wl[kWinImage + threadIdx.x] = (uint8_t)threadIdx.x;
    }
    __syncthreads();
    constexpr int kRuns = kWSize / 4096;  // runs of 4 window bytes per thread
    int cur = 1;
    uint2 c[kRuns], r0[kRuns], r1[kRuns];
    ParBlock bk = {0, 0, 0, -1, 0}, bn = bk;
    // The 4 cells of window bytes [i, i + 4) of a block's window.  The window's first cell is not 8-byte aligned in general:
    // two aligned 8-byte loads (request: unconditional, from clamped addresses, so that the loads of a step are all in
    // flight together, a block ahead of their use) and a funnel shift by the block's wave-uniform misalignment (finish).
    // Where the window reaches back beyond the block the "cell" is the marker of the same byte in the previous window,
    // 0x8000 | (i + out_bytes).
    auto request = [&](const ParBlock &q) {
        const int64_t org = q.out_off + q.out_bytes - kWSize;
        const int off = (int)(org & 3);
#pragma unroll
        for (int j = 0; j < kRuns; j++) {
            const int64_t a = org - off + (threadIdx.x + j * 1024) * 4;  // a multiple of 4 cells (cell_off is one of 64, the array 16-byte aligned)
            // cells before position 0 do not exist; what is loaded in their place is replaced in finish()
            r0[j] = *(const uint2 *)(cl + (a < 0 ? 0 : a));
            r1[j] = *(const uint2 *)(cl + (a + 4 < 0 ? 0 : a + 4));
        }
    };
    auto finish = [&](const ParBlock &q) {
        const int64_t org = q.out_off + q.out_bytes - kWSize;
        const int off = (int)(org & 3), ws = off >> 1, bs = (off & 1) * 16;
#pragma unroll
        for (int j = 0; j < kRuns; j++) {
            const uint32_t w0 = ws ? r0[j].y : r0[j].x, w1 = ws ? r1[j].x : r0[j].y, w2 = ws ? r1[j].y : r1[j].x;
            c[j] = make_uint2(__builtin_amdgcn_alignbit(w1, w0, bs), __builtin_amdgcn_alignbit(w2, w1, bs));
        }
        if (q.out_bytes < kWSize) {  // wave-uniform
#pragma unroll
            for (int j = 0; j < kRuns; j++) {
                const int i = (threadIdx.x + j * 1024) * 4;
                uint32_t v[4] = {c[j].x & 0xFFFFu, c[j].x >> 16, c[j].y & 0xFFFFu, c[j].y >> 16};
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = org + i + u >= q.out_off ? v[u] : 0x8000u | (uint32_t)(i + u + q.out_bytes);
                c[j] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
            }
        }
    };
    if (g.count > 0) {
        bn = blocks[s.blk_off + g.first];
        request(bn);
    }
    for (int k = 0; k < g.count; k++) {
        bk = bn;
        finish(bk);
        if (k + 1 < g.count) {
            bn = blocks[s.blk_off + g.first + k + 1];
            request(bn);
        }
        if (MAP) {
            const uint16_t *pw = wl16 + (cur ^ 1) * (kWinMapImage / 2);
            uint16_t *cw = wl16 + cur * (kWinMapImage / 2) + 256;
            auto look = [pw](uint32_t cell) -> uint32_t { return pw[(cell & 0x8000u) ? 256u + (cell & 0x7FFFu) : cell]; };
#pragma unroll
            for (int j = 0; j < kRuns; j++) {
                const int i = (threadIdx.x + j * 1024) * 4;
                const uint32_t b0 = look(c[j].x & 0xFFFFu), b1 = look(c[j].x >> 16), b2 = look(c[j].y & 0xFFFFu), b3 = look(c[j].y >> 16);
                *(uint2 *)(cw + i) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
            }
        } else {
            const uint8_t *pw = wl + (cur ^ 1) * kWinImage;
            uint8_t *cw = wl + cur * kWinImage + 0x8000;
            uint32_t *dst = (uint32_t *)(win + (int64_t)(g.first + k) * kWSize);
#pragma unroll
            for (int j = 0; j < kRuns; j++) {
                const int i = (threadIdx.x + j * 1024) * 4;
                const uint32_t b0 = pw[c[j].x & 0xFFFFu], b1 = pw[c[j].x >> 16], b2 = pw[c[j].y & 0xFFFFu], b3 = pw[c[j].y >> 16];
                const uint32_t word = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
                *(uint32_t *)(cw + i) = word;
                dst[i >> 2] = word;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (MAP) {
        // the group's map: what the last step left (the identity when the group is empty)
        const uint16_t *fin = wl16 + (cur ^ 1) * (kWinMapImage / 2) + 256;
        for (int i = threadIdx.x; i < kWSize / 2; i += 1024) ((uint32_t *)map_out)[i] = ((const uint32_t *)fin)[i];
    }
}
__global__ __launch_bounds__(1024) void zs_inf_window_kernel(const ParStream *ps, const ParState *st, const WinGroup *groups, const ParBlock *blocks,
                                                             const uint16_t *cells, uint8_t *windows, uint16_t *maps, const uint8_t *entries,
                                                             int pass) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wl[];
    const WinGroup g = groups[blockIdx.x];
    const ParStream s = ps[g.stream];
    const ParState ss = st[g.stream];
    if (!ss.ok) return;
    // pass 0 (launch A): group 0 in byte mode, the others in map mode; pass 1 (launch C): the others in byte mode
    if (pass == 0) {
        if (g.mode == 0) inf_window_pass<false>(wl, s, ss, g, blocks, cells, windows, nullptr, nullptr);
        else inf_window_pass<true>(wl, s, ss, g, blocks, cells, windows, nullptr, maps + (int64_t)g.slot * kWSize);
    } else if (g.mode != 0) {
        inf_window_pass<false>(wl, s, ss, g, blocks, cells, windows, entries + (int64_t)g.slot * kWSize, nullptr);
    }
}
// Launch B: the window before each group >= 1 of a stream, one workgroup per stream.  `sg` lists, per stream, the index
// of its first WinGroup and the number of its groups.
__global__ __launch_bounds__(1024) void zs_inf_winchain_kernel(const ParState *st, const WinGroup *groups, const int2 *sg, const uint8_t *windows,
                                                               const uint16_t *maps, uint8_t *entries) {
    __shared__ uint8_t cur[2][kWSize];
    const int2 q = sg[blockIdx.x];
    if (q.y < 2) return;
    const WinGroup g0 = groups[q.x];
    const ParState ss = st[g0.stream];
    if (!ss.ok) return;
    // the window after group 0 = the window of its last block (written by launch A); zero when group 0 has no block
    const uint8_t *w0 = windows + ((int64_t)ss.win_off + g0.first + g0.count - 1) * kWSize;
    for (int i = threadIdx.x; i < kWSize / 4; i += 1024) ((uint32_t *)cur[0])[i] = g0.count > 0 ? ((const uint32_t *)w0)[i] : 0u;
    __syncthreads();
    int b = 0;
    for (int k = 1; k < q.y; k++) {
        const WinGroup g = groups[q.x + k];
        uint8_t *e = entries + (int64_t)g.slot * kWSize;
        const uint16_t *m = maps + (int64_t)g.slot * kWSize;
        for (int i = threadIdx.x; i < kWSize / 4; i += 1024) ((uint32_t *)e)[i] = ((const uint32_t *)cur[b])[i];
        if (k + 1 < q.y) {
            for (int i = threadIdx.x * 4; i < kWSize; i += 4096) {
                const uint2 mv = *(const uint2 *)(m + i);
                const uint32_t v[4] = {mv.x & 0xFFFFu, mv.x >> 16, mv.y & 0xFFFFu, mv.y >> 16};
                uint32_t word = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) word |= (uint32_t)((v[u] & 0x8000u) ? cur[b][v[u] & 0x7FFFu] : (uint8_t)v[u]) << (8 * u);
                *(uint32_t *)(cur[b ^ 1] + i) = word;
            }
        }
        __syncthreads();
        b ^= 1;
    }
}

// ------------------------------------------------------------------ R
// One workgroup per block: the 32 KiB window before the block staged in LDS once (16-byte loads; the gathers were round trips
// to L1 / L2 for every marker cell), then four cells per lane and trip -- one 8-byte load, four LDS byte gathers, one 4-byte
// store, the lanes of a wave on consecutive cells (the one-cell form was bound by the count of its memory instructions:
// 2-byte loads and 1-byte stores, 2.2 ms per GiB).  The cells and the output share their phase mod 4 when the caller's
// buffer is 4-byte aligned; the few cells in front of the first aligned group and behind the last go one by one.
__global__ __launch_bounds__(256) void zs_inf_resolve_kernel(const ParStream *ps, const ParState *st, const uint2 *work, const ParBlock *blocks,
                                                             const uint16_t *cells, const uint8_t *windows) {
    __shared__ __attribute__((aligned(16))) uint8_t win[kWSize];
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    if (!st[w.x].ok || (int)w.y >= st[w.x].nblk) return;
    const ParBlock k = blocks[s.blk_off + w.y];
    const uint8_t *pw = w.y ? windows + ((int64_t)st[w.x].win_off + w.y - 1) * kWSize : nullptr;
    const uint16_t *cl = cells + s.cell_off + k.out_off;
    uint8_t *o = s.out + k.out_off;
    const int64_t n = k.out_bytes;
    if (pw) {
        for (int i = threadIdx.x * 16; i < kWSize; i += 256 * 16) *(uint4 *)(win + i) = *(const uint4 *)(pw + i);
    } else {
        for (int i = threadIdx.x * 16; i < kWSize; i += 256 * 16) *(uint4 *)(win + i) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    auto one = [&](int64_t i) {
        const uint16_t c = cl[i];
        o[i] = (c & 0x8000) ? win[c & 0x7FFF] : (uint8_t)c;
    };
    // cells are 2 bytes: cl + i is 8-byte aligned where the output address o + i is 4-byte aligned, if the two agree mod 4
    const int64_t head = (int64_t)((4 - ((uintptr_t)o & 3)) & 3);
    const bool same_phase = (((uintptr_t)(cl + head)) & 7) == 0;
    if (!same_phase) {
        for (int64_t i = threadIdx.x; i < n; i += 256) one(i);
        return;
    }
    const int64_t body = n > head ? ((n - head) & ~3LL) : 0;
    if ((int64_t)threadIdx.x < head && (int64_t)threadIdx.x < n) one(threadIdx.x);
    for (int64_t i = head + (int64_t)threadIdx.x * 4; i < head + body; i += 256 * 4) {
        const uint2 v = *(const uint2 *)(cl + i);
        const uint32_t c0 = v.x & 0xFFFFu, c1 = v.x >> 16, c2 = v.y & 0xFFFFu, c3 = v.y >> 16;
        const uint32_t b0 = (c0 & 0x8000u) ? win[c0 & 0x7FFFu] : (c0 & 0xFFu);
        const uint32_t b1 = (c1 & 0x8000u) ? win[c1 & 0x7FFFu] : (c1 & 0xFFu);
        const uint32_t b2 = (c2 & 0x8000u) ? win[c2 & 0x7FFFu] : (c2 & 0xFFu);
        const uint32_t b3 = (c3 & 0x8000u) ? win[c3 & 0x7FFFu] : (c3 & 0xFFu);
        *(uint32_t *)(o + i) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
    for (int64_t i = head + body + threadIdx.x; i < n; i += 256) one(i);
}

}  // namespace zs
