// zs_inflate_tok.hip -- block-parallel inflate, every symbol decoded ONCE (round 5).
//
// Until round 4 a block was decoded twice: zs_inf_measure_sync_kernel decoded it without output to learn where its
// subsequences begin (bit and output position), zs_inf_decode_lane_kernel decoded it again into cells, one lane per
// subsequence, and zs_inf_cellflat_kernel chased the markers the lanes had to leave for sources outside their own
// subsequence (13.3 ms of a 20.4 ms call per GiB; 38 GB of HBM traffic for 1.5 GB of stream and output).
//
//   D1' measure + tokens  (zs_inf_measure_tok_kernel)  the measuring decode writes what it decodes: one 32-bit TOKEN per
//        symbol -- a literal's byte, or a match's (length, distance) -- into a slab of the lane's own.  Tokens do not depend
//        on output positions, so they can be written before the prefix sums exist.  A lane that began at a wrong bit falls
//        into step with the true symbol boundaries after a few symbols (the property the kernel rests on); when its
//        predecessor's proven exit later says where it should have begun, the lane no longer decodes its whole subsequence
//        again: it decodes from the right bit only until it meets a symbol boundary of its first decode (the first
//        kTokBnd of them are kept in LDS) and its token list becomes "the few new tokens, then the old list from that
//        boundary on".  Measured before: 2-3 whole passes per subsequence; now one pass and a short prefix.
//   X  expand  (zs_inf_expand_kernel)  one workgroup per block turns the block's tokens into flattened cells -- a byte, or
//        0x8000 | i = byte i of the 32 KiB before the block -- position by position, all lanes on consecutive cells, whole
//        lines stored: a tile of up to 8192 cells per step, the token that owns a cell from a bitmap of token starts and a
//        prefix popcount, a match cell's source taken from an LDS ring of the block's last 32 Ki flattened cells (one
//        gather, no chase: what lies before the tile is flat already), sources inside the tile by pointer jumping in LDS.
//        The lane decoder's per-lane stores, its read-back of its own cells through L2 and the whole marker-chasing pass are
//        gone.
//
// The window and resolve passes (zs_inflate_par.hip, W and R) read the same cells as before.  A block whose tokens do not
// fit their slab (fewer than 2 bits per symbol, a decode that runs past the next candidate, more than kCkMax
// subsequences) is decoded by the wave decoder (zs_inf_decode_kernel), as blocks without candidates always were.
#include <hip/hip_runtime.h>

namespace zs {

typedef uint32_t sb_u32x4 __attribute__((ext_vector_type(4)));
typedef sb_u32x4 sb_u32x4_a4 __attribute__((aligned(4)));  // 16 bytes at 4-byte alignment: one global_load_dwordx4 / global_store_dwordx4
constexpr int kTabTok = 0x40000000;  // ParCand::tab / ParBlock::tab: the block has tokens (tabs[tab & ~kTabTok]); without the bit: checkpoints for the lane decoder
constexpr int kTokBnd = 24;   // symbol boundaries of a lane's decode kept for a re-entry
constexpr int kTokPre = 48;   // tokens a re-entry may decode before it has to meet the decode it replaces (a multiple of 4)

// token: bit 31 clear -- literal, the byte in bits 0..7; set -- match, length - 3 in bits 0..7, distance - 1 in bits 8..22
__host__ __device__ inline int tok_sub_bits(int64_t span_bits) {  // subsequence length for a block of at most span_bits bits
    const int64_t per = ((span_bits + 63) / 64 + 63) & ~(int64_t)63;
    return per < kSubMinBits ? kSubMinBits : per > kSubMaxBits ? kSubMaxBits : (int)per;
}
// room for 4 bits per symbol and up: the tokens are kept twice (the lanes' slabs, then the block's list in order), and a block
// with a subsequence of shorter symbols -- runs of 258-byte matches -- is decoded again by the lane decoder
__host__ __device__ inline int tok_main_cap(int S) { return ((S >> 2) + 8 + 3) & ~3; }
__host__ __device__ inline int tok_unit(int S) { return kTokPre + tok_main_cap(S); }

struct TokTabs {
    int64_t tok_off;              // the block's tokens, in order, in the list array (the same offset as its slabs in the slab array)
    uint32_t ntok, pad_;
};

// four tokens at 4-byte alignment, one instruction each way
__device__ __forceinline__ void tok_copy4(uint32_t *dst, const uint32_t *src) {
    const sb_u32x4 v = *(const sb_u32x4_a4 *)src;
    *(sb_u32x4_a4 *)dst = v;
}
// a lane's n tokens from src to dst: sixteen at a time with the loads ahead of the stores, then four at a time, then singly
// (never a token more than n: behind them the next lane's begin)
__device__ __forceinline__ void tok_copy(uint32_t *dst, const uint32_t *src, int n) {
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        const sb_u32x4 a = *(const sb_u32x4_a4 *)(src + i), b = *(const sb_u32x4_a4 *)(src + i + 4), c = *(const sb_u32x4_a4 *)(src + i + 8),
                       d = *(const sb_u32x4_a4 *)(src + i + 12);
        *(sb_u32x4_a4 *)(dst + i) = a, *(sb_u32x4_a4 *)(dst + i + 4) = b, *(sb_u32x4_a4 *)(dst + i + 8) = c, *(sb_u32x4_a4 *)(dst + i + 12) = d;
    }
    for (; i + 4 <= n; i += 4) tok_copy4(dst + i, src + i);
    for (; i < n; i++) dst[i] = src[i];
}

// Slabs: candidate i of a stream may decode up to the next candidate's header (its most likely end), in subsequences of S
// bits, each with room for S / 4 tokens and a re-entry's kTokPre.  One workgroup per stream lays the stream's candidates out
// (offsets inside the stream's share), one more adds the shares up (a batch of a thousand streams laid out by one workgroup,
// stream after stream, was 3 ms of barriers).
__global__ __launch_bounds__(1024) void zs_inf_tokalloc_kernel(const ParStream *ps, ParState *st, ParCand *cands) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, si = blockIdx.x;
    if (tid == 0) run = 0;
    __syncthreads();
    const ParStream s = ps[si];
    const int n = st[si].ok ? st[si].ncand : 0;
    const int64_t nbits = s.in_len * 8;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        int64_t need = 0;
        if (i < n) {
            const int64_t bit = cands[s.cand_off + i].bit;
            int64_t hint = i + 1 < n ? cands[s.cand_off + i + 1].bit : nbits;
            if (hint <= bit || hint > nbits) hint = nbits;
            const int S = tok_sub_bits(hint - bit);
            const int64_t nsub = (hint - bit + S - 1) / S + 1;
            need = nsub <= kCkMax ? nsub * tok_unit(S) : 0;
        }
        int64_t v = need;
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int64_t base = run, tot = 0;
        for (int k = 0; k < 16; k++) {
            if (k < wave) base += wsum[k];
            tot += wsum[k];
        }
        if (i < n) {
            ParCand &c = cands[s.cand_off + i];
            c.tok_off = base + v - need;  // (inside the stream's share: the measuring kernel adds ParState::tok_base)
            c.tok_cap = (int32_t)need;
        }
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) st[si].tok_need = run;
}
__global__ __launch_bounds__(1024) void zs_inf_tokbase_kernel(ParState *st, int nstreams, int64_t *total) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nstreams; c0 += 1024) {
        const int i = c0 + tid;
        const int64_t need = i < nstreams ? st[i].tok_need : 0;
        int64_t v = need;
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int64_t base = run, tot = 0;
        for (int k = 0; k < 16; k++) {
            if (k < wave) base += wsum[k];
            tot += wsum[k];
        }
        if (i < nstreams) st[i].tok_base = base + v - need;
        __syncthreads();
        if (tid == 0) run += tot;
        __syncthreads();
    }
    if (tid == 0) *total = run;
}

// A lane's reader of the compressed bits for the measuring decode.  The plain reader (LaneBits) loads 8 bytes whenever its
// lane runs low: in a wave whose lanes all decode their own subsequence SOME lane runs low in nearly every symbol, so the wave
// waited for a load from L2 / HBM per symbol -- the measuring pass was bound by that latency, not by its instructions.  Here the
// dwords a lane is going to need come ahead of time and for the whole wave at once: every kSbEvery symbols each lane asks for
// the kSbWin dwords from where it stands (four 16-byte loads, nobody waits), and the answer to the request before lands in the
// lane's column of an LDS window, from which the lane tops its 64-bit buffer up one dword at a time.  A symbol takes at most
// two dwords (<= 20 bits of literal/length code and extra bits, <= 28 of distance), so the window a lane reads from -- asked
// for at most 2 kSbEvery symbols ago -- covers what it reads: 2 * 2 * kSbEvery = kSbWin dwords.
constexpr int kSbWin = 16, kSbEvery = 4;
// The last dwords of a stream are read through a copy with zeros behind it (zs_inf_tails_kernel), so that a request never
// reaches behind the caller's buffer and needs no branch (a load under a condition is waited for on the spot).
constexpr int kSbTail = 32, kSbTailBuf = 64;  // dwords of the stream in the copy; dwords of the copy
struct SyncBits {
    const __attribute__((address_space(1))) uint32_t *base;  // the stream's first byte lies in base[0]
    const __attribute__((address_space(1))) uint32_t *tail;  // copy of dwords [d_last + 1 - kSbTail, d_last], then zeros
    uint32_t *win;     // LDS, the lane's column: win[j * 64] = dword wbase + j
    int d_last;        // the last dword with stream bytes in it
    int skew;          // bits of base[0] in front of the stream's first byte
    int dw, wbase;     // next dword to take; first dword of the window
    // the position: bit `off` (< 32 between symbols) of the 64 bits hi:lo = dwords dw - 2 and dw - 1; everything 32 bits wide
    // (a 64-bit shift is a quarter-rate instruction, v_alignbit_b32 a full-rate one)
    uint32_t lo, hi;
    int off;
    int pos;           // the same position in bits relative to the caller's origin
    uint32_t nd;       // dword `dw`, read from the window ahead of its use
    uint4 f0, f1, f2, f3;
    int fbase;
    int dbg;
    __device__ __forceinline__ void init(const __attribute__((address_space(1))) uint8_t *in, int64_t n, const uint32_t *tail_copy, uint32_t *lds_col) {
        const uintptr_t a = (uintptr_t)in;
        base = (const __attribute__((address_space(1))) uint32_t *)(a & ~(uintptr_t)3);
        tail = (const __attribute__((address_space(1))) uint32_t *)(uintptr_t)tail_copy;
        skew = (int)(a & 3) * 8;
        d_last = (int)(((int64_t)(a & 3) + n - 1) >> 2);
        win = lds_col;
    }
    __device__ __forceinline__ void ask() {
        fbase = dw;
        int t = dw - (d_last + 1 - kSbTail);  // index into the copy
        t = t > kSbTailBuf - kSbWin ? kSbTailBuf - kSbWin : t;
        const __attribute__((address_space(1))) uint32_t *p = dw + kSbWin - 1 <= d_last ? base + dw : tail + t;
        const sb_u32x4 v0 = *(const __attribute__((address_space(1))) sb_u32x4_a4 *)(p), v1 = *(const __attribute__((address_space(1))) sb_u32x4_a4 *)(p + 4),
                       v2 = *(const __attribute__((address_space(1))) sb_u32x4_a4 *)(p + 8), v3 = *(const __attribute__((address_space(1))) sb_u32x4_a4 *)(p + 12);
        f0 = make_uint4(v0[0], v0[1], v0[2], v0[3]), f1 = make_uint4(v1[0], v1[1], v1[2], v1[3]);
        f2 = make_uint4(v2[0], v2[1], v2[2], v2[3]), f3 = make_uint4(v3[0], v3[1], v3[2], v3[3]);
    }
    __device__ __forceinline__ void land() {
        win[0 * 64] = f0.x, win[1 * 64] = f0.y, win[2 * 64] = f0.z, win[3 * 64] = f0.w;
        win[4 * 64] = f1.x, win[5 * 64] = f1.y, win[6 * 64] = f1.z, win[7 * 64] = f1.w;
        win[8 * 64] = f2.x, win[9 * 64] = f2.y, win[10 * 64] = f2.z, win[11 * 64] = f2.w;
        win[12 * 64] = f3.x, win[13 * 64] = f3.y, win[14 * 64] = f3.z, win[15 * 64] = f3.w;
        wbase = fbase;
    }
    __device__ __forceinline__ uint32_t at(int d) const {
        int j = d - wbase;
        j = j > kSbWin - 1 ? kSbWin - 1 : j;  // (never beyond the window by the schedule; a wrong dword would fail the stream's check, not fault)
        return win[j * 64];
    }
    // every lane that decodes calls these at the same symbol count `it` (0 behind a seek)
    __device__ __forceinline__ void tick(int it) {
        if (dbg == 4) return;
        if ((it & (kSbEvery - 1)) == 0 && it >= kSbEvery) {
            if (it >= 2 * kSbEvery) land();
            ask();
        }
    }
    __device__ __forceinline__ void seek(int64_t bit) {
        const int64_t sb = bit + skew;
        dw = (int)(sb >> 5);
        off = (int)(sb & 31);
        ask();
        land();
        lo = win[0], hi = win[1 * 64];
        dw += 2;
        nd = at(dw);
    }
    __device__ __forceinline__ uint32_t bits32() const { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)off); }  // the 32 bits at the position
    // k <= 32 bits further.  Without a branch: the window is read whether or not a dword is taken (a read under a condition is
    // waited for on the spot; this one is not needed before the next call)
    __device__ __forceinline__ void adv(int k) {
        off += k, pos += k;
        const bool ge = off >= 32;
        lo = ge ? hi : lo, hi = ge ? nd : hi;
        dw += ge ? 1 : 0;
        off -= ge ? 32 : 0;
        nd = at(dw);
    }
};

// tails[s * kSbTailBuf + i]: dword (d_last + 1 - kSbTail + i) of stream s for i < kSbTail, zero behind (see SyncBits)
__global__ __launch_bounds__(64) void zs_inf_tails_kernel(const ParStream *ps, uint32_t *tails) {
    const ParStream s = ps[blockIdx.x];
    const uintptr_t a = (uintptr_t)s.in;
    const uint32_t *base = (const uint32_t *)(a & ~(uintptr_t)3);
    const int64_t d_last = ((int64_t)(a & 3) + s.in_len - 1) >> 2, q = d_last + 1 - kSbTail + threadIdx.x;
    tails[(int64_t)blockIdx.x * kSbTailBuf + threadIdx.x] = (int)threadIdx.x < kSbTail && q >= 0 && s.in_len > 0 ? base[q] : 0u;
}

// A block that was measured but whose tokens found no room -- its decode ran past the next candidate, a header-like bit pattern
// inside it -- is measured once more with a slab for what is now known to be its length (the lane decoder took ~1 ms for one
// such block per stream: one wave, one block).  One thread per candidate: room from a reserve behind the slabs, the candidate
// into the list of the second launch.
constexpr int kTokRetryMax = 1024;
__global__ __launch_bounds__(256) void zs_inf_tokretry_kernel(const ParStream *ps, const uint2 *work, int nwork, ParCand *cands, int64_t reserve_off, int64_t reserve_cap,
                                                              unsigned long long *cursor, int32_t *retry_cnt, int32_t *retry_list) {
    const int wi = blockIdx.x * 256 + threadIdx.x;
    if (wi >= nwork) return;
    const uint2 w = work[wi];
    ParCand &c = cands[ps[w.x].cand_off + w.y];
    if (!c.ok || c.tab < 0 || (c.tab & kTabTok)) return;
    const int64_t span = c.end_bit - c.bit;
    if (span <= 0) return;
    const int S = tok_sub_bits(span);
    const int64_t nsub = (span + S - 1) / S + 1, need = nsub <= kCkMax ? nsub * tok_unit(S) : 0;
    if (!need) return;
    const int64_t off = (int64_t)atomicAdd(cursor, (unsigned long long)need);
    if (off + need > reserve_cap) return;
    const int at = atomicAdd(retry_cnt, 1);
    if (at >= kTokRetryMax) return;
    c.tok_off = reserve_off + off, c.tok_cap = (int32_t)need;
    retry_list[at] = wi;
}

struct TokW {  // a lane's token writer: four tokens leave as one 16-byte store
    uint32_t *base;
    int n, cap;  // tokens put; room (a multiple of 4; 0: nothing is stored)
    uint32_t r0, r1, r2, r3;
    __device__ __forceinline__ void put(uint32_t t) {
        const int k = n & 3;
        r0 = k == 0 ? t : r0, r1 = k == 1 ? t : r1, r2 = k == 2 ? t : r2, r3 = k == 3 ? t : r3;
        n++;
        if (k == 3 && n <= cap) *(uint4 *)(base + n - 4) = make_uint4(r0, r1, r2, r3);
    }
    __device__ __forceinline__ void flush() {
        if ((n & 3) && ((n + 3) & ~3) <= cap) *(uint4 *)(base + (n & ~3)) = make_uint4(r0, r1, r2, r3);
    }
};

// Codes longer than the primary index of their table, without the bit-by-bit canonical walk (15 dependent steps that every
// lane of the wave sat through whenever one lane met such a code): canonical codes ascend with their length, so with the next
// 15 bits first-bit-on-top (rv) a code's length is the number of lengths whose codes end at or below rv, and its symbol follows
// from where its length's codes begin.  end[L]: the left-aligned value behind the last code of length L; idx[L]: symbols
// shorter than L.  rv >= end[15]: no code (an incomplete set).
struct LongCodes {
    uint16_t lend[16], lidx[16], dend[16], didx[16];
};
__device__ __forceinline__ void long_codes_build(const InfTables &T, LongCodes &A) {  // (every lane writes the same values)
    uint32_t le = 0, de = 0, li = 0, di = 0;
    A.lend[0] = 0, A.dend[0] = 0, A.lidx[0] = 0, A.didx[0] = 0;
    for (int L = 1; L <= 15; L++) {
        A.lidx[L] = (uint16_t)li, A.didx[L] = (uint16_t)di;
        le += (uint32_t)T.lcount[L] << (15 - L), de += (uint32_t)T.dcount[L] << (15 - L);
        li += T.lcount[L], di += T.dcount[L];
        A.lend[L] = (uint16_t)(le > 0x8000u ? 0x8000u : le), A.dend[L] = (uint16_t)(de > 0x8000u ? 0x8000u : de);
    }
}

// The ends of the long lengths' codes in registers (the same for every lane and every symbol of the block)
struct LongEnds {
    uint32_t l11, l12, l13, l14, l15, d10, d11, d12, d13, d14, d15;
    __device__ __forceinline__ void load(const LongCodes &A) {
        l11 = A.lend[11], l12 = A.lend[12], l13 = A.lend[13], l14 = A.lend[14], l15 = A.lend[15];
        d10 = A.dend[10], d11 = A.dend[11], d12 = A.dend[12], d13 = A.dend[13], d14 = A.dend[14], d15 = A.dend[15];
    }
};

// One symbol at the reader's position: 0 = a token (olen output bytes), 1 = END_BLOCK, 2 = not decodable.  Written without
// branches where a wave's lanes differ -- a literal here, a match there, in nearly every symbol: both tables are read and both
// kinds worked out by every lane, selects pick -- because a wave runs through every branch any of its lanes takes, pays for
// each with a dozen scalar instructions, and waits for a load under a condition on the spot (round 5 measured 2160 cycles per
// symbol of a wave in the branching form, 530 of them vector instructions).  Only the long codes keep a branch: rare.
__device__ __forceinline__ int tok_symbol(SyncBits &b, const InfTables &T, const LongCodes &A, const LongEnds &E, uint32_t &tok, int &olen) {
    const uint32_t w = b.bits32();
    uint32_t e = T.lit[w & ((1u << kInfLitBits) - 1u)];
    int bad = 0;
    if (e == kInfEsc) {
        static_assert(kInfLitBits == 10, "lengths 11..15 are the long ones");
        const uint32_t rv = __brev(w) >> 17;
        const int cl = 11 + (int)(rv >= E.l11) + (int)(rv >= E.l12) + (int)(rv >= E.l13) + (int)(rv >= E.l14);
        if (rv >= E.l15) bad = 1;
        else e = ((uint32_t)T.lsym[A.lidx[cl] + ((rv - A.lend[cl - 1]) >> (15 - cl))] << 4) | (uint32_t)cl;
    }
    const int sym = (int)(e >> 4), clen = (int)(e & 15);
    const bool is_len = sym > 256;
    int ls = sym - 257;
    ls = ls < 0 ? 0 : ls;
    bad |= ls >= 29 ? 1 : 0;
    // length - 3 = base + extra bits (Trees.cs:58-62, 104-110 in closed form: codes 8.. come four to a number of extra bits)
    const int xl = ls < 8 || ls >= 28 ? 0 : (ls - 4) >> 2;
    const int bl = ls < 8 ? ls : ls >= 28 ? 255 : (4 | (ls & 3)) << xl;
    const uint32_t ml3 = (uint32_t)bl + ((w >> clen) & ((1u << xl) - 1u));  // (code and extra bits: <= 20 of the 32)
    b.adv(bad ? 0 : clen + (is_len ? xl : 0));
    const uint32_t w2 = b.bits32();
    uint32_t d = T.dist[w2 & ((1u << kInfDistBits) - 1u)];
    if (d == kInfEsc && is_len && !bad) {
        static_assert(kInfDistBits == 9, "lengths 10..15 are the long ones");
        const uint32_t rv = __brev(w2) >> 17;
        const int cl = 10 + (int)(rv >= E.d10) + (int)(rv >= E.d11) + (int)(rv >= E.d12) + (int)(rv >= E.d13) + (int)(rv >= E.d14);
        if (rv >= E.d15) bad = 1;
        else d = ((uint32_t)T.dsym[A.didx[cl] + ((rv - A.dend[cl - 1]) >> (15 - cl))] << 4) | (uint32_t)cl;
    }
    const int ds = (int)(d >> 4), dl = (int)(d & 15);
    bad |= is_len && ds >= 30 ? 1 : 0;
    // distance - 1 likewise (Trees.cs:36-41): codes 4.. come two to a number of extra bits
    const int dsc = ds > 29 ? 29 : ds;
    const int xd = dsc < 4 ? 0 : (dsc - 2) >> 1;
    const int bd = dsc < 4 ? dsc : (2 | (dsc & 1)) << xd;
    const uint32_t d1 = (uint32_t)bd + ((w2 >> dl) & ((1u << xd) - 1u));  // (<= 28 of the 32)
    b.adv(is_len && !bad ? dl + xd : 0);
    tok = is_len ? 0x80000000u | ml3 | (d1 << 8) : (uint32_t)sym;
    olen = is_len ? (int)ml3 + 3 : 1;
    return bad ? 2 : sym == 256 ? 1 : 0;  // (the callers compare the position with the stream's last bit: nothing behind it is a symbol)
}

// A lane's decode from `entry` to the first symbol boundary at or after gend (or END_BLOCK), its tokens into w, the state in
// front of its first kTokBnd symbols into bnd[k * 64] (bit position relative to `rel0` | output bytes so far << 16).
// flags as in sub_measure: 0 = crossed gend, 1 = END_BLOCK (exit_bit behind it), 2 = not decodable from here.
__device__ __forceinline__ void sub_decode_tok(SyncBits &b, int64_t nbits, const InfTables &T, const LongCodes &A, const LongEnds &E, int64_t entry, int64_t gend,
                                               int64_t rel0, TokW &w, uint32_t *bnd, int64_t &exit_bit, int &nout, int &nsym, int &flags, int &nb) {
    b.seek(entry);
    b.pos = (int)(entry - rel0);
    const int pos_gend = (int)(gend - rel0), pos_end = nbits - rel0 > 0x3FFFFFFF ? 0x3FFFFFFF : (int)(nbits - rel0);
    int out = 0, ns = 0, fl = 0, rec = 0;
    while (b.pos < pos_gend) {
        b.tick(ns);
        if (b.dbg != 6) bnd[(ns < kTokBnd ? ns : kTokBnd) * 64] = (uint32_t)b.pos | ((uint32_t)out << 16);  // (row kTokBnd: nobody's)
        rec = ns < kTokBnd ? ns + 1 : rec;
        uint32_t tok = 0;
        int olen = 0;
        const int r = tok_symbol(b, T, A, E, tok, olen);
        if (r == 2 || b.pos > pos_end) {
            fl = 2;
            break;
        }
        if (r == 1) {
            fl = 1;
            break;
        }
        if (b.dbg != 5) w.put(tok);
        out += olen, ns++;
    }
    w.flush();
    exit_bit = rel0 + b.pos, nout = out, nsym = ns, flags = fl, nb = rec;
}

// The re-entry: decode from `pe` until the position is one of the nb recorded boundaries of the lane's first decode.
// Returns that boundary's index (pn tokens / pout bytes decoded on the way, into pw), or -1: no meeting within the
// recorded boundaries, kTokPre tokens or the subsequence -- the caller decodes the subsequence again in full.
__device__ __forceinline__ int sub_prefix_tok(SyncBits &b, int64_t nbits, const InfTables &T, const LongCodes &A, const LongEnds &E, int64_t pe, int64_t gend,
                                              int64_t rel0, const uint32_t *bnd, int nb, TokW &pw, int &pn, int &pout) {
    b.seek(pe);
    b.pos = (int)(pe - rel0);
    const int pos_gend = (int)(gend - rel0), pos_end = nbits - rel0 > 0x3FFFFFFF ? 0x3FFFFFFF : (int)(nbits - rel0);
    int kk = 0, out = 0, ns = 0;
    for (;;) {
        const uint32_t rel = (uint32_t)b.pos;
        while (kk < nb && (bnd[kk * 64] & 0xFFFFu) < rel) kk++;
        if (kk >= nb) return -1;
        if ((bnd[kk * 64] & 0xFFFFu) == rel) break;
        if (ns >= kTokPre || b.pos >= pos_gend) return -1;
        b.tick(ns);
        uint32_t tok = 0;
        int olen = 0;
        if (tok_symbol(b, T, A, E, tok, olen) != 0 || b.pos > pos_end) return -1;
        pw.put(tok);
        out += olen, ns++;
    }
    pw.flush();
    pn = ns, pout = out;
    return kk;
}

struct TokLds {
    ParLds L;
    uint32_t bnd[(kTokBnd + 1) * 64];
    uint32_t win[kSbWin * 64];
    LongCodes A;
};

__global__ __launch_bounds__(64) void zs_inf_measure_tok_kernel(const ParStream *ps, const ParState *st, const uint2 *work, ParCand *cands, TokTabs *tabs,
                                                                LaneTabs *ltabs, uint32_t *toks, uint32_t *ctoks, const uint32_t *tails, int32_t *stats, int dbg,
                                                                const int32_t *retry_cnt, const int32_t *retry_list) {
    __shared__ __attribute__((aligned(16))) TokLds M;
    ParLds &L = M.L;
    // (the second launch: the blocks of the retry list, each with its known end for a hint)
    const bool retry = retry_list != nullptr;
    if (retry && (int)blockIdx.x >= (*retry_cnt < kTokRetryMax ? *retry_cnt : kTokRetryMax)) return;
    const int wi = retry ? retry_list[blockIdx.x] : (int)blockIdx.x;
    const uint2 w = work[wi];
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
    if (retry) hint = c.end_bit;
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
    long_codes_build(L.T, M.A);
    __syncthreads();
    LongEnds ends;
    ends.load(M.A);
    if (dbg == 1) {  // (timing: the header alone)
        if (lane == 0) c.ok = 0;
        return;
    }
    const int64_t b0 = inf_tell(hb);  // first symbol of the block
    const int S = tok_sub_bits(hint - cbit);  // (as zs_inf_tokalloc_kernel sized the slab)
    const int unit = tok_unit(S), mcap = tok_main_cap(S);
    const int64_t tok_off = retry ? c.tok_off : st[w.x].tok_base + c.tok_off;  // (a retried block's room comes from the reserve: absolute)
    const int tok_cap = c.tok_cap;
    const __attribute__((address_space(1))) uint8_t *gin = (const __attribute__((address_space(1))) uint8_t *)(uintptr_t)s.in;
    TokTabs &T = tabs[wi];
    LaneTabs &LT = ltabs[wi];  // checkpoints and tables: a block whose tokens found no room is decoded again lane by lane
    uint32_t *bnd = M.bnd + lane;
    SyncBits sbits;
    sbits.init(gin, s.in_len, tails + (int64_t)w.x * kSbTailBuf, M.win + lane);
    sbits.dbg = dbg;
    int64_t entry0 = b0, out_base = 0, total_syms = 0, end_bit = 0;
    int nck = 0, result = 0;  // result: 1 = END_BLOCK reached on the proven chain, 2 = not decodable
    bool store = true, store_ck = true, hint_ok = true;
    int n_full = 0, n_pre = 0;  // statistics: whole decodes beyond the first, re-entries that met the first decode
    for (int round = 0; result == 0; round++) {
        const int64_t g = b0 + ((int64_t)round * 64 + lane) * S, gend = g + S;
        const int sub = round * 64 + lane;
        // the lane's slab: the re-entry's tokens, then the decode's
        const bool room = (int64_t)(sub + 1) * unit <= tok_cap && dbg != 3;  // (dbg 3, timing: no token stores)
        uint32_t *slab = toks + tok_off + (int64_t)sub * unit;
        int64_t entry = lane == 0 ? entry0 : g, exit_bit = -1;
        int nout = 0, nsym = 0, flags = 0, nvalid = 0, lf = 0;
        // the lane's (latest) whole decode, and what of it the proven token list uses
        int64_t m_exit = -1;
        int m_nout = 0, m_nsym = 0, m_flags = 2, m_nb = 0, skip = 0, pre_n = 0;
        bool have_main = false, over = false;
        bool spec = false;  // this lane has decoded its subsequence (from `entry`)
        bool run = lane == 0 || (g < nbits && (g < hint || !hint_ok));
        for (;;) {
            if (run) {
                bool met = false;
                if (have_main && m_flags != 2) {
                    TokW pw{slab, 0, room ? kTokPre : 0, 0, 0, 0, 0};
                    int pn = 0, pout = 0;
                    const int k = sub_prefix_tok(sbits, nbits, L.T, M.A, ends, entry, gend, g, bnd, m_nb, pw, pn, pout);
                    if (k >= 0) {
                        met = true;
                        skip = k, pre_n = pn;
                        nout = pout + m_nout - (int)(bnd[k * 64] >> 16), nsym = pn + m_nsym - k;
                        exit_bit = m_exit, flags = m_flags;
                        n_pre++;
                    }
                }
                if (!met) {
                    TokW mw{slab + kTokPre, 0, room ? mcap : 0, 0, 0, 0, 0};
                    sub_decode_tok(sbits, nbits, L.T, M.A, ends, entry, gend, g, mw, bnd, m_exit, m_nout, m_nsym, m_flags, m_nb);
                    over = mw.n > mw.cap;
                    n_full += have_main ? 1 : 0;
                    have_main = true;
                    skip = 0, pre_n = 0;
                    exit_bit = m_exit, nout = m_nout, nsym = m_nsym, flags = m_flags;
                }
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
            if (dbg == 2) {  // (timing: the header and every lane's first decode)
                lf = 2;
                break;
            }
            // lane nvalid decodes from a proven exit; the lanes behind it whose entry no longer fits their predecessor's exit
            // go again too (their predecessor's exit is usually right already: that is the self-synchronisation)
            run = lane >= nvalid && pspec && pf == 0 && (!spec || entry != pe);
            if (run) entry = pe;
            // the chain has walked past the hint: it was not the block's end, so everyone behind speculates as well
            if (__shfl((int)spec, nvalid) == 0) hint_ok = false;
            if (!hint_ok && !spec && !run && lane > nvalid && g < nbits) run = true;
        }
        // the proven lanes: output positions, token lists
        const bool valid = lane < nvalid;
        int incl = valid ? nout : 0;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        int incl_sy = valid ? nsym : 0;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl_sy, d);
            if (lane >= d) incl_sy += t;
        }
        const int sy = __shfl(incl_sy, 63);
        const bool lane_bad = valid && (over || !room);
        if (__ballot(lane_bad) != 0ull) store = false;
        if (store && nck + nvalid <= kCkMax) {
            // the proven lanes' tokens to the block's list, in order: the re-entry's, then the first decode's from the meeting
            // point on (the lane's own stores have to have landed)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (valid) {
                uint32_t *dst = ctoks + tok_off + (total_syms + incl_sy - nsym);
                tok_copy(dst, slab, pre_n);
                tok_copy(dst + pre_n, slab + kTokPre + skip, nsym - pre_n);
            }
        } else {
            store = false;
        }
        if (store_ck && nck + nvalid <= kCkMax) {
            if (valid) {
                LT.ck_bit[nck + lane] = (uint32_t)(entry - cbit);
                LT.ck_out[nck + lane] = (uint32_t)(out_base + incl - nout);
            }
        } else {
            store_ck = false;
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
            if (entry0 - cbit >= ((int64_t)1 << 32)) store_ck = false;
        }
    }
    if (stats) {
        for (int d = 32; d; d >>= 1) n_full += __shfl_xor(n_full, d), n_pre += __shfl_xor(n_pre, d);
        if (lane == 0) atomicAdd(stats + 0, n_full), atomicAdd(stats + 1, n_pre), atomicAdd(stats + 2, nck), atomicAdd(stats + 3, store && result == 1 ? 0 : 1);
    }
    if (result != 1) {
        if (lane == 0) c.ok = 0;
        return;
    }
    if (!store && store_ck && end_bit - cbit < ((int64_t)1 << 32)) {
        const uint4 *src = (const uint4 *)&L.T;
        uint4 *dst = (uint4 *)&LT;
        for (int i = lane; i < (int)(sizeof(InfTables) / 16); i += 64) dst[i] = src[i];
        if (lane == 0) {
            LT.ck_bit[nck] = (uint32_t)(end_bit - cbit);
            LT.ck_out[nck] = (uint32_t)out_base;
            LT.nsub = nck;
            c.tab = (int32_t)wi;
        }
    }
    if (lane == 0) {
        if (store) {
            T.tok_off = tok_off, T.ntok = (uint32_t)total_syms, T.pad_ = 0;
            c.tab = (int32_t)wi | kTabTok;
        }
        c.end_bit = end_bit;
        c.out_bytes = out_base;
        c.bfinal = bfinal;
        c.ok = 1;
    }
}

// inclusive sum over the 64 lanes with DPP moves: shifts inside the rows of 16, then the rows' last lanes broadcast (a scan by
// __shfl_up is six permutes through the LDS crossbar, one after the other)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// ------------------------------------------------------------------ X
// One workgroup per block, two of them to a CU: a step takes the block's next tokens (at most kExpTok, at most kExpTile cells),
// finds every cell's token (a bitmap of the cells where tokens begin, a prefix popcount over its words), works the cell out --
// a literal's byte; a match's source cell from the ring of flat cells when it lies before the tile (the ring begins as the
// markers of the 32 KiB before the block, so "before the block" is no case of its own), a pointer to the source when it lies in
// the tile -- lets the pointers jump until none is left, and stores the tile: 8 cells = 16 bytes per thread.
constexpr int kExpThreads = 512;
constexpr int kExpTile = 4096;                 // cells a step produces at most: 8 per thread
constexpr int kExpTok = 1024;                  // tokens a step takes at most: 2 per thread
constexpr int kExpRing = kWSize + kExpTile;    // flat cells kept in LDS: the 32 Ki before the tile, and the tile
constexpr int kExpUnres = 0x100;               // [0x100, 0x100 + kExpTile): "the cell at this index of the tile" (not flat yet)
static_assert(kExpUnres + kExpTile <= 0x8000, "tile pointers lie between the bytes and the window markers");
struct ExpLds {
    uint16_t ring[kExpRing];
    uint32_t tok[kExpTok];
    uint16_t tstart[kExpTok];
    uint32_t bits[kExpTile / 32];     // bit c: a token starts at cell c of the tile
    uint16_t wpre[kExpTile / 32];     // token starts in the words before
    uint32_t wsum_a[8], wsum_b[8];
    uint32_t cnt, len;
    uint32_t any[2][8];               // "some lane of wave w still has a pointer", two sets used in turn
};
constexpr int kExpLds = (int)sizeof(ExpLds);
static_assert(2 * (kExpLds + 512) <= 160 * 1024, "two workgroups to a CU");

__global__ __launch_bounds__(kExpThreads) void zs_inf_expand_kernel(const ParStream *ps, const ParState *st, const uint2 *work, const ParBlock *blocks,
                                                                   const TokTabs *tabs, const uint32_t *toks, uint16_t *cells, int32_t *fail, int32_t *stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t exp_smem[];
    ExpLds &E = *(ExpLds *)exp_smem;
    const uint2 w = work[blockIdx.x];
    const ParStream s = ps[w.x];
    if (!st[w.x].ok || (int)w.y >= st[w.x].nblk) return;
    const ParBlock k = blocks[s.blk_off + w.y];
    if (k.tab < 0 || !(k.tab & kTabTok)) return;  // no tokens: the lane decoder's (checkpoints) or the wave decoder's
    const TokTabs &T = tabs[k.tab & ~kTabTok];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *tk = toks + T.tok_off;
    uint16_t *cl = cells + s.cell_off + k.out_off;
    const uint32_t ttot = T.ntok;
    // the ring before the block's first cell: position -i is byte 32768 - i of the window before the block
    for (int i = tid * 2; i < kWSize; i += kExpThreads * 2) *(uint32_t *)&E.ring[kExpTile + i] = (0x8000u | (uint32_t)i) | ((0x8001u | (uint32_t)i) << 16);
    auto fetch = [&](uint32_t t) -> uint32_t { return t < ttot ? tk[t] : 0u; };  // token t of the block (the measuring pass left them in order)
    uint32_t t0 = 0;
    int P = 0, Pm = 0;  // the tile's first cell: block-relative position, ring slot
    bool bad = false;
    int n_steps = 0, n_rounds = 0;
    uint32_t ta = fetch(2u * tid), tb = fetch(2u * tid + 1);
    __syncthreads();
    long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, tc = stats ? clock64() : 0;  // (timing, ZS_DEBUG_INF: cycles of wave 0 per phase)
#define EXP_PH(i)                      \
    if (stats) {                       \
        const long long now_ = clock64(); \
        ph[i] += now_ - tc;            \
        tc = now_;                     \
    }
    while (t0 < ttot) {
        // 1. token lengths and their running sum: where every token's cells begin in the tile
        const bool va = t0 + 2u * tid < ttot, vb = t0 + 2u * tid + 1 < ttot;
        const uint32_t la = !va ? 0u : (ta >> 31) ? (ta & 0xFFu) + 3u : 1u, lb = !vb ? 0u : (tb >> 31) ? (tb & 0xFFu) + 3u : 1u;
        if (tid < kExpTile / 32) E.bits[tid] = 0;
        if (tid == 0) E.cnt = 0, E.len = 0;
        const uint32_t v = wave_incl_sum(la + lb);
        if (lane == 63) E.wsum_a[wave] = v;
        __syncthreads();
        uint32_t base = 0;
        for (int q = 0; q < wave; q++) base += E.wsum_a[q];
        const uint32_t sa = base + v - la - lb, sb = sa + la;
        // 2. the tokens that fit the tile are a prefix of the batch (of every wave's lanes, too: its last fitting lane knows where they end)
        const bool fa = va && sa + la <= (uint32_t)kExpTile, fb = vb && sb + lb <= (uint32_t)kExpTile;
        {
            const uint64_t ma = __ballot(fa), mb = __ballot(fb);
            if (ma) {
                const uint32_t end = fb ? sb + lb : sa + la;
                const uint32_t mx = (uint32_t)__builtin_amdgcn_readlane((int)end, 63 - __builtin_clzll(ma));
                if (lane == 0) {
                    atomicAdd(&E.cnt, (uint32_t)(__builtin_popcountll(ma) + __builtin_popcountll(mb)));
                    atomicMax(&E.len, mx);
                }
            }
        }
        if (fa) {
            E.tok[2 * tid] = ta, E.tstart[2 * tid] = (uint16_t)sa;
            atomicOr(&E.bits[sa >> 5], 1u << (sa & 31));
        }
        if (fb) {
            E.tok[2 * tid + 1] = tb, E.tstart[2 * tid + 1] = (uint16_t)sb;
            atomicOr(&E.bits[sb >> 5], 1u << (sb & 31));
        }
        EXP_PH(0)
        __syncthreads();
        EXP_PH(1)
        const uint32_t cnt = E.cnt;
        const int L = (int)E.len;
        // the next batch is on its way while this one is expanded
        const uint32_t na = fetch(t0 + cnt + 2u * tid), nb = fetch(t0 + cnt + 2u * tid + 1);
        // 3. token starts in the words before each word of the bitmap
        {
            const uint32_t pc = tid < kExpTile / 32 ? (uint32_t)__builtin_popcount(E.bits[tid]) : 0u;
            const uint32_t pv = wave_incl_sum(pc);
            if (lane == 63 && wave < kExpTile / 32 / 64) E.wsum_b[wave] = pv;
            __syncthreads();
            if (tid < kExpTile / 32) {
                uint32_t b2 = 0;
                for (int q = 0; q < wave; q++) b2 += E.wsum_b[q];
                E.wpre[tid] = (uint16_t)(b2 + pv - pc);
            }
            __syncthreads();
        }
        EXP_PH(2)
        // 4. the thread's 8 cells: a byte, a flat cell from the ring, or a pointer into the tile
        const int c0 = tid * 8;
        uint32_t pmask = 0;  // cells that are pointers
        uint32_t cv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool early = k.out_off + P < kWSize;  // (uniform) a distance may still reach before the stream's first byte
        if (c0 + 8 <= L && Pm + c0 + 8 <= kExpRing && !early) {
            // the common case -- the thread's eight cells all belong to the tile, their slots do not wrap: no guards, constant offsets
            const uint32_t word = E.bits[c0 >> 5];
            const int sh = c0 & 31;
            int owner = (int)E.wpre[c0 >> 5] + __builtin_popcount(word & ((1u << sh) - 1u)) - 1;
            const uint32_t sub = word >> sh;
            uint16_t *mine = &E.ring[Pm + c0];
            const int rel = Pm + c0 - 1;  // (ring slot of cell u's source) = rel + u - (distance - 1), wrapped
#pragma unroll
            for (int u = 0; u < 8; u++) {
                owner += (int)((sub >> u) & 1u);
                const uint32_t tkn = E.tok[owner];
                const int d1 = (int)((tkn >> 8) & 0x7FFFu);
                const uint32_t a = (uint32_t)(rel + u - d1), b = a + (uint32_t)kExpRing;
                const uint32_t rv = E.ring[a < b ? a : b];  // (a "negative" a is a huge unsigned one: the smaller of the two is the slot)
                const int sr = c0 + u - 1 - d1;
                const bool is_m = (int32_t)tkn < 0, in_tile = sr >= 0;
                const uint32_t val = !is_m ? tkn & 0xFFu : in_tile ? (uint32_t)(kExpUnres + sr) : rv;
                mine[u] = (uint16_t)val;
                cv[u] = val;
                pmask |= is_m && in_tile ? 1u << u : 0u;
            }
        } else if (c0 < L) {
            const uint32_t word = E.bits[c0 >> 5];
            const int sh = c0 & 31;
            int owner = (int)E.wpre[c0 >> 5] + __builtin_popcount(word & ((1u << sh) - 1u)) - 1;
            uint32_t tkn = owner >= 0 ? E.tok[owner] : 0u;
            int slot = Pm + c0;
            slot = slot >= kExpRing ? slot - kExpRing : slot;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + u;
                if ((word >> (sh + u)) & 1u) {
                    owner++;
                    tkn = E.tok[owner];
                }
                const int dist = (int)((tkn >> 8) & 0x7FFFu) + 1;
                const int sr = c - dist;  // the source, relative to the tile
                int i = Pm + sr;  // (read whether or not it is used: a load under a condition is waited for on the spot)
                i = i < 0 ? i + kExpRing : i;
                i = i >= kExpRing ? i - kExpRing : i;
                const uint32_t rv = E.ring[i];
                const bool is_m = (tkn >> 31) != 0;
                const uint32_t val = !is_m ? tkn & 0xFFu : sr >= 0 ? (uint32_t)(kExpUnres + sr) : rv;
                if (is_m && k.out_off + P + sr < 0) bad = true;  // before the stream's first byte: "invalid distance" (InfCodes.cs:294)
                cv[u] = val;
                if (c < L) {
                    E.ring[slot] = (uint16_t)val;
                    pmask |= (is_m && sr >= 0) ? 1u << u : 0u;
                }
                slot = slot + 1 == kExpRing ? 0 : slot + 1;
            }
        }
        // 5. sources inside the tile: pointer jumping over the cells that are still pointers (a slot read while it is rewritten
        //    holds either form; both say the same)
        n_steps++;
        EXP_PH(3)
        // (the workgroup's "or": a ballot per wave, a flag per wave, one barrier; __syncthreads_or is three barriers and an atomic)
        auto any_left = [&](int set) -> bool {
            const bool mine = __ballot(pmask != 0) != 0ull;
            if (lane == 0) E.any[set][wave] = mine ? 1u : 0u;
            __syncthreads();
            const uint4 f0 = *(const uint4 *)&E.any[set][0], f1 = *(const uint4 *)&E.any[set][4];
            return (f0.x | f0.y | f0.z | f0.w | f1.x | f1.y | f1.z | f1.w) != 0u;
        };
        int aset = 0;
        while (any_left(aset)) {
            aset ^= 1;
            EXP_PH(6)
            n_rounds++;
            if (pmask) {
                // the thread's pointers follow their sources, all of them at once and up to three hops before the workgroup meets
                // again (what a hop reads is a cell as some thread left it: a value, or a pointer further back -- both say the same)
                const uint32_t was = pmask;
#pragma unroll
                for (int hop = 0; hop < 3; hop++) {
                    uint32_t sv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t idx = (pmask >> u) & 1u ? cv[u] - (uint32_t)kExpUnres : 0u;
                        const uint32_t b = (uint32_t)Pm + idx, a = b - (uint32_t)kExpRing;
                        sv[u] = E.ring[a < b ? a : b];  // (no wrap: a is "negative", a huge unsigned number)
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if ((pmask >> u) & 1u) {
                            cv[u] = sv[u];
                            if (!(sv[u] >= (uint32_t)kExpUnres && sv[u] < 0x8000u)) pmask &= ~(1u << u);
                        }
                    if (!pmask) break;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if ((was >> u) & 1u) {
                        int slot = Pm + c0 + u;
                        slot = slot >= kExpRing ? slot - kExpRing : slot;
                        E.ring[slot] = (uint16_t)cv[u];
                    }
            }
        }
        EXP_PH(4)
        // 6. out: 8 cells = one 16-byte store
        if (c0 < L) {
            if (c0 + 8 <= L) {
                store_u4_a2(cl + P + c0, cv[0] | (cv[1] << 16), cv[2] | (cv[3] << 16), cv[4] | (cv[5] << 16), cv[6] | (cv[7] << 16));
            } else {
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (c0 + u < L) cl[P + c0 + u] = (uint16_t)cv[u];
            }
        }
        if (cnt == 0) {  // (cannot happen: a token has at most 258 cells)
            bad = true;
            break;
        }
        t0 += cnt, P += L;
        Pm += L;
        Pm = Pm >= kExpRing ? Pm - kExpRing : Pm;
        ta = na, tb = nb;
        EXP_PH(5)
        __syncthreads();  // the step's reads of tok / bits / cnt are done before the next step rewrites them
        EXP_PH(6)
    }
    if (bad || (int64_t)P != k.out_bytes) fail[w.x] = 1;
    if (stats && tid == 0) {
        atomicAdd(stats + 4, n_steps), atomicAdd(stats + 5, n_rounds), atomicAdd(stats + 6, 1);
        for (int i = 0; i < 7; i++) atomicAdd(stats + 8 + i, (int)(ph[i] >> 8));
    }
}

}  // namespace zs
