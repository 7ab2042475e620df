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

constexpr int kTabTok = 0x40000000;  // ParCand::tab / ParBlock::tab: the block has tokens (tabs[tab & ~kTabTok]); without the bit: checkpoints for the lane decoder
constexpr int kTokBnd = 32;   // symbol boundaries of a lane's decode kept for a re-entry
constexpr int kTokPre = 48;   // tokens a re-entry may decode before it has to meet the decode it replaces (a multiple of 4)

// token: bit 31 clear -- literal, the byte in bits 0..7; set -- match, length - 3 in bits 0..7, distance - 1 in bits 8..22
__host__ __device__ inline int tok_sub_bits(int64_t span_bits) {  // subsequence length for a block of at most span_bits bits
    const int64_t per = ((span_bits + 63) / 64 + 63) & ~(int64_t)63;
    return per < kSubMinBits ? kSubMinBits : per > kSubMaxBits ? kSubMaxBits : (int)per;
}
__host__ __device__ inline int tok_main_cap(int S) { return ((S >> 1) + 8 + 3) & ~3; }  // a symbol has >= 2 bits (a match of two 1-bit codes)
__host__ __device__ inline int tok_unit(int S) { return kTokPre + tok_main_cap(S); }

struct SubTok {
    uint32_t out;                 // block-relative output position of the subsequence's first symbol
    uint32_t pre_off, main_off;   // token offsets inside the candidate's slab: the re-entry's tokens, then the first decode's from the meeting point on
    uint16_t pre_n, main_n;
};
struct TokTabs {
    int32_t nsub, pad_;
    int64_t tok_off;              // the candidate's slab in the token array
    SubTok sub[kCkMax];
};

// Slabs: candidate i of a stream may decode up to the next candidate's header (its most likely end), in subsequences of S
// bits, each with room for S / 2 tokens and a re-entry's kTokPre.  One workgroup lays all candidates of the batch out.
__global__ __launch_bounds__(1024) void zs_inf_tokalloc_kernel(const ParStream *ps, const ParState *st, int nstreams, ParCand *cands, int64_t *total) {
    __shared__ int64_t wsum[16];
    __shared__ int64_t run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) run = 0;
    __syncthreads();
    for (int si = 0; si < nstreams; si++) {
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
                c.tok_off = base + v - need;
                c.tok_cap = (int32_t)need;
            }
            __syncthreads();
            if (tid == 0) run += tot;
            __syncthreads();
        }
    }
    if (tid == 0) *total = run;
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

// one symbol at the reader's position: 0 = a token (olen output bytes), 1 = END_BLOCK, 2 = not decodable
__device__ __forceinline__ int tok_symbol(LaneBits &b, const InfTables &T, uint32_t &tok, int &olen) {
    b.fill();
    int sym, clen;
    {
        const uint16_t e = T.lit[b.peek(kInfLitBits)];
        if (e != kInfEsc) sym = e >> 4, clen = e & 15;
        else sym = lane_slow(b, T.lcount, T.lsym, clen);
    }
    if (sym < 0 || clen > b.cnt) return 2;
    b.drop(clen);
    if (sym < 256) {
        tok = (uint32_t)sym, olen = 1;
        return b.bad ? 2 : 0;
    }
    if (sym == 256) return 1;
    sym -= 257;
    if (sym >= 29) return 2;
    const int mlen = (sym == 28 ? 258 : base_length(sym) + 3) + (int)b.take(extra_lbits(sym));
    b.fill();
    int ds, dl;
    {
        const uint16_t e = T.dist[b.peek(kInfDistBits)];
        if (e != kInfEsc) ds = e >> 4, dl = e & 15;
        else ds = lane_slow(b, T.dcount, T.dsym, dl);
    }
    if (ds < 0 || ds >= 30 || dl > b.cnt) return 2;
    b.drop(dl);
    const int dist = base_dist(ds) + 1 + (int)b.take(extra_dbits(ds));
    tok = 0x80000000u | (uint32_t)(mlen - 3) | ((uint32_t)(dist - 1) << 8), olen = mlen;
    return b.bad ? 2 : 0;
}

// A lane's decode from `entry` to the first symbol boundary at or after gend (or END_BLOCK), its tokens into w, the state in
// front of its first kTokBnd symbols into bnd[k * 64] (bit position relative to `rel0` | output bytes so far << 16).
// flags as in sub_measure: 0 = crossed gend, 1 = END_BLOCK (exit_bit behind it), 2 = not decodable from here.
__device__ __forceinline__ void sub_decode_tok(const __attribute__((address_space(1))) uint8_t *in, int64_t n, const InfTables &T, int64_t entry, int64_t gend,
                                               int64_t rel0, TokW &w, uint32_t *bnd, int64_t &exit_bit, int &nout, int &nsym, int &flags, int &nb) {
    LaneBits b{in, n, 0, 0, 0, false};
    b.seek(entry);
    int out = 0, ns = 0, fl = 0, rec = 0;
    int64_t cur = entry;
    while (cur < gend) {
        if (ns < kTokBnd) bnd[ns * 64] = (uint32_t)(cur - rel0) | ((uint32_t)out << 16), rec = ns + 1;
        uint32_t tok = 0;
        int olen = 0;
        const int r = tok_symbol(b, T, tok, olen);
        if (r == 2) {
            fl = 2;
            break;
        }
        if (r == 1) {
            fl = 1;
            cur = b.tell();
            break;
        }
        w.put(tok);
        out += olen, ns++;
        cur = b.tell();
    }
    w.flush();
    exit_bit = cur, nout = out, nsym = ns, flags = fl, nb = rec;
}

// The re-entry: decode from `pe` until the position is one of the nb recorded boundaries of the lane's first decode.
// Returns that boundary's index (pn tokens / pout bytes decoded on the way, into pw), or -1: no meeting within the
// recorded boundaries, kTokPre tokens or the subsequence -- the caller decodes the subsequence again in full.
__device__ __forceinline__ int sub_prefix_tok(const __attribute__((address_space(1))) uint8_t *in, int64_t n, const InfTables &T, int64_t pe, int64_t gend,
                                              int64_t rel0, const uint32_t *bnd, int nb, TokW &pw, int &pn, int &pout) {
    LaneBits b{in, n, 0, 0, 0, false};
    b.seek(pe);
    int kk = 0, out = 0, ns = 0;
    int64_t cur = pe;
    for (;;) {
        const uint32_t rel = (uint32_t)(cur - rel0);
        while (kk < nb && (bnd[kk * 64] & 0xFFFFu) < rel) kk++;
        if (kk >= nb) return -1;
        if ((bnd[kk * 64] & 0xFFFFu) == rel) break;
        if (ns >= kTokPre || cur >= gend) return -1;
        uint32_t tok = 0;
        int olen = 0;
        if (tok_symbol(b, T, tok, olen) != 0) return -1;
        pw.put(tok);
        out += olen, ns++;
        cur = b.tell();
    }
    pw.flush();
    pn = ns, pout = out;
    return kk;
}

struct TokLds {
    ParLds L;
    uint32_t bnd[kTokBnd * 64];
};

__global__ __launch_bounds__(64) void zs_inf_measure_tok_kernel(const ParStream *ps, const ParState *st, const uint2 *work, ParCand *cands, TokTabs *tabs,
                                                                LaneTabs *ltabs, uint32_t *toks, int32_t *stats) {
    __shared__ __attribute__((aligned(16))) TokLds M;
    ParLds &L = M.L;
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
    const int S = tok_sub_bits(hint - cbit);  // (as zs_inf_tokalloc_kernel sized the slab)
    const int unit = tok_unit(S), mcap = tok_main_cap(S);
    const int64_t tok_off = c.tok_off;
    const int tok_cap = c.tok_cap;
    const __attribute__((address_space(1))) uint8_t *gin = (const __attribute__((address_space(1))) uint8_t *)(uintptr_t)s.in;
    TokTabs &T = tabs[blockIdx.x];
    LaneTabs &LT = ltabs[blockIdx.x];  // checkpoints and tables: a block whose tokens found no room is decoded again lane by lane
    uint32_t *bnd = M.bnd + lane;
    int64_t entry0 = b0, out_base = 0, total_syms = 0, end_bit = 0;
    int nck = 0, result = 0;  // result: 1 = END_BLOCK reached on the proven chain, 2 = not decodable
    bool store = true, store_ck = true, hint_ok = true;
    int n_full = 0, n_pre = 0;  // statistics: whole decodes beyond the first, re-entries that met the first decode
    for (int round = 0; result == 0; round++) {
        const int64_t g = b0 + ((int64_t)round * 64 + lane) * S, gend = g + S;
        const int sub = round * 64 + lane;
        // the lane's slab: the re-entry's tokens, then the decode's
        const bool room = (int64_t)(sub + 1) * unit <= tok_cap;
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
                    const int k = sub_prefix_tok(gin, s.in_len, L.T, entry, gend, g, bnd, m_nb, pw, pn, pout);
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
                    sub_decode_tok(gin, s.in_len, L.T, entry, gend, g, mw, bnd, m_exit, m_nout, m_nsym, m_flags, m_nb);
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
        int sy = valid ? nsym : 0;
        for (int d = 32; d; d >>= 1) sy += __shfl_xor(sy, d);
        const bool lane_bad = valid && (over || !room);
        if (__ballot(lane_bad) != 0ull) store = false;
        if (store && nck + nvalid <= kCkMax) {
            if (valid) {
                SubTok q;
                q.out = (uint32_t)(out_base + incl - nout);
                q.pre_off = (uint32_t)((int64_t)sub * unit);
                q.main_off = (uint32_t)((int64_t)sub * unit + kTokPre + skip);
                q.pre_n = (uint16_t)pre_n, q.main_n = (uint16_t)(nsym - pre_n);
                T.sub[nck + lane] = q;
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
            c.tab = (int32_t)blockIdx.x;
        }
    }
    if (lane == 0) {
        if (store) {
            T.nsub = nck;
            T.tok_off = tok_off;
            c.tab = (int32_t)blockIdx.x | kTabTok;
        }
        c.end_bit = end_bit;
        c.out_bytes = out_base;
        c.bfinal = bfinal;
        c.ok = 1;
    }
}

// ------------------------------------------------------------------ X
constexpr int kExpTile = 8192;                 // cells a step produces at most: 8 per thread
constexpr int kExpTok = 2048;                  // tokens a step takes at most: 2 per thread
constexpr int kExpRing = kWSize + kExpTile;    // flattened cells kept in LDS: the 32 Ki before the tile, and the tile
constexpr int kExpUnres = 0x100;               // [0x100, 0x100 + kExpTile): "the cell at this index of the tile" (not flat yet)
static_assert(kExpUnres + kExpTile <= 0x8000, "tile pointers lie between the bytes and the window markers");
struct ExpLds {
    uint16_t ring[kExpRing];
    uint32_t tok[kExpTok];
    uint16_t tstart[kExpTok];
    uint32_t bits[kExpTile / 32];     // bit c: a token starts at cell c of the tile
    uint16_t wpre[kExpTile / 32];     // token starts in the words before
    uint32_t cum[kCkMax + 1];         // logical tokens before subsequence j
    uint32_t pre_off[kCkMax], main_off[kCkMax];
    uint16_t pre_n[kCkMax];
    uint32_t wsum_a[16], wsum_b[16], wsum_c[16];
    uint32_t cnt, len;
};
constexpr int kExpLds = (int)sizeof(ExpLds);

__global__ __launch_bounds__(1024) void zs_inf_expand_kernel(const ParStream *ps, const ParState *st, const uint2 *work, const ParBlock *blocks,
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
    const int nsub = T.nsub;
    const uint32_t *tk = toks + T.tok_off;
    uint16_t *cl = cells + s.cell_off + k.out_off;
    // logical tokens before every subsequence
    {
        uint32_t n = 0;
        if (tid < nsub) {
            const SubTok q = T.sub[tid];
            n = (uint32_t)q.pre_n + q.main_n;
            E.pre_off[tid] = q.pre_off, E.main_off[tid] = q.main_off, E.pre_n[tid] = q.pre_n;
        }
        uint32_t v = n;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) E.wsum_a[wave] = v;
        __syncthreads();
        uint32_t base = 0;
        for (int q = 0; q < wave; q++) base += E.wsum_a[q];
        if (tid < nsub) E.cum[tid + 1] = base + v;
        if (tid == 0) E.cum[0] = 0;
        __syncthreads();
    }
    const uint32_t ttot = E.cum[nsub];
    // the token at logical index t of the block (0 behind the block's last)
    auto fetch = [&](uint32_t t) -> uint32_t {
        if (t >= ttot) return 0u;
        int lo = 0, hi = nsub - 1;
        while (lo < hi) {  // the subsequence: the last j with cum[j] <= t
            const int mid = (lo + hi + 1) >> 1;
            if (E.cum[mid] <= t) lo = mid;
            else hi = mid - 1;
        }
        const uint32_t r = t - E.cum[lo], pn = E.pre_n[lo];
        return tk[r < pn ? E.pre_off[lo] + r : E.main_off[lo] + (r - pn)];
    };
    uint32_t t0 = 0;
    int P = 0, Pm = 0;  // the tile's first cell: block-relative position, ring slot
    bool bad = false;
    int n_steps = 0, n_rounds = 0;
    uint32_t ta = fetch(2u * tid), tb = fetch(2u * tid + 1);
    while (t0 < ttot) {
        // 1. token lengths and their running sum: where every token's cells begin in the tile
        const bool va = t0 + 2u * tid < ttot, vb = t0 + 2u * tid + 1 < ttot;
        const uint32_t la = !va ? 0u : (ta >> 31) ? (ta & 0xFFu) + 3u : 1u, lb = !vb ? 0u : (tb >> 31) ? (tb & 0xFFu) + 3u : 1u;
        if (tid < kExpTile / 32) E.bits[tid] = 0;
        if (tid == 0) E.cnt = 0, E.len = 0;
        uint32_t v = la + lb;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) E.wsum_a[wave] = v;
        __syncthreads();
        uint32_t base = 0;
        for (int q = 0; q < wave; q++) base += E.wsum_a[q];
        const uint32_t sa = base + v - la - lb, sb = sa + la;
        // 2. the tokens that fit the tile are a prefix of the batch
        const bool fa = va && sa + la <= (uint32_t)kExpTile, fb = vb && sb + lb <= (uint32_t)kExpTile;
        {
            const uint64_t ma = __ballot(fa), mb = __ballot(fb);
            const uint32_t end = fb ? sb + lb : fa ? sa + la : 0u;
            uint32_t mx = end;
            for (int d = 32; d; d >>= 1) {
                const uint32_t o = __shfl_xor(mx, d);
                mx = o > mx ? o : mx;
            }
            if (lane == 0 && (ma | mb)) {
                atomicAdd(&E.cnt, (uint32_t)(__builtin_popcountll(ma) + __builtin_popcountll(mb)));
                atomicMax(&E.len, mx);
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
        __syncthreads();
        const uint32_t cnt = E.cnt;
        const int L = (int)E.len;
        // the next batch is on its way while this one is expanded
        const uint32_t na = fetch(t0 + cnt + 2u * tid), nb = fetch(t0 + cnt + 2u * tid + 1);
        // 3. token starts in the words before each word of the bitmap
        {
            uint32_t pc = tid < kExpTile / 32 ? (uint32_t)__builtin_popcount(E.bits[tid]) : 0u;
            uint32_t pv = pc;
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(pv, d);
                if (lane >= d) pv += t;
            }
            if (lane == 63 && wave < 4) E.wsum_b[wave] = pv;
            __syncthreads();
            if (tid < kExpTile / 32) {
                uint32_t b2 = 0;
                for (int q = 0; q < wave; q++) b2 += E.wsum_b[q];
                E.wpre[tid] = (uint16_t)(b2 + pv - pc);
            }
            __syncthreads();
        }
        // 4. the thread's 8 cells: a byte, a window marker, a flat cell from the ring, or a pointer into the tile
        const int c0 = tid * 8;
        uint32_t cv[8];
        bool pend = false;
        if (c0 < L) {
            const uint32_t word = E.bits[c0 >> 5];
            const int sh = c0 & 31;
            int owner = (int)E.wpre[c0 >> 5] + __builtin_popcount(word & ((1u << sh) - 1u)) - 1;
            uint32_t tkn = owner >= 0 ? E.tok[owner] : 0u;
            int ts = owner >= 0 ? (int)E.tstart[owner] : 0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + u;
                if ((word >> (sh + u)) & 1u) {
                    owner++;
                    tkn = E.tok[owner], ts = (int)E.tstart[owner];
                }
                uint32_t val = tkn & 0xFFu;
                if (tkn >> 31) {
                    const int dist = (int)((tkn >> 8) & 0x7FFFu) + 1;
                    const int sr = c - dist;  // source, relative to the tile
                    if (sr >= 0) {
                        val = (uint32_t)(kExpUnres + sr);
                    } else {
                        const int spb = P + sr;  // block-relative
                        if (spb < 0) {
                            if (k.out_off + spb < 0) bad = true;  // before the stream's first byte: "invalid distance" (InfCodes.cs:294)
                            val = 0x8000u | (uint32_t)(kWSize + spb);
                        } else {
                            int i = Pm + sr;
                            i = i < 0 ? i + kExpRing : i;
                            val = E.ring[i];
                        }
                    }
                }
                cv[u] = val;
                if (c < L) {
                    int i = Pm + c;
                    i = i >= kExpRing ? i - kExpRing : i;
                    E.ring[i] = (uint16_t)val;
                    pend = pend || (val >= (uint32_t)kExpUnres && val < 0x8000u);
                }
            }
        }
        // 5. sources inside the tile: pointer jumping (a slot read while it is rewritten holds either form; both say the same)
        n_steps++;
        while (__syncthreads_or(pend ? 1 : 0)) {
            pend = false;
            n_rounds++;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + u;
                if (c < L && cv[u] >= (uint32_t)kExpUnres && cv[u] < 0x8000u) {
                    int i = Pm + (int)(cv[u] - kExpUnres);
                    i = i >= kExpRing ? i - kExpRing : i;
                    const uint32_t sv = E.ring[i];
                    cv[u] = sv;
                    int j = Pm + c;
                    j = j >= kExpRing ? j - kExpRing : j;
                    E.ring[j] = (uint16_t)sv;
                    pend = pend || (sv >= (uint32_t)kExpUnres && sv < 0x8000u);
                }
            }
        }
        // 6. out: 8 cells = one 16-byte store
        if (c0 + 8 <= L) {
            store_u4_a2(cl + P + c0, cv[0] | (cv[1] << 16), cv[2] | (cv[3] << 16), cv[4] | (cv[5] << 16), cv[6] | (cv[7] << 16));
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (c0 + u < L) cl[P + c0 + u] = (uint16_t)cv[u];
        }
        if (cnt == 0) {  // (cannot happen: a token has at most 258 cells)
            bad = true;
            break;
        }
        t0 += cnt, P += L;
        Pm += L;
        Pm = Pm >= kExpRing ? Pm - kExpRing : Pm;
        ta = na, tb = nb;
        __syncthreads();  // the step's reads of tok / bits / cnt are done before the next step rewrites them
    }
    if (bad || (int64_t)P != k.out_bytes) fail[w.x] = 1;
    if (stats && tid == 0) atomicAdd(stats + 4, n_steps), atomicAdd(stats + 5, n_rounds), atomicAdd(stats + 6, 1);
}

}  // namespace zs
