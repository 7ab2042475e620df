// zs_fast_sweep.hip -- KS: DeflateFast (levels 1-3, Deflate.Fast.cs:20-128) as window-wide sweeps of a workgroup.
// Included by zs_kernels.hip; zs_fast_sweep.h has the formulation and the code shared with the CPU model.
//
// One workgroup of NT threads per stream.  The tile -- bytes, K1's all-position links and the inserted-position bitmap of
// [t0 - 32 512, t0 + TILE) -- is staged in LDS once per TILE - W positions of progress.  A sweep:
//
//   1. search     every thread searches its PPT positions of the window [g0, g0 + W), g0 = w0 rounded down to 64, under the
//                 bitmap as it stands (final below w0, the last sweep's parse behind it, "inserted" where nothing has been
//                 parsed yet): fs_search's walk over the staged links, one chain step per trip of a wave-wide loop
//   2. hops       a wave owns groups of 64 consecutive positions.  next(p) = p + 1 or p + match length; the first position
//                 behind its group that a lane's hops lead to comes from six rounds of pointer doubling with ds_bpermute
//                 (the doubling tables stay in registers), and goes into the exit table in LDS           -- barrier 1
//   3. path       every wave follows the path from w0 through the exit table up to its own groups (at most W / 64 dependent
//                 LDS reads), which gives each group the lane the path enters it at; the lanes on the path -- the window's
//                 loop-tops -- are then found by walking the doubling tables down (six ds_bpermute).  The first loop-top
//                 whose result differs from the sweep before (results are kept in a ring by position): LDS atomic min;
//                 loop-tops per group: LDS                                                                -- barrier 2
//   4. final      loop-tops up to and including that one are final: their symbols leave in order (a wave's rank offset is
//                 the sum of the groups' counts before it), every group writes the bits of its 64 positions as this sweep's
//                 parse has them (loop-tops, the inside of short matches, "inserted" behind the path's end) -- barrier 3
//   5. compress   the links of the positions that have become final are replaced by the distance to the nearest inserted
//                 position of their bucket (fs_compress), in LDS and in the stream's link array; no barrier: walkers find the
//                 same candidates through either link (zs_fast_sweep.h, fact 3), so this overlaps the next sweep's searches
//
// Three barriers per sweep; 1024 positions searched, ~390 made final on text (profiles/r04_fast_jacobi_convergence.txt).
// What it leaves is what K4 / K5 leave for the lazy parse, so the tail engine (restored from the bitmap and the links --
// compressed or not, le_restore_prev_ins finds the same predecessor) and the block kernels go on unchanged.

constexpr int kFsBack = 32512;  // >= kMaxDist, multiple of 64
constexpr int kFsFwd = 272;     // >= kMaxMatch + 8, multiple of 16
constexpr int kFsTile1 = 12288; // positions per tile with one position per thread (W = 1024): 152.7 KiB of LDS
template <int TILE>
struct FsLayout {
    static constexpr int bytes = kFsBack + TILE + kFsFwd, links = kFsBack + TILE, bit_words = links / 32 + 16;
};
template <int NT, int PPT, int TILE>
constexpr int fs_lds_bytes() {
    return FsLayout<TILE>::bytes + 2 * FsLayout<TILE>::links + 4 * FsLayout<TILE>::bit_words + 4 * (2 * NT * PPT) + 4 * (NT * PPT) + 4 * (NT * PPT / 64) + 64;
}

template <int NT, int PPT, int TILE>
__global__ __launch_bounds__(NT) void zs_fast_sweep_kernel(const StreamDesc *sd, StreamState *st, uint16_t *link, uint32_t *syms, int32_t *blk_end,
                                                             int32_t *blk_top, LevelCfg lv, int strategy) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int W = NT * PPT, NG = W / 64, NW = NT / 64, RING = 2 * W;
    constexpr int fsBytes = FsLayout<TILE>::bytes, fsLinks = FsLayout<TILE>::links, fsBitWords = FsLayout<TILE>::bit_words;
    static_assert(TILE % 64 == 0 && TILE >= 2 * W, "tile");
    const StreamDesc s = sd[blockIdx.x];
    if (s.fv_end < 0) return;
    uint8_t *wb = smem;                                  // bytes, index = position - lo
    uint16_t *wl = (uint16_t *)(smem + fsBytes);         // links, 0 = none
    uint32_t *bm = (uint32_t *)(smem + fsBytes + 2 * fsLinks);  // inserted bits, word k = positions [lo + 32 k, + 32)
    uint32_t *ring = bm + fsBitWords;                    // the last sweep's result of position q at q & (RING - 1)
    uint32_t *ex = ring + RING;                          // window index -> first index behind its group on its path | last hop << 16
    uint32_t *cnt = ex + W;                              // loop-tops per group
    uint32_t *shv = cnt + NG;                            // [0] first loop-top with a new result, [1] last loop-top, [2] final ones in its group
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int64_t n = s.n, body_end = s.fv_end;
    const gcbytes in = as_global(s.in);
    uint16_t *lk = link + s.pos_off;
    uint32_t *gbits = s.ins_bits;
    const int kl = s.kl;
    const bool search = strategy != kHuffmanOnly;  // (HuffmanOnly: Longest_match is never called, Deflate.Fast.cs:61-66)
    const int nice = lv.nice, chain = lv.chain, lazy = lv.lazy;
    const bool aligned = (((uintptr_t)in) & 15) == 0;
    // the state that goes from sweep to sweep, held by every thread (all of it is computed from shared values)
    int64_t w0 = 0, nsyms = 0, trigger = kl >= 1 ? read_end_before(1) - (kMinLookahead - 1) : -1, preins = -1, dead_pos = -1, only_pos = -1, ev_end = 0;
    int k_fired = 0;
    int64_t t0 = -(1ll << 40), w0_staged = 0;
    int64_t x_end = 0;  // where the last sweep's parse ended: the bits of [w0, x_end) are that parse's (the guess the results in the ring belong to)
    if (tid == 0) shv[0] = 0xFFFFFFFFu, shv[1] = 0, shv[2] = 0;
    while (w0 <= body_end) {
        int64_t g0 = w0 & ~63LL;
        if (g0 + W > t0 + TILE) {
            // ---- (leave the tile: the bits that became final go back to the stream's bitmap) stage the tile at g0
            __syncthreads();  // the compression of the last sweep reads the tile that is about to be overwritten
            if (t0 >= 0) {
                const int64_t lo_old = t0 - kFsBack;
                for (int64_t wd = (w0_staged >> 5) + tid; wd <= (x_end >> 5); wd += NT) gbits[wd] = bm[wd - (lo_old >> 5)];
                __syncthreads();
            }
            t0 = g0, w0_staged = w0;
            const int64_t lo = t0 - kFsBack;
            for (int i = tid; i < fsBytes / 16; i += NT) {
                const int64_t a = lo + (int64_t)i * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a >= 0 && a + 15 < n && aligned) {
                    const u32x4 t = *(gcu32x4)(in + a);
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                } else if (a + 15 >= 0 && a < n) {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 16; k++) {
                        const int64_t b = a + k;
                        if (b >= 0 && b < n) t[k >> 2] |= (uint32_t)in[b] << (8 * (k & 3));
                    }
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
                ((uint4 *)wb)[i] = v;
            }
            for (int i = tid; i < fsLinks / 8; i += NT) {
                const int64_t a = lo + (int64_t)i * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a >= 0 && a + 7 < n) {
                    v = *(const uint4 *)(lk + a);
                } else if (a + 7 >= 0 && a < n) {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 8; k++) {
                        const int64_t b = a + k;
                        if (b >= 0 && b < n) t[k >> 1] |= (uint32_t)lk[b] << (16 * (k & 1));
                    }
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
                ((uint4 *)wl)[i] = v;
            }
            // the set: the stream's bits below the last parse's end (final below w0, that parse's from there on), "inserted"
            // behind it (the guess for what no sweep has parsed)
            for (int i = tid; i < fsBitWords; i += NT) {
                const int64_t p32 = lo + 32ll * i;  // first position of the word
                uint32_t v = 0xFFFFFFFFu;
                if (p32 + 32 <= x_end) v = p32 >= 0 ? gbits[p32 >> 5] : 0u;
                else if (p32 < x_end) v = gbits[p32 >> 5] | (0xFFFFFFFFu << (uint32_t)(x_end - p32));
                bm[i] = v;
            }
            __syncthreads();
        }
        const int64_t lo = t0 - kFsBack;
        const int min_i = (int)(1 - lo);  // position 0 is never a candidate
        // ---- the read event at loop-top w0: w0 + 1 is inserted first (Deflate.cs:1010-1013).  One per 32 Ki positions.
        if (trigger >= 0 && w0 >= trigger) {
            const bool same = wl[w0 + 1 - lo] == 1;
            __syncthreads();  // (the read above, the compression of the sweep before) before the cut below
            k_fired++;
            preins = w0 + 1;
            if (same) dead_pos = w0, only_pos = w0 + 1;
            else dead_pos = w0 + 1, only_pos = -1;
            if (tid == 0) {
                bm[(w0 + 1 - lo) >> 5] |= 1u << ((w0 + 1 - lo) & 31);
                if (same) wl[w0 - lo] = 0, lk[w0] = 0;  // the reference's prev[w0] = w0 + 1, prev[w0 + 1] = w0
            }
            trigger = k_fired < kl ? read_end_before(k_fired + 1) - (kMinLookahead - 1) : -1;
            __syncthreads();
        }
        int64_t hi = g0 + W;
        if (body_end + 1 < hi) hi = body_end + 1;
        if (trigger >= 0 && trigger < hi) hi = trigger;
        const int w0r = (int)(w0 - g0), hir = (int)(hi - g0), gi = (int)(g0 - lo);
        uint32_t res[PPT], dbl[PPT][6];
        bool act[PPT], agr[PPT];
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int grp = wave + NW * k, self = 64 * grp + lane;  // index in the window
            const int qi = gi + self;                                // index in the tile
            const int64_t q = g0 + self;
            act[k] = self >= w0r && self < hir;
            // ---- 1. search (fs_search, one chain step per trip)
            int found = 0, best = 2, bdist = 0;
            const bool dead = !search || q == dead_pos, only_prev = search && q == only_pos;
            int done = (!act[k] || dead || only_prev) ? 1 : 0;
            if (act[k] && only_prev) {
                int len = 0;
                while (len < kMaxMatch) {
                    const uint64_t y = lds_u64(wb, qi + len) ^ lds_u64(wb, qi - 1 + len);
                    if (y) {
                        len += (int)(__builtin_ctzll(y) >> 3);
                        break;
                    }
                    len += 8;
                }
                len = len < kMaxMatch ? len : kMaxMatch;
                if (len > 2) best = len, bdist = 1;
            }
            if (__ballot(!done)) {
                const uint64_t scan8 = lds_u64(wb, qi);
                int c = qi;
                while (__ballot(!done)) {
                    const int l = wl[c];
                    const int nc = c - l, d = qi - nc;
                    const int maxd = found ? kMaxDist - 1 : kMaxDist;  // hash_head: <= MAX_DIST; later: cur_match > limit
                    const int valid = (done == 0) & (l != 0) & (nc >= min_i) & (d <= maxd);
                    done |= valid ^ 1;
                    c = valid ? nc : qi;  // lanes that are done read their own position (in range)
                    const uint32_t word = bm[c >> 5];
                    const int isin = valid & (int)((word >> (c & 31)) & 1u);
                    if (__ballot(isin)) {
                        int len = 0;
                        if (isin) {
                            const uint64_t x = lds_u64(wb, c) ^ scan8;
                            len = x ? (int)(__builtin_ctzll(x) >> 3) : 8;
                            if (!x) {
                                while (len < kMaxMatch) {
                                    const uint64_t y = lds_u64(wb, qi + len) ^ lds_u64(wb, c + len);
                                    if (y) {
                                        len += (int)(__builtin_ctzll(y) >> 3);
                                        break;
                                    }
                                    len += 8;
                                }
                                len = len < kMaxMatch ? len : kMaxMatch;
                            }
                        }
                        found += isin;
                        const int better = isin & (len > best);
                        best = better ? len : best;
                        bdist = better ? d : bdist;
                        done |= (better & (len >= nice)) | (isin & (found >= chain));
                    }
                }
            }
            const uint32_t r = ((uint32_t)best << 16) | (uint32_t)bdist;
            res[k] = r;
            // the result of the sweep before, if that sweep searched the position (the slot is the position's own)
            const int slot = (int)(q & (RING - 1));
            agr[k] = q < ev_end && ring[slot] == r;
            if (act[k]) ring[slot] = r;
            // ---- 2. hops: the first index behind the group on the lane's path, by pointer doubling; a lane that is not
            //         searched (behind hi: the path ends there) points at itself, with hop length 0
            uint32_t P = act[k] ? (uint32_t)(self + fs_adv(r)) | ((uint32_t)fs_adv(r) << 16) : (uint32_t)self;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                dbl[k][j] = P;
                const int tgt = (int)(P & 0xFFFFu);
                const uint32_t f = (uint32_t)__builtin_amdgcn_ds_bpermute((tgt & 63) << 2, (int)P);
                const uint32_t fl = f >> 16;
                const uint32_t np = (f & 0xFFFFu) | ((fl ? fl : (P >> 16)) << 16);
                P = (tgt >> 6) == grp ? np : P;
            }
            ex[self] = P;
        }
        __syncthreads();  // -------- barrier 1: the exit table
        uint64_t topsm[PPT];
        int lin[PPT], entry[PPT];  // the hop that enters the group (0: none), the index it enters at (-1: the path does not)
        bool term_before[PPT];
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int grp = wave + NW * k, gbase = 64 * grp, self = gbase + lane;
            // ---- 3. the path from w0 up to this group
            int cur = w0r, Lin = 0;
            bool term = false;
            while (cur < gbase) {
                const uint32_t p = ex[cur];
                const int nj = (int)(p & 0xFFFFu);
                if (nj == cur) {
                    term = true;
                    break;
                }
                Lin = (int)(p >> 16), cur = nj;
            }
            cur = __builtin_amdgcn_readfirstlane(cur), Lin = __builtin_amdgcn_readfirstlane(Lin);
            const bool entered = !term && cur < gbase + 64 && gbase + 64 > w0r;
            lin[k] = Lin, entry[k] = entered ? cur : -1, term_before[k] = term;
            // the lanes on the path: from the entry, the doubling tables downwards
            bool top = false;
            if (entered) {
                int at = cur;
#pragma unroll
                for (int j = 5; j >= 0; j--) {
                    const int nx = (int)((uint32_t)__builtin_amdgcn_ds_bpermute((at & 63) << 2, (int)dbl[k][j]) & 0xFFFFu);
                    at = ((nx >> 6) == grp && nx <= self) ? nx : at;
                }
                top = at == self && act[k];
            }
            const uint64_t tm = __ballot(top);
            topsm[k] = tm;
            const uint64_t dis = tm & ~__ballot(agr[k]);
            if (lane == 0) {
                cnt[grp] = (uint32_t)__builtin_popcountll(tm);
                if (dis) atomicMin(&shv[0], (uint32_t)(gbase + (int)__builtin_ctzll(dis)));
                if (tm) atomicMax(&shv[1], (uint32_t)(gbase + 63 - (int)__builtin_clzll(tm)));
            }
        }
        __syncthreads();  // -------- barrier 2: the first loop-top with a new result, the last loop-top, the counts
        const int last_top = (int)shv[1];
        const int tstar = shv[0] != 0xFFFFFFFFu ? (int)shv[0] : last_top;
        const uint32_t r_star = ring[(g0 + tstar) & (RING - 1)], r_last = ring[(g0 + last_top) & (RING - 1)];
        const int64_t w0_new = g0 + tstar + fs_adv(r_star);
        const int Xr = last_top + fs_adv(r_last);  // where the path leaves the searched part of the window
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int grp = wave + NW * k, gbase = 64 * grp, self = gbase + lane;
            const uint64_t tm = topsm[k];
            // ---- 4. the final loop-tops' symbols, in order; block cuts every kBlockSyms symbols (Deflate.cs:910-948)
            if (gbase <= tstar && tm) {
                int before = lane < grp ? (int)cnt[lane] : 0;  // (NG <= 64)
                for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
                const uint64_t fin = tstar - gbase >= 63 ? tm : tm & ((2ull << (tstar - gbase)) - 1ull);
                if ((fin >> lane) & 1ull) {
                    const int64_t g = nsyms + before + __builtin_popcountll(fin & lanemask_lt());
                    const uint32_t r = res[k];
                    const bool match = fs_len(r) >= kMinMatch;
                    syms[s.sym_off + g] = match ? (((uint32_t)fs_dist(r) << 16) | (uint32_t)(fs_len(r) - 3)) : (uint32_t)wb[gi + self];
                    if ((g + 1) % kBlockSyms == 0) {
                        blk_end[s.blk_off + g / kBlockSyms] = (int32_t)(g0 + self + (match ? fs_len(r) : 1));
                        blk_top[s.blk_off + g / kBlockSyms] = (int32_t)(g0 + self);
                    }
                }
                if (lane == 0 && tstar < gbase + 64) shv[2] = (uint32_t)(before + (int)__builtin_popcountll(fin));
            }
            // ---- the next guess: the bits of this sweep's parse for the group's 64 positions
            if (gbase + 64 > w0r) {
                // the last loop-top at or below the lane, and what it inserts; no loop-top below the lane: the hop that enters
                // the group covers it (short matches insert their inside, Deflate.Fast.cs:81-104)
                const uint64_t below = tm & ((2ull << lane) - 1ull);
                const int ti = below ? 63 - (int)__builtin_clzll(below) : 0;
                const int span = __builtin_amdgcn_ds_bpermute(ti << 2, fs_inserted_span(res[k], lazy));
                bool ins = below ? (lane - ti) < span : (lin[k] >= kMinMatch && lin[k] <= lazy);
                if (self >= Xr || term_before[k]) ins = true;  // behind the path's end: not parsed yet
                if (g0 + self == preins) ins = true;
                uint64_t m = __ballot(ins);
                const int wi = (gi + gbase) >> 5;
                if (gbase < w0r) {  // the group of w0: what lies below it is final
                    const uint64_t keep = (1ull << (w0r - gbase)) - 1ull;
                    const uint64_t old = (uint64_t)bm[wi] | ((uint64_t)bm[wi + 1] << 32);
                    m = (m & ~keep) | (old & keep);
                }
                if (lane < 2) bm[wi + lane] = (uint32_t)(m >> (32 * lane));
            }
        }
        // a last match that reaches out of the window: its positions' bits, "inserted" behind it
        if (Xr > W && tid < 9) {
            const int span = fs_inserted_span(r_last, lazy);
            uint32_t v = 0;
            for (int b = 0; b < 32; b++) {
                const int idx = W + 32 * tid + b;
                v |= (uint32_t)((idx >= Xr || idx - last_top < span || g0 + idx == preins) ? 1 : 0) << b;
            }
            bm[((gi + W) >> 5) + tid] = v;
        }
        __syncthreads();  // -------- barrier 3: the bits
        nsyms += shv[2];
        if (tid == 0) shv[0] = 0xFFFFFFFFu, shv[1] = 0;
        // ---- 5. the links of what has become final, compressed (fs_compress); the next sweep's searches run beside this
        const int64_t c_end = w0_new < lo + fsLinks ? w0_new : lo + fsLinks;  // (a last match may reach out of the tile: those links stay as they are)
        for (int64_t c = w0 + tid; c < c_end; c += NT) {
            const int ci = (int)(c - lo);
            int c1 = ci, out = 0;
            for (;;) {
                const int l = wl[c1];
                if (!l) break;
                c1 -= l;
                if (c1 < min_i || ci - c1 > kMaxDist) break;
                if ((bm[c1 >> 5] >> (c1 & 31)) & 1u) {
                    out = ci - c1;
                    break;
                }
            }
            wl[ci] = (uint16_t)out;
            lk[c] = (uint16_t)out;
        }
        ev_end = hi;
        x_end = g0 + Xr;
        w0 = w0_new;
    }
    __syncthreads();
    // ---- leave: the bits that became final go back to the stream's bitmap (the tail engine restores its chains from them)
    if (t0 >= 0) {
        const int64_t lo = t0 - kFsBack;
        for (int64_t wd = (w0_staged >> 5) + tid; wd <= (w0 >> 5); wd += NT) {  // (x_end >= w0: the words of a tile left earlier are all there)
            uint32_t v = bm[wd - (lo >> 5)];
            if (wd == (w0 >> 5)) {  // nothing at or above the hand-over loop-top but the pending pre-insert
                v &= (1u << (w0 & 31)) - 1u;
                if (preins >= w0 && (preins >> 5) == wd) v |= 1u << (preins & 31);
            }
            gbits[wd] = v;
        }
        if (tid == 0 && preins >= w0 && (preins >> 5) != (w0 >> 5)) gbits[preins >> 5] = 1u << (preins & 31);
    }
    if (tid == 0) {
        StreamState &ss = st[blockIdx.x];
        ss.tail_p = (int32_t)w0;
        ss.tail_kind = kR;
        ss.tail_pend = 0;
        ss.k_done = k_fired;
        ss.preins = (int32_t)preins;
        ss.body_syms = (uint32_t)nsyms;
    }
}
