// zs_fast_sweep.hip -- KS: DeflateFast (levels 1-3, Deflate.Fast.cs:20-128) as window-wide sweeps of a workgroup.
// Included by zs_kernels.hip; zs_fast_sweep.h has the formulation and the code shared with the CPU model.
//
// One workgroup of 1024 threads per stream.  The tile -- bytes, K1's all-position links and the inserted-position bitmap of
// [t0 - 32 512, t0 + TILE) -- is staged in LDS once per TILE - W positions of progress.  A link entry in LDS is a halfword:
// bits 0..14 the distance to the previous position of the bucket (0x7FFF: none), bit 15 the position's own membership in the
// set, so a walker gets an entry's membership and the way on with one read.  A sweep (W = 1024 positions from g0 = w0
// rounded down to 64 on; a wave owns the group of 64 positions with its number):
//
//   1. search     every lane searches its position under the set as it stands (final below w0, the last sweep's parse
//                 behind it, "inserted" where nothing has been parsed yet): fs_search, one chain entry per trip, lanes
//                 leaving through the execution mask (the CU is bound by instruction issue here -- 16 waves, every one of
//                 their instructions costs the sweep 4 cycles; forms with fewer, fuller trips were measured and were
//                 slower, profiles/r04_fast_sweep_variants.md).  A window position's first step is its link under the
//                 guess (gl: the distance to the nearest entry the guess has in the set, made once per sweep for all
//                 walkers).  A search whose first candidate lies below w0 rests on final bits alone: its result is
//                 marked and the position is not searched again
//   2. hops       next(p) = p + 1 or p + match length.  Per lane, by six rounds of pointer doubling with ds_bpermute: the
//                 first index behind its group that its hops lead to, the length of the hop that leaves the group and the
//                 number of hops (the doubling tables stay in registers; the result goes to the exit table)  -- barrier 1
//   3. path       ONE wave follows the path from w0 through the exit table, whose rows it holds in registers (a hop from
//                 group to group is a v_readlane), and publishes per group: where the path enters it, the hop that enters,
//                 the loop-tops before it (the other waves wait: four waves share a SIMD's issue, and sixteen copies of the
//                 walk cost more than one)                                                                   -- barrier 2
//   4. loop-tops  per group: the lanes on the path, by walking the doubling tables down from the entry (six ds_bpermute);
//                 the set the parse implies for the group's 64 positions, compared with the guess (the bitmap): the first
//                 group in which they part (LDS atomic min), per group the last loop-top at or below its first
//                 difference                                                                                 -- barrier 3
//   5. final      the loop-tops at or below the first difference are final: their symbols leave in order; every group writes
//                 the bits of its 64 positions as this sweep's parse has them (loop-tops, the inside of short matches; what
//                 lies behind the path's end keeps its guess), into the bitmap and into the link entries     -- barrier 4
//   6. links      the links of the positions that became final are replaced by the distance to the nearest inserted position
//                 of their bucket (fs_compress; in LDS, and in the stream's link array in the stream form); the next window's
//                 positions get their links under the new guess (gl).  Walkers find the same candidates through either
//                 form of a link (zs_fast_sweep.h, fact 3)                                                  -- barrier 5
//
// What it leaves is what K4 / K5 leave for the lazy parse, so the tail engine (restored from the bitmap and the links --
// compressed or not, le_restore_prev_ins finds the same predecessor) and the block kernels go on unchanged.
//
// Two forms (template parameter CH).  CH = false: one workgroup takes a stream from its first position to fv_end, tile after
// tile; a batch of many streams fills the chip that way and parses every position once.  CH = true, the chunk form
// (zs_fast_sweep.h "Rounds"): one workgroup per chunk of a stream -- or per range of consecutive chunks, taken in turn -- and
// round.  The guess below a chunk's first loop-top comes from the chunks before it as the round before left them, or as they
// have just left them if they belong to the workgroup's own range (bit planes by chunk parity and FsMeta::cur; entry
// loop-tops, event cuts and the rounds of last run and last change in FsMeta); the guess from there on is what the chunk's own
// run before left.  K1's links are only read (the history's links are compressed in LDS after staging, an event's cut stays
// in LDS and FsMeta), the symbols go to the chunk's provisional buffer; a chunk none of whose inputs has changed since it
// last read them copies its FsMeta forward.  When a round has changed nothing, zs_fast_commit_scan_kernel /
// zs_fast_commit_kernel put the symbols, block cuts, bits, cuts and the stream's state where the stream form would have left
// them.  64 MiB of text: 1.38 / 1.93 GB/s at levels 1 / 3, 8 MiB 289 / 437 MB/s (one workgroup: 48 / 21 MB/s); kennedy.xls 2.8 ms
// against 67.

constexpr int kFsBack = 32512;   // >= kMaxDist, multiple of 64
constexpr int kFsFwd = 272;      // >= kMaxMatch + 8, multiple of 16
constexpr int kFsTile1 = 12288;  // positions per tile with one position per thread (W = 1024): 152.8 KiB of LDS
constexpr uint32_t kFsNoLink = 0x7FFFu;
template <int TILE>
struct FsLayout {
    static constexpr int bytes = kFsBack + TILE + kFsFwd, links = kFsBack + TILE, bit_words = links / 32 + 16;
};
template <int NT, int TILE>
constexpr int fs_lds_bytes() {
    return FsLayout<TILE>::bytes + 2 * FsLayout<TILE>::links + 2 * 4 * FsLayout<TILE>::bit_words + 4 * (2 * NT) + 4 * NT + 8 * (NT / 64) + 2 * (2 * NT) + 64;
}
// a pair of link halfwords as the link kernel left them -> the staged form (no membership bit yet); p = position of the low one
__device__ __forceinline__ uint32_t fs_stage_links(uint32_t pair, int p) {
    uint32_t a = pair & 0xFFFFu, b = pair >> 16;
    a = (a == 0 || p - (int)a < 1) ? kFsNoLink : a;  // position 0 is never a candidate
    b = (b == 0 || p + 1 - (int)b < 1) ? kFsNoLink : b;
    return a | (b << 16);
}
// the doubling word of a lane: index (12 bits) | last hop's length (9) | hops (7)
__device__ __forceinline__ int fs_pj(uint32_t p) { return (int)(p & 0xFFFu); }
__device__ __forceinline__ int fs_pl(uint32_t p) { return (int)((p >> 12) & 0x1FFu); }
__device__ __forceinline__ int fs_pc(uint32_t p) { return (int)(p >> 21); }

// Runs.  Inside a run of equal bytes every position's bucket is its neighbour's, the all-position chain goes down one position
// at a time, and a long match inserts none of them: a walker at an entry c that is not in the set and whose link is 1 would
// take a step per position (ptt5: 145 per search).  With a bit per position for "link is 1" the next 32 entries are looked
// at together: the stretch below c that is reachable by such links, and the highest member of the set in it.  Returns the
// index to go on from with link 1: one above that member, or one above the stretch's lowest position.
__device__ __forceinline__ int fs_run_skip(int c, const uint32_t *bm, const uint32_t *rb) {
    const int s = c - 1, sb = s & 31, wi = s >> 5;
    const uint32_t cont = (rb[wi] >> 1) | (rb[wi + 1] << 31);  // bit p: one can step from p + 1 to p
    const uint32_t t = cont << (31 - sb);
    const int k = __builtin_clz(~t | 1u) ;                      // positions s, s - 1, ... s - k + 1 are reachable (k >= 1: rb[c] is set)
    const int k1 = k < 1 ? 1 : (k > sb + 1 ? sb + 1 : k);
    const uint32_t reach = (k1 >= 32 ? 0xFFFFFFFFu : ((1u << k1) - 1u)) << (sb + 1 - k1);
    const uint32_t f = bm[wi] & reach;
    const int land = f ? 31 - __builtin_clz(f) : sb + 1 - k1;
    return (wi << 5) + land + 1;
}

// The chunk form's arguments (zs_fast_sweep.h "Rounds"): one workgroup per chunk and round.
struct FsRounds {
    const FsChunk *chunks;
    FsMeta *meta;         // [2][nch]: what the chunks left, by the parity of the round
    uint32_t *planes;     // [2][2][plane_words]: the chunks' bits, by FsMeta::cur and the parity of the chunk's number in the batch
    uint32_t *prov;       // the chunks' symbols until the rounds are over
    uint32_t *counters;   // per round: chunks that left something else than the round before
    int64_t plane_words;
    int nch, round;
    int no_seed;          // (ZS_FR_NO_SEED: the chunk's own bits of the run before are not its first guess)
    int range;            // consecutive chunks a workgroup takes in turn: a chunk reads what the chunks before it in its range have just
                          // left, and of the others what the round before left
};
static_assert(kFsChunkMax + 64 + 63 <= kFsTile1 - 1024 && kFsChunkMax + 64 + 63 + 1024 + 288 <= kFsTile1, "a chunk is covered by one staging of the tile");

template <int NT, int TILE, bool CH>
__global__ __launch_bounds__(NT) void zs_fast_sweep_kernel(const StreamDesc *sd, StreamState *st, uint16_t *link, uint32_t *syms, int32_t *blk_end,
                                                             int32_t *blk_top, LevelCfg lv, int strategy, FsRounds fr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int W = NT, NG = W / 64, RING = 2 * W;
    constexpr int fsBytes = FsLayout<TILE>::bytes, fsLinks = FsLayout<TILE>::links, fsBitWords = FsLayout<TILE>::bit_words;
    static_assert(TILE % 64 == 0 && TILE >= 2 * W && W + 258 < 4096 && NG <= 64, "tile / doubling word");
    uint8_t *wb = smem;                                         // bytes, index = position - lo
    uint16_t *wl = (uint16_t *)(smem + fsBytes);                // link entries
    uint32_t *bm = (uint32_t *)(smem + fsBytes + 2 * fsLinks);  // inserted bits, word k = positions [lo + 32 k, + 32)
    uint32_t *ring = bm + fsBitWords;                           // the last sweep's result of position q at q & (RING - 1)
    uint32_t *ex = ring + RING;                                 // window index -> doubling word of its way out of its group
    uint32_t *gent = ex + W;                                    // per group: entry index (0xFFF: none) | entering hop << 12 | path ended before << 21
    uint32_t *gbef = gent + NG;                                 // per group: loop-tops in the groups before it
    uint32_t *rb = gbef + NG;                                   // bit of a position: its link is 1 (the previous position of its bucket is its neighbour: runs)
    uint16_t *gl = (uint16_t *)(rb + fsBitWords);               // a window position's link under the guess: the distance to the nearest position of its bucket
                                                                // that the guess has in the set (at q & (RING - 1); 0x7FFF: none)
    // [0] first loop-top with a new result, [1] last loop-top, [2] final loop-tops of the sweep
    // (static: the compiler then knows the address space and the atomics below are LDS instructions, not flat ones)
    __shared__ uint32_t shv[4];
    __shared__ int32_t gtop[NG], gtle[NG];  // per group: its last loop-top, its last loop-top at or below the group's first difference (-1: none)
    const int tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform, and the compiler knows)
    // the chunk form: the workgroup's chunks in turn; the stream form: the stream
    const int kc0 = CH ? (int)blockIdx.x * fr.range : (int)blockIdx.x, kc1 = CH ? (kc0 + fr.range < fr.nch ? kc0 + fr.range : fr.nch) : kc0 + 1;
    for (int kc = kc0; kc < kc1; kc++) {
    FsChunk ck = FsChunk();
    if constexpr (CH) {
        // what the chunk before in the range wrote is read from here on: stores of this workgroup's own lanes, and a workgroup's
        // lanes share their CU's L1 -- the barrier's workgroup-scope fence is all it takes (an agent-scope fence writes the
        // XCD's L2 back: 60 us a time on this part, whatever the chunk does)
        if (kc > kc0) __syncthreads();
        ck = fr.chunks[kc];
    }
    const StreamDesc s = sd[CH ? ck.stream : kc];
    if (s.fv_end < 0) continue;
    const int n = s.n, body_end = CH ? ck.b_hi - 1 : s.fv_end;  // (a chunk's last loop-top lies below b_hi)
    const gcbytes in = as_global(s.in);
    uint16_t *lk = link + s.pos_off;
    uint32_t *gbits = s.ins_bits;
    uint32_t *out_syms = CH ? fr.prov + ck.prov_off : syms + s.sym_off;
    const int kl = s.kl;
    // ---- the chunk form: what the round before left -- the loop-top to start from, the chunks whose bits lie in reach (nearest
    //      first) with their first loop-tops, planes and cuts -- and whether any of it has changed
    __shared__ int32_t pE[64], pCut[64];
    __shared__ uint32_t pOff[64];
    __shared__ int32_t pvar[4];  // [0] chunks in reach, [1] something changed, [2] where the chunk before left
    int E0 = CH ? 0 : (int)s.start_pos, np = 0, cut_ev = -1, preins_ev = -1;  // (the stream form of a resumed run begins where the flush left the engine)
    const FsMeta *mp = nullptr;
    FsMeta *mn = nullptr;
    if constexpr (CH) {
        mp = fr.meta + (size_t)((fr.round + 1) & 1) * (size_t)fr.nch, mn = fr.meta + (size_t)(fr.round & 1) * (size_t)fr.nch;
        const int lo_read = ck.b_lo - kFsBack - 64;
        if (threadIdx.x < 64) {
            const int j = kc - 1 - (int)threadIdx.x;
            // a chunk of the workgroup's own range has run in this round already; of the others there is what the round before
            // left (nothing in round 0: their positions count as "inserted")
            bool reach = false, chg = false;
            if (j >= ck.first && fr.chunks[j].b_hi + kMaxMatch > lo_read && (fr.round > 0 || j >= kc0)) {
                reach = true;
                const FsMeta m = j >= kc0 ? mn[j] : mp[j];
                chg = fr.round > 0 && fs_stale(m.chg_round, j, mp[kc].ran_round, mp[kc].seen_lo);
                pE[threadIdx.x] = m.E, pCut[threadIdx.x] = m.cut;
                pOff[threadIdx.x] = (uint32_t)((size_t)(m.cur * 2 + (j & 1)) * (size_t)fr.plane_words + (size_t)(s.pos_off >> 5));
                if (threadIdx.x == 0) pvar[2] = m.X;
            }
            const uint64_t rm = __ballot(reach), cm = __ballot(chg);
            // (the chunks in reach are the nearest ones: lanes 0 .. np - 1)
            if (threadIdx.x == 0) pvar[0] = (int)__builtin_popcountll(rm), pvar[1] = (fr.round == 0 || cm != 0) ? 1 : 0, pvar[3] = (int)(rm & 1ull);
        }
        __syncthreads();
        np = pvar[0];
        if (!pvar[1]) {  // nothing it reads has changed: what it left stays
            if (threadIdx.x == 0) {
                FsMeta m = mp[kc];
                m.changed = 0;
                mn[kc] = m;
            }
            continue;
        }
        E0 = ck.idx == 0 ? (int)s.start_pos : (pvar[3] ? pvar[2] : ck.b_lo);  // (a resumed run's first chunk begins where the flush left the engine)
        if (E0 >= ck.b_hi) {  // (a last span shorter than the match that crosses it)
            if (threadIdx.x == 0) {
                const FsMeta o = mp[kc];
                const int chg = (fr.round == 0 || o.E != E0 || o.X != E0) ? 1 : 0;
                mn[kc] = FsMeta{E0, E0, 0, -1, ck.kfired0, -1, chg, fr.round ? o.cur : 0, chg ? fr.round : o.chg_round, fr.round, kc0, 0};
                if (chg) atomicAdd(&fr.counters[fr.round], 1u);
            }
            continue;
        }
    }
    // the positions from E0 on that the chunk's run before has bits for, and where they are
    int seed_lo = 0, seed_hi = 0;
    size_t seed_off = 0;
    if constexpr (CH) {
        if (fr.round > 0 && !fr.no_seed) {
            const FsMeta o = mp[kc];
            seed_lo = o.E > E0 ? o.E : E0, seed_hi = o.X;
            seed_off = (size_t)(o.cur * 2 + (kc & 1)) * (size_t)fr.plane_words + (size_t)(s.pos_off >> 5);
        }
    }
    // the bits of [p32, p32 + 32) below the chunk's first loop-top, from the planes of the chunks that own the positions
    auto hist_word = [&](int p32) -> uint32_t {
        if (p32 < 0) return 0u;
        uint32_t v = 0, seen = 0;
        for (int t = 0; t < np; t++) {
            const int a = pE[t] > p32 ? pE[t] : p32, b0 = t ? pE[t - 1] : E0, b = b0 < p32 + 32 ? b0 : p32 + 32;
            if (a < b) {
                const uint32_t m = (b - a >= 32 ? 0xFFFFFFFFu : ((1u << (b - a)) - 1u)) << (a - p32);
                v |= fr.planes[(size_t)pOff[t] + (size_t)(p32 >> 5)] & m, seen |= m;
            }
        }
        return v | ~seen;
    };
    const bool search = strategy != kHuffmanOnly;  // (HuffmanOnly: Longest_match is never called, Deflate.Fast.cs:61-66)
    const int nice = lv.nice, chain = lv.chain, lazy = lv.lazy;
    const bool aligned = (((uintptr_t)in) & 15) == 0;
    // the state that goes from sweep to sweep, held by every thread (all of it is computed from shared values); positions are
    // below 2^31 (StreamDesc::n)
    int w0 = E0, nsyms = 0, preins = -1, dead_pos = -1, only_pos = -1, ev_end = 0;
    int k_fired = CH ? ck.kfired0 : 0, next_cut = CH ? 0x7FFFFFFF : kBlockSyms - 1;  // (the symbol with this index ends a block, Deflate.cs:910-948; a chunk's symbols get their places later)
    // read event k fires at the first loop-top at or behind the data end before it - 261; the stream's events are its segment table
    // (one Write: window ends; several: Write ends too)
    auto trig_of = [&](int k) -> int { return k < s.nsegs ? s.seg_after[k - 1] - (kMinLookahead - 1) : -1; };
    int trigger = trig_of(k_fired + 1);
    int t0 = -(1 << 30), w0_staged = E0;
    int x_end = E0;  // where the last sweep's parse ended: the bits of [w0, x_end) are that parse's (the guess the results in the ring belong to)
    if (tid == 0) shv[0] = 0xFFFFFFFFu, shv[1] = 0, shv[2] = 0;
#ifdef ZS_FS_PROF
    long long pf[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_t = wall_clock64(), pf_sweeps = 0, pf_skips = 0, pf_cmps = 0;
#define FS_PF(i) { const long long now_ = wall_clock64(); pf[i] += now_ - pf_t; pf_t = now_; }
#else
#define FS_PF(i)
#endif
    // The distance from tile index ci to the nearest entry of its chain that is in the set (0: none within reach): what a walker
    // from there would step to.  For a final position this is its compressed link (fs_compress); for a window position it
    // is the same walk under the guess, done once per sweep instead of once per walker that comes by (kennedy.xls: 22 chain
    // entries per search, 6 of them to find the first member of the set).
    auto nearest_in_set = [&](int ci) -> int {
        int c1 = ci;
        uint32_t l = wl[ci] & 0x7FFFu;
        for (;;) {
            c1 -= (int)l;
            if (ci - c1 > kMaxDist || c1 < 6) return 0;  // (the tile's first entries are out of every walker's reach: a chunk compresses its whole history)
            const uint32_t v = wl[c1];
            if (v >> 15) return ci - c1;
            l = v & 0x7FFFu;
            if (l == 1u) c1 = fs_run_skip(c1, bm, rb);
        }
    };
    bool gl_stale = true;  // the guess's links have to be made again: a new tile (other positions have links), an event (a forced member, a cut)
    while (w0 <= body_end) {
        const int g0 = w0 & ~63;
        if (CH ? t0 < 0 : g0 + W > t0 + TILE) {
            // ---- (leave the tile: the bits that became final go back to the stream's bitmap) stage the tile at g0
            __syncthreads();
            if (t0 >= 0) {
                const int lo_old = t0 - kFsBack;
                for (int wd = (w0_staged >> 5) + tid; wd <= (x_end >> 5); wd += NT) gbits[wd] = bm[wd - (lo_old >> 5)];
                __syncthreads();
            }
            t0 = g0, w0_staged = w0, gl_stale = true;
            const int lo = t0 - kFsBack;
            for (int i = tid; i < fsBytes / 16; i += NT) {
                const int a = lo + i * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a >= 0 && a + 15 < n && aligned) {
                    const u32x4 t = *(gcu32x4)(in + a);
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                } else if (a + 15 >= 0 && a < n) {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 16; k++) {
                        const int b = a + k;
                        if (b >= 0 && b < n) t[k >> 2] |= (uint32_t)in[b] << (8 * (k & 3));
                    }
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
                ((uint4 *)wb)[i] = v;
            }
            // the set: the stream's bits below the last parse's end (final below w0, that parse's from there on), "inserted"
            // behind it (the guess for what no sweep has parsed)
            for (int i = tid; i < fsBitWords; i += NT) {
                const int p32 = lo + 32 * i;  // first position of the word
                uint32_t v = 0xFFFFFFFFu;
                if constexpr (CH) {
                    if (p32 + 32 <= x_end) v = hist_word(p32);
                    else if (p32 < x_end) v = hist_word(p32) | (0xFFFFFFFFu << (uint32_t)(x_end - p32));
                    // from the chunk's first loop-top on: what its own run before left -- a guess as good as any, and nearly the
                    // parse itself late in the rounds (a sweep then makes its whole window final)
                    if (fr.round > 0 && p32 + 32 > x_end) {
                        const int sa = seed_lo > p32 ? seed_lo : p32, sb = seed_hi < p32 + 32 ? seed_hi : p32 + 32;
                        if (sa < sb) {
                            const uint32_t m = (sb - sa >= 32 ? 0xFFFFFFFFu : ((1u << (sb - sa)) - 1u)) << (sa - p32);
                            v = (v & ~m) | (fr.planes[seed_off + (size_t)(p32 >> 5)] & m);
                        }
                    }
                } else {
                    if (p32 + 32 <= x_end) v = p32 >= 0 ? gbits[p32 >> 5] : 0u;
                    else if (p32 < x_end) v = gbits[p32 >> 5] | (0xFFFFFFFFu << (uint32_t)(x_end - p32));
                }
                bm[i] = v;
            }
            __syncthreads();
            // the link entries: distance (none: 0x7FFF) + the position's bit of the set
            for (int i = tid; i < fsLinks / 8; i += NT) {
                const int a = lo + i * 8;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (a >= 0 && a + 7 < n) {
                    v = *(const uint4 *)(lk + a);
                } else if (a + 7 >= 0 && a < n) {
                    uint32_t t[4] = {0, 0, 0, 0};
                    for (int k = 0; k < 8; k++) {
                        const int b = a + k;
                        if (b >= 0 && b < n) t[k >> 1] |= (uint32_t)lk[b] << (16 * (k & 1));
                    }
                    v = make_uint4(t[0], t[1], t[2], t[3]);
                }
                const uint32_t f = bm[i >> 2] >> (8 * (i & 3));  // the bits of positions 8 i .. 8 i + 7
                v.x = fs_stage_links(v.x, a), v.y = fs_stage_links(v.y, a + 2), v.z = fs_stage_links(v.z, a + 4), v.w = fs_stage_links(v.w, a + 6);
                ((uint8_t *)rb)[i] = (uint8_t)(((v.x & 0xFFFFu) == 1u) | (((v.x >> 16) == 1u) << 1) | (((v.y & 0xFFFFu) == 1u) << 2) | (((v.y >> 16) == 1u) << 3) |
                                               (((v.z & 0xFFFFu) == 1u) << 4) | (((v.z >> 16) == 1u) << 5) | (((v.w & 0xFFFFu) == 1u) << 6) | (((v.w >> 16) == 1u) << 7));
                v.x |= ((f & 1u) << 15) | ((f & 2u) << 30);
                v.y |= ((f & 4u) << 13) | ((f & 8u) << 28);
                v.z |= ((f & 16u) << 11) | ((f & 32u) << 26);
                v.w |= ((f & 64u) << 9) | ((f & 128u) << 24);
                ((uint4 *)wl)[i] = v;
            }
            if (tid < 16) rb[fsLinks / 32 + tid] = 0;
            __syncthreads();
            if constexpr (CH) {
                // the cuts the chunks before left (equal-bucket read events), then the history's links compressed under its set:
                // K1's links stay as they are for everybody else
                if (tid == 0)
                    for (int t = 0; t < np; t++) {
                        const int ci = pCut[t] - lo;
                        if (pCut[t] >= 0 && ci >= 0) wl[ci] |= kFsNoLink, rb[ci >> 5] &= ~(1u << (ci & 31));
                    }
                __syncthreads();
                for (int ci = 6 + tid; ci < w0 - lo; ci += NT) {
                    const int d = nearest_in_set(ci);
                    wl[ci] = (uint16_t)((d ? (uint32_t)d : kFsNoLink) | (wl[ci] & 0x8000u));
                }
                __syncthreads();
            }
            FS_PF(0);
        }
        const int lo = t0 - kFsBack;
        // ---- the read event at loop-top w0: w0 + 1 is inserted first (Deflate.cs:1010-1013).  One per 32 Ki positions.
        if (trigger >= 0 && w0 >= trigger) {
            const bool same = (wl[w0 + 1 - lo] & 0x7FFFu) == 1u;
            __syncthreads();
            k_fired++;
            preins = w0 + 1;
            if (same) dead_pos = w0, only_pos = w0 + 1;
            else dead_pos = w0 + 1, only_pos = -1;
            preins_ev = w0 + 1;
            if (same) cut_ev = w0;
            if (tid == 0) {
                bm[(w0 + 1 - lo) >> 5] |= 1u << ((w0 + 1 - lo) & 31);
                wl[w0 + 1 - lo] |= 0x8000u;
                if (same) {  // the reference's prev[w0] = w0 + 1, prev[w0 + 1] = w0
                    wl[w0 - lo] |= kFsNoLink, rb[(w0 - lo) >> 5] &= ~(1u << ((w0 - lo) & 31));
                    if (!CH) lk[w0] = 0;  // (a chunk's cut goes into K1's links once the rounds are over)
                }
            }
            trigger = trig_of(k_fired + 1);
            gl_stale = true;
            __syncthreads();
        }
        int hi = g0 + W;
        if (body_end + 1 < hi) hi = body_end + 1;
        if (trigger >= 0 && trigger < hi) hi = trigger;
        if (gl_stale) {
            const int g_hi = g0 + W < lo + fsLinks ? g0 + W : lo + fsLinks;
            for (int c = w0 + tid; c < g_hi; c += NT) {
                const int d = nearest_in_set(c - lo);
                gl[c & (RING - 1)] = (uint16_t)(d ? d : (int)kFsNoLink);
            }
            gl_stale = false;
            __syncthreads();
        }
        const int w0r = w0 - g0, hir = hi - g0, gi = g0 - lo;
        const int grp = wave, gbase = 64 * grp, self = gbase + lane;  // the lane's index in the window
        const int qi = gi + self, q = g0 + self;                       // ... in the tile, and its position
        const bool act = self >= w0r && self < hir;
        // ---- 1. search
        const int slot = q & (RING - 1);
        const uint32_t old = ring[slot];
        const bool dead = !search || q == dead_pos, only_prev = search && q == only_pos;
        const bool known = q < ev_end && (old & kFsExact) != 0 && !dead && !only_prev;
        int best = 2, bdist = 0;
        uint32_t exact = (dead || only_prev) ? kFsExact : 0u;
        {
            const uint64_t scan8 = lds_u64(wb, qi);
            bool walking = act && !dead && !known;
            int c = qi, rem = chain, maxd = kMaxDist;  // hash_head: <= MAX_DIST; later: cur_match > limit
            if (qi - (int)(wl[qi] & 0x7FFFu) < gi + w0r) exact = kFsExact;  // (no link: 0x7FFF, below anything)
            uint32_t l = gl[q & (RING - 1)];
            if (only_prev) l = 1, rem = 1;                 // the search sees only q - 1 (equal-bucket refill), whatever the set says
            while (__ballot(walking)) {
#ifdef ZS_FS_PROF
                pf_skips++;
#endif
                if (walking) {
                    const int nc = c - (int)l, d = qi - nc;
                    if (d > maxd) {
                        walking = false;
                    } else {
                        const uint32_t v = wl[nc];
                        c = nc, l = nc >= gi + w0r ? (uint32_t)gl[(lo + nc) & (RING - 1)] : v & 0x7FFFu;  // (inside the window: the guess's link)
                        if ((v >> 15) || only_prev) {
                            const uint64_t x = lds_u64(wb, nc) ^ scan8;
                            int len = (int)(__builtin_ctzll(x) >> 3);
                            if (!x) {
                                len = 8;
                                while (len < kMaxMatch) {
                                    const uint64_t y = lds_u64(wb, qi + len) ^ lds_u64(wb, nc + len);
                                    if (y) {
                                        len += (int)(__builtin_ctzll(y) >> 3);
                                        break;
                                    }
                                    len += 8;
                                }
                                len = len < kMaxMatch ? len : kMaxMatch;
                            }
                            maxd = kMaxDist - 1;
                            if (len > best) {
                                best = len, bdist = d;
                                if (len >= nice) walking = false;
                            }
                            if (--rem == 0) walking = false;
                        } else if (l == 1u) {
                            c = fs_run_skip(nc, bm, rb);
                        }
                    }
                }
            }
        }
        uint32_t r = ((uint32_t)best << 16) | (uint32_t)bdist | exact;
        if (known) r = old;
        if (act) ring[slot] = r;
        FS_PF(1);
        // ---- 2. hops: pointer doubling inside the group; a lane that is not searched (behind hi: the path ends there) points
        //         at itself, with hop length 0 and no hops
        uint32_t dbl[6];
        {
            const int adv = fs_adv(r);
            uint32_t P = act ? (uint32_t)(self + adv) | ((uint32_t)adv << 12) | (1u << 21) : (uint32_t)self;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                dbl[j] = P;
                const int tgt = fs_pj(P);
                const uint32_t f = (uint32_t)__builtin_amdgcn_ds_bpermute((tgt & 63) << 2, (int)P);
                const uint32_t np = (uint32_t)fs_pj(f) | ((uint32_t)(fs_pl(f) ? fs_pl(f) : fs_pl(P)) << 12) | ((uint32_t)(fs_pc(P) + fs_pc(f)) << 21);
                P = (tgt >> 6) == grp ? np : P;
            }
            ex[self] = P;
        }
        FS_PF(2);
        __syncthreads();  // -------- barrier 1: the exit table
        FS_PF(3);
        if (wave == 0) {
            // ---- 3. the path from w0 through the groups: a row of the table (the 64 exits of a group) per register, the hop from
            //         group to group a v_readlane
            uint32_t row[NG];
#pragma unroll
            for (int g = 0; g < NG; g++) row[g] = ex[64 * g + lane];
            // (everything below is uniform: scalar registers, the path's state in three of them; a group's entry goes into lane g
            // of two vector registers)
            int cur = w0r, before = 0;   // cur: 0xFFF once the path has ended
            uint32_t lin_term = 0;       // the hop that entered << 12 | path ended before << 21
            int mine = 0, mine_b = 0;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const bool entered = (cur >> 6) == g;
                const int pub = __builtin_amdgcn_readfirstlane((int)((entered ? (uint32_t)cur : 0xFFFu) | lin_term));
                mine = lane == g ? pub : mine, mine_b = lane == g ? __builtin_amdgcn_readfirstlane(before) : mine_b;
                if (entered) {
                    const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)row[g], cur & 63);
                    const int nj = fs_pj(p);
                    before += fs_pc(p);
                    if ((nj >> 6) == g) cur = 0xFFF, lin_term |= 1u << 21;  // an exit inside the group is a lane at which the path ends (it points at itself)
                    else cur = nj, lin_term = (uint32_t)fs_pl(p) << 12;
                }
            }
            if (lane < NG) gent[lane] = (uint32_t)mine, gbef[lane] = (uint32_t)mine_b;
        }
        FS_PF(4);
        __syncthreads();  // -------- barrier 2: where the path enters the groups
        FS_PF(5);
        // ---- 4. the lanes on the path: from the entry, the doubling tables downwards
        const uint32_t ge = gent[grp];
        const int entry = fs_pj(ge), lin = fs_pl(ge), before = (int)gbef[grp];
        const bool term_before = (ge >> 21) != 0;
        bool top = false;
        if (entry != 0xFFF) {
            int at = entry;
#pragma unroll
            for (int j = 5; j >= 0; j--) {
                const int nx = fs_pj((uint32_t)__builtin_amdgcn_ds_bpermute((at & 63) << 2, (int)dbl[j]));
                at = ((nx >> 6) == grp && nx <= self) ? nx : at;
            }
            top = at == self && act;
        }
        const uint64_t tm = __ballot(top);
        // the set this parse implies for the group's 64 positions: the last loop-top at or below the lane and what it inserts; no
        // loop-top below the lane: the hop that enters the group covers it (short matches insert their inside,
        // Deflate.Fast.cs:81-104)
        bool ins;
        {
            const uint64_t below = tm & ((2ull << lane) - 1ull);
            const int ti = below ? 63 - (int)__builtin_clzll(below) : 0;
            const int span = __builtin_amdgcn_ds_bpermute(ti << 2, fs_inserted_span(r, lazy));
            ins = below ? (lane - ti) < span : (lin >= kMinMatch && lin <= lazy);
            if (q == preins) ins = true;
        }
        // What is final (zs_fast_sweep.h fact 2): a loop-top's search looks at the set below itself only, so the loop-tops up to
        // the first position at which the implied set is not the guess they were searched under have the reference's results.
        // Per group: that position, the last loop-top at or below it, the last loop-top of all.
        const int wi = (gi + gbase) >> 5;
        const uint64_t guess = (uint64_t)bm[wi] | ((uint64_t)bm[wi + 1] << 32);
        {
            int pe = 64;  // the group's lanes below this one are covered by the path (the rest, if any, lies behind its end)
            if (term_before) pe = 0;
            else if (tm) {
                const int lt = 63 - (int)__builtin_clzll(tm);
                pe = lt + __builtin_amdgcn_readlane(fs_adv(r), lt);
            } else if (entry != 0xFFF) pe = entry & 63;
            uint64_t cmp = pe >= 64 ? ~0ull : ((1ull << pe) - 1ull);
            if (gbase < w0r) cmp = w0r - gbase >= 64 ? 0ull : cmp & ~((1ull << (w0r - gbase)) - 1ull);  // (below w0: final)
            const uint64_t diff = (__ballot(ins) ^ guess) & cmp;
            if (lane == 0) {
                const int dg = diff ? (int)__builtin_ctzll(diff) : 64;
                const uint64_t le = dg >= 63 ? tm : tm & ((2ull << dg) - 1ull);
                gtop[grp] = tm ? gbase + 63 - (int)__builtin_clzll(tm) : -1;
                gtle[grp] = le ? gbase + 63 - (int)__builtin_clzll(le) : -1;
                if (diff) atomicMin(&shv[0], (uint32_t)grp);
                if (tm) atomicMax(&shv[1], (uint32_t)(gbase + 63 - (int)__builtin_clzll(tm)));
            }
        }
        FS_PF(6);
        __syncthreads();  // -------- barrier 3: the first group in which guess and parse part, the last loop-top
        FS_PF(7);
        const int last_top = (int)shv[1];
        int tstar = last_top;
        if (shv[0] != 0xFFFFFFFFu) {  // the last loop-top at or below the first difference: in its group, or the last one of a group before
            int g = (int)shv[0];
            tstar = gtle[g];
            while (tstar < 0) tstar = gtop[--g];  // (the group of w0 has w0 itself)
        }
        const uint32_t r_star = ring[(g0 + tstar) & (RING - 1)], r_last = ring[(g0 + last_top) & (RING - 1)];
        const int w0_new = g0 + tstar + fs_adv(r_star);
        const int Xr = last_top + fs_adv(r_last);  // where the path leaves the searched part of the window
        // ---- 5. the final loop-tops' symbols, in order; block cuts every kBlockSyms symbols (Deflate.cs:910-948)
        if (gbase <= tstar && tm) {
            const uint64_t fin = tstar - gbase >= 63 ? tm : tm & ((2ull << (tstar - gbase)) - 1ull);
            if ((fin >> lane) & 1ull) {
                const int g = nsyms + before + (int)__builtin_popcountll(fin & lanemask_lt());
                const bool match = fs_len(r) >= kMinMatch;
                out_syms[g] = match ? (((uint32_t)fs_dist(r) << 16) | (uint32_t)(fs_len(r) - 3)) : (uint32_t)wb[qi];
                if (g == next_cut) {
                    blk_end[s.blk_off + g / kBlockSyms] = q + (match ? fs_len(r) : 1);
                    blk_top[s.blk_off + g / kBlockSyms] = q;
                }
            }
            if (lane == 0 && tstar < gbase + 64) shv[2] = (uint32_t)(before + (int)__builtin_popcountll(fin));
        }
        // ---- the next guess: the bits of this sweep's parse for the group's 64 positions; what lies behind the path's end keeps the
        //      guess it had ("inserted", or what the chunk's run before left there)
        {
            const bool parsed = self < Xr && !term_before;
            uint64_t m = __ballot(ins);
            const uint64_t pm = __ballot(parsed);
            m = (m & pm) | (guess & ~pm);
            if (gbase < w0r) {  // the group of w0: what lies below it is final
                const uint64_t keep = w0r - gbase >= 64 ? ~0ull : (1ull << (w0r - gbase)) - 1ull;
                m = (m & ~keep) | (guess & keep);
            }
            if (lane < 2) bm[wi + lane] = (uint32_t)(m >> (32 * lane));
            if (self >= w0r && parsed) {  // ... and in the position's link entry, where the walkers look
                const uint32_t v = wl[qi];
                wl[qi] = (uint16_t)((v & 0x7FFFu) | (ins ? 0x8000u : 0u));
            }
        }
        // a last match that reaches out of the window: its positions' bits, "inserted" behind it
        if (Xr > W) {
            const int span = fs_inserted_span(r_last, lazy);
            if (tid < 9) {
                uint32_t v = bm[((gi + W) >> 5) + tid];  // (behind the match's end the guess stays)
                for (int b = 0; b < 32; b++) {
                    const int idx = W + 32 * tid + b;
                    if (idx < Xr) v = (v & ~(1u << b)) | ((uint32_t)((idx - last_top < span || g0 + idx == preins) ? 1 : 0) << b);
                }
                bm[((gi + W) >> 5) + tid] = v;
            }
            // (the link entries carry the same bits as the bitmap, position for position: the searches read the one, the comparison
            // of guess and parse the other)
            if (W + tid < Xr && gi + W + tid < fsLinks) {  // (tid < 258; behind the tile's links: the next staging takes the bits)
                const int idx = W + tid;
                const bool ins = idx - last_top < span || g0 + idx == preins;
                const uint32_t v = wl[gi + idx];
                wl[gi + idx] = (uint16_t)((v & 0x7FFFu) | (ins ? 0x8000u : 0u));
            }
        }
        FS_PF(8);
        __syncthreads();  // -------- barrier 4: the bits
        nsyms += (int)shv[2];
        if (nsyms > next_cut) next_cut += kBlockSyms;
        if (tid == 0) shv[0] = 0xFFFFFFFFu, shv[1] = 0;
        // ---- the links of what has become final, compressed (fs_compress), and the links of the next window under the new
        //      guess; a walker that meets an entry another thread is rewriting finds the same candidates through either form
        {
            const int fin_hi = w0_new < lo + fsLinks ? w0_new : lo + fsLinks;  // (a last match may reach out of the tile: those links stay as they are)
            const int g_hi = w0_new + W < lo + fsLinks ? w0_new + W : lo + fsLinks;
            for (int c = w0 + tid; c < g_hi; c += NT) {
                const int ci = c - lo, d = nearest_in_set(ci);
                if (c < fin_hi) {
                    wl[ci] = (uint16_t)((d ? (uint32_t)d : kFsNoLink) | (wl[ci] & 0x8000u));
                    if (!CH) lk[c] = (uint16_t)d;
                } else {
                    gl[c & (RING - 1)] = (uint16_t)(d ? d : (int)kFsNoLink);
                }
            }
        }
        __syncthreads();  // -------- barrier 5: the links
        ev_end = hi;
        x_end = g0 + Xr;
        w0 = w0_new;
        FS_PF(9);
#ifdef ZS_FS_PROF
        pf_sweeps++;
#endif
    }
#ifdef ZS_FS_PROF
    if (tid == 0 && blockIdx.x == 0)
        printf("FSPROF n=%d sweeps=%lld skip trips (wave 0)=%lld compare trips=%lld ticks(100MHz): stage=%lld search=%lld hops=%lld barrier1=%lld path=%lld barrier2=%lld tops=%lld barrier3=%lld final+bits=%lld barrier4=%lld\n",
               n, pf_sweeps, pf_skips, pf_cmps, pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7], pf[8], pf[9]);
#endif
    __syncthreads();
    if constexpr (CH) {
        // ---- leave, chunk form: the bits of [E0, X) into the chunk's other plane; whether anything differs from what it left before
        const int X = w0, lo = t0 - kFsBack;
        const FsMeta o = mp[kc];
        const int cur = fr.round ? 1 - o.cur : 0;
        uint32_t *pn = fr.planes + (size_t)(cur * 2 + (kc & 1)) * (size_t)fr.plane_words + (size_t)(s.pos_off >> 5);
        const uint32_t *po = fr.planes + (size_t)((1 - cur) * 2 + (kc & 1)) * (size_t)fr.plane_words + (size_t)(s.pos_off >> 5);
        int diff = (fr.round == 0 || o.E != E0 || o.X != X || o.cut != cut_ev) ? 1 : 0;
        for (int wd = (E0 >> 5) + tid; wd <= ((X - 1) >> 5); wd += NT) {
            const uint32_t v = bm[wd - (lo >> 5)];
            const int a = E0 > 32 * wd ? E0 - 32 * wd : 0, b = X < 32 * wd + 32 ? X - 32 * wd : 32;
            const uint32_t m = (b - a >= 32 ? 0xFFFFFFFFu : ((1u << (b - a)) - 1u)) << a;
            if (fr.round && ((po[wd] ^ v) & m)) diff = 1;
            pn[wd] = v;
        }
        diff = __syncthreads_or(diff);
#ifdef ZS_FS_PROF
        FS_PF(9);
        if (tid == 0 && (kc == 40 || kc == 41) && (fr.round == 0 || fr.round == 3 || fr.round == 8))
            printf("FSPROF chunk %d round %d [%d, %d) sweeps=%lld ticks(100MHz): prologue+stage+compress=%lld search=%lld hops=%lld barrier1=%lld path=%lld barrier2=%lld tops=%lld barrier3=%lld final+bits=%lld barrier4+links+leave=%lld\n",
                   kc, fr.round, E0, X, pf_sweeps, pf[0], pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7], pf[8], pf[9]);
#endif
        if (tid == 0) {
            mn[kc] = FsMeta{E0, X, nsyms, cut_ev, k_fired, preins_ev, diff, cur, diff ? fr.round : o.chg_round, fr.round, kc0, 0};
            if (diff) atomicAdd(&fr.counters[fr.round], 1u);
        }
        continue;
    }
    // ---- leave: the bits that became final go back to the stream's bitmap (the tail engine restores its chains from them and
    //      the links)
    if (t0 >= 0) {
        const int lo = t0 - kFsBack;
        for (int wd = (w0_staged >> 5) + tid; wd <= (w0 >> 5); wd += NT) {  // (x_end >= w0: the words of a tile left earlier are all there)
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
        ss.tail_p = w0;
        ss.tail_kind = kR;
        ss.tail_pend = 0;
        ss.k_done = k_fired;
        ss.preins = preins;
        ss.body_syms = (uint32_t)nsyms;
    }
    }  // (the workgroup's chunks)
}

// ---- The rounds are over (a round changed nothing: the chunks' symbols, bits and hand-over loop-tops are the reference's).
// zs_fast_commit_scan_kernel, one workgroup per stream: every chunk's place in the stream's symbols, the stream's state for the
// kernels behind.
__global__ __launch_bounds__(1024) void zs_fast_commit_scan_kernel(const StreamDesc *sd, StreamState *st, const FsMeta *mf, int32_t *sym_base) {
    const StreamDesc s = sd[blockIdx.x];
    if (s.fv_end < 0 || s.fr_n <= 0) return;
    __shared__ int sh[1024];
    __shared__ int pre;
    const int tid = threadIdx.x;
    if (tid == 0) pre = -1;
    __syncthreads();
    int total = 0;
    for (int t0 = 0; t0 < s.fr_n; t0 += 1024) {
        const int t = t0 + tid;
        const int v = t < s.fr_n ? mf[s.fr_first + t].nsyms : 0;
        if (t < s.fr_n && mf[s.fr_first + t].preins >= 0) atomicMax(&pre, mf[s.fr_first + t].preins);
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // inclusive prefix sum
            const int x = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += x;
            __syncthreads();
        }
        if (t < s.fr_n) sym_base[s.fr_first + t] = total + sh[tid] - v;
        const int all = sh[1023];
        __syncthreads();
        total += all;
    }
    if (tid == 0) {
        const FsMeta last = mf[s.fr_first + s.fr_n - 1];
        StreamState &ss = st[blockIdx.x];
        ss.tail_p = last.X;
        ss.tail_kind = kR;
        ss.tail_pend = 0;
        ss.k_done = last.kend;
        ss.preins = pre;
        ss.body_syms = (uint32_t)total;
    }
}
// zs_fast_commit_kernel, one workgroup per chunk: the chunk's symbols to their places, the block cut that falls among them
// (every kBlockSyms symbols, Deflate.cs:910-948: its loop-top from the symbols' own lengths), its bits into the stream's bitmap
// (zeroed beforehand; chunks share words at their ends), its event's cut into K1's links -- what the tail engine and the block
// kernels read.
__global__ __launch_bounds__(256) void zs_fast_commit_kernel(const StreamDesc *sd, FsRounds fr, const FsMeta *mf, const int32_t *sym_base, uint16_t *link, uint32_t *syms,
                                                             int32_t *blk_end, int32_t *blk_top) {
    const FsChunk ck = fr.chunks[blockIdx.x];
    const StreamDesc s = sd[ck.stream];
    const FsMeta f = mf[blockIdx.x];
    const int base = sym_base[blockIdx.x], tid = threadIdx.x;
    const uint32_t *src = fr.prov + ck.prov_off;
    uint32_t *dst = syms + s.sym_off + base;
    for (int i = tid; i < f.nsyms; i += 256) dst[i] = src[i];
    const int g = base + (kBlockSyms - 1 - base % kBlockSyms);  // the first symbol index at or behind base that ends a block
    if (g < base + f.nsyms) {
        __shared__ int part[256];
        const int i_cut = g - base;
        int sum = 0;
        for (int i = tid; i < i_cut; i += 256) {
            const uint32_t sy = src[i];
            sum += (sy >> 16) ? (int)(sy & 0xFFFFu) + kMinMatch : 1;
        }
        part[tid] = sum;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) part[tid] += part[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t sy = src[i_cut];
            const int q = f.E + part[0];
            blk_end[s.blk_off + g / kBlockSyms] = q + ((sy >> 16) ? (int)(sy & 0xFFFFu) + kMinMatch : 1);
            blk_top[s.blk_off + g / kBlockSyms] = q;
        }
    }
    const uint32_t *pl = fr.planes + (size_t)(f.cur * 2 + ((int)blockIdx.x & 1)) * (size_t)fr.plane_words + (size_t)(s.pos_off >> 5);
    for (int wd = (f.E >> 5) + tid; f.X > f.E && wd <= ((f.X - 1) >> 5); wd += 256) {
        const int a = f.E > 32 * wd ? f.E - 32 * wd : 0, b = f.X < 32 * wd + 32 ? f.X - 32 * wd : 32;
        const uint32_t m = (b - a >= 32 ? 0xFFFFFFFFu : ((1u << (b - a)) - 1u)) << a;
        atomicOr(&s.ins_bits[wd], pl[wd] & m);
    }
    if (tid == 0) {
        if (f.cut >= 0) link[s.pos_off + f.cut] = 0;
        // nothing at or above the hand-over loop-top but the pending pre-insert
        if (f.preins >= 0 && f.preins >= mf[ck.first + s.fr_n - 1].X) atomicOr(&s.ins_bits[f.preins >> 5], 1u << (f.preins & 31));
    }
}
