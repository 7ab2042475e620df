// zs_model.cpp -- CPU walk-through of the GPU pipeline's algorithm (TEST TOOL).
//
// Runs, with plain host loops, the same stages the HIP kernels run (bucket
// links -> per-position matches for both chain budgets -> chunked lazy-parse
// maps -> compose -> symbol emission -> sequential tail engine -> blocks ->
// trees -> bit emission) using the shared ZS_HD code in zlibstream_amd/csrc,
// and checks every stage against the oracle (oracle/zs_oracle.c): the symbol
// stream, the block table and the final bytes.  It exists to validate the
// absolute-coordinate reformulation on a machine without a GPU; it is not part
// of the product.
//
// usage: zs_model <file> <level> [strategy] [chunk]      -> prints PASS/FAIL
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../oracle/zs_oracle.h"
#include "../../zlibstream_amd/csrc/zs_core.h"
#include "../../zlibstream_amd/csrc/zs_lit_engine.h"
#include "zs_fast_vec.h"
#include "../../zlibstream_amd/csrc/zs_fast_sweep.h"
#include "../../zlibstream_amd/csrc/zs_rle.h"

using namespace zs;

struct OracleTrace {
    std::vector<uint32_t> syms;
    struct Blk {
        int type, nsyms;
        int64_t start;
        int stored_len, eof;
        int64_t bit_start;
    };
    std::vector<Blk> blocks;
    std::vector<int64_t> read_pos;
};
static void on_symbol(void *u, int dist, int lc, int64_t) { ((OracleTrace *)u)->syms.push_back(((uint32_t)dist << 16) | (uint32_t)lc); }
static void on_block(void *u, int type, int nsyms, int64_t start, int stored_len, int eof, int64_t bit_start) {
    ((OracleTrace *)u)->blocks.push_back({type, nsyms, start, stored_len, eof, bit_start});
}
static void on_read(void *u, int64_t s, int, int, int64_t) { ((OracleTrace *)u)->read_pos.push_back(s); }

struct Model {
    const uint8_t *data;
    int64_t n;
    int level, strategy;
    LevelCfg lv;
    std::vector<uint32_t> crc_tab;
    std::vector<uint16_t> link;
    std::vector<uint32_t> mK, mK4;
    std::vector<uint32_t> syms;
    std::vector<BlockRec> blocks;
    std::vector<int64_t> events;  // s_k for k = 1..
    int64_t body_end;             // last body loop-top position (n - 262), or -1
    std::vector<int64_t> wr_end;
    std::vector<ReadEvent> rev;    // the stream's read events (zs_core.h build_read_events); rev[0] = the first read
    Geometry geo;                  // mode "chunk": segments, chunks and read-event clusters (zs_core.h build_geometry)
    int start_slot = 0;            // a run that takes the stream over in the middle (mode resume): chunk 0's entry slot ...
    std::vector<uint32_t> start_syms;  // ... and the symbols of the block in progress, in front of the run's own
    bool poisoned = false;         // the true path met a read the bulk form does not handle: the device falls back
    int flush_mode = 0;            // ZlibOptions.FlushMode of every Write
    std::vector<uint8_t> wr_flush;
    std::vector<int32_t> wr_blk;
    bool incremental = false;  // mode "inc": the literal engine run Write by Write (suspend / re-enter)
    std::vector<uint8_t> ins;  // mode "fvec" (DeflateFast for the lanes of a wave): the inserted positions

    uint32_t bucket(int64_t p) const {
        uint32_t v = (uint32_t)data[p + 2] | ((uint32_t)data[p + 3] << 8) | ((uint32_t)data[p + 4] << 16) | ((uint32_t)data[p + 5] << 24);
        return crc32c_u32_tab(crc_tab.data(), v) & kHashMask;
    }
    void build_links() {
        link.assign((size_t)n + 8, 0);
        std::vector<int64_t> head(kHashSize, -1);
        for (int64_t p = 0; p + 5 < n; p++) {
            uint32_t h = bucket(p);
            int64_t c = head[h];
            link[p] = (c >= 0 && p - c <= 32767) ? (uint16_t)(p - c) : 0;
            head[h] = p;
        }
    }
    int lcp(int64_t p, int64_t c) const {
        int len = 0;
        while (len < kMaxMatch && data[p + len] == data[c + len]) len++;
        return len;
    }
    // floor-2 walk for both budgets: shared with the device code
    void walk(int64_t p, uint32_t &outK, uint32_t &outK4) const {
        outK = outK4 = kNoMatch;
        if (lv.func != 2 || strategy == kHuffmanOnly) return;
        auto lk = [this](int64_t q) { return (int)link[q]; };
        auto lc = [this](int64_t a, int64_t c) { return lcp(a, c); };
        walk_matches(lk, lc, p, lv, outK, outK4);
    }
    uint32_t flt(uint32_t m) const { return m ? filter_match(match_len(m), match_dist(m), strategy) : kNoMatch; }
    void match_all() {
        mK.assign((size_t)n + 8, 0);
        mK4.assign((size_t)n + 8, 0);
        for (int64_t p = 1; p <= body_end; p++) walk(p, mK[p], mK4[p]);
    }
};

// ---- stage A: sequential parse with on-demand matches (validates the rules) ----
static void parse_sequential(Model &m, bool on_demand, int64_t &p_out, int &kind_out, uint32_t &pend_out, int &kdone_out,
                             int64_t &preins_out) {
    int kind = kR;
    int64_t p = 0;
    uint32_t pend = 0;
    int k_fired = 0;
    int kl = (int)m.rev.size() - 1;
    int64_t killed_pos = -1;
    int64_t preins = -1;
    int64_t block_start = 0;
    while (p <= m.body_end) {
        if (k_fired < kl && p >= m.rev[(size_t)k_fired + 1].at - (kMinLookahead - 1)) {
            k_fired++;
            m.events.push_back(p);
            preins = p + 1;
            if (m.bucket(p) == m.bucket(p + 1)) {
                killed_pos = p;
                m.link[p] = 0;  // later walkers stop after evaluating p
            } else {
                killed_pos = p + 1;
            }
        }
        uint32_t cK, cK4;
        if (p == killed_pos || p == 0) cK = cK4 = kNoMatch;
        else if (on_demand) m.walk(p, cK, cK4);
        else cK = m.mK[p], cK4 = m.mK4[p];
        cK = m.flt(cK), cK4 = m.flt(cK4);
        Step st = lazy_step(kind, p, pend, cK, cK4, m.lv);
        if (st.emit) {
            uint32_t sym = st.emit == 1 ? (uint32_t)m.data[p - 1] : (((uint32_t)st.dist << 16) | (uint32_t)(st.len - 3));
            m.syms.push_back(sym);
            if (m.syms.size() % kBlockSyms == 0) {
                int64_t end = st.emit == 1 ? p : p - 1 + st.len;
                BlockRec b;
                b.start = block_start;
                b.sym_start = (int64_t)m.syms.size() - kBlockSyms;
                b.stored_len = (int32_t)(end - block_start);
                b.nsyms = kBlockSyms;
                b.can_store = block_start >= m.rev[(size_t)k_fired].base;
                b.eof = 0;
                m.blocks.push_back(b);
                block_start = end;
            }
        }
        if (st.kind == kXK) pend = cK;
        else if (st.kind == kXK4) pend = cK4;
        kind = st.kind;
        p = st.pos;
    }
    p_out = p;
    kind_out = kind;
    pend_out = pend;
    kdone_out = k_fired;
    preins_out = preins;
    (void)block_start;
}


// ---- DeflateFast as the device runs it (zs_fast_vec.h): windows of 64 positions searched "in parallel" over the
//      all-position chains filtered by the inserted set, then the hops followed through the lanes' results ----
struct FvAcc {
    const Model *m;
    mutable long visits = 0, cmps = 0;
    int link(int64_t c) const { visits++; return (int)m->link[(size_t)c]; }
    bool ins(int64_t c) const { return m->ins[(size_t)c] != 0; }
    int lcp(int64_t q, int64_t c) const { cmps++; return m->lcp(q, c); }
};
// Statistics only (ZS_FV_W=<lanes>): what a wider window would buy -- positions per window and the longest walk in it.
// Read events are left out (one in 32 Ki positions).
static void fv_window_stats(Model &m, int W) {
    m.ins.assign((size_t)m.n + 512, 0);
    FvAcc acc{&m};
    std::vector<FvResult> res((size_t)W);
    long windows = 0, iters = 0;
    int64_t p0 = 0;
    while (p0 <= m.body_end) {
        int limit = W;
        if (m.body_end - p0 + 1 < limit) limit = (int)(m.body_end - p0 + 1);
        long mx = 0;
        for (int i = 0; i < limit; i++) {
            const long v0 = acc.visits;
            res[(size_t)i] = fv_search(acc, p0 + i, p0, m.lv.chain, m.lv.nice, false, false);
            if (acc.visits - v0 > mx) mx = acc.visits - v0;
        }
        int i = 0;
        while (i < limit) {
            const FvResult r = res[(size_t)i];
            if (i > 0 && r.touched) break;
            if (r.len >= kMinMatch) {
                const int k = r.len <= m.lv.lazy ? r.len : 1;
                for (int j = 0; j < k; j++) m.ins[(size_t)(p0 + i + j)] = 1;
                i += r.len;
            } else {
                m.ins[(size_t)(p0 + i)] = 1;
                i += 1;
            }
        }
        p0 += i;
        windows++, iters += mx;
    }
    printf("fvstat W=%d: %ld windows, %.1f positions per window, longest walk per window %.1f entries\n", W, windows, (double)p0 / windows,
           (double)iters / windows);
}

static void parse_fast_vec(Model &m, int64_t &p_out, int &kdone_out, int64_t &preins_out) {
    if (getenv("ZS_FV_W")) fv_window_stats(m, atoi(getenv("ZS_FV_W")));
    m.ins.assign((size_t)m.n + 128, 0);
    FvAcc acc{&m};
    const int kl = (int)m.rev.size() - 1;
    FvState st{0, 0, kl >= 1 ? m.rev[1].at - (kMinLookahead - 1) : -1, 0, -1, 0};
    int64_t block_start = 0;
    const bool search = m.strategy != kHuffmanOnly;
    long fv_windows = 0, fv_iters = 0, fv_max_iters = 0;
    while (st.p <= m.body_end) {
        const int64_t p0 = st.p;
        bool dead0 = false, dead1 = false, only1 = false;
        if (st.trigger >= 0 && p0 >= st.trigger) {
            // the read event at loop-top p0: p0 + 1 is inserted first (Deflate.cs:1010-1013)
            st.k_fired++;
            m.events.push_back(p0);
            st.preins = p0 + 1;
            m.ins[(size_t)p0 + 1] = 1;
            if (m.link[(size_t)p0 + 1] == 1) {
                dead0 = true, only1 = true;
                m.link[(size_t)p0] = 0;  // prev[p0] = p0 + 1, prev[p0 + 1] = p0: nothing older is reachable through them
            } else {
                dead1 = true;
            }
            st.trigger = st.k_fired < kl ? m.rev[(size_t)st.k_fired + 1].at - (kMinLookahead - 1) : -1;
        }
        int limit = kFvLanes;
        if (m.body_end - p0 + 1 < limit) limit = (int)(m.body_end - p0 + 1);
        FvResult res[kFvLanes];
        fv_max_iters = 0;
        for (int i = 0; i < limit; i++) {
            const long v0 = acc.visits;
            res[i] = fv_search(acc, p0 + i, p0, m.lv.chain, m.lv.nice, !search || (i == 0 && dead0) || (i == 1 && dead1), search && i == 1 && only1);
            if (acc.visits - v0 > fv_max_iters) fv_max_iters = acc.visits - v0;
        }
        auto rf = [&](int i) { return res[i]; };
        const FvWindow w = fv_resolve(rf, p0, limit, m.lv.lazy, st.trigger);
        for (int i = 0; i < limit; i++) {
            if (!((w.tops >> i) & 1)) continue;
            const int64_t q = p0 + i;
            const bool match = res[i].len >= kMinMatch;
            m.syms.push_back(match ? (((uint32_t)res[i].dist << 16) | (uint32_t)(res[i].len - 3)) : (uint32_t)m.data[q]);
            if (m.syms.size() % kBlockSyms == 0) {
                const int64_t end = q + (match ? res[i].len : 1);
                BlockRec b;
                b.start = block_start;
                b.sym_start = (int64_t)m.syms.size() - kBlockSyms;
                b.stored_len = (int32_t)(end - block_start);
                b.nsyms = kBlockSyms;
                b.can_store = block_start >= m.rev[(size_t)st.k_fired].base;
                b.eof = 0;
                m.blocks.push_back(b);
                block_start = end;
            }
        }
        for (int k = 0; k < 64; k++)
            if ((w.ins_lo >> k) & 1) m.ins[(size_t)p0 + k] = 1;
        for (int k = 0; k < 32; k++)
            if ((w.ins_hi >> k) & 1) m.ins[(size_t)p0 + 64 + k] = 1;
        st.p = p0 + w.advance;
        fv_windows++, fv_iters += fv_max_iters;
    }
    if (getenv("ZS_FV_STATS")) printf("fvec: %ld windows, %.1f positions per window, max-lane chain visits per window %.1f, visits per position %.1f, compares per position %.2f\n", fv_windows, (double)st.p / (double)(fv_windows ? fv_windows : 1), (double)fv_iters / fv_windows, (double)acc.visits / st.p, (double)acc.cmps / st.p);
    p_out = st.p;
    kdone_out = st.k_fired;
    preins_out = st.preins;
    if (getenv("ZS_MODEL_DEBUG_POS")) {  // the all-position chain of one position with the inserted flags
        int64_t q = atoll(getenv("ZS_MODEL_DEBUG_POS"));
        printf("chain of %lld (handed over at %lld, pre-insert %lld):", (long long)q, (long long)st.p, (long long)st.preins);
        for (int k = 0; k < 60 && m.link[(size_t)q]; k++) {
            q -= m.link[(size_t)q];
            printf(" %lld%s", (long long)q, m.ins[(size_t)q] ? "" : "(-)");
        }
        printf("\n");
    }
}


// ---- DeflateFast as window-wide sweeps (zs_fast_sweep.h): every sweep searches the positions from the first loop-top that
//      is not final on under the current guess of the inserted set, follows the parse through the results, and makes
//      final what lies in front of (and at) the first loop-top whose result is not the sweep before's.  The control flow is
//      the kernel's (zs_fast_sweep_kernel): windows aligned to 64 positions, searches up to the tile's end / the next
//      trigger / the last bulk loop-top, events at a sweep's first loop-top, links compressed behind the final part.
struct FsAcc {
    Model *m;
    mutable long visits = 0, cmps = 0;
    int link(int64_t c) const { visits++; return (int)m->link[(size_t)c]; }
    bool ins(int64_t c) const { return m->ins[(size_t)c] != 0; }
    int lcp(int64_t q, int64_t c) const { cmps++; return m->lcp(q, c); }
};
static void parse_fast_sweep(Model &m, int64_t &p_out, int &kdone_out, int64_t &preins_out) {
    const int W = getenv("ZS_FS_W") ? atoi(getenv("ZS_FS_W")) : 1024;
    const int TILE = getenv("ZS_FS_TILE") ? atoi(getenv("ZS_FS_TILE")) : 12288;
    m.ins.assign((size_t)m.n + 1024, 1);  // the guess for a position no sweep has parsed: inserted
    FsAcc acc{&m};
    const int kl = (int)m.rev.size() - 1;
    FsState st{0, 0, kl >= 1 ? m.rev[1].at - (kMinLookahead - 1) : -1, 0, -1, -1, -1, 0};
    const bool search = m.strategy != kHuffmanOnly;
    std::vector<uint32_t> rprev((size_t)m.n + 1024, kFsFresh), rcur((size_t)W + 64);
    int64_t block_start = 0, t0 = 0;
    long sweeps = 0, skipped = 0, searched = 0;
    std::vector<long> walk_hist;  // chain steps of a search by its distance from w0 (four bands of 256)
    while (st.w0 <= m.body_end) {
        int64_t g0 = st.w0 & ~63LL;
        if (g0 + W > t0 + TILE) t0 = g0;  // the tile is staged again from the window's first group on
        // the read event at loop-top w0: w0 + 1 is inserted first (Deflate.cs:1010-1013)
        if (st.trigger >= 0 && st.w0 >= st.trigger) {
            const int64_t t = st.w0;
            st.k_fired++;
            m.events.push_back(t);
            st.preins = t + 1;
            m.ins[(size_t)t + 1] = 1;
            if (m.link[(size_t)t + 1] == 1) {
                st.dead_pos = t, st.only_pos = t + 1;
                m.link[(size_t)t] = 0;  // prev[t] = t + 1, prev[t + 1] = t: nothing older is reachable through them
            } else {
                st.dead_pos = t + 1, st.only_pos = -1;
            }
            st.trigger = st.k_fired < kl ? m.rev[(size_t)st.k_fired + 1].at - (kMinLookahead - 1) : -1;
        }
        int64_t hi = std::min<int64_t>(std::min<int64_t>(g0 + W, t0 + TILE), m.body_end + 1);
        if (st.trigger >= 0 && st.trigger < hi) hi = st.trigger;  // (the trigger lies behind w0: the event at w0 has fired)
        // ---- search [w0, hi) under the guess
        for (int64_t q = st.w0; q < hi; q++) {
            // a search whose first candidate lies below w0 looks at final bits only: its result stays (kFsExact), and the
            // sweeps behind it do not search the position again
            if (q < st.ev_end && (rprev[(size_t)q] & kFsExact) && q != st.dead_pos && q != st.only_pos) {
                rcur[(size_t)(q - g0)] = rprev[(size_t)q];
                skipped++;
                continue;
            }
            const int l0 = m.link[(size_t)q];
            const bool exact = l0 == 0 || q - l0 < st.w0;
            const long v0 = acc.visits;
            rcur[(size_t)(q - g0)] = fs_search(acc, q, m.lv.chain, m.lv.nice, !search || q == st.dead_pos, search && q == st.only_pos) | (exact ? kFsExact : 0u);
            searched++;
            if (walk_hist.empty()) walk_hist.assign(4 * 64, 0);
            {
                const int b = (int)std::min<int64_t>(3, (q - st.w0) / 256);
                walk_hist[(size_t)b * 64 + (size_t)std::min<long>(63, acc.visits - v0)]++;
            }
        }
        // ---- the parse from w0: loop-tops, the first whose result is new, where the path leaves [w0, hi)
        int64_t t = st.w0, tstar = -1, last_top = -1;
        std::vector<int64_t> tops;
        while (t < hi) {
            const uint32_t r = rcur[(size_t)(t - g0)];
            tops.push_back(t);
            last_top = t;
            t += fs_adv(r);
        }
        const int64_t X = t;  // the path's first position at or behind hi
        // final: the loop-tops up to the first position at which the set this parse implies is not the guess it was searched
        // under (a loop-top's search looks at the set below itself only: zs_fast_sweep.h fact 2)
        {
            std::vector<uint8_t> nb((size_t)(X - st.w0), 0);
            for (int64_t q : tops) {
                const int span = fs_inserted_span(rcur[(size_t)(q - g0)], m.lv.lazy);
                for (int k = 0; k < span; k++) nb[(size_t)(q + k - st.w0)] = 1;
            }
            if (st.preins >= st.w0 && st.preins < X) nb[(size_t)(st.preins - st.w0)] = 1;
            int64_t d = X;
            for (int64_t q = st.w0; q < X; q++)
                if (nb[(size_t)(q - st.w0)] != m.ins[(size_t)q]) {
                    d = q;
                    break;
                }
            for (int64_t q : tops)
                if (q <= d) tstar = q;
        }
        const int64_t w0_new = tstar + fs_adv(rcur[(size_t)(tstar - g0)]);
        // ---- the final loop-tops' symbols; block cuts every kBlockSyms symbols (Deflate.cs:910-948)
        for (int64_t q : tops) {
            if (q > tstar) break;
            const uint32_t r = rcur[(size_t)(q - g0)];
            const bool match = fs_len(r) >= kMinMatch;
            m.syms.push_back(match ? (((uint32_t)fs_dist(r) << 16) | (uint32_t)(fs_len(r) - 3)) : (uint32_t)m.data[q]);
            if (m.syms.size() % kBlockSyms == 0) {
                const int64_t end = q + (match ? fs_len(r) : 1);
                BlockRec b;
                b.start = block_start;
                b.sym_start = (int64_t)m.syms.size() - kBlockSyms;
                b.stored_len = (int32_t)(end - block_start);
                b.nsyms = kBlockSyms;
                b.can_store = block_start >= m.rev[(size_t)st.k_fired].base;
                b.eof = 0;
                m.blocks.push_back(b);
                block_start = end;
            }
        }
        // ---- the next guess: the bits of this parse on [w0, X), "inserted" behind it
        for (int64_t q = st.w0; q < X; q++) m.ins[(size_t)q] = 0;
        for (int64_t q : tops) {
            const int span = fs_inserted_span(rcur[(size_t)(q - g0)], m.lv.lazy);
            for (int k = 0; k < span; k++) m.ins[(size_t)q + k] = 1;
        }
        if (st.preins >= st.w0) m.ins[(size_t)st.preins] = 1;
        for (int64_t q = st.w0; q < hi; q++) rprev[(size_t)q] = rcur[(size_t)(q - g0)];
        st.ev_end = hi;
        // ---- the links of what has become final, compressed
        for (int64_t c = st.w0; c < w0_new; c++) m.link[(size_t)c] = (uint16_t)fs_compress(acc, c);
        st.nsyms += 0;
        st.w0 = w0_new;
        sweeps++;
    }
    // what the tail engine reads: the set below the hand-over loop-top
    for (int64_t q = st.w0; q < m.n + 1024; q++) m.ins[(size_t)q] = 0;
    if (st.preins >= st.w0) m.ins[(size_t)st.preins] = 1;
    if (getenv("ZS_FV_STATS")) printf("fsweep: %ld sweeps, %.1f positions per sweep, %.2f searches (%.2f skipped: exact) %.2f chain steps and %.2f compares per position\n", sweeps, (double)st.w0 / (double)(sweeps ? sweeps : 1), (double)searched / (double)st.w0, (double)skipped / (double)st.w0, (double)acc.visits / (double)st.w0, (double)acc.cmps / (double)st.w0);
    if (getenv("ZS_FV_STATS") && !walk_hist.empty())
        for (int b = 0; b < 4; b++) {
            long tot = 0, sum = 0;
            for (int k = 0; k < 64; k++) tot += walk_hist[(size_t)b * 64 + k], sum += k * walk_hist[(size_t)b * 64 + k];
            long acc2 = 0;
            int p50 = 0, p90 = 0, p99 = 0, p999 = 0;
            for (int k = 0; k < 64; k++) {
                acc2 += walk_hist[(size_t)b * 64 + k];
                if (acc2 * 2 < tot) p50 = k + 1;
                if (acc2 * 10 < tot * 9) p90 = k + 1;
                if (acc2 * 100 < tot * 99) p99 = k + 1;
                if (acc2 * 1000 < tot * 999) p999 = k + 1;
            }
            printf("  searches %d-%d behind w0: %ld, chain steps mean %.1f p50 %d p90 %d p99 %d p99.9 %d (63 = more)\n", 256 * b, 256 * b + 255, tot, tot ? (double)sum / tot : 0.0, p50, p90, p99, p999);
        }
    p_out = st.w0;
    kdone_out = st.k_fired;
    preins_out = st.preins;
}


// ---- DeflateFast as rounds over the chunks of a stream (zs_fast_sweep.h, "Rounds"): what zs_fast_sweep_kernel does in its
//      chunk form and zs_fast_commit_kernel behind it.  Every chunk of a round reads what the round before left -- the
//      entry loop-top, the bits below it from the planes of the chunks that own them, the cuts of equal-bucket events --
//      and leaves its own; K1's links are only read (a chunk works on its own copy).
struct FrAcc {
    const Model *m;
    const uint16_t *lnk;
    const uint8_t *ins_;
    int64_t lo;
    int link(int64_t c) const { return c < lo ? 0 : (int)lnk[(size_t)c]; }
    bool ins(int64_t c) const { return ins_[(size_t)c] != 0; }
    int lcp(int64_t q, int64_t c) const { return m->lcp(q, c); }
};
static void parse_fast_rounds(Model &m, int64_t &p_out, int &kdone_out, int64_t &preins_out) {
    const int W = 1024, kBack = 32512;
    const int target = getenv("ZS_FR_CHUNK") ? atoi(getenv("ZS_FR_CHUNK")) : 4096;
    // a workgroup takes `range` consecutive chunks one after the other: a chunk reads what the chunks before it in its range
    // have just left (this round), and what the round before left of the others
    const int range_max = getenv("ZS_FR_RANGE") ? atoi(getenv("ZS_FR_RANGE")) : 1;
    const bool range_vary = getenv("ZS_FR_RANGE_VARY") != nullptr;  // ... another range every round (the engine chooses them by what the round before changed)
    const int kl = (int)m.rev.size() - 1;
    std::vector<FsChunk> ch;
    fs_build_chunks(0, m.body_end, kl, [&](int k) { return m.rev[(size_t)k].at - (kMinLookahead - 1); }, target, ch);
    const int nch = (int)ch.size();
    const size_t N = (size_t)m.n + 4096;
    std::vector<uint8_t> plane[2][2];
    for (auto &a : plane)
        for (auto &b : a) b.assign(N, 0);
    std::vector<FsMeta> meta[2];
    meta[0].assign((size_t)nch, FsMeta{-1, -1, 0, -1, 0, -1, 0, 0, 0, 0, 0, 0});
    meta[1] = meta[0];
    std::vector<std::vector<uint32_t>> prov((size_t)nch);
    std::vector<uint16_t> lnk(N, 0);
    std::vector<uint8_t> ins(N, 1);
    std::vector<uint32_t> rprev(N, kFsFresh), rcur((size_t)W + 64);
    const bool search = m.strategy != kHuffmanOnly;
    long runs = 0, runs_changed = 0, sweeps_total = 0;
    int round = 0;
    for (;; round++) {
        const std::vector<FsMeta> &mp = meta[(round + 1) & 1];
        std::vector<FsMeta> &mn = meta[round & 1];
        int nchanged = 0;
        const int range = range_vary ? 1 + (int)((uint32_t)(round * 2654435761u) >> 16) % range_max : range_max;
        for (int k = 0; k < nch; k++) {
            const FsChunk &c = ch[(size_t)k];
            const int64_t lo_read = (int64_t)c.b_lo - kBack - 64;
            const int k0 = k / range * range;
            auto view = [&](int j) -> const FsMeta & { return j >= k0 ? mn[(size_t)j] : mp[(size_t)j]; };
            auto valid = [&](int j) { return round > 0 || j >= k0; };
            bool act = round == 0;
            int j0 = k;
            for (int j = k - 1; j >= 0 && (int64_t)ch[(size_t)j].b_hi + kMaxMatch > lo_read && valid(j); j--) {
                j0 = j;
                if (round && fs_stale(view(j).chg_round, j, mp[(size_t)k].ran_round, mp[(size_t)k].seen_lo)) act = true;
            }
            if (!act) {
                mn[(size_t)k] = mp[(size_t)k];
                mn[(size_t)k].changed = 0;
                continue;
            }
            runs++;
            const int64_t E = k == 0 ? 0 : (valid(k - 1) ? view(k - 1).X : c.b_lo);
            const int64_t g00 = E & ~63LL, lo = std::max<int64_t>(0, g00 - kBack), top = std::min<int64_t>((int64_t)N, (int64_t)c.b_hi + 2048);
            // ---- stage: K1's links, the set below E as the chunks that own the positions left it, "inserted" from E on
            std::copy(m.link.begin() + lo, m.link.begin() + std::min<int64_t>(top, (int64_t)m.link.size()), lnk.begin() + lo);
            for (int64_t p = lo; p < E; p++) {
                uint8_t b = 1;
                for (int j = k - 1; j >= j0; j--)
                    if (view(j).E <= p) {
                        b = plane[view(j).cur][j & 1][(size_t)p];
                        break;
                    }
                ins[(size_t)p] = b;
            }
            // from E on: what the chunk's own run before left (a guess as good as any, and nearly right late in the rounds), "inserted"
            // where it has none
            for (int64_t p = E; p < top; p++) ins[(size_t)p] = 1;
            if (round && !getenv("ZS_FR_NO_SEED")) {
                const FsMeta &own = mp[(size_t)k];
                for (int64_t p = std::max<int64_t>(E, own.E); p < own.X; p++) ins[(size_t)p] = plane[own.cur][k & 1][(size_t)p];
            }
            for (int j = j0; j < k; j++)
                if (view(j).cut >= lo) lnk[(size_t)view(j).cut] = 0;
            FrAcc acc{&m, lnk.data(), ins.data(), lo};
            for (int64_t q = std::max<int64_t>(lo, 1); q < E; q++) lnk[(size_t)q] = (uint16_t)fs_compress(acc, q);
            for (int64_t q = E; q < top; q++) rprev[(size_t)q] = kFsFresh;
            // ---- the sweeps of the chunk
            FsState st{E, 0, -1, c.kfired0, -1, -1, -1, 0};
            st.trigger = st.k_fired < kl ? m.rev[(size_t)st.k_fired + 1].at - (kMinLookahead - 1) : -1;
            int64_t cut = -1, preins = -1;
            std::vector<uint32_t> &out = prov[(size_t)k];
            out.clear();
            while (st.w0 < c.b_hi) {
                const int64_t g0 = st.w0 & ~63LL;
                if (st.trigger >= 0 && st.w0 >= st.trigger) {
                    const int64_t t = st.w0;
                    st.k_fired++;
                    st.preins = preins = t + 1;
                    ins[(size_t)t + 1] = 1;
                    if (lnk[(size_t)t + 1] == 1) {
                        st.dead_pos = t, st.only_pos = t + 1;
                        lnk[(size_t)t] = 0;
                        cut = t;
                    } else {
                        st.dead_pos = t + 1, st.only_pos = -1;
                    }
                    st.trigger = st.k_fired < kl ? m.rev[(size_t)st.k_fired + 1].at - (kMinLookahead - 1) : -1;
                }
                int64_t hi = std::min<int64_t>(g0 + W, c.b_hi);
                if (st.trigger >= 0 && st.trigger < hi) hi = st.trigger;
                for (int64_t q = st.w0; q < hi; q++) {
                    if (q < st.ev_end && (rprev[(size_t)q] & kFsExact) && q != st.dead_pos && q != st.only_pos) {
                        rcur[(size_t)(q - g0)] = rprev[(size_t)q];
                        continue;
                    }
                    const int l0 = lnk[(size_t)q];
                    const bool exact = l0 == 0 || q - l0 < st.w0;
                    rcur[(size_t)(q - g0)] = fs_search(acc, q, m.lv.chain, m.lv.nice, !search || q == st.dead_pos, search && q == st.only_pos) | (exact ? kFsExact : 0u);
                }
                int64_t t = st.w0, tstar = -1, last_top = -1;
                std::vector<int64_t> tops;
                while (t < hi) {
                    const uint32_t r = rcur[(size_t)(t - g0)];
                    tops.push_back(t);
                    last_top = t;
                    t += fs_adv(r);
                }
                const int64_t X = t;
                {
                    std::vector<uint8_t> nb((size_t)(X - st.w0), 0);
                    for (int64_t q : tops) {
                        const int span = fs_inserted_span(rcur[(size_t)(q - g0)], m.lv.lazy);
                        for (int i = 0; i < span; i++) nb[(size_t)(q + i - st.w0)] = 1;
                    }
                    if (st.preins >= st.w0 && st.preins < X) nb[(size_t)(st.preins - st.w0)] = 1;
                    int64_t d = X;
                    for (int64_t q = st.w0; q < X; q++)
                        if (nb[(size_t)(q - st.w0)] != ins[(size_t)q]) {
                            d = q;
                            break;
                        }
                    for (int64_t q : tops)
                        if (q <= d) tstar = q;
                    sweeps_total++;
                }
                const int64_t w0_new = tstar + fs_adv(rcur[(size_t)(tstar - g0)]);
                for (int64_t q : tops) {
                    if (q > tstar) break;
                    const uint32_t r = rcur[(size_t)(q - g0)];
                    out.push_back(fs_len(r) >= kMinMatch ? (((uint32_t)fs_dist(r) << 16) | (uint32_t)(fs_len(r) - 3)) : (uint32_t)m.data[q]);
                }
                for (int64_t q = st.w0; q < X; q++) ins[(size_t)q] = 0;
                for (int64_t q : tops) {
                    const int span = fs_inserted_span(rcur[(size_t)(q - g0)], m.lv.lazy);
                    for (int i = 0; i < span; i++) ins[(size_t)q + i] = 1;
                }
                if (st.preins >= st.w0) ins[(size_t)st.preins] = 1;
                for (int64_t q = st.w0; q < hi; q++) rprev[(size_t)q] = rcur[(size_t)(q - g0)];
                st.ev_end = hi;
                for (int64_t q = st.w0; q < w0_new; q++) lnk[(size_t)q] = (uint16_t)fs_compress(acc, q);
                st.w0 = w0_new;
            }
            // ---- what it leaves
            const int64_t X = st.w0;
            const FsMeta &old = mp[(size_t)k];
            const int cur = round ? 1 - old.cur : 0;
            bool diff = round == 0 || old.E != E || old.X != X || old.cut != cut;
            std::vector<uint8_t> &pn = plane[cur][k & 1];
            const std::vector<uint8_t> &po = plane[1 - cur][k & 1];
            for (int64_t q = E; q < X; q++) {
                pn[(size_t)q] = ins[(size_t)q];
                if (round && po[(size_t)q] != ins[(size_t)q]) diff = true;
            }
            mn[(size_t)k] = FsMeta{(int32_t)E, (int32_t)X, (int32_t)out.size(), (int32_t)cut, st.k_fired, (int32_t)preins, diff ? 1 : 0, cur, diff ? round : old.chg_round, round, k0, 0};
            nchanged += diff;
        }
        runs_changed += nchanged;
        if (!nchanged) break;
        if (round > nch + 2) {
            printf("frounds: no fixed point after %d rounds of %d chunks\n", round, nch);
            break;
        }
    }
    // ---- commit: the symbols in order with the block cuts, the set, the cuts in the links, the events
    const std::vector<FsMeta> &mf = meta[round & 1];
    m.ins.assign((size_t)m.n + 1024, 0);
    int64_t block_start = 0;
    preins_out = -1;
    for (int k = 0; k < nch; k++) {
        const FsMeta &f = mf[(size_t)k];
        if (k && f.E != mf[(size_t)k - 1].X) printf("frounds: chunk %d starts at %d, its predecessor left at %d\n", k, f.E, mf[(size_t)k - 1].X);
        if (f.preins >= 0) m.events.push_back(f.preins - 1), preins_out = f.preins;
        if (f.cut >= 0) m.link[(size_t)f.cut] = 0;
        int64_t q = f.E;
        for (uint32_t sy : prov[(size_t)k]) {
            const int adv = (sy >> 16) ? (int)(sy & 0xFFFF) + 3 : 1;
            m.syms.push_back(sy);
            if (m.syms.size() % kBlockSyms == 0) {
                BlockRec b;
                b.start = block_start;
                b.sym_start = (int64_t)m.syms.size() - kBlockSyms;
                b.stored_len = (int32_t)(q + adv - block_start);
                b.nsyms = kBlockSyms;
                b.can_store = block_start >= m.rev[(size_t)f.kend].base;
                b.eof = 0;
                m.blocks.push_back(b);
                block_start = q + adv;
            }
            q += adv;
        }
        if (q != f.X) printf("frounds: chunk %d: its symbols end at %ld, it left at %d\n", k, (long)q, f.X);
        for (int64_t p = f.E; p < f.X; p++) m.ins[(size_t)p] = plane[f.cur][k & 1][(size_t)p];
    }
    p_out = mf[(size_t)nch - 1].X;
    kdone_out = mf[(size_t)nch - 1].kend;
    if (preins_out >= p_out) m.ins[(size_t)preins_out] = 1;
    if (getenv("ZS_FV_STATS")) printf("frounds: %d chunks of ~%d positions, %d rounds, %.2f runs per chunk, %.2f of them left something else than the run before, %.1f sweeps per run\n", nch, target, round + 1, (double)runs / (double)nch, (double)runs_changed / (double)nch, (double)sweeps_total / (double)runs);
}


// ---- CompressionStrategy.Rle without a sequential parse (zs_rle.h): every position's part in the parse from the first position
//      of its run; the loop-tops below the hand-over position in order, blocks cut every kBlockSyms symbols, then the tail engine
static void parse_rle_runs(Model &m, int64_t &p_out, int &kdone_out) {
    const int64_t H = rle_body_end(m.n);
    p_out = 0, kdone_out = 0;
    if (H < 0) return;
    const int kl = (int)m.rev.size() - 1;
    auto byte = [&](int64_t q) { return m.data[q]; };
    int64_t a = 0, block_start = 0, ph = -1;
    for (int64_t p = 0; p < m.n; p++) {
        if (p > 0 && m.data[p] != m.data[p - 1]) a = p;
        const int role = rle_role(byte, p, a);
        if (!role) continue;
        if (p >= H) {
            ph = p;
            break;
        }
        m.syms.push_back(role == 1 ? (uint32_t)m.data[p] : ((1u << 16) | (uint32_t)(role - 3)));
        if (m.syms.size() % kBlockSyms == 0) {
            const int64_t end = p + (role == 1 ? 1 : role);
            BlockRec b;
            b.start = block_start;
            b.sym_start = (int64_t)m.syms.size() - kBlockSyms;
            b.stored_len = (int32_t)(end - block_start);
            b.nsyms = kBlockSyms;
            b.can_store = block_start >= (int64_t)kWSize * rle_refills_fired_at(p, kl);
            b.eof = 0;
            m.blocks.push_back(b);
            block_start = end;
        }
    }
    p_out = ph;
    kdone_out = rle_refills_fired_at(H, kl);
}


// ---- statistics only (mode "spec"): a chunk's lazy parse started `warm` positions in front of it in the state "loop-top, nothing
//      pending" -- does it stand in the true parse's node at the first loop-top at or behind the chunk's beginning?  (What a
//      speculative form of the symbol kernel would need: SURVEY.md hard part 2 says yes on text, no on periodic data.)
static void spec_stats(Model &m) {
    const int64_t be = m.n - kMinLookahead;
    std::vector<uint8_t> kind_at((size_t)m.n + 8, 255);
    std::vector<uint32_t> pend_at((size_t)m.n + 8, 0);
    {
        int kind = kR;
        int64_t p = 0;
        uint32_t pend = 0;
        while (p <= be) {
            kind_at[(size_t)p] = (uint8_t)kind, pend_at[(size_t)p] = pend;
            const uint32_t cK = p ? m.flt(m.mK[(size_t)p]) : 0, cK4 = p ? m.flt(m.mK4[(size_t)p]) : 0;
            const Step st = lazy_step(kind, p, pend, cK, cK4, m.lv);
            pend = st.kind == kXK ? cK : st.kind == kXK4 ? cK4 : 0;
            kind = st.kind, p = st.pos;
        }
    }
    for (int warm : {64, 128, 256, 512, 1024}) {
        long chunks = 0, bad = 0;
        for (int64_t cs = kChunk - (kMinLookahead - 1); cs <= be; cs += kChunk) {
            int kind = kR;
            int64_t p = cs - warm < 1 ? 1 : cs - warm;
            uint32_t pend = 0;
            while (p < cs) {
                const uint32_t cK = m.flt(m.mK[(size_t)p]), cK4 = m.flt(m.mK4[(size_t)p]);
                const Step st = lazy_step(kind, p, pend, cK, cK4, m.lv);
                pend = st.kind == kXK ? cK : st.kind == kXK4 ? cK4 : 0;
                kind = st.kind, p = st.pos;
            }
            chunks++;
            if (p > be) continue;
            if (kind_at[(size_t)p] != kind || pend_at[(size_t)p] != pend) bad++;
        }
        printf("spec: warm-up %4d: %ld of %ld chunks do not stand in the true parse's node at their first loop-top\n", warm, bad, chunks);
    }
}

// ---- stage B: the chunked form the GPU runs ----
struct Sink {
    int64_t base;  // stream-global index of the chunk's first symbol
    std::vector<uint32_t> *syms;
    std::vector<int64_t> *blk_end, *blk_top;
    void operator()(int i, uint32_t sym, int64_t end, int64_t top) {
        int64_t idx = base + i;
        (*syms)[(size_t)idx] = sym;
        if ((idx + 1) % kBlockSyms == 0) {
            size_t b = (size_t)(idx / kBlockSyms);
            (*blk_end)[b] = end;
            (*blk_top)[b] = top;
        }
    }
};
struct ModelAcc {
    const Model *m;
    uint32_t mK(int64_t p) const { return m->flt(m->mK[p]); }
    uint32_t mK4(int64_t p) const { return m->flt(m->mK4[p]); }
    uint8_t byte(int64_t p) const { return m->data[p]; }
    uint32_t bucket(int64_t p) const { return m->bucket(p); }
    int run1(int64_t p) const { return m->lcp(p, p - 1); }
    int link(int64_t p) const { return (int)m->link[(size_t)p]; }
};
static ChunkCtx chunk_ctx(const Model &m, int c) {
    const Geometry &g = m.geo;
    ChunkCtx cx;
    cx.cs = g.cstart[(size_t)c], cx.ce = g.cstart[(size_t)c + 1];
    if (cx.ce > m.body_end + 1) cx.ce = m.body_end + 1;
    cx.cl = nullptr, cx.m = 0, cx.S = 0, cx.after = 0;
    const int h = g.head[(size_t)c];
    if (h) {
        const int k = h - 1;
        cx.cl = g.cl.data() + g.seg_cl[(size_t)k], cx.m = g.seg_cl[(size_t)k + 1] - g.seg_cl[(size_t)k];
        cx.S = g.seg_S[(size_t)k], cx.after = g.seg_after[(size_t)k];
    }
    return cx;
}
static int chunk_of(const Model &m, int64_t p) {  // last chunk whose start is <= p
    const std::vector<int32_t> &cs = m.geo.cstart;
    return (int)(std::upper_bound(cs.begin(), cs.end() - 1, (int32_t)p) - cs.begin()) - 1;
}
static void chunk_walk(Model &m, int c, int slot, int &exit_slot, int &nsyms, Sink *sink) {
    ModelAcc acc{&m};
    const ChunkCtx cx = chunk_ctx(m, c);
    if (sink) walk_chunk(acc, *sink, cx, slot, m.lv, m.strategy, exit_slot, nsyms);
    else {
        NullSink ns;
        walk_chunk(acc, ns, cx, slot, m.lv, m.strategy, exit_slot, nsyms);
    }
}
struct EvList {
    std::vector<std::pair<int64_t, bool>> *v;
    void operator()(int64_t p, bool eq) const { v->push_back({p, eq}); }
};

static void parse_chunked(Model &m, int64_t &p_out, int &kind_out, uint32_t &pend_out, int &kdone_out, int64_t &preins_out) {
    p_out = 0, kind_out = kR, pend_out = 0, kdone_out = 0, preins_out = -1;
    if (m.body_end < 0) return;
    const int nchunks = m.geo.nchunks();
    // K3: maps for every chunk and entry slot (embarrassingly parallel on the GPU)
    std::vector<uint32_t> maps((size_t)nchunks * kSlots);
    std::vector<uint32_t> tbl(kNodeExit3);
    ModelAcc macc{&m};
    auto chunk_maps = [&](int c) {
        // jump table of the chunk, three rows (R, L-or-XK, XK4) as K3 builds it with 512 threads and in-place jumping passes
        const ChunkCtx cx = chunk_ctx(m, c);
        const int64_t cs = cx.cs, ce = cx.ce;
        for (int64_t p = cs; p < ce; p++) {
            uint32_t r[3];
            node_step3_all(macc, p, cs, ce, m.lv, r[0], r[1], r[2]);  // what K3 runs; must agree with the per-node form
            for (int row = 0; row < 3; row++) {
                if (r[row] != node_step3(macc, row, p, cs, ce, m.lv)) { printf("node_step3_all differs at %ld row %d\n", (long)p, row); exit(1); }
                tbl[row * kChunk + (int)(p - cs)] = r[row];
            }
        }
        for (int r = 0; r < 4; r++)  // deliberately unfinished: chunk_exit_by_table3 must not depend on finished entries
            for (int row = 0; row < 3; row++)
                for (int64_t p = cs; p < ce; p++) {
                    int x = row * kChunk + (int)(p - cs);
                    uint32_t v = tbl[x];
                    if (node_succ(v) < kNodeExit3) tbl[x] = node_jump(v, tbl[node_succ(v)]);
                }
        for (int s = 0; s < kSlots; s++) {
            if (!slot_valid(cx, s, m.body_end)) { maps[(size_t)c * kSlots + s] = 0; continue; }
            uint32_t v = chunk_exit_by_table3(macc, tbl, cx, s, m.lv, m.strategy);
            int ex, ns;
            chunk_walk(m, c, s, ex, ns, nullptr);  // cross-check against the plain walk
            if (map_exit(v) != ex || map_count(v) != ns) {
                printf("chunk %d slot %d: table (%d,%d) walk (%d,%d)\n", c, s, map_exit(v), map_count(v), ex, ns);
                exit(1);
            }
            maps[(size_t)c * kSlots + s] = v;
        }
    };
    for (int c = 0; c < nchunks; c++) chunk_maps(c);
    // K4: resolve (one workgroup per stream, sequential over chunks)
    std::vector<int> entry(nchunks);
    std::vector<int64_t> symbase(nchunks);
    std::vector<char> stale(nchunks + 40, 0);
    int slot = m.start_slot, k_fired = 0;
    int64_t total = (int64_t)m.start_syms.size(), preins = -1;
    long n_dirty = 0, n_equal = 0, n_stale = 0, n_events = 0;
    for (int c = 0; c < nchunks; c++) {
        const ChunkCtx cx = chunk_ctx(m, c);
        const int64_t e0 = slot <= 256 ? cx.cs + slot : cx.cs;
        if (cx.m && e0 <= m.body_end) {
            k_fired = m.geo.head[(size_t)c] - 1;
            // the events on the true path from this slot, the cuts of the equal-bucket ones applied in stream order: a cut
            // changes records behind it, so the events behind it are looked for again after every repair
            size_t done = 0;
            for (;;) {
                std::vector<std::pair<int64_t, bool>> evs;
                EvList el{&evs};
                NullSink nsk;
                int kind, ns;
                int64_t pp;
                uint32_t flags;
                chunk_special_prefix(macc, nsk, cx, slot, m.lv, m.strategy, kind, pp, ns, flags, el);
                if (flags & kMapPoisonBit) m.poisoned = true;
                size_t i = 0, neq = 0;
                for (; i < evs.size(); i++)
                    if (evs[i].second && neq++ == done) break;
                if (i == evs.size()) {
                    for (auto &x : evs)
                        if (x.first > 0) m.events.push_back(x.first);
                    n_events += (long)evs.size();
                    if (!evs.empty()) preins = evs.back().first + 1;
                    break;
                }
                done++;
                const int64_t e = evs[i].first;
                n_equal++;
                uint32_t B = m.bucket(e);
                m.link[e] = 0;
                int64_t hi = e + kMaxDist;
                if (hi > m.body_end) hi = m.body_end;
                for (int64_t p = e + 1; p <= hi; p++) {
                    if (m.bucket(p) != B) continue;
                    uint32_t a = m.mK[p], b = m.mK4[p];
                    bool dirty = (a && p - match_dist(a) < e) || (b && p - match_dist(b) < e);
                    if (!dirty) continue;
                    n_dirty++;
                    uint32_t na, nb;
                    m.walk(p, na, nb);
                    if (na != a || nb != b) {
                        m.mK[p] = na, m.mK4[p] = nb;
                        const int cp = chunk_of(m, p);
                        stale[cp] = 1;
                        // the pending match of an X entry at the next chunk's first position reads m[p]
                        if (p + 1 == m.geo.cstart[(size_t)cp + 1]) stale[cp + 1] = 1;
                    }
                }
            }
        }
        entry[c] = slot;
        symbase[c] = total;
        int ex, ns;
        if (stale[c]) {
            n_stale++;
            chunk_walk(m, c, slot, ex, ns, nullptr);
        } else {
            uint32_t v = maps[(size_t)c * kSlots + slot];
            ex = map_exit(v), ns = map_count(v);
            if (v & kMapPoisonBit) m.poisoned = true;
        }
        slot = ex;
        total += ns;
    }
    // K5: emission (one lane per chunk on the GPU)
    m.syms.assign((size_t)total, 0);
    std::copy(m.start_syms.begin(), m.start_syms.end(), m.syms.begin());
    size_t nb = (size_t)(total / kBlockSyms);
    std::vector<int64_t> blk_end(nb), blk_top(nb);
    for (int c = 0; c < nchunks; c++) {
        // the way K5 does it: every chunk on its own from its true entry (the read-event prefix, then plain steps)
        Sink sk{symbase[c], &m.syms, &blk_end, &blk_top};
        int ex, ns;
        chunk_walk(m, c, entry[c], ex, ns, &sk);
        int64_t nxt = c + 1 < nchunks ? symbase[c + 1] : total;
        if (symbase[c] + ns != nxt) { printf("chunk %d: %d symbols, maps said %ld\n", c, ns, (long)(nxt - symbase[c])); exit(1); }
    }
    int64_t bs = 0;
    for (size_t b = 0; b < nb; b++) {
        BlockRec r;
        r.start = bs;
        r.sym_start = (int64_t)b * kBlockSyms;
        r.stored_len = (int32_t)(blk_end[b] - bs);
        r.nsyms = kBlockSyms;
        int fired = refills_fired_at(blk_top[b], 1 << 30);
        r.can_store = bs >= (int64_t)kWSize * fired;
        r.eof = 0;
        m.blocks.push_back(r);
        bs = blk_end[b];
    }
    int64_t ce = m.body_end + 1;
    kind_out = slot <= 256 ? kR : slot - 256;
    p_out = slot <= 256 ? ce + slot : ce;
    pend_out = kind_out == kXK ? m.flt(m.mK[p_out - 1]) : kind_out == kXK4 ? m.flt(m.mK4[p_out - 1]) : 0;
    kdone_out = k_fired;
    preins_out = preins;
    fprintf(stderr, "  chunked: chunks=%d segs=%d events=%ld equal_events=%ld dirty=%ld stale_chunks=%ld%s\n", nchunks, m.geo.nsegs(), n_events, n_equal, n_dirty,
            n_stale, m.poisoned ? " POISONED" : "");
}

static void run_tail(Model &m, int64_t p, int kind, uint32_t pend, int k_done, int64_t preins) {
    LitEngine e;
    memset(&e, 0, sizeof e);
    le_defaults(e);
    std::vector<uint8_t> window(kWindowSize + 512);
    std::vector<uint16_t> head(kHashSize), prev(kWSize);
    e.window = window.data();
    e.head = head.data();
    e.prev = prev.data();
    e.crc_tab = m.crc_tab.data();
    e.data = m.data;
    e.n = m.n;
    e.lv = m.lv;
    e.strategy = m.strategy;
    e.hash_variant = kHashCrc32c;
    e.wr_end = (m.wr_end.size() > 1 || m.flush_mode) ? m.wr_end.data() : nullptr;
    e.n_wr = (int)m.wr_end.size();
    m.wr_flush.assign(m.wr_end.size(), (uint8_t)m.flush_mode);
    m.wr_blk.assign(m.wr_end.size(), 0);
    if (m.flush_mode) e.wr_flush = m.wr_flush.data(), e.wr_blk = m.wr_blk.data();
    size_t body_syms = m.syms.size();
    m.syms.resize(body_syms + 2 * kMinLookahead + 600 + (m.body_end < 0 ? (size_t)m.n : 0));
    e.syms = m.syms.data();
    e.nsyms = (int64_t)body_syms;
    size_t body_blocks = m.blocks.size();
    m.blocks.resize(body_blocks + 8 + m.wr_end.size() + (m.body_end < 0 ? (size_t)m.n / 8000 : 0));
    e.blocks = m.blocks.data();
    e.nblocks = (int)body_blocks;
    e.block_start_abs = body_blocks ? m.blocks[body_blocks - 1].start + m.blocks[body_blocks - 1].stored_len : 0;
    e.block_sym_start = (int64_t)body_blocks * kBlockSyms;
    e.block_syms = m.level == 0 ? (kLitBufsize / 2) - 1 : kBlockSyms;
    int64_t base_in = m.rev.empty() ? 0 : m.rev[(size_t)k_done].base;
    int64_t after_in = m.rev.empty() ? 0 : m.rev[(size_t)k_done].after;
    if (m.geo.nsegs() > 0) base_in = m.geo.seg_base[(size_t)k_done], after_in = m.geo.seg_after[(size_t)k_done];
    const bool preslid = m.lv.func == 2 && m.strategy != kRle && m.strategy != kHuffmanOnly && m.ins.empty() && !m.incremental &&
                         !getenv("ZS_NO_TAIL_RECORDS") && le_tail_preslide(e, p, base_in, after_in, preins);
    if (preslid) base_in += kWSize;  // as zs_tail_kernel: restored in the slid state
    le_restore(e, p, base_in, after_in, kind, pend, m.link.data(), preins, 0, 1, preslid);
    // the tail's searches ahead of its parse (le_tail_record) and the engine without hash heads (LitEngine::no_head), under
    // the conditions of zs_tail_kernel: a slow level, one Write, everything read, no pre-insert pending
    const bool use_rec = m.lv.func == 2 && m.strategy != kRle && m.strategy != kHuffmanOnly && !m.incremental && le_tail_reads_done(e) &&
                         preins < p && m.ins.empty() && !getenv("ZS_NO_TAIL_RECORDS") &&
                         le_tail_record_end(e) > p && le_tail_record_end(e) - p <= kTailRecMax;
    if (e.avail_end > 0) {
        int64_t lo = p - (kWSize - 1);
        if (lo < e.base) lo = e.base;
        if (lo < 0) lo = 0;
        auto insf = [&](int64_t c) { return m.ins[(size_t)c] != 0; };
        for (int64_t q = lo; q < p && q + 5 < m.n; q++) {
            if (!m.ins.empty()) {  // DeflateFast: only what was inserted is in the chains
                if (!m.ins[(size_t)q]) continue;
                le_restore_prev_ins(e, q, m.link.data(), insf);
            } else {
                le_restore_prev(e, q, m.link.data());
            }
            if (!(use_rec && e.final_run && !e.wr_end && le_no_head_ok(e))) e.head[le_bucket(e, q)] = (uint16_t)(q - e.base);  // increasing q: last writer = max
        }
        if (!m.ins.empty()) le_restore_finish(e, p, m.link.data(), preins, insf);
        else le_restore_finish(e, p, m.link.data(), preins);
    }
    std::vector<uint32_t> pre;
    if (use_rec) {
        const int64_t hi = le_tail_record_end(e);
        for (int64_t q = p; q < hi; q++) le_restore_prev(e, q, m.link.data());
        if (e.final_run && !e.wr_end && le_no_head_ok(e)) {  // (as zs_tail_kernel: only where the stream ends in one Write)
            for (int64_t q = p; q < e.n - 5; q++) le_restore_prev(e, q, m.link.data());
            int64_t lo = p - (kWSize - 1);
            if (lo < e.base) lo = e.base;
            if (lo < 1) lo = 1;
            for (int k = 0; k < 3; k++) {
                const uint32_t hb = le_tail_head_bucket(e, k);
                e.tail_head[k] = 0;
                for (int64_t q = lo; q < e.n - 5 + k; q++)
                    if (le_bucket(e, q) == hb) e.tail_head[k] = (int)(q - e.base);
            }
            e.no_head = 1;
            if (getenv("ZS_FV_STATS")) printf("tail heads (window index) %d %d %d, base %ld, p %ld n %ld\n", e.tail_head[0], e.tail_head[1], e.tail_head[2], (long)e.base, (long)p, (long)e.n);
            std::fill(head.begin(), head.end(), (uint16_t)0xDEAD);  // not to be looked at
        }
        {
            pre.assign((size_t)(2 * (hi - p)), 0);
            for (int64_t q = p; q < hi; q++) {
                pre[(size_t)(2 * (q - p))] = le_tail_record(e, (int)(q - e.base), (int)(e.n - q), m.lv.chain);
                pre[(size_t)(2 * (q - p)) + 1] = le_tail_record(e, (int)(q - e.base), (int)(e.n - q), m.lv.chain >> 2);
            }
            e.pre_rec = pre.data(), e.pre_lo = p, e.pre_hi = hi;
            if (getenv("ZS_FV_STATS")) printf("tail records for [%ld, %ld) of n = %ld\n", (long)p, (long)hi, (long)e.n);
        }
    }
    if (m.incremental && !m.wr_end.empty()) {
        // incremental stream: one run per Write (the engine suspends where Deflate.Compress returns for more input and is
        // re-entered with the next Write), then the Finish call as a run of its own without input
        for (size_t r = 0; r <= m.wr_end.size(); r++) {
            const bool last = r == m.wr_end.size();
            e.final_run = last ? 1 : 0;
            e.suspended = 0;
            e.cur_wr = 0;
            if (last) {
                e.wr_end = nullptr, e.n_wr = 1, e.wr_flush = nullptr, e.wr_blk = nullptr;
            } else {
                e.wr_end = m.wr_end.data() + r, e.n_wr = 1;
                e.wr_flush = m.wr_flush.data() + r;  // a NoFlush Write has mode 0
                e.wr_blk = m.wr_blk.data() + r;
                m.wr_blk[r] = e.nblocks;
                e.n = m.wr_end[r];
            }
            le_run(e, m.level, 0, 1);
            if (!last && !e.suspended) printf("run %zu did not suspend\n", r);
        }
    } else {
        le_run(e, m.level, 0, 1);
    }
    m.syms.resize((size_t)e.nsyms);
    m.blocks.resize((size_t)e.nblocks);
}


// ---- stage C: blocks -> trees -> bit offsets -> bytes, the way the GPU does it ----
struct BitOr {
    std::vector<uint8_t> *out;
    int64_t pos;  // bit position
    void operator()(unsigned value, int nbits) {
        for (int i = 0; i < nbits; i++, pos++)
            if ((value >> i) & 1u) (*out)[(size_t)(pos >> 3)] |= (uint8_t)(1u << (pos & 7));
    }
};
static std::vector<uint8_t> emit_stream(Model &m) {
    size_t nb = m.blocks.size();
    std::vector<TreeWork> tw(nb);
    std::vector<int> type(nb);
    std::vector<int64_t> bits(nb), bit_start(nb);
    std::vector<int64_t> sym_off(nb);
    int64_t so = 0;
    for (size_t b = 0; b < nb; b++) {  // K7: one workgroup per block
        so = m.blocks[b].sym_start;
        sym_off[b] = so;
        TreeWork &w = tw[b];
        memset(&w, 0, sizeof w);

        for (int i = 0; i < m.blocks[b].nsyms; i++) {
            uint32_t sy = m.syms[(size_t)(so + i)];
            int dist = (int)(sy >> 16), lc = (int)(sy & 0xFFFF);
            if (dist == 0) w.ltree[lc].fc++;
            else {
                w.ltree[length_code(lc) + kLiterals + 1].fc++;
                w.dtree[dist_code(dist - 1)].fc++;
            }
        }
        w.ltree[kEndBlock].fc = 1;
        so += m.blocks[b].nsyms;
        std::vector<uint32_t> hk(kHeapSize + 8);
        type[b] = build_block_trees(w, hk.data(), m.blocks[b].stored_len, m.blocks[b].can_store != 0, m.strategy);
        if (m.level == 0) type[b] = m.blocks[b].can_store ? 0 : 1;
        bits[b] = type[b] == 1 ? 3 + w.static_len : type[b] == 2 ? 3 + w.opt_len : 0;
    }
    int64_t pos = 16;  // after the 2-byte zlib header
    std::vector<uint8_t> out;
    for (int pass = 0; pass < 2; pass++) {  // K8: sequential per stream; the second pass writes the flush markers
        FlushAcct fa;
        fa_init(fa, 512, m.level, false);
        fa_enter(fa);
        size_t w = 0;
        auto put = [&](int64_t at, uint32_t v, int nbits) {
            if (!pass) return;
            BitOr o{&out, at};
            o(v, nbits);
        };
        for (size_t b = 0; b < nb; b++) {
            while (m.flush_mode && w + 1 < m.wr_blk.size() && (size_t)m.wr_blk[w + 1] <= b) w++, fa_enter(fa);
            pos = fa.bits;
            bit_start[b] = pos;
            if (type[b] == 0) {
                pos += 3;
                pos = (pos + 7) & ~7LL;
                pos += 32 + 8LL * m.blocks[b].stored_len;
            } else {
                pos += bits[b];
            }
            if (m.blocks[b].eof & 1) pos = (pos + 7) & ~7LL;
            fa.bits = pos;
            fa.last_eob_len = type[b] == 0 ? 8 : type[b] == 1 ? 7 : tw[b].ltree[kEndBlock].dl;
            const int f = m.blocks[b].eof >> 1;
            if (f) fa_end_of_write(fa, f, put);
            else fa_after_block(fa);
        }
        pos = fa.bits;
        if (!pass) out.assign((size_t)(pos / 8) + 4 + 8, 0);
    }
    unsigned hdr = zlib_header(m.level);
    out[0] = (uint8_t)(hdr >> 8);
    out[1] = (uint8_t)hdr;
    for (size_t b = 0; b < nb; b++) {  // K9: one workgroup per block
        BitOr put{&out, bit_start[b]};
        put((unsigned)(type[b] << 1) + ((m.blocks[b].eof & 1) ? 1u : 0u), 3);
        if (type[b] == 0) {
            put.pos = (put.pos + 7) & ~7LL;
            unsigned len = (unsigned)m.blocks[b].stored_len;
            put(len & 0xFFFF, 16);
            put(~len & 0xFFFF, 16);
            for (unsigned i = 0; i < len; i++) out[(size_t)(put.pos >> 3) + i] = m.data[m.blocks[b].start + i];
            continue;
        }
        TreeWork &w = tw[b];
        if (type[b] == 2) emit_dyn_header(w, put);
        StaticLTree sl;
        StaticDTree sd;
        for (int i = 0; i < m.blocks[b].nsyms; i++) {
            uint32_t sy = m.syms[(size_t)(sym_off[b] + i)];
            uint64_t v;
            int nbts = type[b] == 2 ? encode_symbol(w.ltree, w.dtree, (int)(sy >> 16), (int)(sy & 0xFFFF), v)
                                    : encode_symbol(sl, sd, (int)(sy >> 16), (int)(sy & 0xFFFF), v);
            put((unsigned)(v & 0xFFFFFFFFu), nbts > 32 ? 32 : nbts);
            if (nbts > 32) put((unsigned)(v >> 32), nbts - 32);
        }
        if (type[b] == 2) put(w.ltree[kEndBlock].fc, w.ltree[kEndBlock].dl);
        else put(sl[kEndBlock].fc, sl[kEndBlock].dl);
        int64_t want_end = bit_start[b] + bits[b];
        if (put.pos != want_end) printf("block %zu: emitted %ld bits, trees said %ld\n", b, (long)(put.pos - bit_start[b]), (long)bits[b]);
    }
    uint32_t ad = zso_adler32(1, m.data, (size_t)m.n);
    size_t tb = (size_t)(pos / 8);
    out[tb] = (uint8_t)(ad >> 24), out[tb + 1] = (uint8_t)(ad >> 16), out[tb + 2] = (uint8_t)(ad >> 8), out[tb + 3] = (uint8_t)ad;
    out.resize(tb + 4);
    return out;
}

// ---- mode "resume" without a flush: one Write; the literal engine stops at the first loop-top at or behind F that no read has
// touched (LitEngine::stop_abs: the warm-up of zs_stream_api.inc for a stream that is not at a flush), and the chunked form
// takes the parse over there -- in the lazy parse's node the engine stands in, with the symbols of the block in progress in
// front of its own, on the engine's chains (GeoStart without at_read).
static int resume_stop_main(std::vector<uint8_t> &buf, int64_t n, int level, int strategy, int64_t F) {
    OracleTrace tr;
    zso_trace t;
    memset(&t, 0, sizeof t);
    t.on_symbol = on_symbol, t.on_block = on_block, t.on_read = on_read, t.user = &tr;
    std::vector<uint8_t> ref(zso_compress_bound((size_t)n) + 4096);
    if (zso_compress_stream(buf.data(), (size_t)n, nullptr, 0, level, strategy, 0, 0, ref.data(), ref.size(), &t) == (size_t)-1) {
        printf("oracle failed\n");
        return 1;
    }
    std::vector<uint32_t> crc(1024);
    for (int tt = 0; tt < 4; tt++)
        for (int i = 0; i < 256; i++) crc[tt * 256 + i] = crc32c_table_entry(tt, (uint32_t)i);
    LitEngine e;
    memset(&e, 0, sizeof e);
    le_defaults(e);
    std::vector<uint8_t> window(kWindowSize + 512);
    std::vector<uint16_t> head(kHashSize), prev(kWSize);
    std::vector<uint32_t> asyms((size_t)n + 1024);
    std::vector<BlockRec> ablocks((size_t)n / 4000 + 64);
    e.window = window.data(), e.head = head.data(), e.prev = prev.data(), e.crc_tab = crc.data();
    e.data = buf.data(), e.n = n, e.lv = level_cfg(level), e.strategy = strategy, e.hash_variant = kHashCrc32c;
    e.syms = asyms.data(), e.nsyms = 0, e.blocks = ablocks.data(), e.nblocks = 0;
    e.block_syms = level == 0 ? (kLitBufsize / 2) - 1 : kBlockSyms;
    e.final_run = 1, e.stop_abs = F;
    le_restore(e, 0, 0, 0, kR, 0, nullptr, -1, 0, 1);
    le_run(e, level, 0, 1);
    if (!e.stopped) {
        printf("PASS (the engine did not stop before the stream's end) n=%ld F=%ld\n", (long)n, (long)F);
        return 0;
    }
    const int64_t p0 = e.base + e.strstart;
    const int64_t done_syms = e.block_sym_start, pending = e.nsyms - e.block_sym_start;  // finished blocks' symbols, the block in progress
    bool ok = true;
    for (int64_t i = 0; i < e.nsyms && ok; i++)
        if ((size_t)i >= tr.syms.size() || asyms[(size_t)i] != tr.syms[(size_t)i]) printf("first run: symbol %ld differs\n", (long)i), ok = false;
    Model m;
    m.data = buf.data(), m.n = n, m.level = level, m.strategy = strategy, m.lv = level_cfg(level), m.crc_tab = crc;
    m.flush_mode = 0, m.incremental = false;
    GeoStart gs;
    gs.resume = true, gs.at_read = false, gs.p0 = p0, gs.E0 = e.avail_end, gs.base0 = e.base;
    const std::vector<int64_t> none;
    const bool bulk = m.lv.func == 2 && strategy != kRle && build_geometry(n, none, m.geo, gs);
    if (!bulk) {
        printf("PASS (not a schedule for the bulk path) n=%ld F=%ld p0=%ld\n", (long)n, (long)F, (long)p0);
        return ok ? 0 : 1;
    }
    m.body_end = m.geo.body_end;
    m.build_links();
    for (int64_t x = p0 - kWSize > e.base ? p0 - kWSize : e.base; x < p0; x++) {
        if (x < 0) continue;
        const int idx = (int)(x - e.base), pv = prev[(size_t)(idx & kWMask)];
        if (pv > idx && prev[(size_t)(pv & kWMask)] != idx) {
            printf("PASS (a forward pointer in prev[] that is not a cycle) n=%ld F=%ld\n", (long)n, (long)F);
            return ok ? 0 : 1;
        }
        int d = (pv != 0 && pv < idx) ? idx - pv : 0;
        if (d > kMaxDist) d = 0;
        m.link[(size_t)x] = (uint16_t)d;
    }
    for (int64_t q = p0; q < p0 + kWSize && q + 5 < n; q++) {
        const int have = m.link[(size_t)q];
        if (have != 0 && q - have >= p0) continue;
        const int hv = head[m.bucket(q)], idx = (int)(q - e.base);
        int d = hv != 0 ? idx - hv : 0;
        if (d < 0 || d > kMaxDist) d = 0;
        m.link[(size_t)q] = (uint16_t)d;
    }
    m.match_all();
    // the node of the lazy parse the engine stands in (as zs_stream_api.inc hands it to the run)
    m.start_slot = e.match_available == 0 ? 0 : e.match_length < kMinMatch ? 256 + kL : e.prev_length >= m.lv.good ? 256 + kXK4 : 256 + kXK;
    m.start_syms.assign(asyms.begin() + done_syms, asyms.begin() + done_syms + pending);
    int64_t p, preins;
    int kind, k_done;
    uint32_t pend;
    parse_chunked(m, p, kind, pend, k_done, preins);
    run_tail(m, p, kind, pend, k_done, preins);
    if ((int64_t)tr.syms.size() != done_syms + (int64_t)m.syms.size()) printf("symbol count %ld + %zu vs oracle %zu\n", (long)done_syms, m.syms.size(), tr.syms.size()), ok = false;
    for (size_t i = 0; i < m.syms.size() && ok; i++)
        if ((size_t)done_syms + i >= tr.syms.size() || m.syms[i] != tr.syms[(size_t)done_syms + i]) {
            printf("run behind the stop: symbol %zu differs: model %08x oracle %08x\n", i, m.syms[i], done_syms + (int64_t)i < (int64_t)tr.syms.size() ? tr.syms[(size_t)done_syms + i] : 0u);
            ok = false;
        }
    printf("%s n=%ld level=%d strat=%d mode=resume stop at %ld (asked %ld) slot=%d base=%ld read to %ld, syms=%ld done + %ld pending + %zu, tail_from=%ld\n",
           ok ? "PASS" : "FAIL", (long)n, level, strategy, (long)p0, (long)F, m.start_slot, (long)e.base, (long)e.avail_end, (long)done_syms, (long)pending,
           m.syms.size() - (size_t)pending, (long)p);
    return ok ? 0 : 1;
}

// ---- mode "resume": a stream flushed after its first F bytes, the rest one Write.  The first Write runs on the literal engine
// (as a run of an incremental stream does, zs_stream_api.inc); the run behind the flush is laid out by build_geometry with
// GeoStart::at_read and parsed in the chunked form on chains taken from that engine -- what zs_import_chains_kernel does on
// the device: prev[] for the 32 Ki positions below the flush, head[] for the first link of every bucket behind it.  The
// symbols of both runs against the oracle's (per-Write flush modes).
static int resume_main(std::vector<uint8_t> &buf, int64_t n, int level, int strategy, int64_t F, int flush, const std::vector<size_t> &behind) {
    if (F <= 0 || F >= n || flush < 0 || flush > 3) {
        printf("resume: need 0 < F < n and a flush mode 0..3\n");
        return 2;
    }
    if (flush == 0) return resume_stop_main(buf, n, level, strategy, F);
    OracleTrace tr;
    zso_trace t;
    memset(&t, 0, sizeof t);
    t.on_symbol = on_symbol, t.on_block = on_block, t.on_read = on_read, t.user = &tr;
    std::vector<uint8_t> ref(zso_compress_bound((size_t)n) + 4096);
    // the Writes: F bytes under the flush mode, then NoFlush ones -- `behind` in turn until the stream ends, or all of it at once
    std::vector<size_t> wl{(size_t)F};
    std::vector<int64_t> wends;  // the ends of the Writes behind the flush
    for (int64_t o = F, k = 0; o < n; k++) {
        const size_t w = behind.empty() ? (size_t)(n - o) : std::min(std::max<size_t>(behind[(size_t)k % behind.size()], 1), (size_t)(n - o));
        wl.push_back(w);
        o += (int64_t)w;
        wends.push_back(o);
    }
    std::vector<int> fm(wl.size(), 0);
    fm[0] = flush;
    if (zso_compress_stream_modes(buf.data(), (size_t)n, wl.data(), wl.size(), level, strategy, 0, fm.data(), 0, ref.data(), ref.size(), &t) == (size_t)-1) {
        printf("oracle failed\n");
        return 1;
    }
    std::vector<uint32_t> crc(1024);
    for (int tt = 0; tt < 4; tt++)
        for (int i = 0; i < 256; i++) crc[tt * 256 + i] = crc32c_table_entry(tt, (uint32_t)i);
    // ---- the first Write on the literal engine, left where Deflate.Compress returns behind the flush
    LitEngine e;
    memset(&e, 0, sizeof e);
    le_defaults(e);
    std::vector<uint8_t> window(kWindowSize + 512);
    std::vector<uint16_t> head(kHashSize), prev(kWSize);
    std::vector<uint32_t> asyms((size_t)F + 1024);
    std::vector<BlockRec> ablocks((size_t)F / 4000 + 64);
    const int64_t wend[1] = {F};
    const uint8_t wflush[1] = {(uint8_t)flush};
    int32_t wblk[1] = {0};
    e.window = window.data(), e.head = head.data(), e.prev = prev.data(), e.crc_tab = crc.data();
    e.data = buf.data(), e.n = F, e.lv = level_cfg(level), e.strategy = strategy, e.hash_variant = kHashCrc32c;
    e.wr_end = wend, e.n_wr = 1, e.wr_flush = wflush, e.wr_blk = wblk;
    e.syms = asyms.data(), e.nsyms = 0, e.blocks = ablocks.data(), e.nblocks = 0;
    e.block_syms = level == 0 ? (kLitBufsize / 2) - 1 : kBlockSyms;
    e.final_run = 0;
    le_restore(e, 0, 0, 0, kR, 0, nullptr, -1, 0, 1);
    le_run(e, level, 0, 1);
    if (!e.suspended || e.base + e.strstart != F || e.lookahead != 0 || e.match_available != 0) {
        printf("the engine is not where a flush leaves it: suspended %d at %ld lookahead %d\n", e.suspended, (long)(e.base + e.strstart), e.lookahead);
        return 1;
    }
    const int64_t nA = e.nsyms;
    bool ok = true;
    for (int64_t i = 0; i < nA && ok; i++)
        if ((size_t)i >= tr.syms.size() || asyms[(size_t)i] != tr.syms[(size_t)i]) printf("first run: symbol %ld differs\n", (long)i), ok = false;
    // ---- the run behind the flush
    Model m;
    m.data = buf.data(), m.n = n, m.level = level, m.strategy = strategy, m.lv = level_cfg(level), m.crc_tab = crc;
    m.flush_mode = 0, m.incremental = false;
    GeoStart gs;
    gs.resume = true, gs.at_read = true, gs.p0 = gs.E0 = F, gs.base0 = e.base;
    const std::vector<int64_t> none;
    if (wends.size() > 1) m.wr_end = wends;
    const bool bulk = m.lv.func == 2 && strategy != kRle && build_geometry(n, wends.size() > 1 ? wends : none, m.geo, gs);
    if (!bulk) {
        printf("PASS (not a schedule for the bulk path) n=%ld F=%ld\n", (long)n, (long)F);
        return ok ? 0 : 1;
    }
    m.body_end = m.geo.body_end;
    m.build_links();
    // the engine's chains: below F its prev[] (a forward pointer that its partner closes into a cycle is a cut) ...
    // (ZS_MODEL_NO_IMPORT: the data's own links, to see what the import is for)
    for (int64_t x = F - kWSize > e.base ? F - kWSize : e.base; x < F && !getenv("ZS_MODEL_NO_IMPORT"); x++) {
        if (x < 0) continue;
        const int idx = (int)(x - e.base), pv = prev[(size_t)(idx & kWMask)];
        if (pv > idx && prev[(size_t)(pv & kWMask)] != idx) {
            printf("PASS (a forward pointer in prev[] that is not a cycle: the stream stays with the literal engine) n=%ld F=%ld\n", (long)n, (long)F);
            return ok ? 0 : 1;
        }
        int d = (pv != 0 && pv < idx) ? idx - pv : 0;
        if (d > kMaxDist) d = 0;
        m.link[(size_t)x] = (uint16_t)d;
    }
    // ... and behind it, for a position whose bucket has no member in [F, q) yet, the engine's head of the bucket
    for (int64_t q = F; q < F + kWSize && q + 5 < n && !getenv("ZS_MODEL_NO_IMPORT"); q++) {
        const int have = m.link[(size_t)q];
        if (have != 0 && q - have >= F) continue;
        const int hv = head[m.bucket(q)], idx = (int)(q - e.base);
        int d = hv != 0 ? idx - hv : 0;
        if (d < 0 || d > kMaxDist) d = 0;
        m.link[(size_t)q] = (uint16_t)d;
    }
    m.match_all();
    int64_t p, preins;
    int kind, k_done;
    uint32_t pend;
    parse_chunked(m, p, kind, pend, k_done, preins);
    run_tail(m, p, kind, pend, k_done, preins);
    if ((int64_t)tr.syms.size() != nA + (int64_t)m.syms.size()) printf("symbol count %ld + %zu vs oracle %zu\n", (long)nA, m.syms.size(), tr.syms.size()), ok = false;
    for (size_t i = 0; i < m.syms.size() && ok; i++)
        if ((size_t)nA + i >= tr.syms.size() || m.syms[i] != tr.syms[(size_t)nA + i]) {
            printf("run behind the flush: symbol %zu differs: model %08x oracle %08x\n", i, m.syms[i], nA + i < tr.syms.size() ? tr.syms[(size_t)nA + i] : 0u);
            ok = false;
        }
    printf("%s n=%ld level=%d strat=%d mode=resume F=%ld flush=%d base=%ld syms=%ld+%zu events=%zu tail_from=%ld\n", ok ? "PASS" : "FAIL", (long)n, level, strategy,
           (long)F, flush, (long)e.base, (long)nA, m.syms.size(), m.events.size(), (long)p);
    return ok ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s file level [strategy] [mode]\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint8_t> buf;
    {
        uint8_t tmp[65536];
        size_t r;
        while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
        fclose(f);
    }
    int level = atoi(argv[2]);
    int strategy = argc > 3 ? atoi(argv[3]) : 0;
    std::string mode = argc > 4 ? argv[4] : "bulk";
    // Write sizes: "N" (every Write N bytes), "a,b,c" (sizes in turn), "rSEED:lo:hi" (random sizes in [lo, hi])
    const std::string wspec = argc > 5 ? argv[5] : "0";
    size_t wchunk = (size_t)atol(wspec.c_str());
    std::vector<size_t> wcycle;
    uint64_t wrand = 0, wlo = 0, whi = 0;
    if (wspec[0] == 'r') {
        sscanf(wspec.c_str(), "r%lu:%lu:%lu", (unsigned long *)&wrand, (unsigned long *)&wlo, (unsigned long *)&whi);
        wrand = wrand * 2654435761u + 88172645463325252ull;
        wchunk = wlo ? wlo : 1;
    } else if (wspec.find(',') != std::string::npos) {
        size_t at = 0;
        while (at < wspec.size()) {
            wcycle.push_back((size_t)atol(wspec.c_str() + at));
            at = wspec.find(',', at);
            if (at == std::string::npos) break;
            at++;
        }
        wchunk = wcycle[0] ? wcycle[0] : 1;
    }
    int flush_mode = argc > 6 ? atoi(argv[6]) : 0;
    int64_t n = (int64_t)buf.size();
    buf.resize(buf.size() + 1024, 0);
    if (mode == "resume")  // "F" or "F,a,b,...": the flushed Write, then the sizes of the NoFlush Writes behind it in turn
        return resume_main(buf, n, level, strategy, (int64_t)wchunk, flush_mode, wcycle.size() > 1 ? std::vector<size_t>(wcycle.begin() + 1, wcycle.end()) : std::vector<size_t>());

    OracleTrace tr;
    zso_trace t;
    memset(&t, 0, sizeof t);
    t.on_symbol = on_symbol;
    t.on_block = on_block;
    t.on_read = on_read;
    t.user = &tr;
    // every Write that ends under a flush mode adds a block end, a marker and now and then an empty block
    std::vector<uint8_t> ref(zso_compress_bound((size_t)n) + (wchunk ? 32 * ((size_t)n / wchunk + 2) : 64));
    std::vector<size_t> wlens;
    std::vector<int64_t> wends;
    if (wchunk) {
        size_t o = 0, i = 0;
        while (o < (size_t)n) {
            size_t w = wchunk;
            if (!wcycle.empty()) w = wcycle[i++ % wcycle.size()];
            else if (whi) {
                wrand ^= wrand << 13, wrand ^= wrand >> 7, wrand ^= wrand << 17;
                w = wlo + (size_t)(wrand % (whi - wlo + 1));
            }
            if (w < 1) w = 1;
            wlens.push_back(std::min(w, (size_t)n - o));
            o += wlens.back();
            wends.push_back((int64_t)o);
        }
    }
    if (flush_mode && wends.empty() && n > 0) wlens.push_back((size_t)n), wends.push_back(n);
    size_t ref_len = zso_compress_stream(buf.data(), (size_t)n, wlens.empty() ? nullptr : wlens.data(), wlens.size(), level, strategy, flush_mode, 0,
                                         ref.data(), ref.size(), &t);
    if (ref_len == (size_t)-1) {
        printf("oracle failed\n");
        return 1;
    }

    Model m;
    m.data = buf.data();
    m.n = n;
    m.level = level;
    m.strategy = strategy;
    m.lv = level_cfg(level);
    m.crc_tab.resize(1024);
    for (int tt = 0; tt < 4; tt++)
        for (int i = 0; i < 256; i++) m.crc_tab[tt * 256 + i] = crc32c_table_entry(tt, (uint32_t)i);
    m.wr_end = wends;
    m.flush_mode = flush_mode;
    m.incremental = mode == "inc";
    // the bulk form needs a regular read schedule: one Write, or NoFlush Writes whose sizes are multiples of kChunk
    // (DeflateFast's sweeps take Write ends anywhere: an event is applied where a sweep starts)
    const bool fast_mode = mode == "fsweep" || mode == "frounds";
    bool regular = build_read_events(n, wends, m.rev, fast_mode) && (wends.size() <= 1 || flush_mode == 0);
    m.body_end = (m.lv.func == 2 && strategy != kRle && regular) ? n - kMinLookahead : -1;
    if (mode == "chunk") {
        // the chunked form takes any NoFlush schedule build_geometry accepts (zs_core.h)
        regular = (wends.size() <= 1 || flush_mode == 0) && m.lv.func == 2 && strategy != kRle && build_geometry(n, wends, m.geo);
        m.body_end = regular ? m.geo.body_end : -1;
        if (!regular) m.geo = Geometry();
    }
    m.build_links();
    int64_t p;
    int kind, k_done;
    uint32_t pend;
    int64_t preins;
    if (mode == "bulk" || mode == "chunk" || mode == "spec") m.match_all();
    if (mode == "spec") {
        spec_stats(m);
        return 0;
    }
    if (mode == "rle") {
        // CompressionStrategy.Rle, one Write: the runs' closed form up to the hand-over loop-top, then the literal engine
        kind = kR, pend = 0, p = 0, k_done = 0, preins = -1;
        m.body_end = -1;
        if (strategy == kRle && level >= 1 && wends.size() <= 1 && flush_mode == 0 && !m.rev.empty()) parse_rle_runs(m, p, k_done);
        if (p > 0) m.body_end = p;  // (run_tail sizes its arrays by this)
    } else if (mode == "fvec" || mode == "fsweep" || mode == "frounds") {
        // DeflateFast, single Write: the vector form up to the last loop-top with a full lookahead, then the literal engine
        m.body_end = (m.lv.func == 1 && strategy != kRle && (wends.size() <= 1 || (fast_mode && regular)) && flush_mode == 0 && n >= kMinLookahead) ? n - kMinLookahead : -1;
        kind = kR, pend = 0, p = 0, k_done = 0, preins = -1;
        if (m.body_end >= 0 && mode == "fsweep") parse_fast_sweep(m, p, k_done, preins);
        else if (m.body_end >= 0 && mode == "frounds") parse_fast_rounds(m, p, k_done, preins);
        else if (m.body_end >= 0) parse_fast_vec(m, p, k_done, preins);
    } else if (mode == "chunk") parse_chunked(m, p, kind, pend, k_done, preins);
    else parse_sequential(m, mode != "bulk", p, kind, pend, k_done, preins);
    run_tail(m, p, kind, pend, k_done, preins);

    bool ok = true;
    if (level == 0 && strategy != kRle) {
        // level 0 on the device: the block list comes from plan_stored_blocks (sizes only) instead of the literal engine;
        // it must be the engine's list, block for block, and the Write -> block counts with it
        std::vector<BlockRec> plan;
        std::vector<int32_t> pwb(m.wr_end.size(), 0);
        const bool wr = m.wr_end.size() > 1 || flush_mode;
        const std::vector<int64_t> no_ends;
        const std::vector<uint8_t> no_flush;
        plan_stored_blocks(n, wr ? m.wr_end : no_ends, flush_mode ? m.wr_flush : no_flush,
                           [&](int64_t start, int32_t blen, int can_store, int eof) { plan.push_back(BlockRec{start, 0, blen, 0, can_store, eof}); },
                           [&](int w, int nb) { pwb[(size_t)w] = nb; });
        if (plan.size() != m.blocks.size()) printf("stored plan: %zu blocks, engine %zu\n", plan.size(), m.blocks.size()), ok = false;
        for (size_t i = 0; ok && i < plan.size(); i++) {
            const BlockRec &a = plan[i], &b = m.blocks[i];
            if (a.start != b.start || a.stored_len != b.stored_len || a.can_store != b.can_store || a.eof != b.eof || b.nsyms != 0) {
                printf("stored plan block %zu: start %ld/%ld len %d/%d can %d/%d eof %d/%d\n", i, (long)a.start, (long)b.start, a.stored_len,
                       b.stored_len, a.can_store, b.can_store, a.eof, b.eof);
                ok = false;
            }
        }
        if (flush_mode && pwb != m.wr_blk) printf("stored plan: wr_blk differs\n"), ok = false;
        m.blocks = plan;  // the bytes below are assembled from the planned list
        if (flush_mode) m.wr_blk = pwb;
    }
    if (m.syms.size() != tr.syms.size()) {
        printf("symbol count %zu vs oracle %zu\n", m.syms.size(), tr.syms.size());
        ok = false;
    }
    size_t lim = m.syms.size() < tr.syms.size() ? m.syms.size() : tr.syms.size();
    for (size_t i = 0; i < lim; i++)
        if (m.syms[i] != tr.syms[i]) {
            printf("symbol %zu differs: model %08x oracle %08x\n", i, m.syms[i], tr.syms[i]);
            ok = false;
            break;
        }
    if (flush_mode) {
        // the oracle's trace also holds the empty blocks of re-entered flushes; the bytes below cover the blocks
    } else if (m.blocks.size() != tr.blocks.size()) {
        printf("block count %zu vs oracle %zu\n", m.blocks.size(), tr.blocks.size());
        ok = false;
    } else {
        for (size_t i = 0; i < m.blocks.size(); i++) {
            const BlockRec &b = m.blocks[i];
            const OracleTrace::Blk &o = tr.blocks[i];
            if (b.start != o.start || b.stored_len != o.stored_len || b.nsyms != o.nsyms || b.eof != o.eof) {
                printf("block %zu differs: start %ld/%ld len %d/%d nsyms %d/%d eof %d/%d\n", i, (long)b.start, (long)o.start,
                       b.stored_len, o.stored_len, b.nsyms, o.nsyms, b.eof, o.eof);
                ok = false;
                break;
            }
        }
    }
    // event loop-tops vs the loop-tops of the oracle's reads (several reads at one loop-top are one event; the reads at 0
    // are the stream's start); the tail engine's own reads are not in the list
    {
        std::vector<int64_t> want;
        for (int64_t r : tr.read_pos)
            if (r > 0 && (want.empty() || want.back() != r)) want.push_back(r);
        for (size_t k = 0; k < m.events.size(); k++)
            if (k >= want.size() || m.events[k] != want[k]) {
                printf("event %zu at %ld, oracle %ld\n", k + 1, (long)m.events[k], k < want.size() ? (long)want[k] : -1L);
                ok = false;
                break;
            }
    }
    {
        std::vector<uint8_t> out = emit_stream(m);
        if (out.size() != ref_len || memcmp(out.data(), ref.data(), ref_len) != 0) {
            size_t i = 0;
            while (i < out.size() && i < ref_len && out[i] == ref[i]) i++;
            printf("bytes differ: model %zu oracle %zu first diff at %zu\n", out.size(), ref_len, i);
            ok = false;
        }
    }
    if (m.poisoned) {
        // the device hands such a stream to the literal engine; nothing to compare here
        printf("PASS (poisoned: literal engine) n=%ld level=%d\n", (long)n, level);
        return 0;
    }
    printf("%s n=%ld level=%d strat=%d mode=%s syms=%zu blocks=%zu events=%zu tail_from=%ld\n", ok ? "PASS" : "FAIL", (long)n, level,
           strategy, mode.c_str(), m.syms.size(), m.blocks.size(), m.events.size(), (long)p);
    return ok ? 0 : 1;
}
