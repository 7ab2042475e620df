// test_geometry.cpp -- structural invariants of zs_core.h build_geometry (host logic, no device): random Write schedules,
// fresh streams and the two ways a run goes on in the middle of one (GeoStart: behind a stop of the literal engine, or at
// the position a flush left).  Every kernel from the chunk maps to the tail engine reads these tables; what they rely on:
//   - the chunks tile [first loop-top, body_end] without gaps, none longer than 2048 positions;
//   - a segment's first chunk is marked with the segment's number iff its cluster's reads fire inside it, and is long enough
//     for every loop-top at which one of them can (the cluster's last data end + 4);
//   - the data ends of all clusters are the Write ends and window ends below the body's end, in order, each once;
//   - window bases move by 32768 per window end, slide thresholds are base + 65274, "data end after" is the next data end.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../zlibstream_amd/csrc/zs_core.h"
using namespace zs;

static int fails = 0;
#define CHECK(c, ...)                          \
    do {                                       \
        if (!(c)) {                            \
            if (fails < 20) {                  \
                printf("FAIL %s: ", #c);       \
                printf(__VA_ARGS__);           \
                printf("\n");                  \
            }                                  \
            fails++;                           \
            return;                            \
        }                                      \
    } while (0)

static void check(int64_t n, const std::vector<int64_t> &ends, const GeoStart &gs, int id) {
    Geometry g;
    const std::vector<int64_t> none;
    if (!build_geometry(n, ends.size() > 1 ? ends : none, g, gs)) return;  // left to the literal engine: nothing to check
    const int64_t off = gs.resume ? gs.base0 : 0, p0 = gs.resume ? gs.p0 : 0;
    const int nc = g.nchunks(), ns = g.nsegs();
    CHECK(nc >= 1 && ns >= 1, "case %d: %d chunks %d segments", id, nc, ns);
    CHECK(g.cstart[0] == p0 && g.cstart[nc] == g.body_end + 1, "case %d: chunks span %d..%d, want %lld..%lld", id, g.cstart[0], g.cstart[nc], (long long)p0,
          (long long)g.body_end + 1);
    CHECK(g.body_end <= n - kMinLookahead && g.body_end >= p0, "case %d: body_end %lld of %lld", id, (long long)g.body_end, (long long)n);
    CHECK((int)g.head.size() == nc && (int)g.seg_after.size() == ns && (int)g.seg_base.size() == ns && (int)g.seg_S.size() == ns && (int)g.seg_cl.size() == ns + 1,
          "case %d: table sizes", id);
    for (int c = 0; c < nc; c++) {
        const int len = g.cstart[c + 1] - g.cstart[c];
        CHECK(len >= 1 && len <= kChunk, "case %d: chunk %d has %d positions", id, c, len);
    }
    // the boundaries the clusters must hold: Write ends and window ends (relative to the window base) in (start, body_end + 262)
    std::vector<int64_t> want;
    {
        size_t w = 0;
        int64_t x = off + kWindowSize, last = gs.resume ? gs.E0 : 0;
        if (gs.resume) want.push_back(gs.E0);
        else want.push_back(0);
        if (last == x) x += kWSize;
        for (;;) {
            while (w < ends.size() && ends[w] <= last) w++;
            const int64_t we = (ends.size() > 1 && w < ends.size()) ? ends[w] : n;
            const int64_t nx = we <= x ? we : x;
            if (nx >= n) break;
            want.push_back(nx);
            last = nx;
            if (nx == x) x += kWSize;
        }
    }
    std::vector<int64_t> got;
    int64_t base = off;
    for (int k = 0; k < ns; k++) {
        const int c0 = g.seg_c0[k];
        CHECK(c0 >= 0 && c0 < nc && (k == 0 || c0 > g.seg_c0[k - 1]), "case %d: segment %d begins at chunk %d", id, k, c0);
        const int m = g.seg_cl[k + 1] - g.seg_cl[k];
        CHECK(m >= 0 && m <= kClusterMax, "case %d: segment %d has %d data ends", id, k, m);
        CHECK(g.head[c0] == (m > 0 ? k + 1 : 0), "case %d: segment %d's first chunk carries %d, cluster of %d", id, k, g.head[c0], m);
        CHECK(g.seg_S[k] == base + kSlideAt, "case %d: segment %d slides at %d, window base %lld", id, k, g.seg_S[k], (long long)base);
        int64_t lastb = -1;
        for (int i = 0; i < m; i++) {
            const uint32_t b = g.cl[g.seg_cl[k] + i];
            const int64_t pos = b & 0x7FFFFFFFu;
            CHECK(pos > lastb, "case %d: segment %d data ends out of order", id, k);
            if (i > 0) CHECK(pos - lastb <= kClusterGap, "case %d: segment %d: gap %lld inside a cluster", id, k, (long long)(pos - lastb));
            lastb = pos;
            got.push_back(pos);
            if (pos - off >= kWindowSize && ((pos - off) & (kWSize - 1)) == 0) base += kWSize;
        }
        if (m > 0) {
            const int64_t first = g.cl[g.seg_cl[k]] & 0x7FFFFFFFu;
            const int64_t seg_start = g.cstart[c0];
            CHECK(seg_start == ((k == 0 && !(gs.resume && !gs.at_read)) ? p0 : first - (kMinLookahead - 1)), "case %d: segment %d begins at %lld, first data end %lld", id, k,
                  (long long)seg_start, (long long)first);
            const int64_t seg_end = k + 1 < ns ? g.cstart[g.seg_c0[k + 1]] : g.body_end + 1;
            const int64_t zone = lastb + 4 - seg_start, len0 = g.cstart[c0 + 1] - seg_start;
            CHECK(len0 >= zone || len0 == seg_end - seg_start, "case %d: segment %d: first chunk %lld positions, the reads need %lld", id, k, (long long)len0, (long long)zone);
            CHECK(lastb - first <= kClusterSpanMax, "case %d: segment %d: cluster spans %lld", id, k, (long long)(lastb - first));
        }
        CHECK(g.seg_base[k] == base, "case %d: segment %d: window base %d after its reads, want %lld", id, k, g.seg_base[k], (long long)base);
        // the data end once the cluster is exhausted: the next boundary (or the stream's end)
        for (int c = c0 + 1; c < (k + 1 < ns ? g.seg_c0[k + 1] : nc); c++) CHECK(g.head[c] == 0, "case %d: chunk %d inside segment %d is marked", id, c, k);
    }
    // a fresh stream whose first read stands alone is not stepped through (its data end 0 is not in the tables)
    size_t wi = 0;
    if (!got.empty() && !want.empty() && got[0] != want[0]) wi = 1;
    if (got.empty()) wi = want.size() ? 1 : 0;
    for (size_t i = 0; i < got.size(); i++, wi++) {
        CHECK(wi < want.size() && got[i] == want[wi], "case %d: data end %zu is %lld, want %lld", id, i, (long long)got[i], wi < want.size() ? (long long)want[wi] : -1LL);
    }
    // what is left out lies behind the body (the tail engine's)
    for (; wi < want.size(); wi++) CHECK(want[wi] > g.body_end, "case %d: data end %lld below the body's end %lld is in no cluster", id, (long long)want[wi], (long long)g.body_end);
    for (int k = 0; k < ns; k++) {
        // seg_after: the first wanted boundary above the cluster's last one, or n
        const int m = g.seg_cl[k + 1] - g.seg_cl[k];
        if (m == 0) continue;
        const int64_t lastb = g.cl[g.seg_cl[k + 1] - 1] & 0x7FFFFFFFu;
        int64_t next = n;
        for (int64_t v : want)
            if (v > lastb) {
                next = v;
                break;
            }
        CHECK(g.seg_after[k] == next, "case %d: segment %d: data end after its reads %d, want %lld", id, k, g.seg_after[k], (long long)next);
    }
}

int main(int argc, char **argv) {
    const int cases = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(12345);
    auto R = [&](int64_t lo, int64_t hi) { return lo + (int64_t)(rng() % (uint64_t)(hi - lo + 1)); };
    int built = 0, built_kind[3] = {0, 0, 0};
    for (int id = 0; id < cases; id++) {
        const int64_t n = R(0, 1) ? R(300, 400000) : R(300, 3000000);
        std::vector<int64_t> ends;
        const int style = (int)R(0, 5);
        int64_t o = 0;
        while (o < n && style > 0) {
            int64_t c = style == 1 ? R(1, 70000) : style == 2 ? R(6000, 400000) : style == 3 ? (int64_t[]){32768, 65536, 65274, 65275, 32506, 98304}[R(0, 5)] - R(0, 300)
                        : style == 4 ? (int64_t[]){1, 3, 100, 261, 262, 263, 1000, 4096, 16385, 81921}[R(0, 9)] : R(200000, 2000000);
            if (c < 1) c = 1;
            o += c;
            ends.push_back(o < n ? o : n);
        }
        if (ends.empty()) ends.push_back(n);
        GeoStart gs;
        const int how = (int)R(0, 2);
        if (how > 0 && n > 70000) {
            gs.resume = true;
            gs.base0 = R(0, 1) ? 0 : R(0, 33000);
            gs.p0 = gs.base0 + R(how == 2 ? 0 : 1, kSlideAt - 1);
            if (gs.p0 >= n - 3000) gs.p0 = gs.base0;
            gs.at_read = how == 2;
            if (gs.at_read) {
                gs.E0 = gs.p0;
            } else {
                // behind a stop of the literal engine: the data end read so far is a Write end or the window's end
                int64_t e = gs.base0 + kWindowSize;
                for (int64_t w : ends)
                    if (w > gs.p0) {
                        if (w < e) e = w;
                        break;
                    }
                gs.E0 = e;
            }
            // the Writes handed to a resumed run are those that end behind what has been read
            std::vector<int64_t> e2;
            for (int64_t w : ends)
                if (w >= (gs.at_read ? gs.E0 + 1 : gs.E0)) e2.push_back(w);
            if (e2.empty()) e2.push_back(n);
            ends = e2;
        }
        const int before = fails;
        Geometry g;
        const std::vector<int64_t> none;
        if (build_geometry(n, ends.size() > 1 ? ends : none, g, gs)) built++, built_kind[gs.resume ? (gs.at_read ? 2 : 1) : 0]++;
        check(n, ends, gs, id);
        if (fails != before && fails <= 20) {
            printf("   case %d: n=%lld resume=%d at_read=%d p0=%lld E0=%lld base0=%lld, %zu Writes:", id, (long long)n, gs.resume, gs.at_read, (long long)gs.p0, (long long)gs.E0,
                   (long long)gs.base0, ends.size());
            for (size_t i = 0; i < ends.size() && i < 8; i++) printf(" %lld", (long long)ends[i]);
            printf("\n");
        }
    }
    printf("%s: %d schedules, %d laid out for the bulk path (%d fresh streams, %d behind a stop of the literal engine, %d at a flush), %d failures\n",
           fails ? "FAIL" : "PASS", cases, built, built_kind[0], built_kind[1], built_kind[2], fails);
    return fails ? 1 : 0;
}
