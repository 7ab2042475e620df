// fast_jacobi.cpp -- CPU experiment (TEST TOOL, not part of the product): how fast does a window-wide fixed-point
// iteration of DeflateFast's parse (Deflate.Fast.cs:20-128) converge?
//
// DeflateFast inserts a position into the hash chains only when it is a loop-top or lies inside a match no longer than
// max_lazy, so the chains -- and the matches -- depend on the parse.  Given a guess G of the inserted-position bitmap the
// search at every position is a function of the data (zs_fast_vec.h fv_search): all positions of a window can be searched
// at once, the hops of the parse followed through the results, and the bitmap the parse implies compared with the guess.
// Everything in front of the first loop-top whose result changed between two consecutive sweeps is final (induction over
// the loop-tops: a loop-top's search only looks below itself).  This tool measures, against the sequential parse:
//
//   mode fix  W       the review's form: a window of W positions iterated to its fixed point from the guess "everything
//                     inserted"; iterations per window
//   mode slide W K T  the sliding form: every sweep searches the W positions behind the last final loop-top under the
//                     current guess, finalises the loop-tops up to the first whose result differs from the sweep before,
//                     and goes on from there; K chip-wide pre-sweeps (each tile of T positions parsed from its own first
//                     position) give the guess for positions no sweep has seen; positions per sweep
//
// Read events (Deflate.cs:1010-1013, one per 32 KiB) are left out on both sides: the statistics do not depend on them.
//
// usage: fast_jacobi <file> <level 1..3> fix <W> | slide <W> <K> <T>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../zlibstream_amd/csrc/zs_core.h"

using namespace zs;

struct Ctx {
    std::vector<uint8_t> data;
    int64_t n = 0, body_end = 0;
    LevelCfg lv;
    std::vector<uint16_t> link;
    std::vector<uint32_t> crc_tab;
    mutable long steps = 0, evals = 0, lcps = 0;

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
        lcps++;
        int len = 0;
        while (len < kMaxMatch && data[p + len] == data[c + len]) len++;
        return len;
    }
    // Longest_match at q with prev_length 2 over the all-position chain filtered by `bits`; (len << 16) | dist, len 2 = none
    uint32_t eval(int64_t q, const std::vector<uint8_t> &bits) const {
        evals++;
        int best = 2, bdist = 0, found = 0;
        int64_t c = q;
        for (;;) {
            const int l = link[(size_t)c];
            if (!l) break;
            steps++;
            c -= l;
            const int64_t d = q - c;
            if (c < 1 || (found == 0 ? d > kMaxDist : d >= kMaxDist)) break;
            if (!bits[(size_t)c]) continue;
            found++;
            const int len = lcp(q, c);
            if (len > best) {
                best = len, bdist = (int)d;
                if (len >= lv.nice) break;
            }
            if (found >= lv.chain) break;
        }
        return ((uint32_t)best << 16) | (uint32_t)bdist;
    }
    int adv(uint32_t r) const { return (int)(r >> 16) >= kMinMatch ? (int)(r >> 16) : 1; }
    void mark(std::vector<uint8_t> &bits, int64_t t, uint32_t r) const {
        const int len = (int)(r >> 16);
        bits[(size_t)t] = 1;
        if (len >= kMinMatch && len <= lv.lazy)
            for (int k = 1; k < len; k++) bits[(size_t)t + k] = 1;
    }
};

static void pct(std::vector<long> &v, const char *what) {
    std::sort(v.begin(), v.end());
    if (v.empty()) return;
    double mean = 0;
    for (long x : v) mean += (double)x;
    mean /= (double)v.size();
    printf("  %s: n=%zu mean %.2f p50 %ld p90 %ld p99 %ld max %ld\n", what, v.size(), mean, v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100],
           v.back());
}

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: fast_jacobi <file> <level> fix <W> | slide <W> <K> <T>\n");
        return 2;
    }
    Ctx cx;
    {
        FILE *f = fopen(argv[1], "rb");
        if (!f) return 2;
        fseek(f, 0, SEEK_END);
        cx.n = ftell(f);
        fseek(f, 0, SEEK_SET);
        cx.data.assign((size_t)cx.n + 600, 0);
        if (fread(cx.data.data(), 1, (size_t)cx.n, f) != (size_t)cx.n) return 2;
        fclose(f);
    }
    const int level = atoi(argv[2]);
    cx.lv = level_cfg(level);
    cx.crc_tab.resize(1024);
    for (int tt = 0; tt < 4; tt++)
        for (int i = 0; i < 256; i++) cx.crc_tab[(size_t)tt * 256 + i] = crc32c_table_entry(tt, (uint32_t)i);
    cx.body_end = cx.n - kMinLookahead;
    cx.build_links();
    const std::string mode = argv[3];
    const int64_t n = cx.n, be = cx.body_end;

    // the sequential parse: the truth
    std::vector<uint8_t> truth((size_t)n + 600, 0);
    std::vector<uint32_t> rtrue((size_t)n + 600, 0xFFFFFFFFu);
    long ntops = 0;
    for (int64_t p = 0; p <= be;) {
        const uint32_t r = cx.eval(p, truth);
        rtrue[(size_t)p] = r;
        cx.mark(truth, p, r);
        p += cx.adv(r);
        ntops++;
    }
    long nins = 0;
    for (int64_t p = 0; p <= be; p++) nins += truth[(size_t)p];
    printf("%s level %d: %lld bytes, %ld loop-tops (%.3f per byte), %ld inserted (%.3f per byte), sequential: %.1f chain steps and %.2f compares per loop-top\n", argv[1], level,
           (long long)n, ntops, (double)ntops / (double)n, nins, (double)nins / (double)n, (double)cx.steps / (double)ntops, (double)cx.lcps / (double)ntops);
    cx.steps = cx.evals = cx.lcps = 0;

    if (mode == "fix") {
        const int W = atoi(argv[4]);
        std::vector<uint8_t> G = truth;  // below the window: final
        std::vector<long> iters;
        std::vector<uint32_t> r((size_t)W + 8);
        int64_t w0 = 0;
        while (w0 <= be) {
            const int64_t hi = std::min<int64_t>(w0 + W, be + 1);
            // guess: everything inserted inside the window
            for (int64_t p = w0; p < hi + 300; p++) G[(size_t)p] = 1;
            int it = 0;
            int64_t next_w0 = hi;
            for (;;) {
                it++;
                for (int64_t p = w0; p < hi; p++) r[(size_t)(p - w0)] = cx.eval(p, G);
                std::vector<uint8_t> nb((size_t)(hi - w0) + 300, 0);
                int64_t t = w0;
                while (t < hi) {
                    const uint32_t x = r[(size_t)(t - w0)];
                    const int len = (int)(x >> 16);
                    nb[(size_t)(t - w0)] = 1;
                    if (len >= kMinMatch && len <= cx.lv.lazy)
                        for (int k = 1; k < len; k++) nb[(size_t)(t - w0) + k] = 1;
                    t += cx.adv(x);
                }
                next_w0 = t;
                bool same = true;
                for (int64_t p = w0; p < hi; p++)
                    if (G[(size_t)p] != nb[(size_t)(p - w0)]) same = false, G[(size_t)p] = nb[(size_t)(p - w0)];
                if (same) break;
                if (it > 100000) break;
            }
            // the window's bits are the fixed point; what a match reaching out of it inserted too
            for (int64_t p = hi; p < hi + 300; p++) G[(size_t)p] = 0;
            {
                int64_t t = w0;
                while (t < hi) {
                    const uint32_t x = r[(size_t)(t - w0)];
                    if (rtrue[(size_t)t] != x) {
                        printf("MISMATCH at %lld\n", (long long)t);
                        return 1;
                    }
                    cx.mark(G, t, x);
                    t += cx.adv(x);
                }
            }
            iters.push_back(it);
            w0 = next_w0;
        }
        printf("fix W=%d: %zu windows, searches per position %.2f\n", W, iters.size(), (double)cx.evals / (double)n);
        pct(iters, "iterations to the fixed point (the last one only confirms)");
        return 0;
    }

    if (mode == "slide") {
        const int W = atoi(argv[4]);
        const int K = argc > 5 ? atoi(argv[5]) : 0;
        const int T = argc > 6 ? atoi(argv[6]) : 4096;
        std::vector<uint8_t> G((size_t)n + 600, 1);
        // chip-wide pre-sweeps: search every position under the guess, parse every tile from its own first position
        std::vector<uint32_t> r((size_t)n + 600, 0xFFFFFFFFu);
        for (int k = 0; k < K; k++) {
            for (int64_t p = 0; p <= be; p++) r[(size_t)p] = cx.eval(p, G);
            std::vector<uint8_t> nb((size_t)n + 600, 0);
            for (int64_t t0 = 0; t0 <= be; t0 += T) {
                int64_t t = t0;
                const int64_t hi = std::min<int64_t>(t0 + T, be + 1);
                while (t < hi) {
                    cx.mark(nb, t, r[(size_t)t]);
                    t += cx.adv(r[(size_t)t]);
                }
                // a match that reaches into the next tile: that tile's own parse decides there
                for (int64_t p = hi; p < std::min<int64_t>(hi + 300, n); p++) nb[(size_t)p] = 0;
            }
            long wrong = 0;
            for (int64_t p = 0; p <= be; p++) wrong += nb[(size_t)p] != truth[(size_t)p];
            printf("  pre-sweep %d: %ld wrong bits (%.3f %%)\n", k + 1, wrong, 100.0 * (double)wrong / (double)n);
            G.swap(nb);
            for (int64_t p = be + 1; p < n + 600; p++) G[(size_t)p] = 1;
        }
        const long pre_evals = cx.evals;
        std::vector<uint32_t> rprev((size_t)n + 600, 0xFFFFFFFFu), rcur((size_t)W + 8);
        std::vector<long> progress;
        int64_t w0 = 0;
        std::vector<uint8_t> fresh((size_t)n + 600, 1);  // G[p] is still the pre-sweeps' (or the all-ones) guess
        while (w0 <= be) {
            const int64_t hi = std::min<int64_t>(w0 + W, be + 1);
            for (int64_t p = w0; p < hi; p++) rcur[(size_t)(p - w0)] = cx.eval(p, G);
            // the parse through the window; final up to and including the first loop-top whose result is new
            int64_t t = w0, next_w0 = -1;
            std::vector<uint8_t> nb((size_t)(hi - w0) + 300, 0);
            while (t < hi) {
                const uint32_t x = rcur[(size_t)(t - w0)];
                if (next_w0 < 0) {
                    if (rtrue[(size_t)t] != x) {
                        printf("MISMATCH at %lld (w0 %lld)\n", (long long)t, (long long)w0);
                        return 1;
                    }
                    if (rprev[(size_t)t] != x) next_w0 = t + cx.adv(x);
                }
                const int len = (int)(x >> 16);
                nb[(size_t)(t - w0)] = 1;
                if (len >= kMinMatch && len <= cx.lv.lazy)
                    for (int k = 1; k < len; k++) nb[(size_t)(t - w0) + k] = 1;
                t += cx.adv(x);
            }
            if (next_w0 < 0) next_w0 = t;  // the whole window agreed with the sweep before
            // the guess for the next sweep: the bits of this parse, up to where it ended (t); behind that, what was there
            for (int64_t p = w0; p < t; p++) G[(size_t)p] = nb[(size_t)(p - w0)];
            for (int64_t p = w0; p < hi; p++) rprev[(size_t)p] = rcur[(size_t)(p - w0)];
            progress.push_back((long)(next_w0 - w0));
            // path compression (ZS_FJ_COMPRESS=1): the link of a final position becomes the distance to the nearest *inserted*
            // position of its bucket below it -- what prev[] holds in the reference -- so that a search walks the all-position
            // chain only inside the window
            if (getenv("ZS_FJ_COMPRESS"))
                for (int64_t c = w0; c < next_w0; c++) {
                    int64_t c1 = c;
                    for (;;) {
                        const int l = cx.link[(size_t)c1];
                        if (!l) { c1 = -1; break; }
                        c1 -= l;
                        if (G[(size_t)c1]) break;
                    }
                    cx.link[(size_t)c] = (c1 < 0 || c - c1 > 32767) ? 0 : (uint16_t)(c - c1);
                }
            w0 = next_w0;
        }
        printf("slide W=%d K=%d T=%d: %zu sweeps, %.1f positions per sweep, searches per position %.2f (+ %.2f in the pre-sweeps), chain steps per search %.1f\n", W, K, T,
               progress.size(), (double)n / (double)progress.size(), (double)(cx.evals - pre_evals) / (double)n, (double)pre_evals / (double)n,
               (double)cx.steps / (double)cx.evals);
        pct(progress, "positions made final per sweep");
        return 0;
    }
    if (mode == "chunks") {
        // the stream cut into chunks of C positions, all parsed at once round after round: chunk k starts at the loop-top its
        // predecessor left through in the round before and reads the bits below it as that round left them; a chunk runs again
        // only if something it reads has changed.  A round without a change is the fixed point, which is the sequential parse
        // (induction from chunk 0, which reads nothing).
        const int64_t C = atoll(argv[4]);
        const int64_t nch = be / C + 1;
        const int64_t back = (kMaxDist + 300 + C - 1) / C + 1;  // chunks whose bits a chunk may read
        std::vector<uint8_t> prev((size_t)n + 600, 1), next, S((size_t)n + 600, 0);
        std::vector<int64_t> ent((size_t)nch), ext((size_t)nch, -1), ent_used((size_t)nch, -1);
        for (int64_t k = 0; k < nch; k++) ent[(size_t)k] = k * C;
        std::vector<uint8_t> changed((size_t)nch, 1), ch2((size_t)nch, 0);
        long runs = 0;
        int rounds = 0;
        for (;;) {
            next = prev;
            std::fill(ch2.begin(), ch2.end(), 0);
            long active = 0, wrongbits = 0;
            for (int64_t k = 0; k < nch; k++) {
                bool act = rounds == 0;
                for (int64_t j = std::max<int64_t>(0, k - back); j < k && !act; j++) act = changed[(size_t)j];
                if (!act) continue;
                active++;
                const int64_t e = k == 0 ? 0 : (rounds == 0 ? k * C : ext[(size_t)k - 1]);
                const int64_t hi = std::min<int64_t>((k + 1) * C, be + 1);
                const int64_t lo = std::max<int64_t>(0, e - kMaxDist - 300);
                std::copy(prev.begin() + lo, prev.begin() + e, S.begin() + lo);
                std::fill(S.begin() + e, S.begin() + std::min<int64_t>(hi + 600, n + 600), 0);
                int64_t t = e;
                while (t < hi) {
                    const uint32_t x = cx.eval(t, S);
                    cx.mark(S, t, x);
                    t += cx.adv(x);
                }
                // what it leaves: its bits on [e, t), its exit
                bool diff = t != ext[(size_t)k] || e != ent_used[(size_t)k];
                for (int64_t p = e; p < t; p++) {
                    if (next[(size_t)p] != S[(size_t)p]) diff = true;
                    next[(size_t)p] = S[(size_t)p];
                }
                ent_used[(size_t)k] = e;
                ext[(size_t)k] = t;
                ch2[(size_t)k] = diff;
            }
            rounds++;
            runs += active;
            long nchanged = 0;
            for (int64_t k = 0; k < nch; k++) nchanged += ch2[(size_t)k];
            for (int64_t p = 0; p <= be; p++) wrongbits += next[(size_t)p] != truth[(size_t)p];
            printf("  round %d: %ld of %lld chunks ran, %ld changed what they leave, %ld wrong bits\n", rounds, active, (long long)nch, nchanged, wrongbits);
            prev.swap(next);
            changed = ch2;
            if (!nchanged) break;
            if (rounds > 4000) break;
        }
        long wrong = 0;
        for (int64_t p = 0; p <= be; p++) wrong += prev[(size_t)p] != truth[(size_t)p];
        printf("chunks C=%lld: %lld chunks, %d rounds, %.2f runs per chunk, wrong bits at the end %ld\n", (long long)C, (long long)nch, rounds, (double)runs / (double)nch, wrong);
        return wrong ? 1 : 0;
    }
    return 2;
}
