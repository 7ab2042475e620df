// zs_fast_vec.h -- DeflateFast (levels 1-3, Deflate.Fast.cs:20-128) for the lanes of a wave: round 3's form, kept for the CPU
// model only (mode "fvec"; the product's kernels are zs_fast_sweep.hip).
//
// DeflateFast inserts only some positions into the hash chains (every loop-top, and the inside of a match no longer than
// max_lazy), so its chains depend on its own parse and the parse cannot be cut into independent pieces (SURVEY.md hard
// part 3).  What can be done in parallel is the search itself.  Two facts, checked against the oracle by the CPU model
// (tests/model, mode "fvec") before the kernel existed:
//
//  1. The reference's chain of a bucket is the chain of ALL positions of that bucket (K1's links, parse-independent)
//     with the positions that were never inserted left out: prev[c] was written when c was inserted and names the
//     nearest inserted position before it.  Given the set of inserted positions below a loop-top p -- a bitmap -- the
//     search at p is a function of the data: walk the all-position chain, skip what is not in the set, count the others
//     against max_chain (fv_search).
//  2. The next loop-top after p is p + 1 or p + match length: a chain of hops inside a window of 64 positions.
//
// So a wave takes the 64 positions from the current loop-top p0 on, every lane searches "its" position as if it were a
// loop-top, using the bitmap of everything below p0; then the hops are followed through the lanes' results (fv_resolve:
// scalar, v_readlane on the device).  A lane whose walk met a candidate at or above p0 -- a position whose membership is
// not known yet -- is not trusted: the window ends there and the next one starts at that loop-top, where the question
// has an answer (the first lane never meets such a candidate, so every window makes progress).  On text that happens
// for a few positions in a hundred; on periodic data nearly always, which is what the speculative chunk runs
// (zs_kernels.hip, zs_fast_run_kernel) are for.
//
// The refill quirk (Deflate.cs:1010-1013: the read at the first loop-top t within 261 bytes of the data end inserts t + 1
// before t) is applied where the resolve reaches such a loop-top: t + 1 joins the set; if t and t + 1 share a bucket (K1:
// link[t + 1] == 1) the search at t is dead, the one at t + 1 sees only t, and the chain is cut behind t (link[t] = 0,
// the 2-cycle prev[t] = t + 1, prev[t + 1] = t of the reference); otherwise the search at t + 1 is dead.
#pragma once
#include "../../zlibstream_amd/csrc/zs_core.h"

namespace zs {

constexpr int kFvLanes = 64;

struct FvResult {
    int len;      // >= 3: match, else no match
    int dist;
    int touched;  // the walk met a position >= p0: result not to be trusted
};

// Longest_match (Deflate.cs:1022-1100) at position q with prev_length 2 over the filtered all-position chain.
//   acc.link(c): distance to the previous position of c's bucket, 0 = none within kMaxDist
//   acc.ins(c):  c is in the inserted set (only asked for c < p0)
//   acc.lcp(q, c): common prefix length of the strings at q and c, at most kMaxMatch
// dead: the search does not happen (refill quirk); only_prev: the search sees just q - 1 (equal-bucket refill).
template <class Acc>
ZS_HD FvResult fv_search(const Acc &acc, int64_t q, int64_t p0, int max_chain, int nice, bool dead, bool only_prev) {
    FvResult r{2, 0, 0};
    if (dead) return r;
    if (only_prev) {
        const int len = acc.lcp(q, q - 1);
        if (len > 2) r.len = len, r.dist = 1;
        return r;
    }
    int found = 0;
    int64_t c = q;
    for (;;) {
        const int l = acc.link(c);
        if (!l) break;
        c -= l;
        const int64_t d = q - c;
        if (c < 1 || (found == 0 ? d > kMaxDist : d >= kMaxDist)) break;  // hash_head: <= MAX_DIST; later: cur_match > limit
        if (c >= p0) {
            r.touched = 1;
            break;
        }
        if (!acc.ins(c)) continue;
        found++;
        const int len = acc.lcp(q, c);
        if (len > r.len) {
            r.len = len, r.dist = (int)d;
            if (len >= nice) break;
        }
        if (found >= max_chain) break;
    }
    return r;
}

// The state that goes from window to window.
struct FvState {
    int64_t p;           // current loop-top
    int64_t nsyms;       // symbols emitted so far
    int64_t trigger;     // the next read event fires at the first loop-top >= trigger (data end before it - 261); < 0: none left
    int k_fired;         // read events fired so far
    int64_t preins;      // position pre-inserted by the last fired event, -1
    int ev_state;        // 0: none; 1: p is an event loop-top whose search is dead (equal buckets); 2: the search at p is
                         // dead (p was pre-inserted, buckets differ); 3: the search at p sees only p - 1
};

// What fv_resolve tells the caller about one window.
struct FvWindow {
    uint64_t tops;       // lanes that are loop-tops of the parse
    uint64_t ins_lo;     // positions p0 + k inserted, k < 64 ...
    uint32_t ins_hi;     // ... and k = 64 .. 95 (the inside of a short match reaching out of the window)
    int advance;         // the next window starts at p0 + advance
    int event_at;        // >= 0: lane whose loop-top fires a read event (the window ends before its symbol is decided)
};

// Follow the parse through one window.  res(i): the lane's FvResult; `limit`: lanes >= limit are not looked at (end of the
// tile / of the bulk part of the stream).  Lane 0 is always a loop-top and always trusted.
template <class Res>
ZS_HD FvWindow fv_resolve(const Res &res, int64_t p0, int limit, int max_lazy, int64_t trigger) {
    FvWindow w{0, 0, 0, 0, -1};
    int i = 0;
    while (i < limit) {
        const FvResult r = res(i);
        if (i > 0 && r.touched) break;
        if (i > 0 && trigger >= 0 && p0 + i >= trigger) {
            w.event_at = i;
            break;
        }
        w.tops |= 1ull << i;
        if (r.len >= kMinMatch) {
            // the loop-top itself is always inserted; the inside of the match only when it is short (Deflate.Fast.cs:81-104)
            const uint64_t m = r.len <= max_lazy ? ((1ull << r.len) - 1) : 1ull;
            w.ins_lo |= m << i;
            if (i + r.len > 64 && r.len <= max_lazy) w.ins_hi |= (uint32_t)(m >> (64 - i));
            i += r.len;
        } else {
            w.ins_lo |= 1ull << i;
            i += 1;
        }
    }
    w.advance = i;
    return w;
}

}  // namespace zs
