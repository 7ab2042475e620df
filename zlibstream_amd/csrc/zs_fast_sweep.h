// zs_fast_sweep.h -- DeflateFast (levels 1-3, Deflate.Fast.cs:20-128) as window-wide sweeps of a workgroup.
//
// DeflateFast inserts a position into the hash chains only when it is a loop-top or lies inside a match no longer than
// max_lazy (Deflate.Fast.cs:81-104), so its chains -- and with them its matches -- depend on its own parse.  Three facts
// make the parse a workgroup's job all the same (checked against the oracle by the CPU model, tests/model mode "fsweep",
// and measured by tests/model/fast_jacobi.cpp before the kernel existed):
//
//  1. The reference's chain of a bucket is the chain of ALL positions of that bucket (K1's links) with the positions that
//     were never inserted left out.  Given a bitmap of inserted positions the search at a position is a function of the
//     data (fs_search: walk the all-position chain, skip what is not in the set, count the rest against max_chain).
//  2. Sweeps.  Let w0 be a loop-top with everything below it final, and G a *guess* of the bitmap from w0 on.  A sweep
//     searches all W positions from w0 on at once under G, follows the hops of the parse from w0 through the results
//     (next loop-top = p + 1 or p + match length) and takes the bitmap that parse implies as the next guess.  A loop-top's
//     search only looks below itself, so by induction over the loop-tops: let d be the first position at which the implied
//     bitmap is not G; every loop-top on the path at or below d was searched under the bitmap of a parse that has been the
//     reference's so far, and its result is the reference's.  They are final; the next sweep starts behind the last of
//     them.  (The first form of the rule compared results with the sweep before's, which is the same thing seen through
//     the results and needs a sweep before: with the bits themselves a guess may come from anywhere -- "inserted"
//     everywhere, another workgroup, the same chunk's parse of the round before.)  Every sweep makes at least one loop-top
//     final, whatever the data and the guess; on text a sweep of 1024 positions makes ~395 final from the guess "inserted"
//     (461 at level 3), one of 4096 ~1100 (profiles/r04_fast_jacobi_convergence.txt), and nearly the whole window from a
//     guess that is nearly the parse.
//  3. Path compression.  Below w0 the set is final, so the link of a final position c may be replaced by the distance to
//     the nearest INSERTED position of its bucket below c -- what prev[] holds in the reference (fs_compress).  A walk that
//     skips uninserted positions finds the same candidates in the same order through either link, so the replacement
//     needs no synchronisation with the walkers; data whose matches are long (little is inserted: kennedy.xls walked
//     ~190 chain entries per search) is then searched in max_chain steps like any other.
//
// The refill quirk (Deflate.cs:1010-1013: the read at the first loop-top t within 261 bytes of the data end inserts t + 1
// before t) is applied when a sweep starts at such a loop-top: a sweep never searches beyond the next trigger, so the
// event's loop-top is always some sweep's w0.  t + 1 joins the set; if t and t + 1 share a bucket (link[t + 1] == 1) the
// search at t is dead, the one at t + 1 sees only t, and the chain is cut behind t (link[t] = 0: the 2-cycle prev[t] = t + 1,
// prev[t + 1] = t of the reference); otherwise the search at t + 1 is dead.
#pragma once
#include "zs_core.h"

namespace zs {

// a search's result: match length << 16 | distance; length 2, distance 0: no match
constexpr uint32_t kFsNone = 2u << 16;
constexpr uint32_t kFsFresh = 0xFFFFFFFFu;  // "not searched by the sweep before"
// flags above the result (length: 9 bits from bit 16): the search looked at final bits only -- its first candidate lay below
// the sweep's first loop-top -- so the result stays whatever later sweeps guess
constexpr uint32_t kFsExact = 1u << 30;
constexpr uint32_t kFsResMask = 0x01FFFFFFu;
ZS_HD int fs_len(uint32_t r) { return (int)((r & kFsResMask) >> 16); }
ZS_HD int fs_dist(uint32_t r) { return (int)(r & 0xFFFFu); }
ZS_HD int fs_adv(uint32_t r) { return fs_len(r) >= kMinMatch ? fs_len(r) : 1; }

// Longest_match (Deflate.cs:1022-1100) at position q with prev_length 2 over the all-position chain filtered by the set.
//   acc.link(c): distance to the previous position of c's bucket -- or, for a final position, to the previous inserted one
//                (fs_compress) --, 0 = none within reach
//   acc.ins(c):  c is in the set (final below the sweep's first loop-top, the guess from there on)
//   acc.lcp(q, c): common prefix length of the strings at q and c, at most kMaxMatch
// dead: the search does not happen (refill quirk); only_prev: the search sees just q - 1 (equal-bucket refill).
template <class Acc>
ZS_HD uint32_t fs_search(const Acc &acc, int64_t q, int max_chain, int nice, bool dead, bool only_prev) {
    if (dead) return kFsNone;
    if (only_prev) {
        const int len = acc.lcp(q, q - 1);
        return len > 2 ? ((uint32_t)len << 16) | 1u : kFsNone;
    }
    int found = 0, best = 2, bdist = 0;
    int64_t c = q;
    for (;;) {
        const int l = acc.link(c);
        if (!l) break;
        c -= l;
        const int64_t d = q - c;
        if (c < 1 || (found == 0 ? d > kMaxDist : d >= kMaxDist)) break;  // hash_head: <= MAX_DIST; later: cur_match > limit
        if (!acc.ins(c)) continue;
        found++;
        const int len = acc.lcp(q, c);
        if (len > best) {
            best = len, bdist = (int)d;
            if (len >= nice) break;
        }
        if (found >= max_chain) break;
    }
    return ((uint32_t)best << 16) | (uint32_t)bdist;
}

// The link of a final position c, compressed: the distance to the nearest inserted position of c's bucket below it, 0 when
// there is none a later search could still use.  Reads links that may or may not be compressed already (same answer).
template <class Acc>
ZS_HD int fs_compress(const Acc &acc, int64_t c) {
    int64_t c1 = c;
    for (;;) {
        const int l = acc.link(c1);
        if (!l) return 0;
        c1 -= l;
        if (c1 < 1 || c - c1 > kMaxDist) return 0;
        if (acc.ins(c1)) return (int)(c - c1);
    }
}

// What a loop-top puts into the set: itself, and the inside of its match when that is short (Deflate.Fast.cs:81-104).
ZS_HD int fs_inserted_span(uint32_t r, int max_lazy) {
    const int len = fs_len(r);
    return (len >= kMinMatch && len <= max_lazy) ? len : 1;
}

// The state that goes from sweep to sweep.
struct FsState {
    int64_t w0;        // first loop-top that is not final
    int64_t nsyms;     // symbols emitted so far
    int64_t trigger;   // the next read event fires at the first loop-top >= trigger (data end before it - 261); < 0: none left
    int k_fired;       // read events fired so far
    int64_t preins;    // position pre-inserted by the last fired event, -1
    int64_t dead_pos;  // position whose search is dead by the last event, -1
    int64_t only_pos;  // position whose search sees only its predecessor (equal-bucket event), -1
    int64_t ev_end;    // positions below this were searched by the sweep before (their results are there to compare with)
};

// ---- Rounds over the chunks of a stream (round 4; zs_fast_sweep_kernel in its chunk form, tests/model mode "frounds").
// A sweep's guess may as well come from another workgroup.  The stream is cut at positions b_0 = 0 < b_1 < ... (every read
// event's trigger is one of them, the spans between are cut into pieces of about `target` positions); chunk k parses from
// the first loop-top at or behind b_k -- the loop-top chunk k - 1 left through in the round before -- to the first at or
// behind b_{k+1}, reading the set below its entry as the chunks before it left it in the round before ("inserted" in round
// 0).  A chunk runs again only when something it reads has changed.  A round in which no chunk changes what it leaves is
// the fixed point, and the fixed point is the reference's parse: chunk 0 reads nothing, and a chunk whose predecessors are
// the reference's leaves the reference's (facts 1 and 2 above).  So the rounds end after at most as many as there are
// chunks -- a sequential parse in the worst case; measured (tests/model/fast_jacobi.cpp, mode chunks): kennedy.xls 4 rounds
// for 250 chunks, ptt5 9 for 126, text 26 (level 1) / 17 (level 3) for 115 chunks of 4096 positions.
struct FsChunk {
    int32_t stream;   // index into the batch
    int32_t b_lo;     // the chunk's loop-tops: from the first at or behind b_lo ...
    int32_t b_hi;     // ... to the last below b_hi (the stream's last chunk: body_end + 1)
    int32_t kfired0;  // read events fired below b_lo (the event whose trigger is b_lo fires at the chunk's first loop-top)
    int32_t first;    // the batch's index of the stream's first chunk
    int32_t idx;      // the chunk's number within its stream
    uint32_t prov_off;  // where its symbols go until the rounds are over (the batch's provisional symbol buffer)
    int32_t pad;
};
// what a chunk leaves (two copies, by the parity of the round)
struct FsMeta {
    int32_t E;        // the loop-top it started from
    int32_t X;        // the loop-top it handed over at (the first at or behind b_hi)
    int32_t nsyms;    // symbols of [E, X)
    int32_t cut;      // the loop-top of an equal-bucket read event (its chain is cut behind it), -1
    int32_t kend;     // read events fired at X
    int32_t preins;   // the position its read event inserted ahead, -1
    int32_t changed;  // this round's run left something else than the run before
    int32_t cur;      // which of its two bit planes holds what it left
    // A workgroup may take a range of consecutive chunks in turn, each reading what the ones before it in the range have
    // just left (this round) and of the others what the round before left; the ranges may differ from round to round.  What a
    // run has seen of a chunk j before it: j's output of the run's own round if j >= seen_lo, else of the round before.
    int32_t chg_round;  // the last round in which its run left something else
    int32_t ran_round;  // the round of its last run
    int32_t seen_lo;    // the first chunk of the range it ran in
    int32_t pad;
};
// Does a chunk that last ran in round `ran` within a range that began at chunk `seen_lo` have to run again because of chunk
// j before it, whose output last changed in round `chg`?
ZS_HD bool fs_stale(int chg, int j, int ran, int seen_lo) { return chg > ran || (chg == ran && j < seen_lo); }
constexpr int kFsChunkMax = 10240;  // a chunk's positions: one staging of the tile covers it (zs_fast_sweep.hip: TILE - W - 258 - 64)
constexpr int kFsChunkMin = 1024;
// trig(k), k = 1 .. ntrig: the trigger of read event k (the data end before it - 261), ascending.  Returns the length of the
// shortest chunk that has a chunk behind it (the kernel looks at 64 chunks in front of one: kFsMinSpan).
constexpr int kFsMinSpan = 520;  // 64 chunks of this length cover the 32 512 + 64 + 258 positions a chunk's tile reaches back
// k0 / s0: a run that takes the stream over at position s0 (behind a flush), whose event k0 - 1 is the read at that very
// position: the chunks begin there, and that event fires at the first chunk's first loop-top.
template <class Vec, class Trig>
inline int64_t fs_build_chunks(int stream, int64_t body_end, int ntrig, const Trig &trig, int target, Vec &out, int k0 = 1, int64_t s0 = 0) {
    const int32_t first = (int32_t)out.size();
    if (target > kFsChunkMax) target = kFsChunkMax;
    if (target < kFsChunkMin) target = kFsChunkMin;
    int64_t s = s0;
    int fired = 0;
    for (int k = k0;; k++) {
        const int64_t tr = k <= ntrig ? (int64_t)trig(k) : body_end + 1;
        const int64_t e = tr <= body_end ? tr : body_end + 1;
        const int64_t len = e - s;
        int64_t pieces = (len + target - 1) / target;
        if (pieces < 1) pieces = 1;
        const int64_t step = ((len + pieces - 1) / pieces + 63) & ~63LL;
        for (int64_t a = s; a < e; a += step) {
            FsChunk c;
            c.stream = stream, c.b_lo = (int32_t)a, c.b_hi = (int32_t)(a + step < e ? a + step : e), c.first = first;
            c.kfired0 = (a == s || s == 0) ? fired : fired + 1;  // (the event of the span's trigger fires in the span's first chunk)
            c.idx = (int32_t)out.size() - first;
            c.prov_off = 0, c.pad = 0;
            out.push_back(c);
        }
        if (e > body_end) break;
        s = e, fired = k - 1;  // (the event k itself fires inside the chunk that starts at its trigger)
    }
    int64_t shortest = 1 << 30;
    for (size_t i = (size_t)first; i + 1 < out.size(); i++) shortest = shortest < out[i].b_hi - out[i].b_lo ? shortest : out[i].b_hi - out[i].b_lo;
    return shortest;
}

}  // namespace zs
