// zs_rle.h -- CompressionStrategy.Rle (Deflate.Rle.cs:18-104) without a sequential parse.
//
// DeflateRle's match at a loop-top s is the run of the byte before it: the bytes from s on that equal data[s - 1], at least 3,
// at most 258 (and at most the lookahead, which only binds in the stream's last bytes).  A match therefore never leaves the
// run of equal bytes it lies in, a literal steps one byte -- so the parse meets every run's first position as a loop-top,
// and inside a run [a, e) it is fixed by a and e alone: a literal at a (the byte before it differs), then from a + 1 on
// matches of 258 while 258 bytes are left, one match of what is left if that is 3 or more, else one literal per byte.  No
// chains, no hash, no dependence from run to run: a position's part in the parse is a function of where its run began
// (a prefix maximum over "run starts here" flags) and of the three bytes around it.  Checked against the oracle's symbol
// trace by the CPU model (tests/model, mode "rle") before the kernels (zs_rle.hip) existed.
//
// Read events (Fill_window is called at the first loop-top with lookahead <= MAX_MATCH = 258, Deflate.Rle.cs:31-38 -- the
// other block functions ask at < MIN_LOOKAHEAD = 262) matter for two things only, since the strategy has no use for the hash
// tables they touch: which window base a block is flushed under (a stored block needs blockStart >= 0, Deflate.cs:953) and
// the state the tail engine is restored in.  For a single Write: event k >= 1 fires at the first loop-top >= 65 536 +
// 32 768 (k - 1) - 258, and always slides.
#pragma once
#include "zs_core.h"

namespace zs {

constexpr int64_t kRleSeg0 = kWindowSize - kMaxMatch;  // 65 278: from this loop-top on the first refill of a single Write has fired

// events of a single Write that have fired once loop-top p has run its Fill_window (cf. refills_fired_at)
ZS_HD int rle_refills_fired_at(int64_t p, int kl) {
    if (p < kRleSeg0) return 0;
    const int64_t k = 1 + (p - kRleSeg0) / kWSize;
    return k > kl ? kl : (int)k;
}

// What position p does in the parse, given the first position a of its run (data[a - 1] != data[a], or a == 0) and the bytes
// around it.  byte(q): data[q] (asked for q in [p, p + 258 + 8) only when p begins a match slot).  Returns 0: not a loop-top;
// 1: a loop-top that emits the literal data[p]; >= 3: a loop-top that emits a match of that length at distance 1.
template <class Byte>
ZS_HD int rle_role(const Byte &byte, int64_t p, int64_t a) {
    const int64_t o = p - a;
    if (o == 0) return 1;  // the byte before differs: no run of it here
    const int64_t k = (o - 1) % kMaxMatch;
    if (k == 0) {
        // a slot of the run begins: what is left of the run from here, up to 258
        const uint8_t c = byte(p);
        int len = 1;
        while (len < kMaxMatch && byte(p + len) == c) len++;
        return len >= kMinMatch ? len : 1;
    }
    // the slot began one byte before: a loop-top only if that slot was a literal, i.e. the run ends behind p
    if (k == 1) return byte(p + 1) != byte(p) ? 1 : 0;
    return 0;
}

// The hand-over loop-top: the body's loop-tops are those below H, the tail engine goes on from the first loop-top at or behind
// it.  No trigger may lie in (H - 258, H]: the events below have then fired before H, the others are the tail engine's.
// Returns H, or -1 when the stream is too short for a body.
inline int64_t rle_body_end(int64_t n) {
    int64_t h = n - 3 * kMinLookahead;
    for (bool moved = true; moved && h > 0;) {
        moved = false;
        for (int64_t t = kRleSeg0; t <= h; t += kWSize)
            if (t > h - kMaxMatch) h = t - kMaxMatch - 1, moved = true;
    }
    return h >= 4096 ? h : -1;
}

}  // namespace zs
