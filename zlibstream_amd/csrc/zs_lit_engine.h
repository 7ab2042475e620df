// zs_lit_engine.h -- the sequential tail of a stream, in the reference's own
// window coordinates.
//
// The bulk kernels treat every loop-top with lookahead >= MIN_LOOKAHEAD in
// absolute coordinates.  The last <= 261 input bytes (where the reference's
// Longest_match depends on prevLength through the `nice_match > lookahead`
// clip, where strings past `max_insert` are not inserted, where hashes and
// Compare256 read stale window bytes past the end of input, and where
// Fill_window may slide without reading) are finished by this engine: it
// rebuilds the reference's window / head / prev state at one loop-top from
// the bulk arrays and then runs DeflateSlow's loop literally.
//
// Reference: Deflate.Slow.cs:18-159, Deflate.cs:866-877 (InsertString),
// :967-1019 (Fill_window), :1022-1100 (Longest_match),
// Deflate.Intrinsics.cs:174-285 (SlideHash).
//
// All functions are ZS_HD; on the device one wave runs an engine instance with
// every lane executing the same scalar code (wave-uniform), lane 0 storing.
#pragma once
#include "zs_core.h"

// On the device one wave runs an engine with all lanes executing the scalar
// code uniformly; loops marked "lane-strided" split their iterations across
// the wave and are fenced by ZS_WAVE_SYNC().
#if defined(__HIP_DEVICE_COMPILE__)
#define ZS_WAVE_SYNC() __syncthreads()
#else
#define ZS_WAVE_SYNC() ((void)0)
#endif

namespace zs {

struct BlockRec {
    int64_t start;      // absolute input position of the block's first byte
    int64_t sym_start;  // index of the block's first symbol in the stream's symbol array
    int32_t stored_len; // bytes covered by the block's symbols
    int32_t nsyms;      // symbols in the block (END_BLOCK not counted)
    int32_t can_store;  // blockStart >= 0 in window coordinates at flush time
    int32_t eof;
};

struct LitEngine {
#ifdef ZS_FV_PROF
    long long pf[5] = {0, 0, 0, 0, 0};  // ticks: refill, insert, longest match, tally / advance, final flush
#endif
    // scratch (per stream)
    uint8_t *window;  // kWindowSize + 512 bytes
    uint16_t *head;   // kHashSize
    uint16_t *prev;   // kWSize
    const uint32_t *crc_tab;  // 4*256 table or nullptr (falls back to the bitwise form)
    // stream
    const uint8_t *data;
    int64_t n;
    // Write boundaries (ZlibOutputStream.WriteCore, ZlibOutputStream.cs:125-168):
    // wr_end[i] is the input offset at which Write i ends; input beyond the
    // current Write is not yet available to Fill_window.  n_wr <= 1 / nullptr
    // means one Write of the whole buffer.
    const int64_t *wr_end;
    int n_wr, cur_wr;
    // FlushMode of each Write (ZlibOptions.FlushMode; 0 NoFlush, 1 Partial, 2 Sync, 3 Full; nullptr: NoFlush).  A
    // Write under a flush mode is parsed to its last byte and closes its block (Deflate.Slow.cs:38-46,147-158);
    // wr_blk[i] (optional) receives the number of blocks flushed before Write i began.
    const uint8_t *wr_flush;
    int32_t *wr_blk;
    LevelCfg lv;
    int strategy, hash_variant;
    // reference state
    int64_t base;       // absolute position of window[0]
    int64_t avail_end;  // absolute end of the input already copied into the window
    int strstart, lookahead, match_length, match_start, match_available, prev_length, prev_match;
    int64_t block_start_abs;
    int defer_start;  // device tail: the start of the first block flushed is filled in later (see zs_body_blocks_kernel)
    int64_t block_sym_start;  // symbols emitted before the current block
    int block_syms;           // litBufsize - 1: 16383, or 8191 at level 0 (memLevel 7, Deflate.cs:246-249,298)
    // speculative chunk runs of DeflateFast (zs_kernels.hip, zs_fast_run_kernel); all off by default
    int64_t stop_abs;         // stop at the first loop-top >= this (no final flush); < 0: run to the end
    int64_t mark_abs;         // record the first loop-top >= this ...
    int64_t mark_pos, mark_nsyms;  // ... its position and the symbols emitted before it (-1 until reached)
    uint32_t *ins_bits;       // bitmap of inserted positions, bit (abs - ins_base); nullptr: not tracked
    int64_t ins_word_idx;     // the bitmap word being filled (-1: none) and its bits so far; le_flush_ins() stores it
    uint32_t ins_word;
    int64_t ins_base;
    int no_blocks;            // do not cut blocks (the caller does it after stitching the runs)
    int64_t *ev_log;          // loop-tops at which a refill read happened (the pre-insert positions - 1), up to 16
    int n_ev;
    // incremental streams (zs_stream_api.inc): a run holds the Writes that have arrived so far.  final_run == 0: the
    // stream goes on after this run's last Write -- where Deflate.Compress would return to its caller for more input
    // (NeedMore under NoFlush, Deflate.Slow.cs:38-46; BlockDone after a flush, Deflate.cs:583-613) the engine stops with
    // `suspended` set, at a loop-top, and a later run re-enters the block function from there
    int final_run, suspended;
    // ... or, for a run that hands the stream over to the bulk pipeline (zs_engine.hip: a long Write behind a flush), at the
    // first loop-top at or behind stop_abs with a full lookahead that no read event has touched (`stopped`): the state there
    // is a node of the lazy parse with the schedule's state a function of the position (zs_core.h build_geometry)
    int stopped;
    int64_t last_event_abs;  // loop-top of the last read event (-1: none in this run)
    // match records computed ahead for the loop-tops [pre_lo, pre_hi) (le_tail_record): two words per position, the
    // record for the full chain budget and the one for a quarter of it; nullptr: every search walks its chain
    const uint32_t *pre_rec;
    int64_t pre_lo, pre_hi;
    // with them, and if the window does not slide any more before the stream ends (le_no_head_ok): the hash heads are not
    // kept at all.  What head[h] would return at the insert of position q is prev[q] as the restore wrote it from K1's links
    // for q <= n - 6, and tail_head[q - (n - 5)] for the up to three positions behind them, whose hashes read past the data
    // (le_tail_head_bucket) -- bytes that a slide would change under them, hence the condition
    int no_head;
    int tail_head[3];
    // a table in device memory (the speculative chunk runs: window + prev fill the LDS) behind a direct-mapped cache in LDS:
    // every insert reads its bucket's head before anything else can go on, and a round trip to the memory side of the chip
    // is 1-2 us with the stores in front of it.  Slot h & (kHeadCache - 1): the value in hc_val, a nibble in hc_nib -- 8 | (h >> 13)
    // when the slot holds bucket h -- and an entry that makes room goes back to the table; hc_pres has a bit per bucket: 0 =
    // the bucket is empty in the table too (a first occurrence costs no load).  nullptr: no cache
    uint16_t *hc_val;
    int64_t n_match;  // loop-tops le_run_fast_hot took one at a time -- matches, literals outside the runs of literals (the speculative runs report it)
    uint32_t *hc_nib, *hc_pres, *hc_claim;  // (hc_claim: a bit per slot, all 0 between two uses: le_run_fast_hot's runs of first occurrences)
    // outputs
    uint32_t *syms;     // symbol i: dist << 16 | lc  (dist 0 = literal)
    int64_t nsyms;      // symbols emitted so far in this stream (body + tail)
    BlockRec *blocks;
    int nblocks;
};

// the optional features are off unless a caller turns them on
ZS_HD void le_defaults(LitEngine &e) {
    e.wr_end = nullptr, e.n_wr = 1, e.cur_wr = 0, e.wr_flush = nullptr, e.wr_blk = nullptr;
    e.stop_abs = -1, e.mark_abs = -1, e.mark_pos = -1, e.mark_nsyms = 0;
    e.ins_bits = nullptr, e.ins_base = 0, e.no_blocks = 0, e.ins_word_idx = -1, e.ins_word = 0;
    e.ev_log = nullptr, e.n_ev = 0;
    e.final_run = 1, e.suspended = 0, e.stopped = 0, e.last_event_abs = -1;
    e.pre_rec = nullptr, e.pre_lo = e.pre_hi = 0;
    e.no_head = 0, e.tail_head[0] = e.tail_head[1] = e.tail_head[2] = 0;
    e.hc_val = nullptr, e.hc_nib = nullptr, e.hc_pres = nullptr, e.hc_claim = nullptr, e.n_match = 0;
    e.block_syms = kBlockSyms, e.block_sym_start = 0, e.block_start_abs = 0, e.defer_start = 0;
    e.nsyms = 0, e.nblocks = 0;
}

// On the device the engine's window, prev and CRC tables are always LDS (zs_tail_kernel, zs_fast_run_kernel), but the
// struct holds generic pointers: an access through one is a flat instruction, whose wait covers the vector-memory
// counter as well -- i.e. the acknowledgement of the head / symbol stores the engine has just issued.  The hot accessors
// say where the memory is.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZS_LDS_PTR(T, p) ((__attribute__((address_space(3))) T *)(p))
#else
#define ZS_LDS_PTR(T, p) (p)
#endif
ZS_HD uint32_t le_hash(const LitEngine &e, uint32_t v) {
    if (e.hash_variant == kHashMul) return hash_mul(v) & kHashMask;
    if (e.crc_tab) return crc32c_u32_tab(ZS_LDS_PTR(const uint32_t, e.crc_tab), v) & kHashMask;
    return crc32c_u32_slow(v) & kHashMask;
}
ZS_HD uint32_t le_load32(const uint8_t *pg) {
    auto p = ZS_LDS_PTR(const uint8_t, pg);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
ZS_HD uint8_t le_wbyte(const LitEngine &e, int i) { return ZS_LDS_PTR(const uint8_t, e.window)[i]; }
// 8 bytes at any alignment, little-endian.  On the device three aligned dword loads and a funnel shift: the window is
// in LDS behind a generic pointer, where byte loads cost a round trip each and misaligned wide loads are slow.
ZS_HD uint64_t le_load64(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto q = ZS_LDS_PTR(const uint32_t, (const uint32_t *)((uintptr_t)p & ~(uintptr_t)3));
    const uint32_t a = q[0], b = q[1], c = q[2], sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
    const uint64_t lo = (uint64_t)a | ((uint64_t)b << 32);
    return sh ? (lo >> sh) | ((uint64_t)c << (64 - sh)) : lo;
#else
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
#endif
}
// number of equal leading bytes of a and b, at most kMaxMatch (Compare256 of the reference's Longest_match); reads up
// to 10 bytes past a + kMaxMatch / b + kMaxMatch (the window carries 512 bytes of padding)
ZS_HD int le_match_len(const uint8_t *a, const uint8_t *b) {
    int len = 0;
    while (len < kMaxMatch) {
        const uint64_t x = le_load64(a + len) ^ le_load64(b + len);
        if (x) {
            int tz = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            tz = (int)__builtin_ctzll(x);
#else
            uint64_t y = x;
            while (!(y & 1)) y >>= 1, tz++;
#endif
            len += tz >> 3;
            break;
        }
        len += 8;
    }
    return len < kMaxMatch ? len : kMaxMatch;
}

// The same for a wave whose 64 lanes all run the engine with the same state (zs_tail_kernel, zs_fast_run_kernel): lane l
// compares bytes [8 l, 8 l + 8), the first lane that differs gives the length -- one LDS round trip for the 258 bytes of
// an image row's match instead of 33 in sequence (sparse64 at level 1: 9 us per symbol, nearly all of it this loop).
ZS_HD int le_match_len_wave(const uint8_t *a, const uint8_t *b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = (int)(threadIdx.x & 63);
    const int off = lane < 33 ? lane * 8 : 0;  // 33 x 8 = 264 >= kMaxMatch; the other lanes repeat lane 0's bytes
    const uint64_t x = le_load64(a + off) ^ le_load64(b + off);
    const uint64_t differ = __ballot(x != 0 && lane < 33);
    if (!differ) return kMaxMatch;
    const int fl = (int)__builtin_ctzll(differ);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, fl), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), fl);
    const uint64_t xf = (uint64_t)lo | ((uint64_t)hi << 32);
    const int len = fl * 8 + (int)(__builtin_ctzll(xf) >> 3);
    return len < kMaxMatch ? len : kMaxMatch;
#else
    return le_match_len(a, b);
#endif
}

ZS_HD void le_flush_ins(LitEngine &e) {
    if (e.ins_bits && e.ins_word_idx >= 0) e.ins_bits[e.ins_word_idx] |= e.ins_word;
    e.ins_word_idx = -1, e.ins_word = 0;
}

constexpr int kHeadCacheBits = 13, kHeadCache = 1 << kHeadCacheBits;  // 8192 slots: 16 KiB of values, 4 KiB of nibbles; 4 KiB of presence bits
constexpr int kHeadCacheLds = 2 * kHeadCache + kHeadCache / 2 + kHashSize / 8 + kHeadCache / 8;
static_assert(kHashSize == 4 * kHeadCache, "a slot's tag is two bits");
// head[h] as InsertString reads it, with `str` left in its place (Deflate.cs:866-877), through the cache; `head` is the table
// in device memory.  One wave, every lane with the same arguments.
template <class Val, class Nib, class Pres, class Head>
ZS_HD int le_head_swap(Val val, Nib nib, Pres pres, Head head, uint32_t h, int str) {
    const uint32_t i = h & (uint32_t)(kHeadCache - 1), sh = 4u * (i & 7u), want = 8u | (h >> kHeadCacheBits);
    const uint32_t nw = nib[i >> 3], pw = pres[h >> 5], cv = val[i];
    const uint32_t have = (nw >> sh) & 15u;
    int cur;
    if (have == want) {
        cur = (int)cv;
    } else {
        cur = ((pw >> (h & 31u)) & 1u) ? (int)head[h] : 0;
        if (have & 8u) head[((have & 3u) << kHeadCacheBits) | i] = (uint16_t)cv;  // (not waited for)
        nib[i >> 3] = (nw & ~(15u << sh)) | (want << sh);
        pres[h >> 5] = pw | (1u << (h & 31u));
    }
    val[i] = (uint16_t)str;
    return cur;
}
// Deflate.cs:866-877
ZS_HD int le_insert(LitEngine &e, int str) {
    if (e.ins_bits) {
        // positions arrive in (almost) increasing order: the current bitmap word is kept in a register and stored when
        // the engine moves past it; only the refill pre-insert (s + 1 before s) can step back across a word boundary
        const int64_t i = e.base + str - e.ins_base;
        const int64_t wi = i >> 5;
        const uint32_t bit = 1u << (i & 31);
        if (wi == e.ins_word_idx) {
            e.ins_word |= bit;
        } else if (wi > e.ins_word_idx) {
            if (e.ins_word_idx >= 0) e.ins_bits[e.ins_word_idx] |= e.ins_word;  // every lane writes the same word: benign
            e.ins_word_idx = wi, e.ins_word = bit;
        } else {
            e.ins_bits[wi] |= bit;
        }
    }
    if (e.no_head) {
        // The table is in HBM scratch: every insert was a round trip (with the acknowledgements of the stores before it),
        // and building it for the tail of a stream took as long as a third of the tail's parse.
        const int64_t qa = e.base + str;
        if (qa <= e.n - 6) return ZS_LDS_PTR(const uint16_t, e.prev)[str & kWMask];
        const int cur = e.tail_head[qa - (e.n - 5)];
        ZS_LDS_PTR(uint16_t, e.prev)[str & kWMask] = (uint16_t)cur;
        return cur;
    }
    uint32_t h = le_hash(e, le_load32(e.window + str + 2));
    if (e.pre_rec) {
        // (with the table kept -- the window will slide before the stream ends:) a loop-top whose search was done ahead from
        // prev[] as the restore left it: what head[h] holds is prev[str], which is written already; only head[h] has to
        // follow, for the few inserts behind the records -- a store is not waited for
        const int64_t qa = e.base + str;
        if (qa >= e.pre_lo && qa < e.pre_hi) {
            e.head[h] = (uint16_t)str;
            return ZS_LDS_PTR(const uint16_t, e.prev)[str & kWMask];
        }
    }
    if (e.hc_val) {
        const int cur = le_head_swap(ZS_LDS_PTR(uint16_t, e.hc_val), ZS_LDS_PTR(uint32_t, e.hc_nib), ZS_LDS_PTR(uint32_t, e.hc_pres), e.head, h, str);
        if (cur != str) ZS_LDS_PTR(uint16_t, e.prev)[str & kWMask] = (uint16_t)cur;
        return cur;
    }
    int cur = e.head[h];
    if (cur != str) {
        ZS_LDS_PTR(uint16_t, e.prev)[str & kWMask] = (uint16_t)cur;
        e.head[h] = (uint16_t)str;
    }
    return cur;
}

// The slide of Fill_window (Deflate.cs:979-999: BlockCopy of the upper window half, SlideHash over head and
// prev).  kWSize is bit 15, so `v >= WSIZE ? v - WSIZE : 0` is "keep the low 15 bits where bit 15 is set"; on the
// device the three arrays (16-byte aligned) move 16 bytes per lane and step.
ZS_HD void le_slide(LitEngine &e, int lane, int nlanes) {
#if defined(__HIP_DEVICE_COMPILE__)
    struct V4 { uint32_t x[4]; };
    auto slide16 = [&](uint16_t *a, int count) {
        V4 *v = (V4 *)a;
        for (int i = lane; i < count / 8; i += nlanes) {
            V4 t = v[i];
#pragma unroll
            for (int k = 0; k < 4; k++) t.x[k] = t.x[k] & 0x7FFF7FFFu & (((t.x[k] >> 15) & 0x00010001u) * 0xFFFFu);
            v[i] = t;
        }
    };
    V4 *w = (V4 *)e.window;
    for (int i = lane; i < kWSize / 16; i += nlanes) w[i] = w[i + kWSize / 16];
    // (a table in device memory -- the speculative runs' -- eight loads in flight per lane: one at a time the 64 round trips of a
    // wave were 50 us of every slide)
    auto slide16_far = [&](uint16_t *a, int count) {
        typedef __attribute__((address_space(1))) V4 *g_v4p;
        const g_v4p v = (g_v4p)a;
        int i = lane;
        for (; i + 7 * nlanes < count / 8; i += 8 * nlanes) {
            V4 t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = v[i + u * nlanes];
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int k = 0; k < 4; k++) t[u].x[k] = t[u].x[k] & 0x7FFF7FFFu & (((t[u].x[k] >> 15) & 0x00010001u) * 0xFFFFu);
                v[i + u * nlanes] = t[u];
            }
        }
        for (; i < count / 8; i += nlanes) {
            V4 t = v[i];
#pragma unroll
            for (int k = 0; k < 4; k++) t.x[k] = t.x[k] & 0x7FFF7FFFu & (((t.x[k] >> 15) & 0x00010001u) * 0xFFFFu);
            v[i] = t;
        }
    };
    if (!e.no_head) {
        if (e.hc_val) slide16_far(e.head, kHashSize);
        else slide16(e.head, kHashSize);
    }
    if (e.hc_val) slide16(e.hc_val, kHeadCache);  // (a slot nobody holds slides too; a bucket that slides to 0 keeps its presence bit: a load finds the 0)
    slide16(e.prev, kWSize);
#else
    for (int i = lane; i < kWSize; i += nlanes) e.window[i] = e.window[i + kWSize];
    if (!e.no_head)
        for (int i = lane; i < kHashSize; i += nlanes) e.head[i] = (uint16_t)(e.head[i] >= kWSize ? e.head[i] - kWSize : 0);
    for (int i = lane; i < kWSize; i += nlanes) e.prev[i] = (uint16_t)(e.prev[i] >= kWSize ? e.prev[i] - kWSize : 0);
    if (e.hc_val)
        for (int i = lane; i < kHeadCache; i += nlanes) e.hc_val[i] = (uint16_t)(e.hc_val[i] >= kWSize ? e.hc_val[i] - kWSize : 0);
#endif
    for (int k = 0; k < 3; k++) e.tail_head[k] = e.tail_head[k] >= kWSize ? e.tail_head[k] - kWSize : 0;  // the heads kept apart (no_head)
}

// Deflate.cs:967-1019.  The whole remaining input is available (single Write
// already issued, flush == Finish).
ZS_HD_NOINLINE inline void le_fill_window(LitEngine &e, int lane, int nlanes) {
    do {
        int more = kWindowSize - e.lookahead - e.strstart;
        if (e.strstart >= kSlideAt) {
            ZS_WAVE_SYNC();
            // lane-strided: each i is read and written by one lane only
            le_slide(e, lane, nlanes);
            ZS_WAVE_SYNC();
            e.match_start -= kWSize;
            e.strstart -= kWSize;
            e.base += kWSize;
            more += kWSize;
        }
        const int64_t limit = (e.wr_end && e.cur_wr < e.n_wr) ? e.wr_end[e.cur_wr] : e.n;
        if (e.avail_end >= limit) return;
        int64_t avail = limit - e.avail_end;
        int cnt = avail < more ? (int)avail : more;
        uint8_t *dst = e.window + e.strstart + e.lookahead;
        const uint8_t *src = e.data + e.avail_end;
        ZS_WAVE_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
        if (e.hc_val && ((((uintptr_t)src) ^ ((uintptr_t)dst)) & 15) == 0 && cnt >= 64) {
            // (the speculative runs: window in LDS, data in device memory, the two equally aligned whenever the caller's buffer is
            // 16-byte aligned -- the window's first byte is a multiple of 32 KiB of the stream) 16 bytes per lane, four loads in
            // flight; byte by byte a read of 32 KiB was 512 round trips
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(1))) u32x4_t *g_c4p;
            typedef const __attribute__((address_space(1))) uint8_t *g_c1p;
            const int head = (int)((16u - ((uint32_t)(uintptr_t)src & 15u)) & 15u);
            const g_c1p sb = (g_c1p)src;
            auto db = ZS_LDS_PTR(uint8_t, dst);
            if (lane < head) db[lane] = sb[lane];
            const int n16 = (cnt - head) / 16;
            const g_c4p sv = (g_c4p)(src + head);
            auto dv = ZS_LDS_PTR(u32x4_t, (u32x4_t *)(dst + head));
            int i = lane;
            for (; i + 3 * nlanes < n16; i += 4 * nlanes) {
                const u32x4_t a = sv[i], b = sv[i + nlanes], c = sv[i + 2 * nlanes], d = sv[i + 3 * nlanes];
                dv[i] = a, dv[i + nlanes] = b, dv[i + 2 * nlanes] = c, dv[i + 3 * nlanes] = d;
            }
            for (; i < n16; i += nlanes) dv[i] = sv[i];
            for (int k = head + 16 * n16 + lane; k < cnt; k += nlanes) db[k] = sb[k];
        } else
#endif
        for (int i = lane; i < cnt; i += nlanes) dst[i] = src[i];  // lane-strided
        ZS_WAVE_SYNC();
        e.avail_end += cnt;
        e.lookahead += cnt;
        if (e.ev_log && e.n_ev < 16) {
            if (lane == 0) e.ev_log[e.n_ev] = e.base + e.strstart;
            e.n_ev++;
        }
        e.last_event_abs = e.base + e.strstart;
        if (e.lookahead >= kMinMatch) le_insert(e, e.strstart + 1);
    } while (e.lookahead < kMinLookahead && e.avail_end < ((e.wr_end && e.cur_wr < e.n_wr) ? e.wr_end[e.cur_wr] : e.n));
}

// `if (lookahead < MIN_LOOKAHEAD) { Fill_window(); if (still short && NoFlush) return NeedMore; }`
// (Deflate.Slow.cs:34-46): with NoFlush the function returns and is re-entered
// by the next Write, whose input the next Fill_window can then read; after the
// last Write the Finish call goes on with whatever lookahead is left.
ZS_HD_NOINLINE inline void le_refill(LitEngine &e, int lane, int nlanes, int &hash_head, int enough = kMinLookahead) {
    for (;;) {
        le_fill_window(e, lane, nlanes);
        if (e.lookahead >= enough) break;
        const bool flushes = e.wr_flush && e.cur_wr < e.n_wr && e.wr_flush[e.cur_wr] != 0;  // flush != NoFlush: go on with what is there
        if (!e.wr_end || e.cur_wr + 1 >= e.n_wr) {
            // no further Write in this run: Finish follows (final run), or the caller is asked for more input
            if (!e.final_run && !flushes) e.suspended = 1;
            break;
        }
        if (flushes) break;
        e.cur_wr++;
        if (e.wr_blk && lane == 0) e.wr_blk[e.cur_wr] = e.nblocks;
        hash_head = 0;  // DeflateSlow's local is reset on every entry (Deflate.Slow.cs:20)
    }
}

// Deflate.cs:1022-1100
ZS_HD_NOINLINE inline int le_longest_match(LitEngine &e, int cur_match) {
    int chain_length = e.lv.chain;
    const uint8_t *scan = e.window + e.strstart;
    int best_len = e.prev_length;
    int limit = e.strstart > kMaxDist ? e.strstart - kMaxDist : 0;
    int nice = e.lv.nice;
    int ms = e.match_start;
    if (best_len == 0) best_len = 1;
    if (e.prev_length >= e.lv.good) chain_length >>= 2;
    if (nice > e.lookahead) nice = e.lookahead;
    // One candidate is one round trip: the four bytes the reference tests (Deflate.cs:1072-1078) and the candidate's link
    // are requested together, the scan's own bytes are kept in registers (the engine is a single dependency chain of LDS
    // round trips; the tests in sequence were four of them)
    auto wl = ZS_LDS_PTR(const uint8_t, e.window);
    auto pl = ZS_LDS_PTR(const uint16_t, e.prev);
    auto sc = wl + e.strstart;
    const uint8_t s0 = sc[0], s1 = sc[1];
    uint8_t sb0 = sc[best_len - 1], sb1 = sc[best_len];
    do {
        if (cur_match >= e.strstart) break;
        const uint8_t *m = e.window + cur_match;
        auto ml = wl + cur_match;
        const uint8_t mb1 = ml[best_len], mb0 = ml[best_len - 1], m0 = ml[0], m1 = ml[1];
        const int next = pl[cur_match & kWMask];
        if (!((mb1 ^ sb1) | (mb0 ^ sb0) | (m0 ^ s0) | (m1 ^ s1))) {
            const int len = le_match_len_wave(scan, m);  // bytes 0 and 1 are known to match; every lane of the wave is here
            if (len > best_len) {
                ms = cur_match;
                best_len = len;
                if (len >= nice) break;
                sb0 = sc[best_len - 1], sb1 = sc[best_len];
            }
        }
        cur_match = next;
    } while (cur_match > limit && --chain_length != 0);
    e.match_start = ms;
    return best_len < e.lookahead ? best_len : e.lookahead;
}

// The searches of the last loop-tops of a stream, ahead of the parse and one position per lane.  For a slow level every
// position below a loop-top q is in the chains when q is searched (Deflate.Slow.cs:58,121-129), so q's chain is prev[] as
// the restore builds it from K1's links -- for the tail positions too (le_restore_prev) -- and, as long as the lookahead
// is at least max_lazy, the result depends on the parse only through prev_length >= good_match (a quarter of the chain
// budget, Deflate.cs:1036-1039): a search starts from best_len = prev_length < max_lazy <= min(nice_match, lookahead), so
// the candidate that ends it early is the same one whatever prev_length is, and the longest of the visited candidates
// (the first of them) is the answer whenever it beats prev_length (le_match_from_record).  This is Longest_match
// (Deflate.cs:1022-1100) for prev_length = 2 with the chain budget given, at loop-top `str` (window index) with `lookahead`
// bytes left; returns pack_match(len, dist) or kNoMatch.  Reads the engine, changes nothing.
ZS_HD_NOINLINE inline uint32_t le_tail_record(const LitEngine &e, int str, int lookahead, int chain_length) {
    int cur_match = ZS_LDS_PTR(const uint16_t, e.prev)[str & kWMask];  // what InsertString(str) will return
    if (cur_match == 0 || str - cur_match > kMaxDist) return kNoMatch;
    const uint8_t *scan = e.window + str;
    int best_len = kMinMatch - 1, ms = 0;
    const int limit = str > kMaxDist ? str - kMaxDist : 0;
    int nice = e.lv.nice;
    if (nice > lookahead) nice = lookahead;
    auto wl = ZS_LDS_PTR(const uint8_t, e.window);
    auto pl = ZS_LDS_PTR(const uint16_t, e.prev);
    auto sc = wl + str;
    const uint8_t s0 = sc[0], s1 = sc[1];
    uint8_t sb0 = sc[best_len - 1], sb1 = sc[best_len];
    do {
        if (cur_match >= str) break;
        const uint8_t *m = e.window + cur_match;
        auto ml = wl + cur_match;
        const uint8_t mb1 = ml[best_len], mb0 = ml[best_len - 1], m0 = ml[0], m1 = ml[1];
        const int next = pl[cur_match & kWMask];
        if (!((mb1 ^ sb1) | (mb0 ^ sb0) | (m0 ^ s0) | (m1 ^ s1))) {
            const int len = le_match_len(scan, m);
            if (len > best_len) {
                ms = cur_match;
                best_len = len;
                if (len >= nice) break;
                sb0 = sc[best_len - 1], sb1 = sc[best_len];
            }
        }
        cur_match = next;
    } while (cur_match > limit && --chain_length != 0);
    if (best_len > lookahead) best_len = lookahead;
    return best_len >= kMinMatch ? pack_match(best_len, str - ms) : kNoMatch;
}
constexpr int kTailRecMax = 288;  // positions of a tail whose searches are done ahead (zs_tail_kernel): <= 261 - max_lazy + a few
// The positions le_tail_record may be asked for when the engine takes over at loop-top p with all input read: links
// exist up to n - 6, and the lookahead must not fall below max_lazy.
ZS_HD int64_t le_tail_record_end(const LitEngine &e) {
    const int64_t a = e.n - 5, b = e.n - e.lv.lazy + 1;
    return a < b ? a : b;
}
// The positions n - 5 .. n - 3 are inserted like every other (Deflate.Slow.cs:58,121-129: lookahead >= MIN_MATCH) but hash
// bytes behind the data, so K1 has no links for them.  What head[] holds for their buckets when they are inserted is the
// nearest position below them with the same bucket -- every position up to n - 3 is in the table by then, in order.
// Does q (an absolute position below n - 5 + k, in the window) qualify for target k?  The callers keep the largest.
ZS_HD uint32_t le_tail_head_bucket(const LitEngine &e, int k) {
    const int64_t t = e.n - 5 + k;
    return t >= 1 && t >= e.base ? le_hash(e, le_load32(e.window + (t - e.base) + 2)) : 0xFFFFFFFFu;
}
// The engine taking over at loop-top p with the window at `base` and everything read would slide first thing (its first
// pass finds lookahead < MIN_LOOKAHEAD and strstart at or behind the slide point, and Fill_window has nothing to read):
// one Write, the stream's last run, no pre-insert pending.  The caller restores it at base + WSIZE instead -- the same
// window image (the upper half keeps its bytes, le_restore's stale-half rule), prev[] and scalars as after the slide.
ZS_HD bool le_tail_preslide(const LitEngine &e, int64_t p, int64_t base, int64_t avail_end, int64_t preins) {
    return e.final_run && !e.wr_end && avail_end >= e.n && e.n - p < kMinLookahead && preins < p && p - base >= kSlideAt;
}
// Can the searches of the rest be done ahead (le_tail_record)?  Everything has been read and the engine is in the run's last
// Write, which the stream ends with (final_run) or which closes under a flush mode: either way the parse goes down to
// lookahead 0 on what is there (Deflate.Slow.cs:38-46), and no read changes the chains on the way.
ZS_HD bool le_tail_reads_done(const LitEngine &e) {
    if (e.avail_end != e.n || e.avail_end <= 0) return false;
    if (e.wr_end && e.cur_wr + 1 < e.n_wr) return false;
    const bool flushes = e.wr_flush && e.cur_wr < e.n_wr && e.wr_flush[e.cur_wr] != 0;
    return e.final_run || flushes;
}
// No slide from here on: every loop-top of the rest stays below the point where Fill_window slides (Deflate.cs:979).
ZS_HD bool le_no_head_ok(const LitEngine &e) { return e.n - e.base <= kSlideAt; }
// Longest_match's result at the current loop-top from its record.
ZS_HD int le_match_from_record(LitEngine &e) {
    const uint32_t *r = e.pre_rec + 2 * (e.base + e.strstart - e.pre_lo);
    const uint32_t rec = e.prev_length >= e.lv.good ? r[1] : r[0];
    int best_len = e.prev_length;
    if (best_len == 0) best_len = 1;
    if (rec != kNoMatch && match_len(rec) > best_len) {
        best_len = match_len(rec);
        e.match_start = e.strstart - match_dist(rec);
    }
    return best_len < e.lookahead ? best_len : e.lookahead;
}

ZS_HD void le_flush_block(LitEngine &e, bool eof, int lane, int flush = 0) {
    int64_t end_abs = e.base + e.strstart;
    if (lane == 0) {
        BlockRec &b = e.blocks[e.nblocks];
        b.start = e.block_start_abs;
        b.sym_start = e.block_sym_start;
        b.stored_len = (int32_t)(end_abs - e.block_start_abs);
        b.nsyms = (int32_t)(e.nsyms - e.block_sym_start);
        b.can_store = e.block_start_abs >= e.base;
        if (e.defer_start) b.can_store = -(1 + (int32_t)(e.base / kWSize));  // start unknown here: leave the window base
        b.eof = (eof ? 1 : 0) | (flush << 1);  // bit 0: last block; bits 1..2: the FlushMode marker that follows the block
    }
    e.nblocks++;
    e.block_start_abs = end_abs;
    e.defer_start = 0;
    e.block_sym_start = e.nsyms;
}

// lookahead == 0 with the current Write ending under FlushMode Partial / Sync / Full?
ZS_HD bool le_write_flushes(const LitEngine &e) { return e.wr_flush && e.cur_wr < e.n_wr && e.wr_flush[e.cur_wr] != 0; }
// The end of such a Write: Flush_block_only(false), BlockDone; Deflate.Compress then sends the marker and, for
// FullFlush, forgets the hash heads (Deflate.cs:583-604).  The next Write (or Finish) enters the block function again.
ZS_HD_NOINLINE inline void le_end_write(LitEngine &e, int lane, int nlanes) {
    const int f = e.wr_flush[e.cur_wr];
    le_flush_block(e, false, lane, f);
    if (f == 3) {
        ZS_WAVE_SYNC();
        for (int i = lane; i < kHashSize; i += nlanes) e.head[i] = 0;
        if (e.hc_val) {
            for (int i = lane; i < kHeadCache / 8; i += nlanes) e.hc_nib[i] = 0;
            for (int i = lane; i < kHashSize / 32; i += nlanes) e.hc_pres[i] = 0;
        }
        ZS_WAVE_SYNC();
    }
    e.cur_wr++;
    if (e.wr_blk && e.cur_wr < e.n_wr && lane == 0) e.wr_blk[e.cur_wr] = e.nblocks;
}

// returns true when the block must be flushed (Deflate.cs:910-948)
ZS_HD bool le_tally(LitEngine &e, int dist, int lc, int lane) {
    if (lane == 0) e.syms[e.nsyms] = ((uint32_t)dist << 16) | (uint32_t)lc;
    e.nsyms++;
    return !e.no_blocks && (e.nsyms - e.block_sym_start) == e.block_syms;
}

// Deflate.Slow.cs:18-159 with flush == Finish, run to the end of the stream.
#if defined(ZS_FV_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define LE_PF_T0() long long pf_t_ = wall_clock64()
#define LE_PF(i) { const long long now_ = wall_clock64(); e.pf[i] += now_ - pf_t_; pf_t_ = now_; }
#else
#define LE_PF_T0()
#define LE_PF(i)
#endif
ZS_HD_NOINLINE inline void le_run_slow(LitEngine &e, int lane, int nlanes) {
    int hash_head = 0;
    for (;;) {
        LE_PF_T0();
        if (e.lookahead < kMinLookahead) {
            // the visits of a stream's last 261 loop-tops find nothing to read and nothing to slide: not worth the call
            const bool idle = !e.wr_end && e.final_run && e.avail_end >= e.n && e.strstart < kSlideAt;
            if (!idle) le_refill(e, lane, nlanes, hash_head);
            if (e.suspended) return;
            if (e.lookahead == 0) {
                if (!le_write_flushes(e)) break;
                if (e.match_available != 0) {
                    le_tally(e, 0, le_wbyte(e, e.strstart - 1), lane);
                    e.match_available = 0;
                }
                le_end_write(e, lane, nlanes);
                hash_head = 0;
                continue;
            }
        }
        LE_PF(0);
        if (e.lookahead >= kMinMatch) hash_head = le_insert(e, e.strstart);
        LE_PF(1);
        e.prev_length = e.match_length;
        e.prev_match = e.match_start;
        e.match_length = kMinMatch - 1;
        if (hash_head != 0 && e.prev_length < e.lv.lazy && e.strstart - hash_head <= kMaxDist) {
            if (e.strategy != kHuffmanOnly) {
                const int64_t qa = e.base + e.strstart;
                e.match_length = (e.pre_rec && qa >= e.pre_lo && qa < e.pre_hi) ? le_match_from_record(e) : le_longest_match(e, hash_head);
            }
            if (e.match_length <= 5 &&
                (e.strategy == kFiltered || (e.match_length == kMinMatch && e.strstart - e.match_start > kTooFar)))
                e.match_length = kMinMatch - 1;
        }
        LE_PF(2);
        if (e.prev_length >= kMinMatch && e.match_length <= e.prev_length) {
            int max_insert = e.strstart + e.lookahead - kMinMatch;
            bool bflush = le_tally(e, e.strstart - 1 - e.prev_match, e.prev_length - kMinMatch, lane);
            e.lookahead -= e.prev_length - 1;
            e.prev_length -= 2;
            if (e.no_head) {
                // without the hash heads the inserts inside a match have nothing to do (their prev[] entries are written, what
                // they return is not looked at) except for the three positions behind K1's links, which leave their prev[]
                const int last = e.strstart + e.prev_length;  // the last position the loop below would visit
                int s0 = (int)(e.n - 5 - e.base);
                if (s0 <= e.strstart) s0 = e.strstart + 1;
                for (int str = s0; str <= last && str <= max_insert; str++) (void)le_insert(e, str);
                // hash_head outlives the iteration (it is only assigned while lookahead >= MIN_MATCH): what the last insert returned
                const int li = last <= max_insert ? last : max_insert;
                if (li > e.strstart) hash_head = le_insert(e, li);
                e.strstart = last;
                e.prev_length = 0;
            } else {
                do {
                    if (++e.strstart <= max_insert) hash_head = le_insert(e, e.strstart);
                } while (--e.prev_length != 0);
            }
            e.match_available = 0;
            e.match_length = kMinMatch - 1;
            e.strstart++;
            if (bflush) le_flush_block(e, false, lane);
        } else if (e.match_available != 0) {
            bool bflush = le_tally(e, 0, le_wbyte(e, e.strstart - 1), lane);
            if (bflush) le_flush_block(e, false, lane);
            e.strstart++;
            e.lookahead--;
        } else {
            e.match_available = 1;
            e.strstart++;
            e.lookahead--;
        }
        LE_PF(3);
        if (e.stop_abs >= 0 && e.base + e.strstart >= e.stop_abs && e.lookahead >= kMinLookahead && e.base + e.strstart > e.last_event_abs + 1) {
            e.stopped = 1;
            return;
        }
    }
    LE_PF_T0();
    if (e.match_available != 0) {
        le_tally(e, 0, le_wbyte(e, e.strstart - 1), lane);
        e.match_available = 0;
    }
    le_flush_block(e, true, lane);
    LE_PF(4);
}

// Deflate.Fast.cs:20-128 with flush == Finish, run to the end of the stream.
ZS_HD_NOINLINE inline void le_run_fast(LitEngine &e, int lane, int nlanes) {
    for (;;) {
        LE_PF_T0();
        if (e.lookahead < kMinLookahead) {
            int dummy = 0;
            le_refill(e, lane, nlanes, dummy);
            if (e.suspended) return;
            if (e.lookahead == 0) {
                if (!le_write_flushes(e)) break;
                le_end_write(e, lane, nlanes);
                continue;
            }
        }
        if (e.mark_abs >= 0 && e.mark_pos < 0 && e.base + e.strstart >= e.mark_abs) e.mark_pos = e.base + e.strstart, e.mark_nsyms = e.nsyms;
        if (e.stop_abs >= 0 && e.base + e.strstart >= e.stop_abs) return;  // a speculative run ends at a loop-top, nothing is flushed
        LE_PF(0);
        int hash_head = 0;
        if (e.lookahead >= kMinMatch) hash_head = le_insert(e, e.strstart);
        LE_PF(1);
        if (hash_head != 0 && e.strstart - hash_head <= kMaxDist) {
            if (e.strategy != kHuffmanOnly) e.match_length = le_longest_match(e, hash_head);
        }
        LE_PF(2);
        bool bflush;
        if (e.match_length >= kMinMatch) {
            bflush = le_tally(e, e.strstart - e.match_start, e.match_length - kMinMatch, lane);
            e.lookahead -= e.match_length;
            if (e.match_length <= e.lv.lazy && e.lookahead >= kMinMatch) {
                e.match_length--;
                do {
                    e.strstart++;
                    le_insert(e, e.strstart);
                } while (--e.match_length != 0);
                e.strstart++;
            } else {
                e.strstart += e.match_length;
                e.match_length = 0;
            }
        } else {
            bflush = le_tally(e, 0, le_wbyte(e, e.strstart), lane);
            e.lookahead--;
            e.strstart++;
        }
        LE_PF(3);
        if (bflush) le_flush_block(e, false, lane);
        LE_PF(4);
    }
    le_flush_block(e, true, lane);
}


#if defined(__HIPCC__)
// le_run_fast for the speculative chunk runs (zs_fast_run_kernel), statement for statement, with the state that every
// symbol touches in registers.  The engine's struct lives in the wave's scratch memory -- it goes to the out-of-line pieces by
// reference -- and a symbol of the generic loop is some forty dependent round trips to it: 0.9 us, of which the searches
// and compares are a tenth (sparse64 at level 1: 5.8 of 7.2 ms).  Here the struct is read at entry and written back where the
// rare pieces need it (a read: once per 32 Ki positions; the stream's last 261 positions go to the generic loop).  The hashes
// of 64 consecutive positions are made at once, a position per lane, and picked with v_readlane: one LDS round trip for the
// window bytes and one for the CRC tables per 64 inserts instead of per insert; the bitmap of inserted positions is
// written with atomics nobody waits for.  Needs: one Write, no block cuts (no_blocks), the head cache.
__device__ inline void le_run_fast_hot(LitEngine &e, int lane) {
    typedef __attribute__((address_space(1))) uint32_t *g_u32p;
    typedef __attribute__((address_space(1))) uint16_t *g_u16p;
    if (!e.hc_val || !e.no_blocks || e.wr_end || e.pre_rec || e.no_head || !e.ins_bits || (e.hash_variant != kHashMul && !e.crc_tab)) {
        le_run_fast(e, lane, 64);
        return;
    }
    auto wl = ZS_LDS_PTR(const uint8_t, e.window);
    auto pl = ZS_LDS_PTR(uint16_t, e.prev);
    auto cval = ZS_LDS_PTR(uint16_t, e.hc_val);
    auto cnib = ZS_LDS_PTR(uint32_t, e.hc_nib);
    auto cpres = ZS_LDS_PTR(uint32_t, e.hc_pres);
    auto crc = ZS_LDS_PTR(const uint32_t, e.crc_tab);
    const uint8_t *const win = e.window;
    const g_u16p head_g = (g_u16p)e.head;
    const g_u32p syms = (g_u32p)e.syms, bits = (g_u32p)e.ins_bits;
    const bool mul = e.hash_variant == kHashMul, search = e.strategy != kHuffmanOnly;
    const int lazy = e.lv.lazy, nice_cfg = e.lv.nice;
    // (prev_length is not touched by DeflateFast: Longest_match starts from what the restore left)
    const int best0 = e.prev_length == 0 ? 1 : e.prev_length, chain_eff = e.prev_length >= e.lv.good ? e.lv.chain >> 2 : e.lv.chain;
    const int64_t mark_abs = e.mark_abs, stop_abs = e.stop_abs;
    int strstart = e.strstart, lookahead = e.lookahead, match_length = e.match_length, match_start = e.match_start;
    int nsyms = (int)e.nsyms, ins_idx = (int)e.ins_word_idx, n_match = 0;
    uint32_t ins_word = e.ins_word;
    bool marked = e.mark_pos >= 0;
    int ins_off = 0, mark_rel = 0, stop_rel = 0;
    auto rel = [](int64_t a, int64_t base) -> int {
        const int64_t r = a - base;
        return r > (1 << 30) ? (1 << 30) : r < -(1 << 30) ? -(1 << 30) : (int)r;
    };
    auto rebase = [&]() { ins_off = (int)(e.base - e.ins_base), mark_rel = rel(mark_abs, e.base), stop_rel = rel(stop_abs, e.base); };
    rebase();
    auto write_back = [&]() {
        e.strstart = strstart, e.lookahead = lookahead, e.match_length = match_length, e.match_start = match_start;
        e.nsyms = nsyms, e.ins_word_idx = ins_idx, e.ins_word = ins_word, e.n_match = n_match;
    };
    int hb = -(1 << 20);  // the hashes of positions [hb, hb + 64), lane l: position hb + l; and their bytes
    uint32_t hv = 0, lit = 0;
    auto fill = [&](int at) {
        hb = at;
        const uint32_t v = le_load32(win + hb + lane + 2);
        lit = wl[hb + lane];
        hv = (mul ? hash_mul(v) : crc32c_u32_tab(crc, v)) & kHashMask;
    };
    auto cclaim = ZS_LDS_PTR(uint32_t, e.hc_claim);
    int lit_streak = 0;  // literals in a row whose buckets had no head
    auto insert = [&](int str) -> int {
        {
            const int i = ins_off + str, wi = i >> 5;
            const uint32_t bit = 1u << (i & 31);
            if (wi == ins_idx) {
                ins_word |= bit;
            } else if (wi > ins_idx) {
                if (ins_idx >= 0) (void)__hip_atomic_fetch_or(bits + ins_idx, ins_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ins_idx = wi, ins_word = bit;
            } else {
                (void)__hip_atomic_fetch_or(bits + wi, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if ((unsigned)(str - hb) >= 64u) fill(str);
        const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hv, __builtin_amdgcn_readfirstlane(str - hb));
        const int cur = le_head_swap(cval, cnib, cpres, head_g, h, str);
        if (cur != str) pl[str & kWMask] = (uint16_t)cur;
        return cur;
    };
#ifdef ZS_FV_PROF
    long long hp[5] = {0, 0, 0, 0, 0}, hp_t = wall_clock64();
#define HOT_PF(i) { const long long now_ = wall_clock64(); hp[i] += now_ - hp_t; hp_t = now_; }
#define HOT_PF_OUT() { for (int k_ = 0; k_ < 5; k_++) e.pf[k_] += hp[k_]; }
#else
#define HOT_PF(i)
#define HOT_PF_OUT()
#endif
    for (;;) {
        if (lookahead < kMinLookahead) {
            write_back();
            int dummy = 0;
            le_refill(e, lane, 64, dummy);
            if (e.suspended) return;
            if (e.lookahead < kMinLookahead) {  // no more to read: the stream's last loop-tops
                HOT_PF_OUT();
                le_run_fast(e, lane, 64);
                return;
            }
            strstart = e.strstart, lookahead = e.lookahead, match_start = e.match_start;
            ins_idx = (int)e.ins_word_idx, ins_word = e.ins_word;  // (the read's pre-insert)
            rebase();
            hb = -(1 << 20);  // (the window may have slid)
            HOT_PF(4);
        }
        if (mark_abs >= 0 && !marked && strstart >= mark_rel) e.mark_pos = e.base + strstart, e.mark_nsyms = nsyms, marked = true;
        if (stop_abs >= 0 && strstart >= stop_rel) {
            write_back();
            HOT_PF_OUT();
            return;
        }
        HOT_PF(0);
        // A run of first occurrences (the first period of an image row: 256 literals): positions whose buckets have no head at
        // all -- the presence bit says so without a load -- are literals whatever else happens, and as long as no two of them
        // share a slot of the cache their inserts do not meet either: up to 64 loop-tops at once, a position per lane.  A lane
        // is out if its bucket has a head, if a lane below it claimed its slot (a returning LDS atomic on a bit per slot), or if
        // its loop-top would read, mark or stop; the lanes below the first such lane go.
        if (lit_streak >= 2 && strstart > 0 && e.hc_claim) {
            if (hb != strstart) fill(strstart);
            int cap = lookahead - kMinLookahead + 1;
            if (mark_abs >= 0 && !marked && mark_rel - strstart < cap) cap = mark_rel - strstart;
            if (stop_abs >= 0 && stop_rel - strstart < cap) cap = stop_rel - strstart;
            const uint32_t h = hv, i = h & (uint32_t)(kHeadCache - 1), sh = 4u * (i & 7u), cbit = 1u << (i & 31u);
            const uint32_t pw = cpres[h >> 5], have = (cnib[i >> 3] >> sh) & 15u, cv = cval[i];
            const uint32_t was = __hip_atomic_fetch_or(cclaim + (i >> 5), cbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // no head anywhere, or the slot holds the bucket and its head is out of reach (the same strings 64 KiB earlier): no search
            const int str = strstart + lane;
            const bool absent = ((pw >> (h & 31u)) & 1u) == 0, hit = have == (8u | (h >> kHeadCacheBits));
            const bool far = hit && (cv == 0 || str - (int)cv > kMaxDist);
            const bool bad = !(absent || far) || (was & cbit) != 0 || lane >= cap;
            const uint64_t bm = __ballot(bad);
            (void)__hip_atomic_fetch_and(cclaim + (i >> 5), ~cbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int m = bm ? (int)__builtin_ctzll(bm) : 64;
            if (m > 0) {
                if (lane < m) {
                    if (!hit) {
                        if (have & 8u) head_g[((have & 3u) << kHeadCacheBits) | i] = (uint16_t)cv;
                        (void)__hip_atomic_fetch_and(cnib + (i >> 3), ~(15u << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_or(cnib + (i >> 3), (8u | (h >> kHeadCacheBits)) << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_or(cpres + (h >> 5), 1u << (h & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    cval[i] = (uint16_t)str;
                    pl[str & kWMask] = hit ? (uint16_t)cv : (uint16_t)0;
                    syms[nsyms + lane] = lit;
                }
                // the bits of positions [strstart, strstart + m), as m inserts in a row leave them
                for (int i0 = ins_off + strstart, left = m; left > 0;) {
                    const int wi = i0 >> 5, take = 32 - (i0 & 31) < left ? 32 - (i0 & 31) : left;
                    const uint32_t mask = (take >= 32 ? 0xFFFFFFFFu : ((1u << take) - 1u)) << (i0 & 31);
                    if (wi == ins_idx) {
                        ins_word |= mask;
                    } else if (wi > ins_idx) {
                        if (ins_idx >= 0) (void)__hip_atomic_fetch_or(bits + ins_idx, ins_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ins_idx = wi, ins_word = mask;
                    } else {
                        (void)__hip_atomic_fetch_or(bits + wi, mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    i0 += take, left -= take;
                }
                strstart += m, lookahead -= m, nsyms += m;
                HOT_PF(3);
                continue;
            }
            lit_streak = 0;
        }
        int hash_head = 0;
        n_match++;  // (a loop-top taken by itself: a match, or a literal outside the runs of literals)
        if (lookahead >= kMinMatch) hash_head = insert(strstart);
        HOT_PF(1);
        if (hash_head != 0 && strstart - hash_head <= kMaxDist && search) {
            // Longest_match (le_longest_match).  The reference looks at four bytes of a candidate before it compares
            // (Deflate.cs:1072-1078); a candidate that fails them has at most best_len equal bytes, so comparing every candidate
            // gives the same answer -- and here the compare is one LDS round trip for all 258 bytes (a lane per 8 bytes), the same
            // round trip that brings the candidate's link: one per candidate instead of two
            int chain_length = chain_eff, best_len = best0, cur_match = hash_head;
            const int limit = strstart > kMaxDist ? strstart - kMaxDist : 0;
            const int nice = nice_cfg > lookahead ? lookahead : nice_cfg;
            const int off = lane < 33 ? lane * 8 : 0;  // 33 x 8 = 264 >= kMaxMatch; the other lanes repeat lane 0's bytes
            const uint64_t sx = le_load64(win + strstart + off);
            do {
                if (cur_match >= strstart) break;
                const int next = pl[cur_match & kWMask];
                const uint64_t x = le_load64(win + cur_match + off) ^ sx;
                const uint64_t differ = __ballot(x != 0 && lane < 33);
                int len = kMaxMatch;
                if (differ) {
                    const int fl = (int)__builtin_ctzll(differ);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, fl), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), fl);
                    len = fl * 8 + (int)(__builtin_ctzll((uint64_t)lo | ((uint64_t)hi << 32)) >> 3);
                    len = len < kMaxMatch ? len : kMaxMatch;
                }
                if (len > best_len) {
                    match_start = cur_match;
                    best_len = len;
                    if (len >= nice) break;
                }
                cur_match = next;
            } while (cur_match > limit && --chain_length != 0);
            match_length = best_len < lookahead ? best_len : lookahead;
        }
        HOT_PF(2);
        if (match_length >= kMinMatch) {
            if (lane == 0) syms[nsyms] = ((uint32_t)(strstart - match_start) << 16) | (uint32_t)(match_length - kMinMatch);
            nsyms++;
            lookahead -= match_length;
            if (match_length <= lazy && lookahead >= kMinMatch) {
                match_length--;
                do {
                    strstart++;
                    (void)insert(strstart);
                } while (--match_length != 0);
                strstart++;
            } else {
                strstart += match_length;
                match_length = 0;
            }
            lit_streak = 0;
        } else {
            if (lane == 0) syms[nsyms] = (uint32_t)wl[strstart];
            nsyms++;
            lookahead--;
            strstart++;
            lit_streak = (hash_head == 0 || strstart - 1 - hash_head > kMaxDist) ? lit_streak + 1 : 0;
        }
        HOT_PF(3);
    }
}
#endif

// Deflate.Stored.cs:24-84 (level 0; memLevel 7 -> pending 32 KiB -> max_block_size 32763), flush == Finish.
ZS_HD_NOINLINE inline void le_run_stored(LitEngine &e, int lane, int nlanes) {
    const int max_block_size = 32768 - 5;
    for (;;) {
        if (e.lookahead <= 1) {
            int dummy = 0;
            le_refill(e, lane, nlanes, dummy, 1);  // NoFlush returns only while lookahead == 0
            if (e.suspended) return;
            if (e.lookahead == 0) {
                if (!le_write_flushes(e)) break;
                le_end_write(e, lane, nlanes);
                continue;
            }
        }
        e.strstart += e.lookahead;
        e.lookahead = 0;
        const int block_start = (int)(e.block_start_abs - e.base);
        const int max_start = block_start + max_block_size;
        if (e.strstart == 0 || e.strstart >= max_start) {
            e.lookahead = e.strstart - max_start;
            e.strstart = max_start;
            le_flush_block(e, false, lane);
        }
        if (e.strstart - (int)(e.block_start_abs - e.base) >= kMaxDist) le_flush_block(e, false, lane);
    }
    le_flush_block(e, true, lane);
}

// Deflate.Rle.cs:18-104 (CompressionStrategy.Rle), flush == Finish.
ZS_HD_NOINLINE inline void le_run_rle(LitEngine &e, int lane, int nlanes) {
    for (;;) {
        if (e.lookahead <= kMaxMatch) {
            int dummy = 0;
            le_refill(e, lane, nlanes, dummy, kMaxMatch + 1);
            if (e.suspended) return;
        }
        if (e.lookahead == 0) {
            if (!le_write_flushes(e)) break;
            le_end_write(e, lane, nlanes);
            continue;
        }
        e.match_length = 0;
        if (e.lookahead >= kMinMatch && e.strstart > 0) {
            const uint8_t *w = e.window + e.strstart;
            const uint8_t prev = w[-1];
            if (prev == w[0] && prev == w[1] && prev == w[2]) {
                const int len = le_match_len_wave(w, w - 1);  // the run of `prev` = what w shares with itself one byte back
                e.match_length = len < e.lookahead ? len : e.lookahead;
            }
        }
        bool bflush;
        if (e.match_length >= kMinMatch) {
            bflush = le_tally(e, 1, e.match_length - kMinMatch, lane);
            e.lookahead -= e.match_length;
            e.strstart += e.match_length;
            e.match_length = 0;
        } else {
            bflush = le_tally(e, 0, le_wbyte(e, e.strstart), lane);
            e.lookahead--;
            e.strstart++;
        }
        if (bflush) le_flush_block(e, false, lane);
    }
    le_flush_block(e, true, lane);
}

// the strategy / level dispatch of Deflate.Compress (Deflate.cs:535-558)
ZS_HD void le_run(LitEngine &e, int level, int lane, int nlanes) {
    if (e.strategy == kRle) le_run_rle(e, lane, nlanes);
    else if (level == 0) le_run_stored(e, lane, nlanes);
    else if (e.lv.func == 1) le_run_fast(e, lane, nlanes);
    else le_run_slow(e, lane, nlanes);
}

// Rebuild the reference state at absolute loop-top `p` of a stream whose read
// events so far have left the window at absolute position `base` and the data
// read up to `avail_end` (both 0 when nothing has been read yet: the engine
// then performs the first read itself), from the bulk arrays:
//   link[q]  distance from q to the previous position in q's hash bucket
//            (0 = none within 32767), valid for q <= n - 6, with link[s] = 0 at
//            every resolved equal-bucket refill position s;
//   kind / pend  the lazy-parse node at p (zs_core.h) and its pending match;
//   preins  the position that the last refill pre-inserted (s_k + 1) or -1.
// `lane`/`nlanes` split the copy loops across a wave on the device.
ZS_HD_NOINLINE inline void le_restore(LitEngine &e, int64_t p, int64_t base, int64_t avail_end, int kind, uint32_t pend,
                                      const uint16_t *link, int64_t preins, int lane, int nlanes, bool preslid = false) {
    e.base = base;
    e.avail_end = avail_end;
    if (e.avail_end > e.n) e.avail_end = e.n;
    bool started = (p > 0 || base > 0 || kind != kR);
    if (!started) e.avail_end = 0;  // nothing read yet: the engine performs read 0 itself
    // the Write whose data the next Fill_window continues with
    e.cur_wr = 0;
    if (e.wr_end) {  // the first Write whose end is not below avail_end (a stream of scanlines has thousands of Writes)
        int lo = 0, hi = e.n_wr - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (e.wr_end[mid] < e.avail_end) lo = mid + 1;
            else hi = mid;
        }
        e.cur_wr = lo;
    }
    // preslid (le_tail_preslide): the image was filled at base - WSIZE and has slid since -- the lower half is what the upper
    // half was, the upper half is what it was (bytes behind the data included: zeros in a window that was never full)
    const int64_t ib = preslid ? e.base - kWSize : e.base;
    const int64_t valid = e.avail_end - ib;
    for (int w = lane; w < kWindowSize + 512; w += nlanes) {
        const int wi = preslid && w < kWSize ? w + kWSize : w;
        uint8_t v = 0;
        if (wi < valid) v = e.data[ib + wi];
        else if (ib >= kWSize && wi < kWindowSize) v = e.data[ib + wi - kWSize];  // stale upper half
        e.window[w] = v;
    }
    for (int i = lane; i < kHashSize; i += nlanes) e.head[i] = 0;
    for (int i = lane; i < kWSize; i += nlanes) e.prev[i] = 0;
    e.strstart = (int)(p - e.base);
    e.lookahead = (int)(e.avail_end - p);
    e.match_available = (kind == kR) ? 0 : 1;
    e.match_length = (kind == kXK || kind == kXK4) ? match_len(pend) : kMinMatch - 1;
    e.match_start = (kind == kXK || kind == kXK4) ? (int)((p - 1 - match_dist(pend)) - e.base) : 0;
    e.prev_length = kMinMatch - 1;
    e.prev_match = 0;
}

// Second phase of the restore (after a barrier on the device): prev[] for the
// positions q in [q0, q1) whose bucket depends on real input only (q <= n-6),
// straight from the bulk links.  Parallel over q.
ZS_HD void le_restore_prev(LitEngine &e, int64_t q, const uint16_t *link) {
    int w = (int)(q - e.base);
    int64_t c = link[q] ? q - (int64_t)link[q] : -1;
    e.prev[w & kWMask] = (uint16_t)(c >= e.base ? c - e.base : 0);
}
// The same for DeflateFast, whose chains hold only the inserted positions (`ins(q)`, zs_fast_vec.h): prev[q] is the
// nearest inserted position before q on the all-position chain.
template <class Ins>
ZS_HD void le_restore_prev_ins(LitEngine &e, int64_t q, const uint16_t *link, const Ins &ins) {
    int64_t c = q;
    for (;;) {
        const int l = link[c];
        if (!l) {
            c = -1;
            break;
        }
        c -= l;
        if (c < e.base || ins(c)) break;
    }
    e.prev[(int)(q - e.base) & kWMask] = (uint16_t)(c >= e.base ? c - e.base : 0);
}
// bucket of an already-restored window position
ZS_HD uint32_t le_bucket(const LitEngine &e, int64_t q) { return le_hash(e, le_load32(e.window + (q - e.base) + 2)); }

// Third phase, sequential: the few positions below p that were inserted with
// hashes reaching past the end of input (q in [n-5, p), at most 3 of them are
// <= max_insert = n-3), then the pending pre-insert of the last refill.
// `ins` (DeflateFast: zs_fast_vec.h): which positions are in the chains -- what the pending pre-insert finds as its bucket's
// head is the nearest inserted position on the all-position chain, as in le_restore_prev_ins; nullptr-like AllIns: all are.
struct AllIns {
    ZS_HD bool operator()(int64_t) const { return true; }
};
template <class Ins>
ZS_HD_NOINLINE inline void le_restore_finish(LitEngine &e, int64_t p, const uint16_t *link, int64_t preins, const Ins &ins) {
    int64_t q = e.n - 5;
    if (q < 0) q = 0;
    if (q < e.base) q = e.base;
    // (DeflateFast: only what the parse inserted -- a body whose last match ends four bytes in front of the stream's end hands over
    // at n - 4, and n - 5 is no loop-top)
    for (; q < p && q <= e.n - kMinMatch; q++)
        if (ins(q)) le_insert(e, (int)(q - e.base));
    if (preins >= p && preins >= 1) {
        // the refill at loop-top preins-1 inserted preins before preins-1 (Deflate.cs:1010-1013)
        int w = (int)(preins - e.base);
        uint32_t h1 = le_hash(e, le_load32(e.window + w + 2));
        uint32_t h0 = le_hash(e, le_load32(e.window + w - 1 + 2));
        if (h0 == h1) {
            // head[h] == preins-1 already; the reference left prev[preins-1] = preins
            e.prev[(w - 1) & kWMask] = (uint16_t)w;
        } else {
            int64_t c = preins;
            for (;;) {
                const int l = link[c];
                if (!l) {
                    c = -1;
                    break;
                }
                c -= l;
                if (c < e.base || ins(c)) break;
            }
            e.prev[w & kWMask] = (uint16_t)(c >= e.base ? c - e.base : 0);
            e.head[h1] = (uint16_t)w;
        }
    }
}
ZS_HD_NOINLINE inline void le_restore_finish(LitEngine &e, int64_t p, const uint16_t *link, int64_t preins) {
    le_restore_finish(e, p, link, preins, AllIns());
}

}  // namespace zs
