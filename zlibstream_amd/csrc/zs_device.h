// zs_device.h -- device-visible descriptors shared by the kernels and the host
// pipeline (zs_engine.hip).
#pragma once
#include <cstddef>
#include "zs_core.h"
#include "zs_lit_engine.h"
#include "zs_fast_sweep.h"
#include "zs_rle.h"

namespace zs {

// Match tile: positions per workgroup of the match kernel; its LDS holds the
// input bytes [t0 - kMatchBack, t0 + kMatchTile + kMatchFwd) and the links of
// [t0 - kMatchBack, t0 + kMatchTile).
constexpr int kMatchTile = 16384;
constexpr int kMatchBack = 32512;  // >= kMaxDist, multiple of 16
constexpr int kMatchFwd = 272;     // >= kMaxMatch + 8, multiple of 16
constexpr int kMatchLdsBytes = kMatchBack + kMatchTile + kMatchFwd;
constexpr int kMatchLdsLinks = kMatchBack + kMatchTile;
constexpr int kMatchLds = kMatchLdsBytes + 2 * kMatchLdsLinks;  // 146,960 B of the CU's 160 KiB
constexpr int kAdlerPiece = 65536;

// per-stream tail-engine scratch layout (bytes)
constexpr int64_t kScratchWindow = 0;
constexpr int64_t kScratchHead = 66560;                         // window: 65536 + 512, rounded
constexpr int64_t kScratchPrev = kScratchHead + 2 * kHashSize;  // u16 head
constexpr int64_t kScratchHead32 = kScratchPrev + 2 * kWSize;   // u16 prev
constexpr int64_t kScratchBytes = kScratchHead32 + 4 * kHashSize;

struct StreamDesc {
    const uint8_t *in;
    uint8_t *out;
    int64_t out_cap;
    int64_t pos_off;   // index of position 0 in link / mK / mK4
    int64_t sym_off;   // index of symbol 0 in syms
    int32_t n;
    int32_t body_end;  // last loop-top handled by the bulk path (n - 262), -1 if none
    int32_t kl;        // refills (reads k >= 1) of the single-Write schedule
    int32_t nchunks;   // parse chunks covering [0, body_end]
    int32_t chunk_off; // index of chunk 0 in maps / entry / symbase / stale
    int32_t blk_off;   // index of block 0 in block arrays
    int32_t max_blocks;
    int32_t adler_off; // index of piece 0 in adler pieces
    int32_t n_adler;
    int32_t fast_runs; // > 0: DeflateFast by speculative chunk runs (zs_fast_run_kernel); the tail kernel skips the stream
    int32_t run_off;   // index of run 0 in the run arrays
    int32_t run_slots; // run slots its runs take in the arrays (= fast_runs; more for the one run of a whole stream, whose symbols and bits take the slots in a row)
    int32_t run_chunk; // bytes of the stream per run (64 .. 256 KiB: as many runs as the chip has CUs, if the batch is long enough for that)
    int32_t seg_off;   // index of parse segment 0 in segmap / seg_entry / seg_symbase / seg_stale
    int32_t nsegs;
    int32_t sup_off;   // index of the stream's first row in supmap (one row per kSupSegs parse segments, zs_supmap_kernel)
    // parse segments (one per cluster of read events, zs_core.h build_geometry), already offset to this stream: first chunk
    // of each, data end once its events have fired, window base after them, slide threshold at its entry, and where its
    // cluster's boundaries are in `cl` (nsegs + 1 offsets); per chunk: its first position (nchunks + 1 entries) and, for the
    // first chunk of a segment whose cluster fires inside it, the segment's number + 1 (else 0)
    const int32_t *seg_c0, *seg_after, *seg_base, *seg_S, *seg_cl;
    const uint32_t *cl;
    const int32_t *cstart;
    const int32_t *head;
    int32_t grid_chunks;     // the chunks are on the single Write's grid (chunk c >= 1 begins at 2048 c - 261): chunk_of in closed form
    int32_t n_wr;            // > 1: several Writes -> the whole stream runs on the literal engine
    const int64_t *wr_end;   // device array of n_wr cumulative Write ends (or nullptr)
    // FlushMode Partial / Sync / Full (nullptr: every Write is NoFlush): the mode of each Write, the number of blocks
    // flushed before each Write began (written by the literal engine, read by the offsets kernel), the caller's output
    // chunk (ZlibOutputStream's 512) and whether the stream is raw deflate (no header bytes were delivered)
    const uint8_t *wr_flush;
    int32_t *wr_blk;
    int32_t out_chunk, raw;
    // level 0 (DeflateStored): the block list is a function of the sizes alone and comes from the host
    // (zs_core.h plan_stored_blocks); the literal engine is not run, the bytes are moved by the bit-emission kernel
    const BlockRec *plan_blk;
    int32_t plan_nblk;
    // incremental streams (zs_stream_api.inc): the run is not the end of the stream (final_run == 0) and / or continues one
    // whose engine state lives in `persist` (cont != 0: the input buffer holds the stream from position abs_off on -- the
    // last 64 KiB already read, then the new bytes -- and nothing is parsed in bulk).  The Adler-32 of the whole stream is then the caller's (adler_stream).
    int32_t final_run;
    struct LitPersist *persist;
    int64_t abs_off;
    int32_t cont;
    // A run that takes a stream over from the literal engine in the middle (zs_engine.hip: the bulk pipeline behind the
    // warm-up after a flush): `resume` -- the bulk parse begins at chunk 0's entry slot start_slot with start_syms symbols of
    // the block in progress already there (copied from `persist`), the block began at start_block, the window stood at
    // base0; all of them buffer positions, persist_off = the stream position of buffer position 0 (abs_off is 0 for such a
    // run).  cont_bits: the run's output goes on in the middle of the stream's bits (a continued or a resumed run).
    // stop_abs (a continued run): the literal engine stops at the first clean loop-top at or behind it (LitEngine::stopped).
    int32_t resume, start_slot, cont_bits;
    int32_t mid_write;  // the run begins in the middle of a Write (behind a stop of the literal engine): no new Deflate call begins with it
    uint32_t start_syms;
    int64_t base0, start_block, persist_off, stop_abs;
    int64_t start_pos;  // resume: the run's first loop-top (buffer position); the engine before it ran up to there
    uint32_t adler_stream;
    uint32_t carry_byte;  // cont: the bits of the stream's last, incomplete byte from the run before
    // DeflateFast as sweeps of a workgroup (zs_fast_sweep.h, zs_fast_sweep_kernel): last loop-top it handles (n - 262), -1: not
    // this stream; the stream's inserted-position bitmap (bit q of the array = position q)
    int32_t fv_end;
    uint32_t *ins_bits;
    // the cuts of equal-bucket events collected by the resolve kernel's dry passes (batched cut rounds): where the stream's
    // entries begin in the two cut lists, and how many there is room for (one per read boundary)
    int32_t cut_off, cut_cap;
    // CompressionStrategy.Rle over the chip (zs_rle.h, zs_rle.hip): the loop-tops below rle_end are the body's, the tail engine
    // goes on from the first one at or behind it (-1: not this stream); the stream's first tile in the batch's tile arrays
    int32_t rle_end, rle_tile_off;
    // DeflateFast as rounds over the chunks of the stream (zs_fast_sweep.h "Rounds"): the stream's chunks in the batch's list (fr_n == 0: the
    // stream is one workgroup's, from its first position to fv_end)
    int32_t fr_first, fr_n;
};

// where each kernel's work items start in the work array (zs_worklist_kernel): 9 lists, then the total
struct WorkOffsets {
    uint32_t off[10];
};
constexpr int kK5Feeders = 2;     // waves of the symbol kernel's workgroup that stage the walking wave's records in LDS
constexpr int kK5Threads = 64 * (1 + kK5Feeders);
constexpr int kSupSegs = 16;      // parse segments composed into one row of supmap ahead of the resolve kernel
// the sweep kernel's last tile stages bitmap words for its tile and a match's reach past the stream's end: the bitmap array carries
// that much room behind the last stream
constexpr size_t kFvBitSlack = (16384 + 512) / 8 + 64;

// What a suspended literal engine keeps between runs (device memory, one per zs_deflate stream): the reference's own state
// -- window, prev, head, the Deflate fields (Deflate.cs:128-226) -- plus the symbols of the block in progress and the
// chunk accounting of the output protocol (zs_core.h FlushAcct).
struct LitPersist {
    int64_t base, avail_end, block_start_abs;
    int32_t strstart, lookahead, match_length, match_start, match_available, prev_length, prev_match;
    int32_t pending_syms;
    FlushAcct fa;
    int32_t fa_valid, stopped;  // stopped: the engine was left at a clean loop-top for the bulk pipeline to go on from (LitEngine::stopped)
    int32_t good_prev, pad2_[3];  // ... where the search before ran on the reduced chain budget (prev_length >= good_match); the arrays below stay 16-byte aligned
    uint8_t window[kWindowSize + 512];
    uint16_t prev[kWSize];
    uint16_t head[kHashSize];
    uint32_t syms[kLitBufsize];
};
static_assert(offsetof(LitPersist, window) % 16 == 0 && offsetof(LitPersist, prev) % 16 == 0 && offsetof(LitPersist, head) % 16 == 0, "16-byte copies");

struct StreamState {
    // written by the resolve kernel
    int32_t tail_p, tail_kind;
    uint32_t tail_pend;
    int32_t k_done, preins;
    uint32_t body_syms;
    // written by the tail kernel
    uint32_t nsyms;
    int32_t nblocks;
    // written by the offsets kernel
    int64_t out_len;   // bytes of output; a run that is not the stream's end: complete bytes only
    uint32_t adler;
    int32_t status;
    int64_t end_bits;  // bit position in the stream (zlib header included) behind the run's last block or marker
    // the resolve kernel's position when it is run part by part behind the match kernel (one long stream: zs_engine.hip):
    // next segment, entry slot, symbols so far, last segment whose events fired and its entry slot; and the equal-bucket cuts whose repair reaches into
    // positions whose matches were not computed yet (cut position, last position repaired)
    int32_t r_seg, r_slot, r_kfired, r_kslot;
    uint32_t r_total;
    int32_t r_ncut;
    int32_t r_cut_e[8], r_cut_done[8];
    // data whose refills are equal-bucket ones by the thousand (zeros pages, runs): the resolve kernel gives the stream up
    // (deferred = 1: the kernels behind it skip the stream, the host runs the batch again in rounds) or, in a round, stops
    // at a cut whose repair is worth the whole chip (deferred = 2: zs_repair_kernel, zs_stalemaps_kernel, and on it goes;
    // r_scan: the walk resumes behind an applied cut)
    int32_t deferred, r_scan;
    int32_t r_cutidx;  // equal-bucket events of the current segment's cluster that have been cut already
    // batched cut rounds (zs_engine.hip): a stream with more cuts than one CU should repair one after the other is given up
    // (deferred = 1) where it stands; dry passes of the resolve kernel then collect the cuts of the rest of the stream from
    // the records as they are (nc[pass & 1] of them), the chip repairs them all at once, and the passes go on until one finds
    // the cuts of the pass before (cuts_same).  cut_diff_idx / cut_diff_pos: the first cut that differs between the last two
    // passes and the position from which the records are restored and repaired again.
    int32_t nc[2], cuts_same, cut_diff_idx, cut_diff_pos;
    // the true path met a read whose pre-insert hashes bytes behind the data (zs_core.h kMapPoisonBit): the host runs the
    // stream on the literal engine
    int32_t poison;
};
constexpr int kDeferBudget = 8;   // cuts with positions to walk again that a stream may repair on its one CU before it is given up
constexpr int kCutBudget = 12;    // ... and cuts of any kind (a cut without such positions is a scan of 32 Ki records on one CU)

// DeflateFast (levels 1-3) as speculative chunk runs: run j re-parses kFastWarm bytes before its chunk with an
// "everything inserted" history, then its chunk; it is exact iff its state at the first loop-top of the chunk
// (position + set of inserted strings in the 32 KiB before it) equals the previous run's final state.
constexpr int kFastChunk = 262144;
constexpr int kFastWarm = 65536;
constexpr int kFastMinInput = 4 << 20;  // below this the vector form alone is faster than a failed attempt plus the vector form, unless the data is all period
constexpr int kFastMinPeriodic = 64 << 10;  // ... in which case the runs are tried from here on (zeros, image rows, a short period: the sweeps settle one range a round on them)
constexpr int64_t kFastRunSyms = kFastWarm + kFastChunk + 1024;             // symbol slots per run
constexpr int64_t kFastRunBitWords = (kFastWarm + kFastChunk + 2048) / 32;  // inserted-position bitmap words per run
constexpr int64_t kFastRunScratch = 2 * kHashSize + 4 * kHashSize;          // u16 head + u32 head32
struct FastRunOut {
    int64_t mark_pos, mark_nsyms;  // first loop-top inside the run's own chunk, symbols emitted before it
    int64_t end_pos, nsyms;        // loop-top at which the run stopped (n for the last run), symbols emitted
    int64_t final_base;            // window base at the end (last run: decides whether the last block may be stored)
    int64_t sym_dst;               // where the run's own symbols go in the stream's symbol array (set by the stitch kernel)
    int64_t ev[16];                // loop-tops of the refills the run performed
    int64_t n_match;               // symbols that took the engine a loop-top of their own (warm-up included): what one run over the whole stream would cost
    int32_t ok, n_ev;
};

struct BlockInfo {
    int32_t type;  // 0 stored, 1 static, 2 dynamic
    int32_t bits;  // compressed blocks: 3 + opt_len / static_len
    int64_t bit_start;
};

}  // namespace zs
