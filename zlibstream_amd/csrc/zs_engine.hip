// zs_engine.hip -- host side of the MI355X deflate engine: workspace, kernel
// pipeline, and the C ABI of include/zsgpu.h.  No CPU fallback: without a
// usable HIP device every compressing entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zsgpu.h"
#include "zs_kernels.hip"
#include "zs_inflate_par.hip"
#include "zs_inflate_tok.hip"

using namespace zs;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

enum Stage {
    kStClear, kStAdler, kStLinks, kStMatch, kStChunkMap, kStSegMap, kStResolve, kStExpand, kStEmitSyms, kStTail, kStTrees,
    kStOffsets, kStEmitBits, kStCount
};
const char *const kStageNames[kStCount] = {"clear", "adler", "links", "match", "chunkmap", "segmap", "resolve", "expand",
                                           "emit_syms", "tail", "trees", "offsets", "emit_bits"};

}  // namespace

struct zs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;  // second stream: tree building of the finished blocks runs beside the tail engine
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pre0 = nullptr, ev_pre = nullptr;
    hipEvent_t ev_part[16] = {};          // one long stream run part by part: part k's maps are ready
    std::vector<hipEvent_t> ev_pool;      // timing pairs of the part-wise launches (profiling)
    std::string err;
    bool profiling = false;
    int fast_fallbacks = 0;  // speculative DeflateFast batches that had to be redone sequentially
    int round_runs = 0;      // batches with streams in the batched cut rounds (their cuts were not one CU's job)
    int cut_rounds = 0;      // rounds of those
    int lit_fallbacks = 0;   // batches run again with a stream on the literal engine (zs_core.h kMapPoisonBit)
    int64_t lit_engine_bytes = 0;  // input bytes parsed by the one-wave literal engine beyond the streams' last 261 (zs_ctx_counter)
    // inflate: compressed bytes each stream of the last call used, trailer included (0: unknown / not ended); and, for a
    // probing call (zs_inflate asking whether the stream's end has arrived), where the block chain ended
    std::vector<int64_t> inf_used;
    int64_t *inf_probe = nullptr;
    int fast_rounds = 0;  // rounds the last call's DeflateFast took over its chunks (0: one workgroup per stream)
    bool no_rounds_once = false;  // the next plan takes one workgroup per stream (set when the rounds gave up)
    int last_op = 0;  // 0: deflate stages, 1: block-parallel inflate stages, 2: deflate at levels 1-3 (for zs_ctx_stage_name)
    hipEvent_t ev[kStCount + 1] = {};
    double stage_ms[kStCount] = {};
    uint32_t *crc_tab = nullptr;
    DevBuf sd, st, work, wpre, geo, link, mm, maps, chunk_far, segmap, supmap, seg_entry, seg_symbase, seg_stale, entry, symbase, stale, syms, blk_end, blk_top, blocks, trees, info, pieces, scratch,
        stage_in, stage_out, wr, inf_desc, inf_state, par_ps, par_st, par_work, par_cbits, par_ccnt, par_surv, par_scnt, par_cands, par_tabs, par_toktabs, par_toks, par_ctoks, par_tokstat, par_tails, par_retry, par_fxtab, par_blocks, par_cells,
        par_windows, par_fail, run_syms, run_bits, run_scratch, run_outs, run_fail, adl_tr, adl_res, plan_blk, ins_bits, mm_bak, cut_pos, cut_bkt, win_groups, win_sg, win_maps, win_entries, persist_bak, resume_flag, rle_tiles, own_in, fr_chunks, fr_meta, fr_planes, fr_prov, fr_base, fr_counters;
    bool resume_poisoned = false;  // a resumed run met a read the bulk form does not handle: the caller goes on with the literal engine
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    // the Stream API's pinned input buffer (zs_stream_api.inc InBuf): one deflate stream of the context at a time holds it; `own_in` is
    // that stream's input on the device, sent ahead of its first run
    void *pin_io = nullptr;
    size_t pin_io_cap = 0;
    bool pin_io_busy = false;
    void *pin_out = nullptr;  // ... and an inflate stream's decoded output (OutBuf)
    size_t pin_out_cap = 0;
    bool pin_out_busy = false;
};

namespace {

bool fail(zs_ctx *c, const char *what, hipError_t e) {
    c->err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define ZS_HIP(c, call)                                  \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return fail(c, #call, e_); \
    } while (0)

bool ensure(zs_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap) return true;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    // an eighth of headroom so that a slightly larger batch reuses the buffer; without it when the device is nearly full
    size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        want = bytes + 4096;
        if (hipMalloc(&b.p, want) != hipSuccess) {
            (void)hipGetLastError();
            b.p = nullptr;
            char msg[160];
            snprintf(msg, sizeof msg, "workspace: %zu bytes do not fit the device's free memory (the pipeline needs ~17 bytes per input byte: split the batch)", bytes);
            c->err = msg;
            return false;
        }
    }
    b.cap = want;
    return true;
}
bool ensure_pinned(zs_ctx *c, size_t bytes) {
    if (bytes <= c->pinned_cap) return true;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr;
    c->pinned_cap = 0;
    ZS_HIP(c, hipHostMalloc(&c->pinned, bytes + 4096, hipHostMallocDefault));
    c->pinned_cap = bytes + 4096;
    return true;
}

// A kernel's work items: (stream, index) for index < count(stream), stream after stream.  The host keeps the running totals
// (one per stream); the pairs themselves are written on the device (zs_worklist_kernel) -- for a 64 MiB stream they are
// 46 K pairs, whose making and upload were 0.1 ms of host time per call.
struct WorkList {
    std::vector<int32_t> pre;  // pre[i] = items of the streams before i; one more entry = the total
    void add(int stream, int64_t count) {
        while ((int)pre.size() <= stream) pre.push_back(pre.empty() ? 0 : pre.back());
        pre.push_back(pre.back() + (int32_t)count);
    }
    void finish(int n) {
        if (pre.empty()) pre.push_back(0);
        while ((int)pre.size() <= n) pre.push_back(pre.back());
    }
    size_t size() const { return pre.empty() ? 0 : (size_t)pre.back(); }
    bool empty() const { return size() == 0; }
};
constexpr int kWorkLists = 9;
struct Plan {
    std::vector<StreamDesc> sd;
    WorkList w_clear, w_adler, w_links, w_match, w_chunks, w_segs, w_sups, w_blocks, w_runs;
    int64_t n_pos = 0, n_syms = 0;
    int64_t n_chunks = 0, n_segs = 0, n_sups = 0, n_blocks = 0, n_pieces = 0, n_runs = 0;
    bool any_fv = false, any_rle = false;
    bool any_dual = false;  // levels 1-3, a few streams below 4 MiB: planned for the sweeps and for the speculative runs, the links decide (zs_fast_probe_kernel)
    int64_t n_rle_tiles = 0;
    int64_t lit_bytes = 0;           // input bytes of this plan that only the literal engine parses
    std::vector<FsChunk> fr_chunks;  // levels 1-3 as rounds over the chunks of the streams (zs_fast_sweep.h "Rounds"); empty: one workgroup per stream
    size_t fr_prov = 0;              // symbols of room in the chunks' provisional buffer
    int fr_max_n = 0;                // the most chunks a stream has
    int64_t n_cuts = 0;     // entries of a cut list (batched cut rounds): one per read boundary
    // parse-segment tables (zs_core.h build_geometry), all streams: per segment (seg_off order); seg_cl and cstart hold one
    // entry more per stream (stream i's lists begin at seg_off + i / chunk_off + i); seg_cl's values index `cl`
    std::vector<int32_t> seg_c0, seg_after, seg_base, seg_S, seg_cl, cstart, head;
    std::vector<uint32_t> cl;
    std::vector<BlockRec> plan_blk;                    // level 0: the stored blocks of every stream, stream after stream
    std::vector<int32_t> plan_wr_blk;                  // level 0 under a flush mode: blocks flushed before each Write began
};

template <class T>
T *dev(DevBuf &b) {
    return (T *)b.p;
}

// The Writes of one stream (ZlibOutputStream.WriteCore): cumulative ends, the FlushMode of each, the caller's output
// chunk size and whether the stream is raw deflate.
struct WriteSpec {
    std::vector<int64_t> ends;
    std::vector<uint8_t> flush;
    int chunk = 512;
    bool raw = false;
    bool flushing() const {
        for (uint8_t f : flush)
            if (f) return true;
        return false;
    }
};

// One run of an incremental stream (zs_stream_api.inc; one stream per call).  final_run == false: the stream goes on after
// this run's Writes, the engine is left in `persist` (device memory) where Deflate.Compress would return to its caller, and
// only what is final -- whole blocks and flush markers, complete bytes -- is output.  cont: the run continues from `persist`;
// the input buffer then holds the stream from position abs_off on (64 KiB already read, then the new bytes) and `writes`
// holds absolute ends.
struct RunOpts {
    bool final_run = true, cont = false;
    int64_t hist = 0;  // leading bytes of the buffer that earlier runs have parsed (history: up to 64 KiB)
    LitPersist *persist = nullptr;
    int64_t abs_off = 0;
    uint32_t adler_stream = 1, carry_byte = 0;
    int64_t end_bits = 0;  // out: stream bit position behind the run's last block or marker
    // cont: the literal engine stops at the first clean loop-top at or behind this stream position (< 0: runs on) ...
    int64_t stop_abs = -1;
    // ... and the run behind such a stop goes on from there on the bulk pipeline (`resume`): loop-top p0 with the input read up to
    // E0, the window at base0, the block in progress begun at start_block -- buffer positions, the buffer beginning at
    // stream position abs_off -- with start_syms symbols, in the lazy parse's node `slot` of a chunk that begins at p0
    // at_read: the stream was flushed at p0 (E0 == p0, nothing pending): the run's first pass through the loop reads, like a
    // stream's first one, and the run begins with a new Deflate call
    bool resume = false, mid_write = false, at_read = false;
    int64_t p0 = 0, E0 = 0, base0 = 0, start_block = 0;
    int slot = 0;
    uint32_t start_syms = 0;
};

// `writes` (optional, one stream only): the Writes of a multi-Write stream, or of one whose Writes carry a flush mode
bool run_pipeline(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                  int64_t *out_len, int *status, int level, int strategy, int hash_variant, hipStream_t stream,
                  const WriteSpec *writes = nullptr, int force_seq = 0, RunOpts *ro = nullptr, bool rounds = false,
                  const std::vector<uint8_t> *force_lit = nullptr) {
    if (level == -1) level = 6;
    LevelCfg lv = level_cfg(level);
    static const bool host_times = getenv("ZS_HOST_TIMES") != nullptr;  // (where the host's share of a call goes: stderr, microseconds)
    const auto ht0 = std::chrono::steady_clock::now();
    auto ht_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ht0).count(); };
    double ht_plan = 0, ht_launched = 0;
    for (int i = 0; i < n; i++) {  // what the caller sees if a HIP call fails before the results are known
        out_len[i] = 0;
        if (status) status[i] = ZS_STREAM_ERROR;
    }
    // CompressionStrategy.Rle does not look at Write ends: Deflate.Rle.cs leaves a Deflate call as soon as fewer than MAX_MATCH bytes
    // are ahead under NoFlush (:24-38), so every match is measured with a full lookahead whatever the Writes are, and nothing is
    // inserted anywhere.  What a Write end can move is the loop-top at which the window slides -- when it lies within 262 bytes
    // below a window end -- and with it a block's permission to be stored.  Any other NoFlush schedule is the single Write's
    // stream (checked against the oracle on random data and schedules, tests/test_oracle.py) and runs as one.
    WriteSpec rle_one;
    if (writes && writes->ends.size() > 1 && strategy == kRle && level >= 1 && !writes->flushing() && !ro && n == 1 && !getenv("ZS_NO_RLE_MULTI")) {
        bool safe = true;
        for (size_t k = 0; k + 1 < writes->ends.size() && safe; k++) {
            const int64_t E = writes->ends[k];
            safe = !(E >= kWindowSize - kMinLookahead && (E % kWSize) >= kWSize - kMinLookahead);
        }
        if (safe) {
            rle_one = *writes;
            rle_one.ends.assign(1, in_len[0]);
            rle_one.flush.assign(1, 0);
            writes = &rle_one;
        }
    }
    // Levels 1-3, a few streams of 256 KiB .. 4 MiB: data that is all period (zeros, image rows, a short period) takes the sweeps'
    // rounds one range a round -- 1 MiB of zeros 57 ms -- and the speculative runs' engine 7; anything else is better off with
    // the sweeps, and which it is the links tell (zs_fast_probe_kernel).  Such a batch is planned for both; the kernels behind
    // the link kernel see the one the probe chose.
    static const int64_t fast_min_input = getenv("ZS_FAST_MIN_INPUT") ? atoll(getenv("ZS_FAST_MIN_INPUT")) : kFastMinPeriodic;
    bool allow_dual = lv.func == 1 && strategy != kRle && strategy != kHuffmanOnly && !writes && !ro && !rounds && force_seq == 0 && !force_lit && n <= 16 && !getenv("ZS_NO_FAST_VEC") &&
                      !getenv("ZS_FAST_NO_ROUNDS");
    for (int i = 0; i < n && allow_dual; i++) allow_dual = in_len[i] >= fast_min_input && in_len[i] < kFastMinInput;
    bool dual_broken = false;
plan_again:
    Plan pl;
    pl.sd.resize((size_t)n);
    // positions per workgroup of the link kernel (each replays 32 Ki positions of warm-up first): long spans for a
    // big batch, shorter ones when that is what it takes to give every CU a workgroup
    int64_t total_len = 0;
    for (int i = 0; i < n; i++) total_len += in_len[i];
    const int64_t link_span = total_len >= (48ll << 20) ? 262144 : total_len >= (24ll << 20) ? 131072 : 65536;
    // a speculative run's share of its stream (levels 1-3, image-like data): a run is one wave on one CU, so a batch shorter than
    // 256 runs of 256 KiB is cut finer, down to 64 KiB a run (64 KiB of warm-up in front of each: at most twice the work)
    const int run_chunk = getenv("ZS_FAST_CHUNK") ? std::max(65536, std::min((int)kFastChunk, atoi(getenv("ZS_FAST_CHUNK")) & ~65535))
                                                  : (int)std::max<int64_t>(65536, std::min<int64_t>(kFastChunk, ((total_len / 256 + 65535) >> 16) << 16));
    for (int i = 0; i < n; i++) {
        StreamDesc &s = pl.sd[(size_t)i];
        int64_t len = in_len[i];
        s.in = (const uint8_t *)in[i];
        s.out = (uint8_t *)out[i];
        s.out_cap = out_cap[i];
        s.n = (int32_t)len;
        const bool multi = writes && writes->ends.size() > 1;
        // an incremental run always takes the block-by-block output accounting (its state is carried from run to run)
        const bool flushing = writes && (writes->flushing() || ro);
        const bool resume = ro && ro->resume;
        const bool cont = ro && ro->cont && !resume, final_run = !ro || ro->final_run;
        // the bulk pipeline takes the NoFlush schedules build_geometry accepts (zs_core.h: any Write sizes but streams written
        // a few bytes at a time); a stream the resolve kernel flagged (force_lit: a read whose pre-insert hashes bytes behind the
        // data) and other streams of several Writes run on the literal engine.  Levels 1-3 keep the single Write's events.
        std::vector<ReadEvent> rev;
        Geometry geo;
        const int64_t one_write[1] = {len};
        const bool real_flush = writes && writes->flushing();
        const bool lit_forced = force_lit && (*force_lit)[(size_t)i];
        const std::vector<int64_t> no_ends_;
        GeoStart gs;
        if (resume) gs.resume = true, gs.at_read = ro->at_read, gs.p0 = ro->p0, gs.E0 = ro->E0, gs.base0 = ro->base0;
        // (a flush mode on the run's last Write alone is the tail engine's business: it closes the block there)
        bool inner_flush = false;
        if (writes)
            for (size_t k = 0; k + 1 < writes->flush.size(); k++) inner_flush = inner_flush || writes->flush[k] != 0;
        const bool slow_ok = lv.func == 2 && strategy != kRle && !cont && !lit_forced && !(multi && inner_flush) &&
                             build_geometry(len, multi ? writes->ends : no_ends_, geo, gs);
        // levels 1-3 take several NoFlush Writes too when every read brings a full lookahead and no Write ends where its loop-top
        // may or may not slide the window (zs_core.h build_read_events: Stream.CopyTo's 81 920-byte Writes do; a Write every few
        // bytes does not)
        const bool fast_multi = multi && !inner_flush && !cont && !resume && lv.func == 1 && strategy != kRle && !getenv("ZS_NO_FAST_MULTI") &&
                                build_read_events(len, writes->ends, rev, true);
        // levels 1-3 behind a flush (round 5): the run goes on from the suspended engine's chains too -- zs_import_chains_kernel's
        // links ARE DeflateFast's chains for the history (prev[] names inserted positions only), so with "inserted" for every
        // position the chains reach, the sweeps start at p0 as they start at 0; one Write, the engine standing at the flush
        // (several NoFlush Writes behind the flush as well, on the conditions of a stream's several Writes: every read brings a
        // full lookahead, no Write ends where its loop-top may or may not slide the window)
        const bool fast_resume = resume && ro->at_read && lv.func == 1 && strategy != kRle && !inner_flush && !lit_forced && !getenv("ZS_NO_FAST_RESUME") &&
                                 len - ro->p0 >= 2 * kMinLookahead &&
                                 (multi ? build_read_events(len, writes->ends, rev, true, ro->p0, ro->base0)
                                        : build_read_events(len, std::vector<int64_t>(one_write, one_write + 1), rev, true, ro->p0, ro->base0));
        const bool regular = fast_resume ? true : cont ? false : multi ? fast_multi : build_read_events(len, std::vector<int64_t>(one_write, one_write + 1), rev);
        s.body_end = slow_ok ? (int32_t)geo.body_end : -1;
        // levels 1-3, one Write: the speculative chunk runs for large streams (they verify on periodic data and are parallel
        // inside a stream), else -- and when they did not verify (force_seq) -- DeflateFast for the lanes of a wave
        // (round 5: also the first run of a stream that flushes -- one Write, the flush at its end: the tail engine closes the
        // block and is left suspended as it is behind the slow levels' bulk runs)
        const bool fast_one = lv.func == 1 && strategy != kRle && regular && len >= kMinLookahead && !cont && !resume &&
                              ((!flushing && final_run && !ro && (!multi || fast_multi)) || (ro && (!multi || fast_multi) && !inner_flush && !lit_forced && !getenv("ZS_NO_FAST_RESUME")));
        // (force_seq: 1 -- the runs did not verify, or the data does not look periodic: the sweeps; 2 -- they did not verify and the
        // stream is few symbols: one run of the engine for the whole stream, below)
        // (a run's buffers are 1.7 MB whatever the stream's length: streams below 4 MiB only in batches of a few -- allow_dual)
        // (HuffmanOnly: nothing is searched, every position a literal -- the sweeps take that at 7 GB/s, the runs' engine symbol by symbol)
        const bool fast_par = fast_one && !multi && force_seq != 1 && !ro && strategy != kHuffmanOnly &&
                              (len >= kFastMinInput || allow_dual || (force_seq == 2 && n <= 16 && len >= fast_min_input));
        if (allow_dual && !fast_par) dual_broken = true;
        s.fv_end = ((fast_one && (!fast_par || allow_dual) && !getenv("ZS_NO_FAST_VEC")) || fast_resume) ? (int32_t)(len - kMinLookahead) : -1;
        s.ins_bits = nullptr;
        if (s.fv_end >= 0) {
            pl.any_fv = true;
        }
        // CompressionStrategy.Rle, one Write: the parse is a function of where the runs of equal bytes begin (zs_rle.h)
        s.rle_end = -1, s.rle_tile_off = 0;
        s.fr_first = 0, s.fr_n = 0;
        if (strategy == kRle && level >= 1 && !multi && !flushing && final_run && !ro && regular && !getenv("ZS_NO_RLE_RUNS")) {
            s.rle_end = (int32_t)rle_body_end(len);
            if (s.rle_end >= 0) {
                pl.any_rle = true;
                s.rle_tile_off = (int32_t)pl.n_rle_tiles;
                pl.n_rle_tiles += (s.rle_end + kMaxMatch + 1 + 4095) / 4096;
            }
        }
        s.n_wr = (multi || flushing) ? (int32_t)writes->ends.size() : 1;  // 0: a run without input (Finish alone)
        s.wr_end = nullptr;
        s.wr_flush = nullptr, s.wr_blk = nullptr, s.out_chunk = writes ? writes->chunk : 512, s.raw = writes && writes->raw;
        s.kl = num_refills(len - (resume ? ro->base0 : 0));
        s.resume = resume && (slow_ok || fast_resume) ? 1 : 0, s.cont_bits = (cont || resume) ? 1 : 0, s.mid_write = (ro && (ro->mid_write || (resume && !ro->at_read))) ? 1 : 0;
        if (resume && !slow_ok && !fast_resume) {
            c->err = "the run cannot go on in the bulk pipeline from where the literal engine stopped";
            return false;
        }
        s.start_slot = resume ? ro->slot : 0, s.start_syms = resume ? ro->start_syms : 0, s.base0 = resume ? ro->base0 : 0;
        s.start_block = resume ? ro->start_block : 0, s.start_pos = resume ? ro->p0 : 0, s.persist_off = resume ? ro->abs_off : 0, s.stop_abs = (ro && cont) ? ro->stop_abs : -1;
        s.nchunks = s.body_end >= 0 ? geo.nchunks() : 0;
        s.pos_off = pl.n_pos;
        pl.n_pos += (len + 64 + 63) & ~63LL;
        s.sym_off = pl.n_syms;
        pl.n_syms += len + 64 + (ro ? kLitBufsize : 0);  // a continued run starts with the symbols of the block in progress
        s.chunk_off = (int32_t)pl.n_chunks;
        pl.n_chunks += s.nchunks;
        s.seg_off = (int32_t)pl.n_segs;
        s.nsegs = 0;
        s.cut_off = (int32_t)pl.n_cuts, s.cut_cap = 0;
        s.grid_chunks = 0;
        if (s.body_end >= 0) {
            s.grid_chunks = 1;
            for (size_t k = 1; k + 1 < geo.cstart.size() && s.grid_chunks; k++)
                if (geo.cstart[k] != (int32_t)((int64_t)k * kChunk - (kMinLookahead - 1))) s.grid_chunks = 0;
            // ... and the segments those of a single Write's window ends (chunk_ctx's closed form for the chunk -> segment marks)
            if (geo.cstart.empty() || geo.cstart[0] != 0) s.grid_chunks = 0;
            for (size_t k = 0; k < geo.head.size() && s.grid_chunks; k++)
                if (geo.head[k] != ((k >= 32 && ((k - 32) & 15) == 0) ? (int32_t)((k - 32) >> 4) + 2 : 0)) s.grid_chunks = 0;
            s.cut_cap = (int32_t)(geo.cl.size() + (size_t)geo.nsegs() + 8);
            pl.n_cuts += s.cut_cap;
            s.nsegs = geo.nsegs();
            const int32_t cl0 = (int32_t)pl.cl.size();
            pl.seg_c0.insert(pl.seg_c0.end(), geo.seg_c0.begin(), geo.seg_c0.end());
            pl.seg_after.insert(pl.seg_after.end(), geo.seg_after.begin(), geo.seg_after.end());
            pl.seg_base.insert(pl.seg_base.end(), geo.seg_base.begin(), geo.seg_base.end());
            pl.seg_S.insert(pl.seg_S.end(), geo.seg_S.begin(), geo.seg_S.end());
            for (int32_t v : geo.seg_cl) pl.seg_cl.push_back(v + cl0);
            pl.cl.insert(pl.cl.end(), geo.cl.begin(), geo.cl.end());
            pl.cstart.insert(pl.cstart.end(), geo.cstart.begin(), geo.cstart.end());
            pl.head.insert(pl.head.end(), geo.head.begin(), geo.head.end());
        } else {
            if (fast_resume) {  // segment 0: the engine as the flush left it -- the data ends at p0, so the first event fires at the run's first loop-top
                // (an event's trigger is the data end before it - 261, and a negative one means "none": p0 below 261 counts as 261)
                pl.seg_c0.push_back(0), pl.seg_after.push_back((int32_t)std::max<int64_t>(ro->p0, kMinLookahead - 1)), pl.seg_base.push_back((int32_t)ro->base0);
                pl.seg_S.push_back(0), pl.seg_cl.push_back((int32_t)pl.cl.size());
                s.nsegs++;
            }
            if (s.fv_end >= 0 || s.rle_end >= 0)  // levels 1-3, Rle: the events of the single Write (the tail engine takes base and data end from them)
                for (size_t k = 0; k < rev.size(); k++) {
                    const int64_t at = (k || fast_resume) ? rev[k].at - (s.rle_end >= 0 ? kMaxMatch : kMinLookahead - 1) : 0;  // where the segment starts
                    if (at > (s.rle_end >= 0 ? s.rle_end : s.fv_end)) break;
                    pl.seg_c0.push_back(0), pl.seg_after.push_back((int32_t)rev[k].after), pl.seg_base.push_back((int32_t)rev[k].base);
                    pl.seg_S.push_back(0), pl.seg_cl.push_back((int32_t)pl.cl.size());
                    s.nsegs++;
                }
            pl.seg_cl.push_back((int32_t)pl.cl.size());
            pl.cstart.push_back(0);
        }
        s.seg_c0 = s.seg_after = s.seg_base = s.seg_S = s.seg_cl = nullptr, s.cl = nullptr, s.cstart = nullptr, s.head = nullptr;
        pl.n_segs += s.nsegs;
        s.sup_off = (int32_t)pl.n_sups;
        pl.n_sups += (s.nsegs + kSupSegs - 1) / kSupSegs;
        // levels 1-3, one Write, large enough: speculative chunk runs instead of one sequential engine
        s.run_chunk = run_chunk;
        s.fast_runs = fast_par ? (force_seq == 2 ? 1 : (int32_t)((len + run_chunk - 1) / run_chunk)) : 0;
        s.run_slots = fast_par ? (force_seq == 2 ? (int32_t)((len + 4096) / kFastChunk + 1) : s.fast_runs) : 0;
        s.run_off = (int32_t)pl.n_runs;
        pl.n_runs += s.run_slots;
        pl.w_runs.add(i, s.fast_runs);
        s.blk_off = (int32_t)pl.n_blocks;
        // level 0 runs with memLevel 7: a block is flushed every 8191 symbols (only Rle tallies symbols there)
        s.max_blocks = (int32_t)(level == 0 ? len / 8191 + len / 32506 + 4 : len / kBlockSyms + 2);
        if (flushing) s.max_blocks += (int32_t)writes->ends.size() + 1;  // every Write under a flush mode closes a block
        s.plan_blk = nullptr, s.plan_nblk = 0;
        s.final_run = final_run ? 1 : 0, s.cont = cont ? 1 : 0, s.persist = ro ? ro->persist : nullptr;
        s.abs_off = (ro && !resume) ? ro->abs_off : 0, s.adler_stream = ro ? ro->adler_stream : 1, s.carry_byte = ro ? ro->carry_byte : 0;
        if (level == 0 && strategy != kRle && !ro) {
            // DeflateStored: block boundaries from the sizes alone (zs_core.h plan_stored_blocks); s.plan_blk holds the
            // offset into the batch's list until the device address is known
            const size_t first = pl.plan_blk.size();
            const std::vector<int64_t> no_ends;
            const std::vector<uint8_t> no_flush;
            if (flushing) pl.plan_wr_blk.assign(writes->ends.size(), 0);
            plan_stored_blocks(len, (multi || flushing) ? writes->ends : no_ends, flushing ? writes->flush : no_flush,
                               [&](int64_t start, int32_t blen, int can_store, int eof) {
                                   pl.plan_blk.push_back(BlockRec{start, 0, blen, 0, can_store, eof});
                               },
                               [&](int w, int nb) {
                                   if (flushing) pl.plan_wr_blk[(size_t)w] = nb;
                               });
            s.plan_nblk = (int32_t)(pl.plan_blk.size() - first);
            s.plan_blk = (const BlockRec *)(uintptr_t)first;
            s.max_blocks = s.plan_nblk + 1;
        }
        pl.n_blocks += s.max_blocks;
        s.adler_off = (int32_t)pl.n_pieces;
        s.n_adler = ro ? 0 : (int32_t)((len + kAdlerPiece - 1) / kAdlerPiece);  // an incremental stream's checksum is its owner's
        pl.n_pieces += s.n_adler;
        pl.w_clear.add(i, (s.out_cap + 65535) / 65536);
        pl.w_adler.add(i, s.n_adler);
        if (s.body_end >= 0 || s.fast_runs > 0 || s.fv_end >= 0)
            pl.w_links.add(i, len - 5 > 0 ? (len - 5 + link_span - 1) / link_span : 0);
        if (s.body_end >= 0) {
            pl.w_match.add(i, (int64_t)s.body_end / kMatchTile + 1);
            pl.w_chunks.add(i, s.nchunks);
            pl.w_segs.add(i, s.nsegs);
            if (s.nsegs > kSupSegs) pl.w_sups.add(i, (s.nsegs + kSupSegs - 1) / kSupSegs);  // shorter streams are resolved row by row
        }
        pl.w_blocks.add(i, s.max_blocks);
        // what none of the parallel forms takes is the one-wave literal engine's: the whole run, not just its last 261 bytes
        if (s.body_end < 0 && s.fv_end < 0 && s.rle_end < 0 && s.fast_runs == 0 && s.plan_nblk == 0 && !(level == 0 && strategy != kRle && !ro)) {
            const int64_t from = resume ? ro->p0 : ro ? ro->hist : 0;
            pl.lit_bytes += std::max<int64_t>(0, len - from - (kMinLookahead - 1));
        }
    }
    if (allow_dual && dual_broken) {  // (a stream the fast forms do not take after all: the batch as it always was planned)
        allow_dual = false, dual_broken = false;
        goto plan_again;
    }
    pl.any_dual = allow_dual;
    c->lit_engine_bytes += pl.lit_bytes;
    c->fast_rounds = 0;
    const bool no_rounds_this_call = c->no_rounds_once;
    c->no_rounds_once = false;
    if (pl.any_fv && !getenv("ZS_FAST_NO_ROUNDS") && !no_rounds_this_call && !(ro && ro->resume && getenv("ZS_NO_RESUME_ROUNDS"))) {
        // levels 1-3: every stream's parse as rounds over its chunks, all chunks of the batch at once (zs_fast_sweep.h "Rounds"),
        // when that is the shorter way.  One workgroup per stream takes as long as the longest stream at 46 / 35 / 20 MB/s (levels
        // 1 / 2 / 3 on text; kennedy.xls and ptt5 are slower).  The rounds take the whole batch through the chip at ~3.5 GB/s a few
        // times over and then wait for the last disturbances to die out -- up to ~16 ms on text, less on anything else, and never
        // longer than the longest stream takes one workgroup (profiles/r04_fast_batch_shapes.log: 4 x 4 MiB 26 against 88 ms,
        // 32 x 4 MiB 46 against 91, 32 x 1 MiB 21 against 23, 64 x 1 MiB 26 against 23, 128 x 256 KiB 10 against 6.6 at level 1; at
        // level 3 the rounds win up to 64 x 1 MiB: 31 against 49).
        int64_t pos_fv = 0, max_fv = 0;
        for (int i = 0; i < n; i++)
            if (pl.sd[(size_t)i].fv_end >= 0) {
                pos_fv += pl.sd[(size_t)i].fv_end + 1 - pl.sd[(size_t)i].start_pos;
                max_fv = std::max<int64_t>(max_fv, pl.sd[(size_t)i].fv_end + 1 - pl.sd[(size_t)i].start_pos);
            }
        const double t_stream = (double)max_fv / (level >= 3 ? 20e6 : level == 2 ? 35e6 : 46e6);
        const double t_rounds = std::min(0.016, (double)max_fv / 60e6) + (double)pos_fv / 3.5e9;
        if (getenv("ZS_FR_RATIO") ? (double)pos_fv <= atof(getenv("ZS_FR_RATIO")) * (double)max_fv : t_rounds < t_stream) {
            // chunks of 2048 positions or more (a run's fixed cost -- staging 32 K positions of history -- is 15-30 us, a sweep makes
            // ~390 positions final in 8 us), at most what one staging of the tile covers; between the two, as many chunks as fit
            // the chip at once: a round of 260 chunks takes the 256 CUs twice as long as one of 250
            int64_t target = getenv("ZS_FR_CHUNK") ? atoll(getenv("ZS_FR_CHUNK")) : pos_fv / 250;
            target = target < 2048 ? 2048 : target > kFsChunkMax ? kFsChunkMax : target;
            bool too_short = false;
            for (;;) {
                pl.fr_chunks.clear();
                pl.fr_max_n = 0;
                too_short = false;
                for (int i = 0; i < n; i++) {
                    StreamDesc &s = pl.sd[(size_t)i];
                    if (s.fv_end < 0) continue;
                    s.fr_first = (int32_t)pl.fr_chunks.size();
                    {
                        // the stream's read events are its segments (seg_after of event k - 1 = the data end before event k)
                        const int32_t *after = pl.seg_after.data() + s.seg_off;
                        // (a resumed run: segment 0 is the engine as the flush left it, event 1 the read at p0 itself)
                        const bool res = s.resume != 0;
                        const int64_t shortest = fs_build_chunks(i, (int64_t)s.fv_end, s.nsegs - 1, [&](int k) { return (int64_t)after[k - 1] - (kMinLookahead - 1); }, (int)target, pl.fr_chunks,
                                                                 res ? 2 : 1, res ? s.start_pos : 0);
                        if (shortest < kFsMinSpan) too_short = true;  // (Write ends a few hundred bytes apart: more chunks in a chunk's reach than it looks at)
                    }
                    s.fr_n = (int32_t)pl.fr_chunks.size() - s.fr_first;
                    pl.fr_max_n = s.fr_n > pl.fr_max_n ? s.fr_n : pl.fr_max_n;
                }
                if (pl.fr_chunks.size() <= 256 || target >= kFsChunkMax || getenv("ZS_FR_CHUNK")) break;
                target += 128;
            }
            if (too_short) {  // one workgroup per stream then
                pl.fr_chunks.clear();
                pl.fr_max_n = 0;
                for (int i = 0; i < n; i++) pl.sd[(size_t)i].fr_first = pl.sd[(size_t)i].fr_n = 0;
            }
            for (FsChunk &ck : pl.fr_chunks) {
                ck.prov_off = (uint32_t)pl.fr_prov;
                pl.fr_prov += (size_t)(ck.b_hi - ck.b_lo) + kMaxMatch + 64;  // (its loop-tops lie in [b_lo, b_hi + 258))
            }
            if (pl.fr_prov > 0xFFFF0000u) {  // (prov_off is 32 bits: a batch beyond ~4 G positions takes one workgroup per stream)
                pl.fr_chunks.clear();
                pl.fr_max_n = 0, pl.fr_prov = 0;
                for (int i = 0; i < n; i++) pl.sd[(size_t)i].fr_first = pl.sd[(size_t)i].fr_n = 0;
            }
        }
    }
    WorkList *const lists[kWorkLists] = {&pl.w_clear, &pl.w_adler, &pl.w_links, &pl.w_match, &pl.w_chunks, &pl.w_segs, &pl.w_sups, &pl.w_blocks, &pl.w_runs};
    for (WorkList *l : lists) l->finish(n);
    // ---- workspace ----
    size_t n_work = pl.w_clear.size() + pl.w_adler.size() + pl.w_links.size() + pl.w_match.size() + pl.w_chunks.size() +
                    pl.w_segs.size() + pl.w_sups.size() + pl.w_blocks.size() + pl.w_runs.size();
    if (!ensure(c, c->sd, sizeof(StreamDesc) * (size_t)n) || !ensure(c, c->st, sizeof(StreamState) * (size_t)n) ||
        !ensure(c, c->work, sizeof(uint2) * (n_work + 1)) || !ensure(c, c->link, 2 * (size_t)pl.n_pos + 64) ||
        !ensure(c, c->mm, 8 * (size_t)pl.n_pos + 256) ||  // the symbol kernel stages whole 128-byte lines
        !ensure(c, c->maps, 4 * (size_t)(pl.n_chunks + 1) * kSlots) || !ensure(c, c->segmap, 8 * (size_t)(pl.n_segs + 1) * kSlots) ||
        !ensure(c, c->supmap, 8 * (size_t)(pl.n_sups + 1) * kSlots) || !ensure(c, c->chunk_far, 2 * (size_t)pl.n_chunks + 64) ||
        !ensure(c, c->seg_entry, 2 * (size_t)(pl.n_segs + 2)) || !ensure(c, c->seg_symbase, 4 * (size_t)(pl.n_segs + 2)) ||
        !ensure(c, c->seg_stale, (size_t)pl.n_segs + 64) || !ensure(c, c->entry, 2 * (size_t)(pl.n_chunks + 2)) ||
        !ensure(c, c->symbase, 4 * (size_t)(pl.n_chunks + 2)) || !ensure(c, c->stale, (size_t)pl.n_chunks + 64) ||
        !ensure(c, c->syms, 4 * (size_t)pl.n_syms + 64) || !ensure(c, c->blk_end, 4 * (size_t)pl.n_blocks + 64) ||
        !ensure(c, c->blk_top, 4 * (size_t)pl.n_blocks + 64) || !ensure(c, c->blocks, sizeof(BlockRec) * (size_t)pl.n_blocks) ||
        !ensure(c, c->trees, sizeof(TreeWork) * (size_t)pl.n_blocks) || !ensure(c, c->info, sizeof(BlockInfo) * (size_t)pl.n_blocks) ||
        !ensure(c, c->pieces, 4 * (size_t)pl.n_pieces + 64) || !ensure(c, c->scratch, (size_t)kScratchBytes * (size_t)n) ||
        !ensure(c, c->cut_pos, 8 * (size_t)pl.n_cuts + 64) || !ensure(c, c->cut_bkt, 8 * (size_t)pl.n_cuts + 64))
        return false;
    // parse-segment tables: [seg_c0 | seg_after | seg_base | seg_S : int32 x n_segs each][seg_cl : int32 x (n_segs + n)]
    // [cstart : int32 x (n_chunks + n)][head : int32 x n_chunks][cl : u32 x n_cl]
    const size_t o_segcl = 16 * (size_t)pl.n_segs, o_cstart = o_segcl + 4 * ((size_t)pl.n_segs + (size_t)n),
                 o_head = o_cstart + 4 * ((size_t)pl.n_chunks + (size_t)n), o_cl = o_head + 4 * (size_t)pl.n_chunks;
    const size_t geo_bytes = o_cl + 4 * pl.cl.size();
    if (!ensure(c, c->geo, geo_bytes + 64)) return false;
    for (int i = 0; i < n; i++) {
        StreamDesc &s = pl.sd[(size_t)i];
        const int32_t *g = (const int32_t *)c->geo.p;
        const uint8_t *gb = (const uint8_t *)c->geo.p;
        s.seg_c0 = g + s.seg_off, s.seg_after = g + pl.n_segs + s.seg_off, s.seg_base = g + 2 * pl.n_segs + s.seg_off;
        s.seg_S = g + 3 * pl.n_segs + s.seg_off;
        s.seg_cl = (const int32_t *)(gb + o_segcl) + s.seg_off + i;
        s.cstart = (const int32_t *)(gb + o_cstart) + s.chunk_off + i;
        s.head = (const int32_t *)(gb + o_head) + s.chunk_off;
        s.cl = (const uint32_t *)(gb + o_cl);
    }
    if (pl.any_fv) {
        // one bit per position (pos_off is a multiple of 64: every stream's bitmap starts on a word)
        if (!ensure(c, c->ins_bits, (size_t)pl.n_pos / 8 + kFvBitSlack)) return false;
        for (int i = 0; i < n; i++)
            if (pl.sd[(size_t)i].fv_end >= 0) pl.sd[(size_t)i].ins_bits = dev<uint32_t>(c->ins_bits) + pl.sd[(size_t)i].pos_off / 32;
    }
    if (!pl.plan_blk.empty()) {
        if (!ensure(c, c->plan_blk, sizeof(BlockRec) * pl.plan_blk.size())) return false;
        for (int i = 0; i < n; i++)
            if (pl.sd[(size_t)i].plan_nblk) pl.sd[(size_t)i].plan_blk = dev<BlockRec>(c->plan_blk) + (uintptr_t)pl.sd[(size_t)i].plan_blk;
    }
    if (pl.n_runs &&
        (!ensure(c, c->run_syms, 4 * (size_t)pl.n_runs * kFastRunSyms) || !ensure(c, c->run_bits, 4 * (size_t)pl.n_runs * kFastRunBitWords) ||
         !ensure(c, c->run_scratch, (size_t)pl.n_runs * kFastRunScratch) || !ensure(c, c->run_outs, sizeof(FastRunOut) * (size_t)pl.n_runs) ||
         !ensure(c, c->run_fail, 4 * (size_t)n + 64)))
        return false;
    // rounds: the call goes on behind the batched cut rounds of the call that made it -- same plan, same workspace; nothing
    // is uploaded or zeroed again and the kernels up to the resolve kernel are not run
    if (!rounds && writes && (writes->ends.size() > 1 || writes->flushing() || ro)) {
        // [ends: int64 x nw][blocks before each Write: int32 x nw][flush modes: u8 x nw]
        const size_t nw = writes->ends.size();
        if (!ensure(c, c->wr, 13 * nw + 64)) return false;
        ZS_HIP(c, hipMemsetAsync(c->wr.p, 0, 13 * nw + 64, stream));
        if (nw) ZS_HIP(c, hipMemcpyAsync(c->wr.p, writes->ends.data(), sizeof(int64_t) * nw, hipMemcpyHostToDevice, stream));
        pl.sd[0].wr_end = (const int64_t *)c->wr.p;
        if (writes->flushing() || ro) {
            if (nw) ZS_HIP(c, hipMemcpyAsync((uint8_t *)c->wr.p + 12 * nw, writes->flush.data(), nw, hipMemcpyHostToDevice, stream));
            pl.sd[0].wr_blk = (int32_t *)((uint8_t *)c->wr.p + 8 * nw);
            pl.sd[0].wr_flush = (const uint8_t *)c->wr.p + 12 * nw;
        }
        if (!pl.plan_wr_blk.empty())
            ZS_HIP(c, hipMemcpyAsync((uint8_t *)c->wr.p + 8 * nw, pl.plan_wr_blk.data(), 4 * nw, hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));  // `writes` is caller-owned pageable memory
    }
    if (!rounds && !pl.plan_blk.empty()) {
        ZS_HIP(c, hipMemcpyAsync(c->plan_blk.p, pl.plan_blk.data(), sizeof(BlockRec) * pl.plan_blk.size(), hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));  // pageable source
    }
    // ---- upload descriptors and work lists (one pinned staging copy) ----
    const size_t pre_bytes = 4 * (size_t)kWorkLists * (size_t)(n + 1);
    size_t up_bytes = sizeof(StreamDesc) * (size_t)n + pre_bytes + geo_bytes + 16;
    if (!ensure_pinned(c, std::max(up_bytes, sizeof(StreamState) * (size_t)n)) || !ensure(c, c->wpre, pre_bytes + 64)) return false;
    uint8_t *hp = (uint8_t *)c->pinned;
    memcpy(hp, pl.sd.data(), sizeof(StreamDesc) * (size_t)n);
    int32_t *hpre = (int32_t *)(hp + sizeof(StreamDesc) * (size_t)n);
    size_t o_clear = 0, o_adler, o_links, o_match, o_chunks, o_segs, o_sups, o_blocks, o_runs;
    WorkOffsets wo;
    {
        size_t *const offs[kWorkLists] = {&o_clear, &o_adler, &o_links, &o_match, &o_chunks, &o_segs, &o_sups, &o_blocks, &o_runs};
        size_t at = 0;
        for (int l = 0; l < kWorkLists; l++) {
            *offs[l] = at;
            wo.off[l] = (uint32_t)at;
            memcpy(hpre + (size_t)l * (size_t)(n + 1), lists[l]->pre.data(), 4 * (size_t)(n + 1));
            at += lists[l]->size();
        }
        wo.off[kWorkLists] = (uint32_t)at;
    }
    if (geo_bytes) {
        uint8_t *hg = hp + sizeof(StreamDesc) * (size_t)n + pre_bytes;
        if (pl.n_segs) {
            memcpy(hg, pl.seg_c0.data(), 4 * (size_t)pl.n_segs);
            memcpy(hg + 4 * (size_t)pl.n_segs, pl.seg_after.data(), 4 * (size_t)pl.n_segs);
            memcpy(hg + 8 * (size_t)pl.n_segs, pl.seg_base.data(), 4 * (size_t)pl.n_segs);
            memcpy(hg + 12 * (size_t)pl.n_segs, pl.seg_S.data(), 4 * (size_t)pl.n_segs);
        }
        memcpy(hg + o_segcl, pl.seg_cl.data(), 4 * pl.seg_cl.size());
        memcpy(hg + o_cstart, pl.cstart.data(), 4 * pl.cstart.size());
        if (pl.n_chunks) memcpy(hg + o_head, pl.head.data(), 4 * (size_t)pl.n_chunks);
        if (!pl.cl.empty()) memcpy(hg + o_cl, pl.cl.data(), 4 * pl.cl.size());
        if (!rounds) ZS_HIP(c, hipMemcpyAsync(c->geo.p, hg, geo_bytes, hipMemcpyHostToDevice, stream));
    }
    if (!rounds) ZS_HIP(c, hipMemcpyAsync(c->sd.p, hp, sizeof(StreamDesc) * (size_t)n, hipMemcpyHostToDevice, stream));
    if (n_work && !rounds) {
        ZS_HIP(c, hipMemcpyAsync(c->wpre.p, hpre, pre_bytes, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(zs_worklist_kernel, dim3((unsigned)((n_work + 255) / 256)), dim3(256), 0, stream, dev<int32_t>(c->wpre), n, wo,
                           dev<uint2>(c->work));
    }
    if (!rounds) {
        // the per-run flags and state in one launch (the link array is not cleared: K1 writes every entry that is read)
        ZeroRegions z;
        auto reg = [&](int r, DevBuf &b, size_t bytes) {
            z.p[r] = b.p;
            z.n16[r] = (uint32_t)(((bytes + 15) / 16 < b.cap / 16) ? (bytes + 15) / 16 : b.cap / 16);
        };
        reg(0, c->stale, (size_t)pl.n_chunks + 64);
        reg(1, c->seg_stale, (size_t)pl.n_segs + 64);
        reg(2, c->st, sizeof(StreamState) * (size_t)n);
        if (pl.any_fv) reg(3, c->ins_bits, (size_t)pl.n_pos / 8 + kFvBitSlack);
        else z.p[3] = nullptr, z.n16[3] = 0;
        uint32_t most = 0;
        for (int r = 0; r < 4; r++) most = z.n16[r] > most ? z.n16[r] : most;
        unsigned blocks = (most + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(zs_zero_kernel, dim3(blocks), dim3(256), 0, stream, z);
        if (ro && ro->resume && pl.sd[0].fv_end >= 0) {
            // a resumed run at levels 1-3: below p0 the set is what the suspended engine's chains contain -- every position they
            // reach is in it (and no walk meets another one): all ones
            // (exactly the bits below p0: the chunk form's commit ORs the run's own bits in)
            const int64_t lo = std::max<int64_t>(0, ro->p0 - kWSize - 64) / 32, hi = ro->p0 / 32;
            if (hi > lo) ZS_HIP(c, hipMemsetAsync(pl.sd[0].ins_bits + lo, 0xFF, (size_t)(hi - lo) * 4, stream));
            ZS_HIP(c, hipMemsetD32Async((hipDeviceptr_t)(pl.sd[0].ins_bits + hi), (int)((1u << (ro->p0 & 31)) - 1u), 1, stream));
        }
    }

    const StreamDesc *d_sd = dev<StreamDesc>(c->sd);
    StreamState *d_st = dev<StreamState>(c->st);
    const uint2 *d_work = dev<uint2>(c->work);
    const bool prof = c->profiling;
    const int k5_ahead = getenv("ZS_K5_AHEAD") ? atoi(getenv("ZS_K5_AHEAD")) : 1;  // lines the symbol kernel's helper wave asks for ahead of a lane
    auto mark = [&](int i) {
        if (prof) (void)hipEventRecord(c->ev[i], stream);
    };
    // the output buffers' clearing and the Adler-32 pieces are wanted by the last two kernels only: they run on the second
    // stream beside the link and match kernels (their stages are reported as 0: 0.015 and 0.03 ms on english64 alone, and
    // their events are not recorded).  The host enqueues them behind the first launches of the critical path: every call
    // in front of K1 is ~5 us in which the device waits for the host.
    ht_plan = ht_us();
    ZS_HIP(c, hipEventRecord(c->ev_pre0, stream));
    auto side_work = [&]() -> bool {
        ZS_HIP(c, hipStreamWaitEvent(c->aux, c->ev_pre0, 0));
        if (!pl.w_clear.empty())
            hipLaunchKernelGGL(zs_clear_kernel, dim3((unsigned)pl.w_clear.size()), dim3(256), 0, c->aux, d_sd, d_work + o_clear);
        if (!pl.w_adler.empty())
            hipLaunchKernelGGL(zs_adler_kernel, dim3((unsigned)pl.w_adler.size()), dim3(256), 0, c->aux, d_sd, d_work + o_adler,
                               dev<uint32_t>(c->pieces));
        ZS_HIP(c, hipEventRecord(c->ev_pre, c->aux));
        return true;
    };
    // (ZS_NO_DEFER: every cut on the stream's own CU; ZS_FORCE_ROUNDS: the rounds from the first cut on -- for the tests)
    const int cut_budget = getenv("ZS_FORCE_ROUNDS") ? 0 : kCutBudget;
    const int defer_mode = ((ro || getenv("ZS_NO_DEFER")) ? 0 : 1) | (getenv("ZS_DEBUG_CUTS") ? 0x100 : 0) | (cut_budget << 16);
    const int cut_stride = (int)pl.n_cuts;
    const bool use_sup = !pl.w_sups.empty() && !getenv("ZS_NO_SUPMAP");
    auto launch_resolve = [&](int mode, int iter) {
        hipLaunchKernelGGL(zs_resolve_kernel, dim3((unsigned)n), dim3(1024), kResolveLds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                           dev<uint2>(c->mm), dev<uint32_t>(c->maps), dev<uint2>(c->segmap),
                           dev<uint16_t>(c->seg_entry), dev<uint32_t>(c->seg_symbase), dev<uint8_t>(c->stale),
                           dev<uint8_t>(c->seg_stale), c->crc_tab, lv, strategy, hash_variant, 0x7FFFFFFF, 0x7FFFFFFF,
                           use_sup ? dev<uint2>(c->supmap) : (const uint2 *)nullptr, dev<uint16_t>(c->chunk_far), mode, dev<int32_t>(c->cut_pos),
                           dev<uint32_t>(c->cut_bkt), cut_stride, iter);
    };
    // Batched cut rounds, for the streams the resolve kernel has given up (StreamState::deferred = 1: data whose read events
    // are equal-bucket ones by the dozen -- zeros, runs, zero pages -- or with thousands of positions to walk again behind
    // each).  Where a cut falls depends on the parse up to it, and the parse on the repairs of the cuts before: one CU doing
    // cut, repair, cut, repair is what made such streams 30-170 times slower than text.  Instead: a dry pass of the resolve
    // kernel walks the rest of the stream on the records as they are and collects every cut on its way; the records from the
    // first cut that differs from the pass before are put back as they were (a copy made when the rounds began) and all
    // the cuts of the pass are repaired at once over the chip; the maps of the chunks that changed are made again; and so
    // on until a pass finds the cuts of the pass before -- then the records it walked are the ones those cuts leave, which
    // is what the one-after-the-other order gives (the first cut never depends on a repair, the second only on the
    // first's, ...: a pass fixes at least one more cut, in practice nearly all of them).
    auto run_cut_rounds = [&]() -> bool {
        // (the records as they are now, and behind them the chunks' largest match distances: what a restore puts back)
        const size_t far_off = 8 * (size_t)pl.n_pos + 256;
        if (!ensure(c, c->mm_bak, far_off + 2 * (size_t)pl.n_chunks + 64)) return false;
        ZS_HIP(c, hipMemcpyAsync(c->mm_bak.p, c->mm.p, 8 * (size_t)pl.n_pos + 256, hipMemcpyDeviceToDevice, stream));
        ZS_HIP(c, hipMemcpyAsync((uint8_t *)c->mm_bak.p + far_off, c->chunk_far.p, 2 * (size_t)pl.n_chunks + 64, hipMemcpyDeviceToDevice, stream));
        ZS_HIP(c, hipMemsetAsync(c->cut_pos.p, 0xFF, 8 * (size_t)pl.n_cuts + 64, stream));  // no cut in any slot (the pass before the first)
        StreamState *hr = (StreamState *)c->pinned;
        for (int iter = 0;; iter++) {
            launch_resolve(3 | (defer_mode & 0x100), iter);
            ZS_HIP(c, hipMemcpyAsync(hr, d_st, sizeof(StreamState) * (size_t)n, hipMemcpyDeviceToHost, stream));
            ZS_HIP(c, hipStreamSynchronize(stream));
            bool all_same = true;
            int max_nc = 0;
            for (int i = 0; i < n; i++)
                if (hr[i].deferred == 1 && !hr[i].cuts_same) all_same = false, max_nc = std::max(max_nc, hr[i].nc[iter & 1]);
            hipLaunchKernelGGL(zs_cuts_apply_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, d_st, dev<uint16_t>(c->link), dev<int32_t>(c->cut_pos),
                               cut_stride, iter);  // (the streams that are through leave the rounds)
            if (getenv("ZS_DEBUG")) {
                int nd = 0, nsame = 0;
                for (int i = 0; i < n; i++) nd += hr[i].deferred == 1, nsame += hr[i].deferred == 1 && hr[i].cuts_same;
                fprintf(stderr, "zs: cut round %d: %d streams in the rounds, %d settled, most cuts %d, stream 0: %d cuts, first difference at cut %d (position %d)\n", iter,
                        nd, nsame, max_nc, hr[0].nc[iter & 1], hr[0].cut_diff_idx, hr[0].cut_diff_pos);
            }
            if (all_same && getenv("ZS_DEBUG_CUTS")) {  // the cuts the rounds settled on, stream 0
                const int ncap = hr[0].nc[iter & 1];
                std::vector<int32_t> cp((size_t)ncap);
                std::vector<uint32_t> cb((size_t)ncap);
                (void)hipMemcpy(cp.data(), dev<int32_t>(c->cut_pos) + (size_t)(iter & 1) * (size_t)cut_stride + pl.sd[0].cut_off, 4 * (size_t)ncap, hipMemcpyDeviceToHost);
                (void)hipMemcpy(cb.data(), dev<uint32_t>(c->cut_bkt) + (size_t)(iter & 1) * (size_t)cut_stride + pl.sd[0].cut_off, 4 * (size_t)ncap, hipMemcpyDeviceToHost);
                fprintf(stderr, "zs: cuts of stream 0 (slot: position / bucket):");
                for (int k = 0; k < ncap; k++)
                    if (cp[(size_t)k] >= 0) fprintf(stderr, " %d: %d / %u", k, cp[(size_t)k], cb[(size_t)k]);
                fprintf(stderr, "\n");
            }
            if (all_same) break;
            if (iter >= 4096) {
                c->err = "the cut rounds did not settle";
                return false;
            }
            c->cut_rounds++;
            if (iter > 0)  // (before the first round's repairs the records are the copy)
            hipLaunchKernelGGL(zs_cut_restore_kernel, dim3(2048, (unsigned)std::min(n, 65535)), dim3(256), 0, stream, d_sd, d_st, dev<uint2>(c->mm), (const uint2 *)c->mm_bak.p,
                               dev<uint8_t>(c->stale), dev<uint8_t>(c->seg_stale), dev<uint16_t>(c->chunk_far), (const uint16_t *)((const uint8_t *)c->mm_bak.p + far_off), n);
            if (max_nc > 0) {
                // few cuts: 128 workgroups each (every 128th position behind the cut); many: fewer, larger ones
                if (max_nc <= 64)
                    hipLaunchKernelGGL((zs_cuts_repair_kernel<256, 1>), dim3(128, (unsigned)max_nc, (unsigned)std::min(n, 65535)), dim3(256), kRepairLds, stream, d_sd, d_st,
                                       dev<uint16_t>(c->link), dev<uint2>(c->mm), dev<uint8_t>(c->stale), dev<uint8_t>(c->seg_stale),
                                       dev<uint16_t>(c->chunk_far), c->crc_tab, lv, hash_variant, dev<int32_t>(c->cut_pos), dev<uint32_t>(c->cut_bkt),
                                       cut_stride, iter, n);
                else if (max_nc <= 2048)
                    hipLaunchKernelGGL((zs_cuts_repair_kernel<256, 8>), dim3(16, (unsigned)max_nc, (unsigned)std::min(n, 65535)), dim3(256), kRepairLds, stream, d_sd, d_st,
                                       dev<uint16_t>(c->link), dev<uint2>(c->mm), dev<uint8_t>(c->stale), dev<uint8_t>(c->seg_stale),
                                       dev<uint16_t>(c->chunk_far), c->crc_tab, lv, hash_variant, dev<int32_t>(c->cut_pos), dev<uint32_t>(c->cut_bkt),
                                       cut_stride, iter, n);
                else
                    hipLaunchKernelGGL((zs_cuts_repair_kernel<1024, 32>), dim3(1, (unsigned)std::min(max_nc, 65535), (unsigned)std::min(n, 65535)), dim3(1024), kRepairLds, stream, d_sd,
                                       d_st, dev<uint16_t>(c->link), dev<uint2>(c->mm), dev<uint8_t>(c->stale), dev<uint8_t>(c->seg_stale),
                                       dev<uint16_t>(c->chunk_far), c->crc_tab, lv, hash_variant, dev<int32_t>(c->cut_pos), dev<uint32_t>(c->cut_bkt),
                                       cut_stride, iter, n);
            }
            hipLaunchKernelGGL(zs_chunkmap_kernel, dim3((unsigned)pl.w_chunks.size()), dim3(512), 0, stream, d_sd, d_work + o_chunks,
                               dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint32_t>(c->maps), c->crc_tab, lv, strategy,
                               hash_variant, dev<uint16_t>(c->chunk_far), dev<uint8_t>(c->stale), (const StreamState *)d_st);
            hipLaunchKernelGGL(zs_segmap_kernel, dim3((unsigned)pl.w_segs.size()), dim3(320), 0, stream, d_sd, d_work + o_segs,
                               dev<uint32_t>(c->maps), dev<uint2>(c->segmap));
            ZS_HIP(c, hipMemsetAsync(c->seg_stale.p, 0, (size_t)pl.n_segs + 64, stream));
            if (use_sup)
                hipLaunchKernelGGL(zs_supmap_kernel, dim3((unsigned)pl.w_sups.size()), dim3(320), 0, stream, d_sd, d_work + o_sups,
                                   dev<uint2>(c->segmap), dev<uint8_t>(c->seg_stale), dev<uint2>(c->supmap));
        }
        return true;
    };
    // One long stream: the position-parallel kernels (links, matches, chunk maps) fill the chip, the kernels that follow
    // the parse (resolve: one workgroup; symbols: one lane per chunk, a latency chain) leave it idle.  ZS_PIPE_PARTS=k cuts
    // the stream into k parts at parse-segment boundaries: while the match kernel works on part i + 1 the second HIP stream
    // resolves part i and emits its symbols (the resolve kernel keeps its place between launches, StreamState r_*).  Same
    // kernels, same bytes -- and, measured on english64, the same time (5.82 ms with 2 parts, 5.90 with 4, against 5.85): the
    // symbol kernel's ~0.65 ms latency floor is paid again behind the last part and the match kernel loses 0.2-0.3 ms to the
    // company.  Off unless asked for.
    int n_parts = 0;
    if (n == 1 && !ro && pl.sd[0].body_end >= 0 && pl.sd[0].nsegs >= 256 && strategy != kHuffmanOnly && !pl.any_fv) {
        const char *e = getenv("ZS_PIPE_PARTS");
        n_parts = e ? atoi(e) : 0;
        if (n_parts > 16) n_parts = 16;
        if (n_parts < 2) n_parts = 0;
    }
    struct TimedPair {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<TimedPair> pairs;
    size_t pool_used = 0;
    if (n_parts) {
        if (!side_work()) return false;
        for (int i = 2; i <= 9; i++) mark(i);  // stages 2..9 are timed launch by launch below; their marks only have to exist
        auto timed = [&](int stage, hipStream_t st_, auto &&launch) {
            if (!prof) {
                launch();
                if (getenv("ZS_DEBUG")) {
                    hipError_t e_ = hipGetLastError();
                    if (e_ != hipSuccess) fprintf(stderr, "zs: launch of stage %s failed: %s\n", kStageNames[stage], hipGetErrorString(e_));
                }
                return;
            }
            while (c->ev_pool.size() < pool_used + 2) {
                hipEvent_t ev = nullptr;
                (void)hipEventCreate(&ev);
                c->ev_pool.push_back(ev);
            }
            hipEvent_t a = c->ev_pool[pool_used++], b = c->ev_pool[pool_used++];
            (void)hipEventRecord(a, st_);
            launch();
            (void)hipEventRecord(b, st_);
            pairs.push_back(TimedPair{stage, a, b});
        };
        const StreamDesc &s0 = pl.sd[0];
        const int nsegs = s0.nsegs, nchunks = s0.nchunks;
        const int64_t n_tiles = (int64_t)pl.w_match.size(), n_spans = (int64_t)pl.w_links.size();
        int64_t tiles_done = 0, spans_done = 0;
        for (int k = 0; k < n_parts; k++) {
            const int sa = (int)((int64_t)nsegs * k / n_parts), sb = (int)((int64_t)nsegs * (k + 1) / n_parts);
            const int ca = pl.seg_c0[(size_t)sa], cb = sb < nsegs ? pl.seg_c0[(size_t)sb] : nchunks;
            // the chunks below cb need the match records of the positions below their end
            const int64_t q = sb < nsegs ? (int64_t)pl.cstart[(size_t)cb] : (int64_t)s0.body_end + 1;
            int64_t tiles_end = (q + kMatchTile - 1) / kMatchTile;
            if (tiles_end > n_tiles || sb == nsegs) tiles_end = n_tiles;
            int64_t spans_end = (tiles_end * kMatchTile + link_span - 1) / link_span;
            if (spans_end > n_spans || sb == nsegs) spans_end = n_spans;
            // the link kernel's time is that of one workgroup's span whatever the number of spans: all of them at once, up front
            if (k == 0) spans_end = n_spans;
            if (spans_end > spans_done)
                timed(kStLinks, stream, [&] {
                    hipLaunchKernelGGL(zs_links_kernel, dim3((unsigned)(spans_end - spans_done)), dim3(1024), kLkLds, stream, d_sd,
                                       d_work + o_links + spans_done, dev<uint16_t>(c->link), c->crc_tab, hash_variant, (int)link_span);
                });
            if (k == 0 && ro && ro->resume)  // (one stream: a resumed run's chains are the ones the engine before it left)
                hipLaunchKernelGGL(zs_import_chains_kernel, dim3(64), dim3(1024), 0, stream, d_sd, 0, dev<uint16_t>(c->link), c->crc_tab, hash_variant, ro->p0);
            if (tiles_end > tiles_done)
                timed(kStMatch, stream, [&] {
                    hipLaunchKernelGGL(zs_match_kernel, dim3((unsigned)(tiles_end - tiles_done)), dim3(1024), kMatchLds + 16, stream, d_sd,
                                       d_work + o_match + tiles_done, dev<uint16_t>(c->link), dev<uint2>(c->mm), lv, strategy);
                });
            spans_done = spans_end > spans_done ? spans_end : spans_done, tiles_done = tiles_end > tiles_done ? tiles_end : tiles_done;
            timed(kStChunkMap, stream, [&] {
                hipLaunchKernelGGL(zs_chunkmap_kernel, dim3((unsigned)(cb - ca)), dim3(512), 0, stream, d_sd, d_work + o_chunks + ca,
                                   dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint32_t>(c->maps), c->crc_tab, lv, strategy, hash_variant, dev<uint16_t>(c->chunk_far));
            });
            timed(kStSegMap, stream, [&] {
                hipLaunchKernelGGL(zs_segmap_kernel, dim3((unsigned)(sb - sa)), dim3(320), 0, stream, d_sd, d_work + o_segs + sa,
                                   dev<uint32_t>(c->maps), dev<uint2>(c->segmap));
            });
            ZS_HIP(c, hipEventRecord(c->ev_part[k], stream));
            ZS_HIP(c, hipStreamWaitEvent(c->aux, c->ev_part[k], 0));
            const int64_t mm_limit = tiles_end * kMatchTile - 1;
            timed(kStResolve, c->aux, [&] {
                hipLaunchKernelGGL(zs_resolve_kernel, dim3(1), dim3(1024), kResolveLds, c->aux, d_sd, d_st, dev<uint16_t>(c->link),
                                   dev<uint2>(c->mm), dev<uint32_t>(c->maps), dev<uint2>(c->segmap), dev<uint16_t>(c->seg_entry),
                                   dev<uint32_t>(c->seg_symbase), dev<uint8_t>(c->stale), dev<uint8_t>(c->seg_stale), c->crc_tab, lv,
                                   strategy, hash_variant, sb, (int)(mm_limit > 0x7FFFFFFF ? 0x7FFFFFFF : mm_limit), (const uint2 *)nullptr,
                                   dev<uint16_t>(c->chunk_far), 0, (int32_t *)nullptr, (uint32_t *)nullptr, 0, 0);
            });
            if (k == n_parts - 1) {
                // the tail engine needs what the last resolve launch left: it runs on the first stream beside the last part's symbols
                ZS_HIP(c, hipEventRecord(c->ev_fork, c->aux));
                ZS_HIP(c, hipStreamWaitEvent(stream, c->ev_fork, 0));
                timed(kStTail, stream, [&] {
                    hipLaunchKernelGGL(zs_tail_kernel, dim3(1), dim3(1024), kTailLds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                                       dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks),
                                       dev<uint8_t>(c->scratch), c->crc_tab, lv, strategy, hash_variant, level);
                });
            }
            timed(kStExpand, c->aux, [&] {
                hipLaunchKernelGGL(zs_expand_kernel, dim3((unsigned)((sb - sa + 63) / 64)), dim3(64), 0, c->aux, d_sd, d_st, d_work + o_segs + sa,
                                   sb - sa, dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint32_t>(c->maps), dev<uint16_t>(c->seg_entry),
                                   dev<uint32_t>(c->seg_symbase), dev<uint8_t>(c->stale), dev<uint16_t>(c->entry), dev<uint32_t>(c->symbase),
                                   c->crc_tab, lv, strategy, hash_variant);
            });
            timed(kStEmitSyms, c->aux, [&] {
                hipLaunchKernelGGL(zs_emit_syms_lane_kernel<4>, dim3((unsigned)((cb - ca + 63) / 64)), dim3(kK5Threads), 0, c->aux, d_sd, d_st,
                                   d_work + o_chunks + ca, cb - ca, dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint16_t>(c->entry), dev<uint32_t>(c->symbase),
                                   dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), c->crc_tab, lv, strategy,
                                   hash_variant, k5_ahead);
            });
        }
        ZS_HIP(c, hipEventRecord(c->ev_join, c->aux));
        ZS_HIP(c, hipStreamWaitEvent(stream, c->ev_join, 0));
        hipLaunchKernelGGL(zs_body_blocks_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, d_st, dev<int32_t>(c->blk_end),
                           dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks), dev<uint32_t>(c->syms));
    } else {
    if (!rounds) {
    mark(2);
    if (!pl.w_links.empty())
        hipLaunchKernelGGL(zs_links_kernel, dim3((unsigned)pl.w_links.size()), dim3(1024), kLkLds, stream, d_sd, d_work + o_links,
                           dev<uint16_t>(c->link), c->crc_tab, hash_variant, (int)link_span);
    if (ro && ro->resume)
        hipLaunchKernelGGL(zs_import_chains_kernel, dim3(64), dim3(1024), 0, stream, d_sd, 0, dev<uint16_t>(c->link), c->crc_tab, hash_variant, ro->p0);
    if (pl.any_dual) {
        // the links are there: sweeps or speculative runs (above, allow_dual)?  The other plan is struck from the descriptors
        bool periodic = true;
        if (!getenv("ZS_FAST_NO_PROBE")) {
            hipLaunchKernelGGL(zs_fast_probe_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, dev<uint16_t>(c->link), dev<int32_t>(c->run_fail));
            std::vector<int32_t> np((size_t)n, 0);
            ZS_HIP(c, hipMemcpyAsync(np.data(), c->run_fail.p, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
            ZS_HIP(c, hipStreamSynchronize(stream));
            for (int i = 0; i < n; i++) periodic = periodic && np[(size_t)i] == 0;
        }
        for (int i = 0; i < n; i++) {
            StreamDesc &sk = pl.sd[(size_t)i];
            if (periodic) sk.fv_end = -1, sk.fr_first = 0, sk.fr_n = 0;
            else sk.fast_runs = 0, sk.run_slots = 0;
        }
        if (periodic) pl.any_fv = false, pl.fr_chunks.clear(), pl.fr_max_n = 0;
        else pl.n_runs = 0;
        memcpy(c->pinned, pl.sd.data(), sizeof(StreamDesc) * (size_t)n);  // (the staging copy's uploads are through: the stream was just waited for)
        ZS_HIP(c, hipMemcpyAsync(c->sd.p, c->pinned, sizeof(StreamDesc) * (size_t)n, hipMemcpyHostToDevice, stream));
        if (getenv("ZS_DEBUG")) fprintf(stderr, "zs: %d stream(s) below 4 MiB at level %d: %s\n", n, level, periodic ? "all period, the speculative runs" : "the sweeps");
    }
    mark(3);
    if (strategy == kHuffmanOnly) {
        // Longest_match is never called (Deflate.Slow.cs:66-71): every position has no match
        ZS_HIP(c, hipMemsetAsync(c->mm.p, 0, 8 * (size_t)pl.n_pos + 64, stream));
    } else if (!pl.w_match.empty())
        hipLaunchKernelGGL(zs_match_kernel, dim3((unsigned)pl.w_match.size()), dim3(1024), kMatchLds + 16, stream, d_sd, d_work + o_match,
                           dev<uint16_t>(c->link), dev<uint2>(c->mm), lv, strategy);
    mark(4);
    if (!side_work()) return false;
    if (!pl.w_chunks.empty())
        hipLaunchKernelGGL(zs_chunkmap_kernel, dim3((unsigned)pl.w_chunks.size()), dim3(512), 0, stream, d_sd, d_work + o_chunks,
                           dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint32_t>(c->maps), c->crc_tab, lv, strategy,
                           hash_variant, dev<uint16_t>(c->chunk_far));
    mark(5);
    if (!pl.w_segs.empty())
        hipLaunchKernelGGL(zs_segmap_kernel, dim3((unsigned)pl.w_segs.size()), dim3(320), 0, stream, d_sd, d_work + o_segs,
                           dev<uint32_t>(c->maps), dev<uint2>(c->segmap));
    // the segment maps composed 16 at a time, for the resolve kernel's short way through a long stream (ZS_NO_SUPMAP: without)
    if (use_sup)
        hipLaunchKernelGGL(zs_supmap_kernel, dim3((unsigned)pl.w_sups.size()), dim3(320), 0, stream, d_sd, d_work + o_sups,
                           dev<uint2>(c->segmap), dev<uint8_t>(c->seg_stale), dev<uint2>(c->supmap));
    mark(6);
    // A stream whose refills are equal-bucket ones with positions to walk again by the thousand (zero pages, runs) is not
    // one CU's job: the resolve kernel gives it up after kDeferBudget such cuts (the kernels behind it skip the stream) and the
    // batch is run again in rounds -- the kernel stops at every such cut, the repair and the chunk maps behind it run over the
    // chip, and the kernel goes on (it keeps its place in StreamState).
    // (Only where a walk is long -- chains of 1024 and 4096, levels 8 and 9: a round costs ~0.1 ms of launches and a
    // synchronisation, which is what a cut's repair takes on one CU at level 6.  64 MiB of short runs: level 9 7.3 s -> 2.0 s,
    // level 6 175 ms inline against 237 in rounds.)
    // (ZS_DEFER_ALL: at every level, for the tests)
    launch_resolve(defer_mode, 0);
    }
    mark(7);
    // fork: the tail engine (sequential, one workgroup per stream) needs only what the resolve kernel left, so it runs
    // on the second stream beside the symbol kernels
    if (pl.any_fv) {
        // window-wide sweeps of a workgroup (zs_fast_sweep.hip), one workgroup per stream
        constexpr int fs_lds = fs_lds_bytes<1024, kFsTile1>();
        if (!pl.fr_chunks.empty()) {
            // rounds over the chunks of the streams: every round one workgroup per chunk, until a round changes nothing
            const size_t nch = pl.fr_chunks.size(), plane_words = (size_t)pl.n_pos / 32 + kFvBitSlack / 4 + 64;
            // the switches of the measurements, read once (not per round)
            static const int env_max_rounds = getenv("ZS_FR_MAX_ROUNDS") ? atoi(getenv("ZS_FR_MAX_ROUNDS")) : 0;
            static const int env_range = getenv("ZS_FR_RANGE") ? std::max(1, atoi(getenv("ZS_FR_RANGE"))) : 0;
            static const int env_group = getenv("ZS_FR_GROUP") ? atoi(getenv("ZS_FR_GROUP")) : 4;
            static const int env_tail = getenv("ZS_FR_TAIL_RANGE") ? atoi(getenv("ZS_FR_TAIL_RANGE")) : 1;
            static const bool env_no_seed = getenv("ZS_FR_NO_SEED") != nullptr, env_fixed = getenv("ZS_FR_RANGE_FIXED") != nullptr, env_dbg = getenv("ZS_DEBUG") != nullptr;
            const int max_rounds = env_max_rounds ? env_max_rounds : pl.fr_max_n + 2;  // (chunk r of a stream is the reference's after round r at the latest)
            if (!ensure(c, c->fr_chunks, nch * sizeof(FsChunk)) || !ensure(c, c->fr_meta, 2 * nch * sizeof(FsMeta)) || !ensure(c, c->fr_planes, 16 * plane_words) ||
                !ensure(c, c->fr_prov, 4 * pl.fr_prov + 64) || !ensure(c, c->fr_base, 4 * nch + 64) || !ensure(c, c->fr_counters, 4 * ((size_t)max_rounds + 16))) {
                // no room for the rounds' planes and provisional symbols: one workgroup per stream needs none of them
                c->err.clear();
                ZS_HIP(c, hipStreamSynchronize(c->aux));
                ZS_HIP(c, hipStreamSynchronize(stream));
                c->no_rounds_once = true;
                return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, force_seq, ro, false, force_lit);
            }
            ZS_HIP(c, hipMemcpyAsync(c->fr_chunks.p, pl.fr_chunks.data(), nch * sizeof(FsChunk), hipMemcpyHostToDevice, stream));
            ZS_HIP(c, hipMemsetAsync(c->fr_counters.p, 0, 4 * ((size_t)max_rounds + 16), stream));
            // more chunks than CUs: a workgroup takes a range of consecutive chunks in turn, every chunk reading what the chunks before
            // it in the range have just left -- within a range the parse is sequential, and the rounds only have to settle what
            // crosses the ranges' ends (1 MiB of text in chunks of 8192: 43 rounds and 27 runs per chunk with ranges of one, 21 and
            // 11.5 with ranges of four, 6 and 3.5 with ranges of sixteen; tests/model mode frounds)
            // A round's ranges are chosen from what the round before changed: while most chunks still change, as many ranges as CUs;
            // once the chunks that will run again fit the chip (a changed chunk wakes the ~4 behind it), one chunk per workgroup --
            // two active chunks of one range would wait for each other.
            // (below two chipfuls of chunks ranges of one: the corpus, 336 chunks of 11 files, 13.9 / 19.1 ms against 16.5 / 23.6)
            const int range_max = env_range ? env_range : nch >= 512 ? (int)((nch + 255) / 256) : 1;
            FsRounds fr{dev<FsChunk>(c->fr_chunks), dev<FsMeta>(c->fr_meta), dev<uint32_t>(c->fr_planes), dev<uint32_t>(c->fr_prov), dev<uint32_t>(c->fr_counters),
                        (int64_t)plane_words, (int)nch, 0, env_no_seed ? 1 : 0, range_max};
            const int group = env_group;  // rounds between two looks at the counter (ranges of one)
            uint32_t changed = 1;
            int r = 0;
            std::string trace;
            while (r < max_rounds && changed) {
                const int g_n = fr.range > 1 ? 1 : group;
                const unsigned n_wg = (unsigned)((nch + (size_t)fr.range - 1) / (size_t)fr.range);
                for (int g = 0; g < g_n && r < max_rounds; g++, r++) {
                    fr.round = r;
                    hipLaunchKernelGGL((zs_fast_sweep_kernel<1024, kFsTile1, true>), dim3(n_wg), dim3(1024), fs_lds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                                       dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), lv, strategy, fr);
                }
                ZS_HIP(c, hipMemcpyAsync(&changed, dev<uint32_t>(c->fr_counters) + (r - 1), 4, hipMemcpyDeviceToHost, stream));
                ZS_HIP(c, hipStreamSynchronize(stream));
                if (env_dbg) {
                    static auto t_last = std::chrono::steady_clock::now();
                    const auto t_now = std::chrono::steady_clock::now();
                    char b[64];
                    snprintf(b, sizeof b, " %u/%d(%.2f)", changed, fr.range, r <= g_n ? 0.0 : std::chrono::duration<double, std::milli>(t_now - t_last).count());
                    t_last = t_now;
                    trace += b;
                }
                if (!env_fixed) {
                    static const double env_wake = getenv("ZS_FR_WAKE") ? atof(getenv("ZS_FR_WAKE")) : 4.0;
                    const size_t awake = std::min<size_t>(nch, (size_t)(env_wake * (double)changed));
                    const int tail = env_tail;
                    fr.range = std::min<int>(range_max, std::max<int>(tail, (int)((awake + 255) / 256)));
                }
            }
            c->fast_rounds = r;
            if (env_dbg) fprintf(stderr, "zs: DeflateFast over %zu chunks of %d streams, up to %d to a workgroup: %d rounds (changed/range:%s)\n", nch, n, range_max, r, trace.c_str());
            if (changed) {
                // (cannot happen: chunk r of a stream is the reference's after round r at the latest.  Should it, the batch takes one
                // workgroup per stream instead of failing the call)
                ZS_HIP(c, hipStreamSynchronize(c->aux));
                c->no_rounds_once = true;
                return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, force_seq, ro, false, force_lit);
            }
            const FsMeta *mf = dev<FsMeta>(c->fr_meta) + (size_t)((r - 1) & 1) * nch;
            hipLaunchKernelGGL(zs_fast_commit_scan_kernel, dim3((unsigned)n), dim3(1024), 0, stream, d_sd, d_st, mf, dev<int32_t>(c->fr_base));
            hipLaunchKernelGGL(zs_fast_commit_kernel, dim3((unsigned)nch), dim3(256), 0, stream, d_sd, fr, mf, dev<int32_t>(c->fr_base), dev<uint16_t>(c->link),
                               dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top));
        } else {
            hipLaunchKernelGGL((zs_fast_sweep_kernel<1024, kFsTile1, false>), dim3((unsigned)n), dim3(1024), fs_lds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                               dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), lv, strategy, FsRounds());
        }
    }
    if (pl.any_rle) {
        // CompressionStrategy.Rle: the body's symbols from the runs of equal bytes (zs_rle.hip), the tail engine behind them
        const size_t nt = (size_t)pl.n_rle_tiles;
        if (!ensure(c, c->rle_tiles, 4 * (4 * nt + (size_t)n) + 256)) return false;
        int32_t *rb = dev<int32_t>(c->rle_tiles);
        RleTiles rt{rb, rb + nt, rb + 2 * nt, rb + 3 * nt, rb + 4 * nt};
        int max_tiles = 0;
        for (int i = 0; i < n; i++)
            if (pl.sd[(size_t)i].rle_end >= 0) max_tiles = std::max(max_tiles, (pl.sd[(size_t)i].rle_end + kMaxMatch + 1 + 4095) / 4096);
        // (a grid's y ends at 65 535: the tile kernels take the streams in slices)
        auto tiles = [&](auto &&launch) {
            for (int s0 = 0; s0 < n; s0 += 65535) launch(dim3((unsigned)((max_tiles + 3) / 4), (unsigned)std::min(n - s0, 65535)), s0);
        };
        tiles([&](dim3 tg, int s0) { hipLaunchKernelGGL(zs_rle_starts_kernel, tg, dim3(256), 0, stream, d_sd, rt, s0); });
        hipLaunchKernelGGL(zs_rle_scan_kernel, dim3((unsigned)n), dim3(1024), 0, stream, d_sd, rt);
        tiles([&](dim3 tg, int s0) {
            hipLaunchKernelGGL(zs_rle_pass_kernel<0>, tg, dim3(256), 0, stream, d_sd, rt, dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), s0);
        });
        hipLaunchKernelGGL(zs_rle_sums_kernel, dim3((unsigned)n), dim3(1024), 0, stream, d_sd, d_st, rt);
        tiles([&](dim3 tg, int s0) {
            hipLaunchKernelGGL(zs_rle_pass_kernel<1>, tg, dim3(256), 0, stream, d_sd, rt, dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), s0);
        });
    }
    // the engine is left for a later run, or took the block in progress over from one: it needs K5's symbols and block ends
    const bool tail_late = ro && (!ro->final_run || ro->resume);
    // Beside the symbol kernel the tails of a few streams are free; those of hundreds are not: 256 tail workgroups of 1024
    // threads and 133 KiB of LDS each, one per CU, cost the symbol kernel of 256 x 1 MiB 3 ms (6.8 against 3.9), more than
    // they take alone (0.5).  From 64 streams of 96 KiB or more on the tails run behind the symbols instead (thousands of
    // small streams are the other way round: 16 rounds of tails, a short symbol kernel -- beside each other 18.9 ms for
    // 4096 x 32 KiB, one after the other 20.1).
    // (1024 x 128 KiB: 2.8 ms one after the other, 3.0 side by side; 4096 x 32 KiB: 5.7 against 4.9)
    const bool tail_serial = !tail_late && n >= 64 && (pl.n_pos / n >= (96 << 10) || getenv("ZS_TAIL_SERIAL")) && !getenv("ZS_TAIL_FORK");
    if (!tail_late && !tail_serial) {
        ZS_HIP(c, hipEventRecord(c->ev_fork, stream));
        ZS_HIP(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        hipLaunchKernelGGL(zs_tail_kernel, dim3((unsigned)n), dim3(1024), kTailLds, c->aux, d_sd, d_st, dev<uint16_t>(c->link),
                           dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks),
                           dev<uint8_t>(c->scratch), c->crc_tab, lv, strategy, hash_variant, level);
        ZS_HIP(c, hipEventRecord(c->ev_join, c->aux));
    }
    if (!pl.w_segs.empty())
        hipLaunchKernelGGL(zs_expand_kernel, dim3((unsigned)((pl.w_segs.size() + 63) / 64)), dim3(64), 0, stream, d_sd, d_st,
                           d_work + o_segs, (int)pl.w_segs.size(), dev<uint2>(c->mm), dev<uint16_t>(c->link),
                           dev<uint32_t>(c->maps), dev<uint16_t>(c->seg_entry), dev<uint32_t>(c->seg_symbase),
                           dev<uint8_t>(c->stale), dev<uint16_t>(c->entry), dev<uint32_t>(c->symbase), c->crc_tab, lv, strategy,
                           hash_variant);
    mark(8);
    if (!pl.w_chunks.empty())
        hipLaunchKernelGGL(zs_emit_syms_lane_kernel<4>, dim3((unsigned)((pl.w_chunks.size() + 63) / 64)), dim3(kK5Threads), 0, stream, d_sd, d_st,
                           d_work + o_chunks, (int)pl.w_chunks.size(), dev<uint2>(c->mm), dev<uint16_t>(c->link), dev<uint16_t>(c->entry),
                           dev<uint32_t>(c->symbase), dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top),
                           c->crc_tab, lv, strategy, hash_variant, k5_ahead);
    mark(9);
    if (tail_serial)
        hipLaunchKernelGGL(zs_tail_kernel, dim3((unsigned)n), dim3(1024), kTailLds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                           dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks),
                           dev<uint8_t>(c->scratch), c->crc_tab, lv, strategy, hash_variant, level);
    else if (!tail_late) ZS_HIP(c, hipStreamWaitEvent(stream, c->ev_join, 0));
    hipLaunchKernelGGL(zs_body_blocks_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, d_st, dev<int32_t>(c->blk_end),
                       dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks), dev<uint32_t>(c->syms));
    if (tail_late)
        hipLaunchKernelGGL(zs_tail_kernel, dim3((unsigned)n), dim3(1024), kTailLds, stream, d_sd, d_st, dev<uint16_t>(c->link),
                           dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks),
                           dev<uint8_t>(c->scratch), c->crc_tab, lv, strategy, hash_variant, level);
    }
    if (pl.n_runs) {
        // DeflateFast by speculative chunk runs; a run whose hand-over state does not verify sends the batch to the
        // sequential engine (the result is the reference's bytes either way)
        if (!getenv("ZS_FAST_NO_PROBE") && force_seq == 0 && !pl.any_dual) {
            // only data that looks periodic is worth the attempt (zs_fast_probe_kernel); anything else goes to the sweeps right away
            hipLaunchKernelGGL(zs_fast_probe_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, dev<uint16_t>(c->link), dev<int32_t>(c->run_fail));
            std::vector<int32_t> np((size_t)n, 0);
            ZS_HIP(c, hipMemcpyAsync(np.data(), c->run_fail.p, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
            ZS_HIP(c, hipStreamSynchronize(stream));
            for (int i = 0; i < n; i++)
                if (np[(size_t)i]) {
                    ZS_HIP(c, hipStreamSynchronize(c->aux));  // (the forked passes read the workspace that is about to be reused)
                    return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, 1, ro, false, force_lit);
                }
        }
        ZS_HIP(c, hipMemsetAsync(c->run_fail.p, 0, 4 * (size_t)n + 64, stream));
        hipLaunchKernelGGL(zs_fast_run_kernel, dim3((unsigned)pl.w_runs.size()), dim3(1024), kFastRunLds, stream, d_sd, d_work + o_runs,
                           dev<uint16_t>(c->link), dev<uint32_t>(c->run_syms), dev<uint32_t>(c->run_bits), dev<uint8_t>(c->run_scratch),
                           dev<FastRunOut>(c->run_outs), c->crc_tab, lv, strategy, hash_variant);
        hipLaunchKernelGGL(zs_fast_verify_kernel, dim3((unsigned)pl.w_runs.size()), dim3(256), 0, stream, d_sd, d_work + o_runs,
                           dev<uint32_t>(c->run_bits), dev<FastRunOut>(c->run_outs), dev<int32_t>(c->run_fail));
        std::vector<int32_t> rfail((size_t)n, 0);
        ZS_HIP(c, hipMemcpyAsync(rfail.data(), c->run_fail.p, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));
        for (int i = 0; i < n; i++)
            if (rfail[(size_t)i]) {
                ZS_HIP(c, hipStreamSynchronize(c->aux));  // the forked tree pass reads the workspace that is about to be reused
                c->fast_fallbacks++;
                std::vector<FastRunOut> outs((size_t)pl.n_runs);
                ZS_HIP(c, hipMemcpy(outs.data(), c->run_outs.p, sizeof(FastRunOut) * outs.size(), hipMemcpyDeviceToHost));
                if (getenv("ZS_DEBUG"))
                    for (int j = 0; j < pl.sd[(size_t)i].fast_runs; j++) {
                        const FastRunOut &o = outs[(size_t)pl.sd[(size_t)i].run_off + j];
                        fprintf(stderr, "zs: stream %d run %d ok=%d mark=%lld (+%lld syms) end=%lld nsyms=%lld n_ev=%d\n", i, j, o.ok,
                                (long long)o.mark_pos, (long long)o.mark_nsyms, (long long)o.end_pos, (long long)o.nsyms, o.n_ev);
                    }
                // Data whose parse does not fall back into step (a period that is no divisor of anything, zeros, 8 KiB rows): the
                // rounds settle one range a round on it -- the stream at one workgroup's 46 / 35 / 20 MB/s -- while one run of
                // the engine over the whole stream costs ~1 us a symbol, and the runs have just counted the symbols: 5 MiB of a
                // 7-byte period is 20 K matches, 20 ms against 375; ptt5 nine times over is 600 K symbols, 600 ms against 53.  One run per
                // stream when that is the shorter way for the batch by a margin (both are estimates).
                double t_engine = 0, t_sweeps = 0;
                for (int k = 0; k < n; k++) {
                    const StreamDesc &sk = pl.sd[(size_t)k];
                    if (sk.fast_runs <= 0) continue;
                    // (a run's warm-up is a fifth of it: the counts of whole runs, scaled)
                    int64_t syms = 0, matches = 0;
                    for (int j = 0; j < sk.fast_runs; j++) syms += outs[(size_t)sk.run_off + j].nsyms, matches += outs[(size_t)sk.run_off + j].n_match;
                    const double scale = sk.fast_runs > 1 ? 0.8 : 1.0;
                    const double te = scale * (1.2e-6 * (double)matches + 0.03e-6 * (double)(syms - matches)) + (double)sk.n / 1.5e9;
                    if (getenv("ZS_DEBUG")) fprintf(stderr, "zs: stream %d: ~%.0f symbols one at a time, ~%.0f in runs of literals, %d bytes\n", k, scale * (double)matches, scale * (double)(syms - matches), sk.n);
                    t_engine = std::max(t_engine, te);
                    t_sweeps = std::max(t_sweeps, (double)sk.n / (level >= 3 ? 20e6 : level == 2 ? 35e6 : 46e6));
                }
                const int mode = (force_seq == 0 && !getenv("ZS_NO_WHOLE_RUNS") && (getenv("ZS_WHOLE_RUNS") || t_engine < 0.7 * t_sweeps)) ? 2 : 1;
                if (getenv("ZS_DEBUG")) fprintf(stderr, "zs: the runs did not verify: %s (engine %.1f ms, sweeps up to %.1f ms)\n", mode == 2 ? "one run per stream" : "the sweeps", t_engine * 1e3, t_sweeps * 1e3);
                return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, mode, ro, false, force_lit);
            }
        hipLaunchKernelGGL(zs_fast_plan_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, d_st, dev<FastRunOut>(c->run_outs), n);
        hipLaunchKernelGGL(zs_fast_stitch_kernel, dim3((unsigned)pl.w_runs.size()), dim3(256), 0, stream, d_sd, d_work + o_runs,
                           dev<uint32_t>(c->run_syms), dev<FastRunOut>(c->run_outs), dev<uint32_t>(c->syms), dev<int32_t>(c->blk_end),
                           dev<int32_t>(c->blk_top));
        hipLaunchKernelGGL(zs_fast_blocks_kernel, dim3((unsigned)n), dim3(64), 0, stream, d_sd, d_st, dev<FastRunOut>(c->run_outs),
                           dev<int32_t>(c->blk_end), dev<int32_t>(c->blk_top), dev<BlockRec>(c->blocks), n);
    }
    mark(10);
    // the tree kernel is one latency chain per block: many small blocks (a batch of small streams) want more resident
    // workgroups, a long stream's 16 Ki-symbol blocks a wider histogram
    const int trees_threads = pl.w_blocks.size() > 8192 ? 128 : 256;
    if (n >= 64 && !getenv("ZS_NO_LIVE_LIST")) {
        // many streams: the block list rewritten with the blocks that exist in front (zs_live_scan_kernel)
        hipLaunchKernelGGL(zs_live_scan_kernel, dim3(1), dim3(1024), 0, stream, d_st, n, dev<int32_t>(c->wpre));
        hipLaunchKernelGGL(zs_live_fill_kernel, dim3((unsigned)((pl.w_blocks.size() + 255) / 256)), dim3(256), 0, stream, dev<int32_t>(c->wpre), n,
                           (uint32_t)pl.w_blocks.size(), dev<uint2>(c->work) + o_blocks);
    }
    hipLaunchKernelGGL(zs_trees_kernel, dim3((unsigned)pl.w_blocks.size()), dim3(trees_threads), 0, stream, d_sd, d_st, d_work + o_blocks,
                       dev<uint32_t>(c->syms), dev<BlockRec>(c->blocks), dev<TreeWork>(c->trees), dev<BlockInfo>(c->info), strategy, level, 2);
    mark(11);
    ZS_HIP(c, hipStreamWaitEvent(stream, c->ev_pre, 0));  // the cleared output and the Adler pieces (second stream, above)
    hipLaunchKernelGGL(zs_offsets_kernel, dim3((unsigned)n), dim3(256), 0, stream, d_sd, d_st, dev<BlockRec>(c->blocks),
                       dev<BlockInfo>(c->info), dev<TreeWork>(c->trees), dev<uint32_t>(c->pieces), level, n);
    mark(12);
    hipLaunchKernelGGL(zs_emit_bits_kernel, dim3((unsigned)pl.w_blocks.size()), dim3(256), 0, stream, d_sd, d_st, d_work + o_blocks,
                       dev<uint32_t>(c->syms), dev<BlockRec>(c->blocks), dev<TreeWork>(c->trees), dev<BlockInfo>(c->info));
    mark(13);
    ZS_HIP(c, hipGetLastError());
    StreamState *hst = (StreamState *)c->pinned;
    ZS_HIP(c, hipMemcpyAsync(hst, d_st, sizeof(StreamState) * (size_t)n, hipMemcpyDeviceToHost, stream));
    ht_launched = ht_us();
    ZS_HIP(c, hipStreamSynchronize(stream));
    if (host_times) fprintf(stderr, "zs: host: plan + uploads %.0f us, all launched at %.0f us, device done at %.0f us\n", ht_plan, ht_launched, ht_us());
    c->last_op = lv.func == 1 ? 2 : 0;
    if (getenv("ZS_DEBUG_CHAIN") && n == 1) {  // the chain of one position as the kernels saw it (buffer positions)
        const int64_t q0 = atoll(getenv("ZS_DEBUG_CHAIN")) - (ro ? ro->abs_off : 0);
        if (q0 > 0 && q0 < in_len[0]) {
            std::vector<uint16_t> lk((size_t)in_len[0]);
            (void)hipMemcpy(lk.data(), dev<uint16_t>(c->link) + pl.sd[0].pos_off, 2 * (size_t)in_len[0], hipMemcpyDeviceToHost);
            fprintf(stderr, "[zs] run of %lld bytes (stream position %lld on), resume %d at %lld: chain of %lld:", (long long)in_len[0], (long long)(ro ? ro->abs_off : 0),
                    pl.sd[0].resume, (long long)pl.sd[0].start_pos, (long long)q0);
            int64_t q = q0;
            int hops = 0;
            const int64_t want = getenv("ZS_DEBUG_CHAIN_WANT") ? atoll(getenv("ZS_DEBUG_CHAIN_WANT")) - (ro ? ro->abs_off : 0) : -1;
            int want_at = -1;
            while (lk[(size_t)q] && hops < 100000) {
                q -= lk[(size_t)q], hops++;
                if (hops <= 12) fprintf(stderr, " %lld", (long long)q);
                if (q == want) want_at = hops;
            }
            fprintf(stderr, " ... %d hops, ends at %lld; position %lld is hop %d\n", hops, (long long)q, (long long)want, want_at);
        }
    }
    if (getenv("ZS_DEBUG_FA") && writes && pl.sd[0].wr_blk) {  // the flush accounting's inputs: blocks flushed before each Write began
        const size_t nw = writes->ends.size();
        std::vector<int32_t> wb(nw);
        (void)hipMemcpy(wb.data(), pl.sd[0].wr_blk, 4 * nw, hipMemcpyDeviceToHost);
        fprintf(stderr, "[zs] run of %lld bytes, %zu Writes, %d blocks, body ends at %d, resume %d; blocks before each Write:", (long long)in_len[0], nw, hst[0].nblocks,
                pl.sd[0].body_end, pl.sd[0].resume);
        for (size_t k = 0; k < nw && k < 24; k++) fprintf(stderr, " %d(end %lld, flush %d)", wb[k], (long long)writes->ends[k], writes->flush.empty() ? 0 : writes->flush[k]);
        fprintf(stderr, "\n");
    }
    if (prof) {
        for (int i = 0; i < kStCount; i++) {
            float ms = 0;
            if (i >= kStLinks) (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]);  // clear / adler: beside the others, unmarked
            c->stage_ms[i] = ms;
        }
        if (n_parts) {  // part-wise launches: every launch has its own pair of events, a stage is the sum of its launches
            for (int i = kStLinks; i <= kStTail; i++) c->stage_ms[i] = 0;
            for (const TimedPair &t : pairs) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, t.a, t.b);
                c->stage_ms[t.stage] += ms;
            }
        }
    }
    {
        // a stream whose true path met a read the bulk form does not handle (zs_core.h kMapPoisonBit): the batch again with
        // those streams on the literal engine
        bool any_poison = false;
        std::vector<uint8_t> fl(force_lit ? *force_lit : std::vector<uint8_t>((size_t)n, 0));
        for (int i = 0; i < n; i++)
            if (hst[i].poison) any_poison = true, fl[(size_t)i] = 1;
        if (any_poison) {
            ZS_HIP(c, hipStreamSynchronize(c->aux));
            if (ro && ro->resume) {  // (the engine's state in `persist` has been run over: the caller has kept a copy)
                c->err = "the resumed run met a read the bulk pipeline leaves to the literal engine";
                c->resume_poisoned = true;
                return false;
            }
            c->lit_fallbacks++;
            return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, force_seq, ro, false, &fl);
        }
    }
    if (!ro && !rounds) {
        // streams the resolve kernel gave up (zs_device.h, StreamState::deferred): the batched cut rounds, then the kernels
        // behind the resolve kernel once more (they skipped those streams): this call again, from there
        bool gave_up = false;
        for (int i = 0; i < n; i++) gave_up = gave_up || hst[i].deferred == 1;
        if (gave_up) {
            ZS_HIP(c, hipStreamSynchronize(c->aux));
            c->round_runs++;
            if (!run_cut_rounds()) return false;
            return run_pipeline(c, n, in, in_len, out, out_cap, out_len, status, level, strategy, hash_variant, stream, writes, force_seq, ro, true, force_lit);
        }
    }
    if (ro) ro->end_bits = hst[0].end_bits;
    // every stream's length and code are reported; the first failing one sets the message and the return value
    bool all_ok = true;
    for (int i = 0; i < n; i++) {
        out_len[i] = hst[i].out_len;
        if (status) status[i] = hst[i].status;
        if (hst[i].status != 0 && all_ok) {
            c->err = hst[i].status == ZS_BUF_ERROR ? "buffer error"
                                                   : "stored block larger than the reference's pending buffer (the managed engine throws here)";
            all_ok = false;
        }
    }
    return all_ok;
}

// Adler-32 (seed 1) of m device buffers in one pass: the piece kernel over all of them, one combining workgroup per
// buffer, one copy back.  `trailers` (optional, device pointers or null entries): the 4 big-endian bytes each result is
// compared with (the check of Inflate.cs:300-345); ok[i] = 1 when they agree.
bool device_adlers(zs_ctx *c, int m, const void *const *bufs, const int64_t *lens, const void *const *trailers, uint32_t *adlers,
                   int *ok, hipStream_t stream) {
    if (m <= 0) return true;
    std::vector<StreamDesc> sd((size_t)m);
    std::vector<uint2> work;
    int64_t n_pieces = 0;
    for (int i = 0; i < m; i++) {
        StreamDesc &s = sd[(size_t)i];
        memset(&s, 0, sizeof s);
        s.in = (const uint8_t *)bufs[i];
        s.n = (int32_t)lens[i];
        s.adler_off = (int32_t)n_pieces;
        s.n_adler = (int32_t)((lens[i] + kAdlerPiece - 1) / kAdlerPiece);
        n_pieces += s.n_adler;
        for (int k = 0; k < s.n_adler; k++) work.push_back(make_uint2((unsigned)i, (unsigned)k));
    }
    const size_t b_sd = sizeof(StreamDesc) * (size_t)m, b_work = sizeof(uint2) * work.size(), b_tr = sizeof(void *) * (size_t)m;
    if (!ensure(c, c->sd, b_sd) || !ensure(c, c->work, b_work + 16) || !ensure(c, c->pieces, 4 * (size_t)n_pieces + 64) ||
        !ensure(c, c->adl_tr, b_tr) || !ensure(c, c->adl_res, sizeof(uint2) * (size_t)m) ||
        !ensure_pinned(c, b_sd + b_work + b_tr + sizeof(uint2) * (size_t)m))
        return false;
    uint8_t *hp = (uint8_t *)c->pinned;
    memcpy(hp, sd.data(), b_sd);
    if (b_work) memcpy(hp + b_sd, work.data(), b_work);
    if (trailers) memcpy(hp + b_sd + b_work, trailers, b_tr);
    ZS_HIP(c, hipMemcpyAsync(c->sd.p, hp, b_sd, hipMemcpyHostToDevice, stream));
    if (b_work) ZS_HIP(c, hipMemcpyAsync(c->work.p, hp + b_sd, b_work, hipMemcpyHostToDevice, stream));
    if (trailers) ZS_HIP(c, hipMemcpyAsync(c->adl_tr.p, hp + b_sd + b_work, b_tr, hipMemcpyHostToDevice, stream));
    if (!work.empty())
        hipLaunchKernelGGL(zs_adler_kernel, dim3((unsigned)work.size()), dim3(256), 0, stream, dev<StreamDesc>(c->sd), dev<uint2>(c->work),
                           dev<uint32_t>(c->pieces));
    hipLaunchKernelGGL(zs_adler_finish_kernel, dim3((unsigned)m), dim3(256), 0, stream, dev<StreamDesc>(c->sd), dev<uint32_t>(c->pieces),
                       trailers ? (const uint8_t *const *)c->adl_tr.p : nullptr, dev<uint2>(c->adl_res));
    uint2 *hres = (uint2 *)(hp + b_sd + b_work + b_tr);
    ZS_HIP(c, hipMemcpyAsync(hres, c->adl_res.p, sizeof(uint2) * (size_t)m, hipMemcpyDeviceToHost, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    for (int i = 0; i < m; i++) {
        if (adlers) adlers[i] = hres[i].x;
        if (ok) ok[i] = (int)hres[i].y;
    }
    return true;
}

// The longest run of streams from `lo` on whose device memory -- per input byte: `per_byte` bytes of workspace (and staging),
// plus the output capacity when it is staged -- fits what the device has free now plus what the context already holds (its
// buffers are reused); at least one stream.
int batch_prefix_that_fits(zs_ctx *c, int n, const int64_t *in_len, const int64_t *out_cap, int lo, int per_byte) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return n;
    const DevBuf *held[] = {&c->link, &c->mm, &c->maps, &c->segmap, &c->supmap, &c->syms, &c->trees, &c->blocks, &c->info, &c->stage_in, &c->stage_out,
                            &c->scratch, &c->ins_bits, &c->mm_bak};
    size_t have = 0;
    for (const DevBuf *b : held) have += b->cap;
    const double budget = 0.55 * ((double)free_b + (double)have);  // (the buffers are grown with an eighth of headroom each)
    double need = 0;
    int hi = lo;
    while (hi < n) {
        const double add = (double)in_len[hi] * per_byte + (per_byte > 19 ? (double)out_cap[hi] : 0.0) + 4.0e6;
        if (hi > lo && need + add > budget) break;
        need += add;
        hi++;
    }
    return hi;
}

bool check_args(zs_ctx *c, int n, const int64_t *in_len, int level, int strategy) {
    if (!c) return false;
    if (n < 0 || level < -1 || level > 9 || strategy < 0 || strategy > 4) {
        c->err = "stream error";
        return false;
    }
    for (int i = 0; i < n; i++)
        if (in_len[i] < 0 || in_len[i] > 0x7FFFFFFF - 1024) {
            c->err = "stream error";
            return false;
        }
    return true;
}

}  // namespace

extern "C" {

void zs_ctx_destroy(zs_ctx *c);
int zs_ctx_create(int device, zs_ctx **out) {
    if (!out) return ZS_STREAM_ERROR;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZS_STREAM_ERROR;
    if (hipSetDevice(device) != hipSuccess) return ZS_STREAM_ERROR;
    zs_ctx *c = new zs_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        zs_ctx_destroy(c);  // releases whatever was created so far
        return ZS_MEM_ERROR;
    }
    for (auto &e : c->ev) (void)hipEventCreate(&e);
    {
        // the link kernel relies on the LDS applying the lanes of one DS_MSKOR_RTN_B32 in lane order: check it here
        int *d_ok = nullptr, ok = 0;
        if (hipMalloc((void **)&d_ok, sizeof(int)) != hipSuccess) {
            zs_ctx_destroy(c);  // releases whatever was created so far
            return ZS_MEM_ERROR;
        }
        hipLaunchKernelGGL(zs_lds_order_kernel, dim3(1), dim3(64), 0, c->stream, d_ok);
        const bool copied = hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                            hipStreamSynchronize(c->stream) == hipSuccess;
        (void)hipFree(d_ok);
        if (!copied || !ok) {
            fprintf(stderr, "zsgpu: LDS lane-order self-test failed on device %d; this build cannot run there\n", device);
            zs_ctx_destroy(c);
            return ZS_STREAM_ERROR;
        }
    }
    if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pre0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming) != hipSuccess) {
        zs_ctx_destroy(c);  // releases whatever was created so far
        return ZS_MEM_ERROR;
    }
    for (auto &e : c->ev_part)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            zs_ctx_destroy(c);
            return ZS_MEM_ERROR;
        }
    std::vector<uint32_t> tab(1024);
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 256; i++) tab[(size_t)t * 256 + i] = crc32c_table_entry(t, (uint32_t)i);
    if (hipMalloc((void **)&c->crc_tab, 4096) != hipSuccess ||
        hipMemcpy(c->crc_tab, tab.data(), 4096, hipMemcpyHostToDevice) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMatchLds + 16) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_resolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kResolveLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_links_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLkLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kTailLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_fast_run_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kFastRunLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_fast_sweep_kernel<1024, kFsTile1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (fs_lds_bytes<1024, kFsTile1>())) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_fast_sweep_kernel<1024, kFsTile1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (fs_lds_bytes<1024, kFsTile1>())) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_inf_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWinMapLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_inf_chain_par_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kChainParLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_inf_expand_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kExpLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_cuts_repair_kernel<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kRepairLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_cuts_repair_kernel<256, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, kRepairLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_cuts_repair_kernel<1024, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, kRepairLds) != hipSuccess ||
        hipFuncSetAttribute((const void *)zs_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kInfLds) != hipSuccess) {
        zs_ctx_destroy(c);
        return ZS_MEM_ERROR;
    }
    // the match kernel addresses its tile from LDS address 0 (lds0_u32 ...): true as long as it has no static LDS
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)zs_match_kernel) != hipSuccess || fa.sharedSizeBytes != 0) {
        fprintf(stderr, "zsgpu: zs_match_kernel has static LDS; its tile addressing assumes none\n");
        zs_ctx_destroy(c);
        return ZS_STREAM_ERROR;
    }
    *out = c;
    return ZS_OK;
}

void zs_ctx_destroy(zs_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    DevBuf *bufs[] = {&c->sd, &c->st, &c->work, &c->wpre, &c->geo, &c->link, &c->mm, &c->maps, &c->chunk_far, &c->segmap, &c->supmap, &c->seg_entry, &c->seg_symbase, &c->seg_stale, &c->entry, &c->symbase, &c->stale, &c->syms,
                      &c->blk_end, &c->blk_top, &c->blocks, &c->trees, &c->info, &c->pieces, &c->scratch, &c->stage_in, &c->stage_out, &c->wr, &c->inf_desc, &c->inf_state, &c->par_ps, &c->par_st, &c->par_work, &c->par_cbits, &c->par_ccnt, &c->par_surv, &c->par_scnt,
                      &c->par_cands, &c->par_tabs, &c->par_toktabs, &c->par_toks, &c->par_ctoks, &c->par_tokstat, &c->par_tails, &c->par_retry, &c->par_fxtab, &c->par_blocks, &c->par_cells, &c->par_windows, &c->par_fail, &c->run_syms, &c->run_bits,
                      &c->run_scratch, &c->run_outs, &c->run_fail, &c->adl_tr, &c->adl_res, &c->plan_blk, &c->ins_bits, &c->mm_bak, &c->cut_pos, &c->cut_bkt, &c->win_groups, &c->win_sg, &c->win_maps, &c->win_entries, &c->persist_bak, &c->resume_flag, &c->rle_tiles, &c->own_in, &c->fr_chunks, &c->fr_meta, &c->fr_planes, &c->fr_prov, &c->fr_base, &c->fr_counters};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    if (c->crc_tab) (void)hipFree(c->crc_tab);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pin_io) (void)hipHostFree(c->pin_io);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_part)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->ev_pool)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_pre0) (void)hipEventDestroy(c->ev_pre0);
    if (c->ev_pre) (void)hipEventDestroy(c->ev_pre);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *zs_ctx_last_error(const zs_ctx *c) { return c ? c->err.c_str() : "no context"; }

int64_t zs_deflate_bound(int64_t n) { return n + (n >> 3) + 1024; }

void zs_ctx_set_profiling(zs_ctx *c, int enable) {
    if (c) c->profiling = enable != 0;
}
int zs_ctx_stage_count(const zs_ctx *) { return kStCount; }
int64_t zs_ctx_counter(const zs_ctx *c, const char *name) {
    if (!c || !name) return -1;
    const std::string k(name);
    if (k == "fast_rounds") return c->fast_rounds;        // rounds of the last call's DeflateFast over its chunks (0: a workgroup per stream)
    if (k == "fast_fallbacks") return c->fast_fallbacks;  // speculative DeflateFast batches redone sequentially
    if (k == "round_runs") return c->round_runs;          // batches with streams in the batched cut rounds
    if (k == "cut_rounds") return c->cut_rounds;
    if (k == "lit_fallbacks") return c->lit_fallbacks;    // batches run again with a stream on the literal engine
    if (k == "lit_engine_bytes") return c->lit_engine_bytes;  // input bytes the one-wave literal engine has parsed (all calls)
    return -1;
}

const char *zs_ctx_stage_name(const zs_ctx *c, int s) {
    static const char *const inf_names[6] = {"inf_find", "inf_measure", "inf_chain", "inf_decode", "inf_windows", "inf_resolve"};
    if (c && c->last_op == 1) return s >= 0 && s < 6 ? inf_names[s] : "";
    // levels 1-3: DeflateFast for the lanes of a wave runs where the lazy parse has its expand stage, and the speculative
    // chunk runs (run / verify / stitch) are timed with the tail engine
    if (c && c->last_op == 2 && s == kStExpand) return "fast_sweep";
    if (c && c->last_op == 2 && s == kStTail) return "fast_runs+tail";
    return s >= 0 && s < kStCount ? kStageNames[s] : "";
}
double zs_ctx_stage_ms(const zs_ctx *c, int s) { return c && s >= 0 && s < kStCount ? c->stage_ms[s] : 0.0; }

int zs_deflate_batch_device(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out,
                            const int64_t *out_cap, int64_t *out_len, int *status, int level, int strategy, int hash_variant,
                            void *hip_stream) {
    if (!check_args(c, n, in_len, level, strategy)) return ZS_STREAM_ERROR;
    if (n == 0) return ZS_OK;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // a batch whose workspace (~17 bytes per input byte) does not fit the device's free memory runs in sub-batches: the
    // streams are independent, the bytes the same
    int rc = ZS_OK;
    for (int lo = 0; lo < n;) {
        const int hi = batch_prefix_that_fits(c, n, in_len, out_cap, lo, 19);
        if (!run_pipeline(c, hi - lo, in + lo, in_len + lo, out + lo, out_cap + lo, out_len + lo, status ? status + lo : nullptr, level, strategy,
                          hash_variant, s) && rc == ZS_OK)
            rc = c->err == "buffer error" ? ZS_BUF_ERROR : ZS_STREAM_ERROR;
        lo = hi;
    }
    return rc;
}

int zs_deflate_writes_device(zs_ctx *c, const void *in, int64_t in_len, const int64_t *write_ends, int64_t n_writes, void *out,
                             int64_t out_cap, int64_t *out_len, int level, int strategy, int hash_variant, void *hip_stream) {
    if (!check_args(c, 1, &in_len, level, strategy)) return ZS_STREAM_ERROR;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    WriteSpec ws;
    int64_t prev = 0;
    for (int64_t i = 0; i < n_writes; i++) {
        if (write_ends[i] < prev || write_ends[i] > in_len) {
            c->err = "stream error";
            return ZS_STREAM_ERROR;
        }
        if (write_ends[i] > prev) ws.ends.push_back(write_ends[i]);  // an empty Write never reaches Deflate (ZlibOutputStream.cs:127-130)
        prev = write_ends[i];
    }
    if (prev != in_len) {
        c->err = "stream error";
        return ZS_STREAM_ERROR;
    }
    ws.flush.assign(ws.ends.size(), 0);
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    int st = 0;
    if (!run_pipeline(c, 1, &in, &in_len, &out, &out_cap, out_len, &st, level, strategy, hash_variant, s, ws.ends.size() > 1 ? &ws : nullptr))
        return c->err == "buffer error" ? ZS_BUF_ERROR : ZS_STREAM_ERROR;
    return ZS_OK;
}

namespace {
int deflate_batch_host_once(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                            int64_t *out_len, int *status, int level, int strategy, int hash_variant) {
    std::vector<const void *> din((size_t)n);
    std::vector<void *> dout((size_t)n);
    std::vector<int64_t> dcap((size_t)n);
    size_t tin = 0, tout = 0;
    for (int i = 0; i < n; i++) {
        tin += ((size_t)in_len[i] + 255) & ~(size_t)255;
        dcap[(size_t)i] = out_cap[i];
        tout += ((size_t)out_cap[i] + 255) & ~(size_t)255;
    }
    auto chk = [&](hipError_t e, const char *what) {
        if (e != hipSuccess) fail(c, what, e);
        return e == hipSuccess;
    };
    if (!ensure(c, c->stage_in, tin + 256) || !ensure(c, c->stage_out, tout + 256)) return ZS_MEM_ERROR;
    size_t oi = 0, oo = 0;
    for (int i = 0; i < n; i++) {
        din[(size_t)i] = (uint8_t *)c->stage_in.p + oi;
        dout[(size_t)i] = (uint8_t *)c->stage_out.p + oo;
        if (in_len[i] &&
            !chk(hipMemcpyAsync((void *)din[(size_t)i], in[i], (size_t)in_len[i], hipMemcpyHostToDevice, c->stream), "H2D"))
            return ZS_STREAM_ERROR;
        oi += ((size_t)in_len[i] + 255) & ~(size_t)255;
        oo += ((size_t)out_cap[i] + 255) & ~(size_t)255;
    }
    std::vector<int> st_local;
    if (!status) st_local.assign((size_t)n, 0), status = st_local.data();
    const bool ok = run_pipeline(c, n, din.data(), in_len, dout.data(), dcap.data(), out_len, status, level, strategy, hash_variant, c->stream);
    const std::string first_err = c->err;
    // streams that succeeded are delivered even when another stream of the batch failed (status[i] says which)
    for (int i = 0; i < n; i++)
        if (status[i] == 0 && out_len[i] > 0 && out_len[i] <= out_cap[i] &&
            !chk(hipMemcpyAsync(out[i], dout[(size_t)i], (size_t)out_len[i], hipMemcpyDeviceToHost, c->stream), "D2H"))
            return ZS_STREAM_ERROR;
    if (!chk(hipStreamSynchronize(c->stream), "sync")) return ZS_STREAM_ERROR;
    if (!ok) {
        c->err = first_err;
        return first_err == "buffer error" ? ZS_BUF_ERROR : ZS_STREAM_ERROR;
    }
    return ZS_OK;
}
}  // namespace

int zs_deflate_batch(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                     int64_t *out_len, int *status, int level, int strategy, int hash_variant) {
    if (!check_args(c, n, in_len, level, strategy)) return ZS_STREAM_ERROR;
    if (n == 0) return ZS_OK;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    // a batch that does not fit the device (staging for the inputs and outputs + ~17 bytes of workspace per input byte) runs
    // in sub-batches, one after the other through the same buffers: the streams are independent, the bytes the same
    int rc = ZS_OK;
    std::string first_err;
    for (int lo = 0; lo < n;) {
        const int hi = batch_prefix_that_fits(c, n, in_len, out_cap, lo, 20);
        const int r = deflate_batch_host_once(c, hi - lo, in + lo, in_len + lo, out + lo, out_cap + lo, out_len + lo, status ? status + lo : nullptr, level,
                                              strategy, hash_variant);
        if (r != ZS_OK && rc == ZS_OK) rc = r, first_err = c->err;
        lo = hi;
    }
    if (rc != ZS_OK) c->err = first_err;
    return rc;
}

int zs_adler32_device(zs_ctx *c, const void *d_buf, int64_t len, uint32_t seed, uint32_t *out, void *hip_stream) {
    if (!c || !out || len < 0 || len > 0x7FFFFFFF - 1024) return ZS_STREAM_ERROR;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    uint32_t ad = 1;
    if (!device_adlers(c, 1, &d_buf, &len, nullptr, &ad, nullptr, s)) return ZS_STREAM_ERROR;
    *out = adler_combine(seed, ad, (uint64_t)len);
    return ZS_OK;
}

}  // extern "C"


// ------------------------------------------------------------------ inflate
namespace {
const char *const kInfMessages[kInfMsgCount] = {
    "", "unknown compression method", "invalid window size", "incorrect header check", "need dictionary", "invalid block type",
    "invalid stored block lengths", "too many length or distance symbols", "invalid bit length repeat",
    "oversubscribed dynamic bit lengths tree", "incomplete dynamic bit lengths tree", "oversubscribed literal/length tree",
    "incomplete literal/length tree", "oversubscribed distance tree", "incomplete distance tree", "empty distance tree with lengths",
    "invalid literal/length code", "invalid distance code", "buffer error", "buffer error", "incorrect data check"};

bool run_inflate_seq(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                 int64_t *out_len, int *status, hipStream_t stream, uint32_t *adler_out = nullptr) {
    std::vector<InfDesc> d((size_t)n);
    for (int i = 0; i < n; i++) d[(size_t)i] = {(const uint8_t *)in[i], (uint8_t *)out[i], in_len[i], out_cap[i]};
    if (!ensure(c, c->inf_desc, sizeof(InfDesc) * (size_t)n) || !ensure(c, c->inf_state, sizeof(InfState) * (size_t)n)) return false;
    std::vector<InfState> st((size_t)n);
    ZS_HIP(c, hipMemcpyAsync(c->inf_desc.p, d.data(), sizeof(InfDesc) * (size_t)n, hipMemcpyHostToDevice, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));  // `d` is pageable host memory
    const bool prof = c->profiling;
    if (prof) (void)hipEventRecord(c->ev[0], stream);
    hipLaunchKernelGGL(zs_inflate_kernel, dim3((unsigned)n), dim3(64), kInfLds, stream, dev<InfDesc>(c->inf_desc),
                       dev<InfState>(c->inf_state));
    if (prof) (void)hipEventRecord(c->ev[1], stream);
    ZS_HIP(c, hipGetLastError());
    ZS_HIP(c, hipMemcpyAsync(st.data(), c->inf_state.p, sizeof(InfState) * (size_t)n, hipMemcpyDeviceToHost, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    if (prof) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
        for (double &v : c->stage_ms) v = 0;
        c->stage_ms[0] = ms;  // reported under stage 0 for inflate calls
    }
    // Adler-32 of the produced bytes against the stored trailer, on the device, all finished streams in one pass
    // (Inflate.cs:300-345)
    std::vector<const void *> abuf;
    std::vector<int64_t> alen;
    std::vector<int> aidx;
    for (int i = 0; i < n; i++)
        if (st[(size_t)i].status == ZS_STREAM_END) abuf.push_back(out[i]), alen.push_back(st[(size_t)i].out_len), aidx.push_back(i);
    if (c->inf_probe == nullptr && n == 1) c->inf_used.assign(1, st[0].status == ZS_STREAM_END ? st[0].in_used : 0);
    std::vector<uint32_t> ads(abuf.size(), 1u);
    if (!device_adlers(c, (int)abuf.size(), abuf.data(), alen.data(), nullptr, ads.data(), nullptr, stream)) return false;
    for (size_t k = 0; k < aidx.size(); k++) {
        InfState &q = st[(size_t)aidx[k]];
        if (ads[k] != q.adler_stored) q.status = ZS_DATA_ERROR, q.msg = kInfBadCheck;
        else if (adler_out) adler_out[aidx[k]] = ads[k];
    }
    bool all_ok = true;
    for (int i = 0; i < n; i++) {
        out_len[i] = st[(size_t)i].out_len;
        const int code = st[(size_t)i].status;
        if (status) status[i] = code;
        if (code != ZS_STREAM_END) {
            if (all_ok) c->err = kInfMessages[st[(size_t)i].msg];
            all_ok = false;
        }
    }
    return all_ok;
}

// One piece of a stream that is decoded as it arrives (zs_inflate): the complete blocks in [start_bit, 8 in_len) of the
// device buffer `in`, behind `hist_have` bytes of history (device), into `out`.  status: ZS_OK -- stopped in front of a
// block that is not complete --, ZS_STREAM_END -- the trailer is behind it (its Adler-32 in *adler_stored; the caller
// keeps the running checksum) --, or the stream's error; good_bits: the bit behind the last complete block.
bool run_inflate_piece(zs_ctx *c, const void *in, int64_t in_len, bool first, int64_t start_bit, const void *hist, int hist_have, void *out, int64_t out_cap,
                       int64_t *out_len, int64_t *good_bits, int64_t *in_used, int *status, uint32_t *adler_stored, hipStream_t stream) {
    InfDesc d{(const uint8_t *)in, (uint8_t *)out, in_len, out_cap, (const uint8_t *)hist, first ? 0 : start_bit, hist_have, first ? 1 : 2};
    if (!ensure(c, c->inf_desc, sizeof(InfDesc)) || !ensure(c, c->inf_state, sizeof(InfState))) return false;
    InfState st;
    ZS_HIP(c, hipMemcpyAsync(c->inf_desc.p, &d, sizeof d, hipMemcpyHostToDevice, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    hipLaunchKernelGGL(zs_inflate_kernel, dim3(1), dim3(64), kInfLds, stream, dev<InfDesc>(c->inf_desc), dev<InfState>(c->inf_state));
    ZS_HIP(c, hipGetLastError());
    ZS_HIP(c, hipMemcpyAsync(&st, c->inf_state.p, sizeof st, hipMemcpyDeviceToHost, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    *out_len = st.out_len, *good_bits = st.good_bits, *in_used = st.in_used, *status = st.status, *adler_stored = st.adler_stored;
    if (st.status != ZS_OK && st.status != ZS_STREAM_END) c->err = kInfMessages[st.msg];
    return true;
}

// shorter streams go straight to the one-wave decoder (4 MB/s: 0.25 ms for a kilobyte of compressed text, what the block-parallel
// pass costs before it has decoded anything; until late in round 5 the line was drawn at 256 KiB -- a 600 KiB text stream took 120 ms,
// it takes 1.05 now, 256 streams of 64 KiB 1.25 instead of 14.2)
constexpr int64_t kParMinInput = 1024;

// Block-parallel path for the streams listed in `idx`; streams it cannot handle are appended to `rest`.
bool run_inflate_par(zs_ctx *c, const std::vector<int> &idx, const void *const *in, const int64_t *in_len, void *const *out,
                     const int64_t *out_cap, int64_t *out_len, int *status, hipStream_t stream, std::vector<int> &rest,
                     uint32_t *adler_out = nullptr) {
    const int m = (int)idx.size();
    if (m == 0) return true;
    std::vector<ParStream> ps((size_t)m);
    int64_t nchunks = 0, ncand = 0, nblk = 0, ncells = 0, nfx = 0;
    std::vector<uint2> w_find;
    for (int j = 0; j < m; j++) {
        const int i = idx[(size_t)j];
        ParStream &p = ps[(size_t)j];
        p.in = (const uint8_t *)in[i], p.out = (uint8_t *)out[i], p.in_len = in_len[i], p.out_cap = out_cap[i];
        p.chunk_off = (int32_t)nchunks, p.nchunks = (int32_t)((in_len[i] + kFindChunk - 1) / kFindChunk);
        nchunks += p.nchunks;
        p.cand_off = (int32_t)ncand, p.max_cand = (int32_t)(in_len[i] / 128 + 64);
        ncand += p.max_cand;
        p.blk_off = (int32_t)nblk, p.max_blk = (int32_t)(in_len[i] / 96 + 64);
        nblk += p.max_blk;
        p.cell_off = ncells;
        p.fx_off = (int32_t)nfx, p.fx_regions = (int32_t)(in_len[i] * 8 / kFxRegionBits + 1);
        nfx += p.fx_regions;
        if (c->inf_probe) p.out_cap = (int64_t)0x7FFFFFFF - 1024;  // a probing call decodes nothing: any output size is fine
        else ncells += (out_cap[i] + 63) & ~63LL;
        for (int k = 0; k < p.nchunks; k++) w_find.push_back(make_uint2((unsigned)j, (unsigned)k));
    }
    if (!ensure(c, c->par_ps, sizeof(ParStream) * (size_t)m) || !ensure(c, c->par_st, sizeof(ParState) * (size_t)m) ||
        !ensure(c, c->par_work, sizeof(uint2) * (size_t)std::max<int64_t>(std::max<int64_t>(nchunks, ncand), nblk) + 64) ||
        !ensure(c, c->par_cbits, 8 * (size_t)nchunks * kFindMaxCand + 64) || !ensure(c, c->par_ccnt, 4 * (size_t)nchunks + 64) ||
        !ensure(c, c->par_surv, 4 * (size_t)nchunks * kFindMaxSurv + 64) || !ensure(c, c->par_scnt, 4 * (size_t)nchunks + 64) ||
        !ensure(c, c->par_cands, sizeof(ParCand) * (size_t)ncand) || !ensure(c, c->par_blocks, sizeof(ParBlock) * (size_t)nblk) ||
        !ensure(c, c->par_cells, 2 * (size_t)ncells + 64) || !ensure(c, c->par_fail, 4 * (size_t)m + 64))
        return false;
    std::vector<ParState> st((size_t)m);
    ZS_HIP(c, hipMemcpyAsync(c->par_ps.p, ps.data(), sizeof(ParStream) * (size_t)m, hipMemcpyHostToDevice, stream));
    ZS_HIP(c, hipMemcpyAsync(c->par_work.p, w_find.data(), sizeof(uint2) * w_find.size(), hipMemcpyHostToDevice, stream));
    ZS_HIP(c, hipMemsetAsync(c->par_fail.p, 0, 4 * (size_t)m + 64, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));  // ps / w_find are pageable
    const ParStream *d_ps = dev<ParStream>(c->par_ps);
    ParState *d_st = dev<ParState>(c->par_st);
    const bool prof = c->profiling;
    auto mark = [&](int k) {
        if (prof) (void)hipEventRecord(c->ev[k], stream);
    };
    mark(0);
    hipLaunchKernelGGL(zs_inf_prefilter_kernel, dim3((unsigned)w_find.size()), dim3(256), 0, stream, d_ps, dev<uint2>(c->par_work),
                       dev<int32_t>(c->par_surv), dev<int32_t>(c->par_scnt));
    hipLaunchKernelGGL(zs_inf_check_kernel, dim3((unsigned)((w_find.size() + 1) / 2)), dim3(64), 0, stream, d_ps, dev<uint2>(c->par_work), (int)w_find.size(),
                       dev<int32_t>(c->par_surv), dev<int32_t>(c->par_scnt), dev<int64_t>(c->par_cbits), dev<int32_t>(c->par_ccnt));
    hipLaunchKernelGGL(zs_inf_flatten_kernel, dim3((unsigned)m), dim3(256), 0, stream, d_ps, d_st, dev<int64_t>(c->par_cbits),
                       dev<int32_t>(c->par_ccnt), dev<ParCand>(c->par_cands), m);
    // decode once (zs_inflate_tok.hip): the measuring pass writes tokens, the expand pass turns them into cells.
    // ZS_INF_LANE_DECODE=1: round 4's form (measure, then decode again lane by lane, then chase the lanes' markers)
    static const bool tok_mode = !getenv("ZS_INF_LANE_DECODE") && !getenv("ZS_INF_WAVE_MEASURE") && !getenv("ZS_INF_WAVE_DECODE");
    int64_t tok_total = 0;
    if (tok_mode) {
        if (!ensure(c, c->par_tokstat, 128) || !ensure(c, c->par_tails, 4 * (size_t)kSbTailBuf * (size_t)m)) return false;
        hipLaunchKernelGGL(zs_inf_tails_kernel, dim3((unsigned)m), dim3(64), 0, stream, d_ps, dev<uint32_t>(c->par_tails));
        if (!c->inf_probe) {  // (a probing call decodes nothing: no slabs, the measuring pass stores no tokens)
            hipLaunchKernelGGL(zs_inf_tokalloc_kernel, dim3((unsigned)m), dim3(1024), 0, stream, d_ps, d_st, dev<ParCand>(c->par_cands));
            hipLaunchKernelGGL(zs_inf_tokbase_kernel, dim3(1), dim3(1024), 0, stream, d_st, m, dev<int64_t>(c->par_tokstat));
            ZS_HIP(c, hipMemcpyAsync(&tok_total, c->par_tokstat.p, 8, hipMemcpyDeviceToHost, stream));
        }
    }
    mark(1);
    ZS_HIP(c, hipMemcpyAsync(st.data(), d_st, sizeof(ParState) * (size_t)m, hipMemcpyDeviceToHost, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    std::vector<uint2> w;
    bool wave_measure = false;
    // (a stream whose candidate lists overflowed -- ok == 0: it goes to the sequential decoder -- is not measured: what
    // its list holds behind the overflow is whatever was in the buffer)
    for (int j = 0; j < m; j++)
        for (int k = 0; st[(size_t)j].ok && k < st[(size_t)j].ncand; k++) w.push_back(make_uint2((unsigned)j, (unsigned)k));
    if (getenv("ZS_DEBUG_INF")) {  // the finder's candidates per stream: count, range, order
        for (int j = 0; j < m; j++) {
            const int nc = st[(size_t)j].ncand;
            std::vector<ParCand> hc((size_t)std::max(nc, 1));
            if (nc) (void)hipMemcpy(hc.data(), dev<ParCand>(c->par_cands) + ps[(size_t)j].cand_off, sizeof(ParCand) * (size_t)nc, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int k = 0; k < nc; k++) bad += hc[(size_t)k].bit < 16 || hc[(size_t)k].bit >= ps[(size_t)j].in_len * 8 || (k && hc[(size_t)k].bit <= hc[(size_t)k - 1].bit);
            fprintf(stderr, "[zs] inflate stream %d: %lld bytes, %d candidates (room for %d), ok %d, out of range or order: %d, first %lld last %lld\n", j,
                    (long long)ps[(size_t)j].in_len, nc, ps[(size_t)j].max_cand, st[(size_t)j].ok, bad, nc ? (long long)hc[0].bit : -1LL,
                    nc ? (long long)hc[(size_t)nc - 1].bit : -1LL);
        }
    }
    if (!w.empty()) {
        ZS_HIP(c, hipMemcpyAsync(c->par_work.p, w.data(), sizeof(uint2) * w.size(), hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));
        // one wave per candidate, its lanes on subsequences of the block (self-synchronising decode); ZS_INF_WAVE_MEASURE
        // selects the plain wave decoder (one dependency chain per block) for comparison
        if (getenv("ZS_INF_WAVE_MEASURE")) {
            wave_measure = true;
            hipLaunchKernelGGL(zs_inf_measure_kernel, dim3((unsigned)w.size()), dim3(64), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                               dev<ParCand>(c->par_cands));
        } else if (tok_mode) {
            if (!ensure(c, c->par_toktabs, sizeof(TokTabs) * w.size()) || !ensure(c, c->par_tabs, sizeof(LaneTabs) * w.size())) return false;
            // the tokens twice (the lanes' slabs, the blocks' lists); without room for them the blocks are decoded again lane by lane
            int mdbg = getenv("ZS_INF_MEASURE_DBG") ? atoi(getenv("ZS_INF_MEASURE_DBG")) : 0;
            // (behind the slabs: room for the blocks that are measured a second time, zs_inf_tokretry_kernel)
            const int64_t tok_reserve = c->inf_probe ? 0 : std::max<int64_t>(tok_total / 16, 2 << 20);
            if (!ensure(c, c->par_toks, 4 * (size_t)(tok_total + tok_reserve) + 64) || !ensure(c, c->par_ctoks, 4 * (size_t)(tok_total + tok_reserve) + 64)) {
                c->err.clear();
                if (!ensure(c, c->par_toks, 64) || !ensure(c, c->par_ctoks, 64)) return false;
                mdbg = 3;
            }
            int32_t *stats = nullptr;
            if (getenv("ZS_DEBUG_INF") && !c->inf_probe) {
                ZS_HIP(c, hipMemsetAsync((uint8_t *)c->par_tokstat.p + 16, 0, 64, stream));
                stats = (int32_t *)((uint8_t *)c->par_tokstat.p + 16);
            }
            hipLaunchKernelGGL(zs_inf_measure_tok_kernel, dim3((unsigned)w.size()), dim3(64), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                               dev<ParCand>(c->par_cands), dev<TokTabs>(c->par_toktabs), dev<LaneTabs>(c->par_tabs), dev<uint32_t>(c->par_toks), dev<uint32_t>(c->par_ctoks), dev<uint32_t>(c->par_tails), stats, mdbg, nullptr, nullptr);
            if (mdbg != 3 && !c->inf_probe) {
                // blocks measured without room for their tokens: once more, with room for their known length
                if (!ensure(c, c->par_retry, 4 * (size_t)kTokRetryMax + 64)) return false;
                uint8_t *rb = (uint8_t *)c->par_retry.p;
                unsigned long long *d_cursor = (unsigned long long *)(rb + 4 * (size_t)kTokRetryMax);
                int32_t *d_rcnt = (int32_t *)(rb + 4 * (size_t)kTokRetryMax + 8);
                ZS_HIP(c, hipMemsetAsync(d_cursor, 0, 16, stream));
                hipLaunchKernelGGL(zs_inf_tokretry_kernel, dim3((unsigned)((w.size() + 255) / 256)), dim3(256), 0, stream, d_ps, dev<uint2>(c->par_work), (int)w.size(),
                                   dev<ParCand>(c->par_cands), tok_total, tok_reserve, d_cursor, d_rcnt, (int32_t *)rb);
                hipLaunchKernelGGL(zs_inf_measure_tok_kernel, dim3((unsigned)std::min<size_t>(w.size(), (size_t)kTokRetryMax)), dim3(64), 0, stream, d_ps, d_st,
                                   dev<uint2>(c->par_work), dev<ParCand>(c->par_cands), dev<TokTabs>(c->par_toktabs), dev<LaneTabs>(c->par_tabs), dev<uint32_t>(c->par_toks),
                                   dev<uint32_t>(c->par_ctoks), dev<uint32_t>(c->par_tails), (int32_t *)nullptr, mdbg, d_rcnt, (int32_t *)rb);
            }
            if (stats) {
                int32_t hs[4] = {0, 0, 0, 0};
                ZS_HIP(c, hipMemcpyAsync(hs, stats, 16, hipMemcpyDeviceToHost, stream));
                ZS_HIP(c, hipStreamSynchronize(stream));
                fprintf(stderr, "[zs] inflate measure: %zu candidates, %d subsequences, %d re-entries met their first decode, %d decoded again in full, %d blocks without tokens; %lld tokens of room\n",
                        w.size(), hs[2], hs[1], hs[0], hs[3], (long long)tok_total);
            }
        } else if (!ensure(c, c->par_tabs, sizeof(LaneTabs) * w.size())) {
            return false;
        } else {
            hipLaunchKernelGGL(zs_inf_measure_sync_kernel, dim3((unsigned)w.size()), dim3(64), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                               dev<ParCand>(c->par_cands), dev<LaneTabs>(c->par_tabs));
        }
    }
    mark(2);
    // measured blocks carry checkpoints: they are decoded by sub-blocks, one lane each
    const bool lane_decode = !w.empty() && !wave_measure && !getenv("ZS_INF_WAVE_DECODE");
    const bool chain_par = !getenv("ZS_INF_CHAIN_WALK");
    if (chain_par)
        hipLaunchKernelGGL(zs_inf_chain_par_kernel, dim3((unsigned)m), dim3(1024), kChainParLds, stream, d_ps, d_st, dev<ParCand>(c->par_cands),
                           dev<ParBlock>(c->par_blocks), lane_decode ? 1 : 0);
    // a stream the finder's blocks do not chain (fixed-code or stored blocks among them): its fixed blocks found ahead of the walk,
    // a wave per 64 KiB (zs_inf_fixed_scan_kernel; the waves of the streams that chained leave at once)
    const FxEntry *d_fx = nullptr;
    static const bool fx_scan = !getenv("ZS_INF_NO_FIXED_SCAN");
    if (chain_par && fx_scan && !c->inf_probe && m <= 65535 && nfx > 0 && nfx < (1 << 22) &&
        ensure(c, c->par_fxtab, sizeof(FxEntry) * (size_t)nfx * kFxEntries + 64)) {
        int max_regions = 1;
        for (int j = 0; j < m; j++) max_regions = std::max(max_regions, (int)ps[(size_t)j].fx_regions);
        d_fx = dev<FxEntry>(c->par_fxtab);
        hipLaunchKernelGGL(zs_inf_fixed_scan_kernel, dim3((unsigned)max_regions, (unsigned)m), dim3(64), 0, stream, d_ps, d_st, dev<FxEntry>(c->par_fxtab));
    }
    hipLaunchKernelGGL(zs_inf_chain_kernel, dim3((unsigned)m), dim3(64), 0, stream, d_ps, d_st, dev<ParCand>(c->par_cands),
                       dev<ParBlock>(c->par_blocks), lane_decode ? 1 : 0, (chain_par ? 1 : 0) | (c->inf_probe ? 2 : 0), d_fx);
    mark(3);
    ZS_HIP(c, hipMemcpyAsync(st.data(), d_st, sizeof(ParState) * (size_t)m, hipMemcpyDeviceToHost, stream));
    ZS_HIP(c, hipStreamSynchronize(stream));
    if (c->inf_probe) {
        // a probing call (zs_inflate: has the stream's end arrived?): the chain of blocks reached a final block and its
        // trailer lies inside the bytes -- or not yet; nothing is decoded
        const int64_t tb = (st[0].end_bit + 7) >> 3;
        *c->inf_probe = (m == 1 && st[0].ok && tb + 4 <= in_len[idx[0]]) ? tb + 4 : 0;
        return true;
    }
    w.clear();
    // windows are laid out by the block counts the chain found (not by the per-stream bounds the other tables use)
    int64_t total_blk = 0;
    for (int j = 0; j < m; j++)
        if (st[(size_t)j].ok) {
            for (int k = 0; k < st[(size_t)j].nblk; k++) w.push_back(make_uint2((unsigned)j, (unsigned)k));
            st[(size_t)j].win_off = (int32_t)total_blk;
            total_blk += st[(size_t)j].nblk;
        }
    std::vector<int32_t> bfail((size_t)m, 0);
    if (!w.empty()) {
        if (total_blk > 0x7FFFFFFF / 2 || !ensure(c, c->par_windows, (size_t)total_blk * kWSize + 64)) {
            // no room for the windows (streams cut into very many small blocks): the one-wave decoder needs none
            c->err.clear();
            for (int j = 0; j < m; j++) rest.push_back(idx[(size_t)j]);
            return true;
        }
        // the window pass runs over groups of blocks (zs_inflate_par.hip, W): enough groups to give every CU one, none
        // shorter than a few blocks
        std::vector<WinGroup> groups;
        std::vector<int2> sgv((size_t)m, make_int2(0, 0));
        int n_ok = 0, n_slots = 0;
        for (int j = 0; j < m; j++) n_ok += st[(size_t)j].ok ? 1 : 0;
        for (int j = 0; j < m; j++) {
            if (!st[(size_t)j].ok) continue;
            const int nb = st[(size_t)j].nblk;
            int G = (256 + n_ok - 1) / n_ok;
            if (G > 16) G = 16;
            if (G > nb / 4) G = nb / 4;
            if (G < 1) G = 1;
            sgv[(size_t)j] = make_int2((int)groups.size(), G);
            for (int k = 0; k < G; k++) {
                const int first = (int)((int64_t)nb * k / G), last = (int)((int64_t)nb * (k + 1) / G);
                groups.push_back(WinGroup{j, first, last - first, k == 0 ? 0 : 1, k == 0 ? 0 : n_slots, 0});
                if (k) n_slots++;
            }
        }
        if (!ensure(c, c->win_groups, sizeof(WinGroup) * groups.size() + 64) || !ensure(c, c->win_sg, sizeof(int2) * (size_t)m + 64) ||
            !ensure(c, c->win_maps, 2 * (size_t)kWSize * (size_t)n_slots + 64) || !ensure(c, c->win_entries, (size_t)kWSize * (size_t)n_slots + 64))
            return false;
        ZS_HIP(c, hipMemcpyAsync(c->win_groups.p, groups.data(), sizeof(WinGroup) * groups.size(), hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipMemcpyAsync(c->win_sg.p, sgv.data(), sizeof(int2) * (size_t)m, hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipMemcpyAsync(d_st, st.data(), sizeof(ParState) * (size_t)m, hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipMemcpyAsync(c->par_work.p, w.data(), sizeof(uint2) * w.size(), hipMemcpyHostToDevice, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));
        bool lane_work = lane_decode && !tok_mode;
        for (int j = 0; j < m && tok_mode; j++) lane_work = lane_work || (st[(size_t)j].ok && st[(size_t)j].lane_blocks);
        if (lane_decode && tok_mode)
            hipLaunchKernelGGL(zs_inf_expand_kernel, dim3((unsigned)w.size()), dim3(kExpThreads), kExpLds, stream, d_ps, d_st, dev<uint2>(c->par_work),
                               dev<ParBlock>(c->par_blocks), dev<TokTabs>(c->par_toktabs), dev<uint32_t>(c->par_ctoks), dev<uint16_t>(c->par_cells),
                               dev<int32_t>(c->par_fail), getenv("ZS_DEBUG_INF") ? (int32_t *)((uint8_t *)c->par_tokstat.p + 16) : nullptr);
        if (lane_work)
            hipLaunchKernelGGL(zs_inf_decode_lane_kernel, dim3((unsigned)((w.size() + kDecBlocks - 1) / kDecBlocks)), dim3(64), 0, stream, d_ps,
                               d_st, dev<uint2>(c->par_work), (int)w.size(), dev<ParBlock>(c->par_blocks), dev<LaneTabs>(c->par_tabs),
                               dev<uint16_t>(c->par_cells), dev<int32_t>(c->par_fail));
        hipLaunchKernelGGL(zs_inf_decode_kernel, dim3((unsigned)w.size()), dim3(64), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                           dev<ParBlock>(c->par_blocks), dev<uint16_t>(c->par_cells), dev<int32_t>(c->par_fail));
        if (lane_work)
            hipLaunchKernelGGL(zs_inf_cellflat_kernel, dim3((unsigned)w.size()), dim3(256), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                               dev<ParBlock>(c->par_blocks), dev<LaneTabs>(c->par_tabs), dev<uint16_t>(c->par_cells));
        mark(4);
        if (!groups.empty()) {
            hipLaunchKernelGGL(zs_inf_window_kernel, dim3((unsigned)groups.size()), dim3(1024), kWinMapLds, stream, d_ps, d_st, dev<WinGroup>(c->win_groups),
                               dev<ParBlock>(c->par_blocks), dev<uint16_t>(c->par_cells), dev<uint8_t>(c->par_windows), dev<uint16_t>(c->win_maps),
                               dev<uint8_t>(c->win_entries), 0);
            if (n_slots) {
                hipLaunchKernelGGL(zs_inf_winchain_kernel, dim3((unsigned)m), dim3(1024), 0, stream, d_st, dev<WinGroup>(c->win_groups), dev<int2>(c->win_sg),
                                   dev<uint8_t>(c->par_windows), dev<uint16_t>(c->win_maps), dev<uint8_t>(c->win_entries));
                hipLaunchKernelGGL(zs_inf_window_kernel, dim3((unsigned)groups.size()), dim3(1024), kWinMapLds, stream, d_ps, d_st,
                                   dev<WinGroup>(c->win_groups), dev<ParBlock>(c->par_blocks), dev<uint16_t>(c->par_cells), dev<uint8_t>(c->par_windows),
                                   dev<uint16_t>(c->win_maps), dev<uint8_t>(c->win_entries), 1);
            }
        }
        mark(5);
        hipLaunchKernelGGL(zs_inf_resolve_kernel, dim3((unsigned)w.size()), dim3(256), 0, stream, d_ps, d_st, dev<uint2>(c->par_work),
                           dev<ParBlock>(c->par_blocks), dev<uint16_t>(c->par_cells), dev<uint8_t>(c->par_windows));
        mark(6);
        ZS_HIP(c, hipGetLastError());
        ZS_HIP(c, hipMemcpyAsync(bfail.data(), c->par_fail.p, 4 * (size_t)m, hipMemcpyDeviceToHost, stream));
        ZS_HIP(c, hipStreamSynchronize(stream));
        if (tok_mode && getenv("ZS_DEBUG_INF")) {
            int32_t hs[16] = {};
            (void)hipMemcpy(hs, (uint8_t *)c->par_tokstat.p + 16, 64, hipMemcpyDeviceToHost);
            fprintf(stderr, "[zs] inflate expand: %d blocks with tokens, %d steps, %d rounds of pointer jumping; cycles / 256 of wave 0 in phases: scan %d, barrier %d, fetch+words %d, cells %d, rounds %d, out %d, barrier %d\n",
                    hs[6], hs[4], hs[5], hs[8], hs[9], hs[10], hs[11], hs[12], hs[13], hs[14]);
        }
        if (prof) {
            c->last_op = 1;
            for (double &v : c->stage_ms) v = 0;
            for (int k = 0; k < 6; k++) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]);
                c->stage_ms[k] = ms;  // stages 0..5 of an inflate call: find, measure, chain, decode, windows, resolve
            }
        }
    }
    // Adler-32 trailer after the last block, byte aligned (Inflate.cs:300-345): all streams in one device pass
    std::vector<const void *> abuf, atr;
    std::vector<int64_t> alen;
    std::vector<int> aidx;
    for (int j = 0; j < m; j++) {
        const int i = idx[(size_t)j];
        const ParState &q = st[(size_t)j];
        const int64_t tb = (q.end_bit + 7) >> 3;
        if (!q.ok || bfail[(size_t)j] || tb + 4 > in_len[i]) {
            rest.push_back(i);  // not decodable here, or truncated: the sequential decoder classifies it
            continue;
        }
        abuf.push_back(out[i]), alen.push_back(q.out_len), atr.push_back((const uint8_t *)in[i] + tb), aidx.push_back(j);
    }
    std::vector<uint32_t> ads(abuf.size(), 1u);
    std::vector<int> aok(abuf.size(), 0);
    if (!device_adlers(c, (int)abuf.size(), abuf.data(), alen.data(), atr.data(), ads.data(), aok.data(), stream)) return false;
    for (size_t k = 0; k < aidx.size(); k++) {
        const int j = aidx[k], i = idx[(size_t)j];
        out_len[i] = st[(size_t)j].out_len;
        status[i] = ZS_STREAM_END;
        if (m == 1) c->inf_used.assign(1, ((st[(size_t)j].end_bit + 7) >> 3) + 4);
        if (adler_out) adler_out[i] = ads[k];
        if (!aok[k]) {
            status[i] = ZS_DATA_ERROR;
            c->err = kInfMessages[kInfBadCheck];
        }
    }
    return true;
}

bool run_inflate(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                 int64_t *out_len, int *status, hipStream_t stream, uint32_t *adler_out = nullptr) {
    std::vector<int> par, seq;
    static const int64_t par_min = getenv("ZS_INF_PAR_MIN") ? atoll(getenv("ZS_INF_PAR_MIN")) : kParMinInput;
    for (int i = 0; i < n; i++) (in_len[i] >= par_min ? par : seq).push_back(i);
    std::vector<int> st((size_t)n, 0);
    if (!run_inflate_par(c, par, in, in_len, out, out_cap, out_len, st.data(), stream, seq, adler_out)) return false;
    bool ok = true;
    for (int i : par)
        if (std::find(seq.begin(), seq.end(), i) == seq.end() && st[(size_t)i] != ZS_STREAM_END) ok = false;
    if (!seq.empty()) {
        const int m = (int)seq.size();
        std::vector<const void *> sin((size_t)m);
        std::vector<void *> sout((size_t)m);
        std::vector<int64_t> slen((size_t)m), scap((size_t)m), solen((size_t)m);
        std::vector<int> sst((size_t)m);
        std::vector<uint32_t> sad((size_t)m, 1u);
        for (int j = 0; j < m; j++) sin[(size_t)j] = in[seq[(size_t)j]], sout[(size_t)j] = out[seq[(size_t)j]], slen[(size_t)j] = in_len[seq[(size_t)j]], scap[(size_t)j] = out_cap[seq[(size_t)j]];
        if (!run_inflate_seq(c, m, sin.data(), slen.data(), sout.data(), scap.data(), solen.data(), sst.data(), stream, sad.data())) ok = false;
        for (int j = 0; j < m; j++) {
            out_len[seq[(size_t)j]] = solen[(size_t)j], st[(size_t)seq[(size_t)j]] = sst[(size_t)j];
            if (adler_out) adler_out[seq[(size_t)j]] = sad[(size_t)j];
        }
    }
    if (status) memcpy(status, st.data(), sizeof(int) * (size_t)n);
    return ok;
}
}  // namespace

extern "C" {

int zs_inflate_batch_device(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out,
                            const int64_t *out_cap, int64_t *out_len, int *status, void *hip_stream) {
    if (!c || n < 0) return ZS_STREAM_ERROR;
    if (n == 0) return ZS_OK;
    // what the caller sees if anything fails before the results are known (as run_pipeline does for deflate)
    for (int i = 0; i < n; i++) {
        out_len[i] = 0;
        if (status) status[i] = ZS_STREAM_ERROR;
    }
    for (int i = 0; i < n; i++)
        if (in_len[i] < 0 || in_len[i] > 0x7FFFFFFF - 1024 || out_cap[i] < 0 || out_cap[i] > 0x7FFFFFFF - 1024) {
            c->err = "stream error";
            return ZS_STREAM_ERROR;
        }
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    std::vector<int> st((size_t)n, ZS_STREAM_ERROR);
    bool ok = run_inflate(c, n, in, in_len, out, out_cap, out_len, st.data(), s);
    if (status) memcpy(status, st.data(), sizeof(int) * (size_t)n);
    if (ok) return ZS_OK;
    for (int v : st)
        if (v != ZS_STREAM_END) return v == ZS_OK ? ZS_STREAM_ERROR : v;
    return ZS_STREAM_ERROR;
}

int zs_inflate_batch(zs_ctx *c, int n, const void *const *in, const int64_t *in_len, void *const *out, const int64_t *out_cap,
                     int64_t *out_len, int *status) {
    if (!c || n < 0) return ZS_STREAM_ERROR;
    if (n == 0) return ZS_OK;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    size_t tin = 0, tout = 0;
    for (int i = 0; i < n; i++) {
        out_len[i] = 0;
        if (status) status[i] = ZS_STREAM_ERROR;
    }
    for (int i = 0; i < n; i++) {
        if (in_len[i] < 0 || in_len[i] > 0x7FFFFFFF - 1024 || out_cap[i] < 0 || out_cap[i] > 0x7FFFFFFF - 1024) {
            c->err = "stream error";
            return ZS_STREAM_ERROR;
        }
        tin += ((size_t)in_len[i] + 255) & ~(size_t)255, tout += ((size_t)out_cap[i] + 255) & ~(size_t)255;
    }
    if (!ensure(c, c->stage_in, tin + 256) || !ensure(c, c->stage_out, tout + 256)) return ZS_MEM_ERROR;
    std::vector<const void *> din((size_t)n);
    std::vector<void *> dout((size_t)n);
    size_t oi = 0, oo = 0;
    for (int i = 0; i < n; i++) {
        din[(size_t)i] = (uint8_t *)c->stage_in.p + oi;
        dout[(size_t)i] = (uint8_t *)c->stage_out.p + oo;
        if (in_len[i] && hipMemcpyAsync((void *)din[(size_t)i], in[i], (size_t)in_len[i], hipMemcpyHostToDevice, c->stream) != hipSuccess)
            return ZS_STREAM_ERROR;
        oi += ((size_t)in_len[i] + 255) & ~(size_t)255;
        oo += ((size_t)out_cap[i] + 255) & ~(size_t)255;
    }
    std::vector<int> st((size_t)n, ZS_STREAM_ERROR);
    int rc = zs_inflate_batch_device(c, n, din.data(), in_len, dout.data(), out_cap, out_len, st.data(), c->stream);
    if (status) memcpy(status, st.data(), sizeof(int) * (size_t)n);
    // only what a stream that ended cleanly produced goes back, and never more than the caller's buffer holds
    for (int i = 0; i < n; i++)
        if (st[(size_t)i] == ZS_STREAM_END && out_len[i] > 0 && out_len[i] <= out_cap[i] &&
            hipMemcpyAsync(out[i], dout[(size_t)i], (size_t)out_len[i], hipMemcpyDeviceToHost, c->stream) != hipSuccess)
            return ZS_STREAM_ERROR;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return ZS_STREAM_ERROR;
    return rc;
}

}  // extern "C"

// ------------------------------------------------------------------ PNG scanline filtering (the caller path of the sparse case)
extern "C" int zs_png_filter_device(zs_ctx *c, const void *pixels, int64_t row_bytes, int64_t height, int bpp, int filter, void *out,
                                    void *hip_stream) {
    if (!c || !pixels || !out || row_bytes <= 0 || height <= 0 || bpp < 1 || bpp > 8 || filter < 0 || filter > 5 || height > 0x7FFFFFFF)
        return ZS_STREAM_ERROR;
    if (hipSetDevice(c->device) != hipSuccess) return ZS_STREAM_ERROR;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    hipLaunchKernelGGL(zs_png_filter_kernel, dim3((unsigned)height), dim3(256), 0, s, (const uint8_t *)pixels, row_bytes, height, bpp, filter,
                       (uint8_t *)out);
    if (hipGetLastError() != hipSuccess) return ZS_STREAM_ERROR;
    return hip_stream ? ZS_OK : (hipStreamSynchronize(s) == hipSuccess ? ZS_OK : ZS_STREAM_ERROR);
}

// ------------------------------------------------------------------ multi-GPU batch entry points
// BASELINE north_star: "independent input buffers shard embarrassingly across the 8 GPUs of one node (no RCCL needed)".
// The unit of sharding is the buffer (a zlib stream cannot be split bit-exactly: 32 KiB history, sequential parse,
// bit-contiguous blocks).  Longest-processing-time partition by size, one host thread and one context per device, results
// in input order; no collective, no peer traffic.
extern "C" {

int zs_device_count(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

int zs_partition(const int64_t *sizes, int n, int n_parts, int *part_of) {
    if (!sizes || !part_of || n < 0 || n_parts <= 0) return ZS_STREAM_ERROR;
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] > sizes[b]; });  // ties keep input order
    std::vector<int64_t> load((size_t)n_parts, 0);
    for (int i : order) {
        int best = 0;
        for (int k = 1; k < n_parts; k++)
            if (load[(size_t)k] < load[(size_t)best]) best = k;  // ties go to the lowest part
        part_of[i] = best;
        load[(size_t)best] += sizes[i];
    }
    return ZS_OK;
}

}  // extern "C"

namespace {
template <class Fn>  // Fn(ctx, count, in, in_len, out, out_cap, out_len, status) -> code
int run_multi(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
              const int64_t *out_cap, int64_t *out_len, int *status, const int64_t *weights, const int *part_of, Fn fn) {
    if (!ctxs || n_ctx <= 0 || n < 0) return ZS_STREAM_ERROR;
    for (int k = 0; k < n_ctx; k++)
        if (!ctxs[k]) return ZS_STREAM_ERROR;
    // a context is one workspace and one pair of HIP streams: two host threads in it at once would corrupt both shares
    for (int k = 0; k < n_ctx; k++)
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) {
                ctxs[k]->err = "stream error: the same context passed twice";
                return ZS_STREAM_ERROR;
            }
    if (n == 0) return ZS_OK;
    std::vector<int> part((size_t)n);
    if (part_of) {
        for (int i = 0; i < n; i++) {
            if (part_of[i] < 0 || part_of[i] >= n_ctx) return ZS_STREAM_ERROR;
            part[(size_t)i] = part_of[i];
        }
    } else if (zs_partition(weights, n, n_ctx, part.data()) != ZS_OK) return ZS_STREAM_ERROR;
    std::vector<std::vector<int>> idx((size_t)n_ctx);
    for (int i = 0; i < n; i++) idx[(size_t)part[(size_t)i]].push_back(i);
    std::vector<int> rc((size_t)n_ctx, ZS_OK);
    std::vector<std::thread> th;
    for (int k = 0; k < n_ctx; k++) {
        if (idx[(size_t)k].empty()) continue;
        th.emplace_back([&, k]() {
            const std::vector<int> &ix = idx[(size_t)k];
            const size_t m = ix.size();
            std::vector<const void *> sin(m);
            std::vector<void *> sout(m);
            std::vector<int64_t> slen(m), scap(m), solen(m, 0);
            std::vector<int> sst(m, ZS_STREAM_ERROR);
            for (size_t j = 0; j < m; j++) sin[j] = in[ix[j]], sout[j] = out[ix[j]], slen[j] = in_len[ix[j]], scap[j] = out_cap[ix[j]];
            rc[(size_t)k] = fn(ctxs[k], (int)m, sin.data(), slen.data(), sout.data(), scap.data(), solen.data(), sst.data());
            for (size_t j = 0; j < m; j++) {
                out_len[ix[j]] = solen[j];
                if (status) status[ix[j]] = sst[j];
            }
        });
    }
    for (std::thread &t : th) t.join();
    for (int k = 0; k < n_ctx; k++)
        if (rc[(size_t)k] != ZS_OK) return rc[(size_t)k];
    return ZS_OK;
}
}  // namespace

extern "C" {

int zs_deflate_batch_multi(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                           const int64_t *out_cap, int64_t *out_len, int *status, int level, int strategy, int hash_variant) {
    return run_multi(ctxs, n_ctx, n, in, in_len, out, out_cap, out_len, status, in_len, nullptr,
                     [=](zs_ctx *c, int m, const void *const *i, const int64_t *il, void *const *o, const int64_t *oc, int64_t *ol, int *st) {
                         return zs_deflate_batch(c, m, i, il, o, oc, ol, st, level, strategy, hash_variant);
                     });
}

int zs_deflate_batch_multi_device(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                                  const int64_t *out_cap, int64_t *out_len, int *status, const int *part_of, int level, int strategy,
                                  int hash_variant) {
    if (!part_of) return ZS_STREAM_ERROR;
    return run_multi(ctxs, n_ctx, n, in, in_len, out, out_cap, out_len, status, in_len, part_of,
                     [=](zs_ctx *c, int m, const void *const *i, const int64_t *il, void *const *o, const int64_t *oc, int64_t *ol, int *st) {
                         return zs_deflate_batch_device(c, m, i, il, o, oc, ol, st, level, strategy, hash_variant, nullptr);
                     });
}

int zs_inflate_batch_multi(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                           const int64_t *out_cap, int64_t *out_len, int *status) {
    // balanced by output capacity: the decoded size is what an inflate costs
    return run_multi(ctxs, n_ctx, n, in, in_len, out, out_cap, out_len, status, out_cap, nullptr,
                     [](zs_ctx *c, int m, const void *const *i, const int64_t *il, void *const *o, const int64_t *oc, int64_t *ol, int *st) {
                         return zs_inflate_batch(c, m, i, il, o, oc, ol, st);
                     });
}

// ... and over device pointers: every stream and its output already resident on the GPU of the context `part_of` names (the
// caller partitions with zs_partition -- by the decoded sizes -- and places them), so the data path of N GPUs has no PCIe in it
int zs_inflate_batch_multi_device(zs_ctx *const *ctxs, int n_ctx, int n, const void *const *in, const int64_t *in_len, void *const *out,
                                  const int64_t *out_cap, int64_t *out_len, int *status, const int *part_of) {
    if (!part_of) return ZS_STREAM_ERROR;
    return run_multi(ctxs, n_ctx, n, in, in_len, out, out_cap, out_len, status, out_cap, part_of,
                     [](zs_ctx *c, int m, const void *const *i, const int64_t *il, void *const *o, const int64_t *oc, int64_t *ol, int *st) {
                         return zs_inflate_batch_device(c, m, i, il, o, oc, ol, st, nullptr);
                     });
}

}  // extern "C"

#include "zs_stream_api.inc"
