// zs_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// deflate engine.  Integer / byte work only: no MFMA.  See DESIGN.md for the
// pipeline; every kernel names the reference code whose result it reproduces.
//
//   K0 zs_clear_kernel     zero the output buffers (bit emission ORs into them)
//   K1 zs_links_kernel     per-position bucket links  (InsertString, Deflate.cs:866-877)
//   K2 zs_match_kernel     per-position Longest_match for both chain budgets (Deflate.cs:1022-1100)
//   K3 zs_chunkmap_kernel  lazy-parse transfer maps per 2 Ki-position chunk (Deflate.Slow.cs:34-145)
//   K4 zs_resolve_kernel   compose the maps along the true path, refill quirk fix-ups (Deflate.cs:1010-1013)
//   K5 zs_emit_syms_lane_kernel symbols + block cuts along the true path (Tr_tally_*, Deflate.cs:910-948)
//   K6 zs_tail_kernel      last <= 261 bytes by the literal engine (zs_lit_engine.h)
//   K7 zs_trees_kernel     per-block histograms + exact Build_tree replay (Trees.cs:404-643)
//   K8 zs_offsets_kernel   block bit offsets, zlib header / Adler trailer (Deflate.cs:464-493,627-635)
//   K9 zs_emit_bits_kernel Huffman bit packing (Compress_block / Send_bits, Trees.cs:956-989, Deflate.cs:799-821)
//   KA zs_adler_kernel     Adler-32 per 64 KiB piece (Adler32.cs:270-326)
#include <hip/hip_runtime.h>

#include <type_traits>

#include "zs_device.h"

namespace zs {

typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
// The stream buffers are reached through pointers stored in StreamDesc, which the compiler has to treat as generic
// addresses: flat loads, which count on the LDS counter as well and so make every wait for an LDS read a wait for
// all outstanding global loads.  They are global memory: say so.
typedef const __attribute__((address_space(1))) uint8_t *gcbytes;
typedef __attribute__((address_space(1))) uint8_t *gbytes_w;
__device__ __forceinline__ gcbytes as_global(const uint8_t *p) { return (gcbytes)(uintptr_t)p; }
__device__ __forceinline__ gbytes_w as_global(uint8_t *p) { return (gbytes_w)(uintptr_t)p; }
typedef const __attribute__((address_space(1))) uint32_t __attribute__((aligned(1))) *gcu32u;
typedef const __attribute__((address_space(1))) uint64_t __attribute__((aligned(1))) *gcu64u;
// (Only for addresses that differ from lane to lane: where the compiler can prove an address wave-uniform it makes the
// load a scalar one, and a scalar dword load reads from the address rounded down to 4 whatever the type says -- found in
// zs_repair_kernel.  Uniform or possibly uniform addresses take g_u32_bytes.)
__device__ __forceinline__ uint32_t g_u32_bytes(gcbytes p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4 *gcu32x4;
// Unaligned 4 / 8 bytes out of an LDS byte array (16-byte aligned base, readable one dword past the last index
// used): aligned dword reads + v_alignbyte.  A ds_read whose address is not a multiple of its size is legal but
// the LDS pipe handles the misaligned lanes of a wave one at a time.
__device__ __forceinline__ uint32_t lds_u32(const uint8_t *base, int idx) {
    const uint32_t *q = (const uint32_t *)(base + (idx & ~3));
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)idx & 3u);
}
__device__ __forceinline__ uint64_t lds_u64(const uint8_t *base, int idx) {
    const uint32_t *q = (const uint32_t *)(base + (idx & ~3));
    const uint32_t a = q[0], b = q[1], c = q[2], sh = (uint32_t)idx & 3u;
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, sh) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, sh) << 32);
}

// The same for a byte array that starts at LDS address 0, with the index as the address: no base to add (K2's tile;
// zs_match_kernel has no static LDS, so its dynamic allocation starts at 0 -- zs_ctx_create checks that).  v_alignbyte
// looks at the low two bits of its shift operand only, so the index goes in as it is.
typedef const __attribute__((address_space(3))) uint32_t *lds_cu32;
typedef const __attribute__((address_space(3))) uint16_t *lds_cu16;
__device__ __forceinline__ uint32_t lds0_u32(int idx) {
    lds_cu32 q = (lds_cu32)(uintptr_t)(uint32_t)(idx & ~3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)idx);
}
__device__ __forceinline__ uint64_t lds0_u64(int idx) {
    lds_cu32 q = (lds_cu32)(uintptr_t)(uint32_t)(idx & ~3);
    const uint32_t a = q[0], b = q[1], c = q[2];
    return (uint64_t)__builtin_amdgcn_alignbyte(b, a, (uint32_t)idx) | ((uint64_t)__builtin_amdgcn_alignbyte(c, b, (uint32_t)idx) << 32);
}
__device__ __forceinline__ int lds0_link(int idx) { return *(lds_cu16)(uintptr_t)(uint32_t)(kMatchLdsBytes + 2 * idx); }

// Match records in `mm` (uint2 per position): x = record for budget K, y = for budget K >> 2, each dist | (len-3) << 16
// (zs_core.h pack_match); bits 24..31 of x carry the input byte of the position, so that the symbol kernel gets its
// literals with the records it reads anyway.
constexpr uint32_t kRecMask = 0x00FFFFFFu;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ uint32_t dev_bucket(const uint32_t *tab, uint32_t v, int hash_variant) {
    return (hash_variant == kHashMul ? hash_mul(v) : crc32c_u32_tab(tab, v)) & kHashMask;
}
__device__ __forceinline__ void load_crc_tab(uint32_t *tab, const uint32_t *g) {
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = g[i];
}

// ------------------------------------------------------------------ work items
// Every kernel below takes (stream, index) pairs from a list; list l holds, stream after stream, the indices 0 .. count - 1
// of that stream.  pre[l * (n + 1) + i] = items of list l that belong to the streams before i.
__global__ __launch_bounds__(256) void zs_worklist_kernel(const int32_t *pre, int n, WorkOffsets wo, uint2 *work) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= wo.off[9]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 9; k++) l += g >= wo.off[k] ? 1 : 0;
    const int32_t j = (int32_t)(g - wo.off[l]);
    const int32_t *p = pre + (size_t)l * (size_t)(n + 1);
    int lo = 0, hi = n - 1;  // the last stream whose first item is <= j (streams without items share their successor's start)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (p[mid] <= j) lo = mid;
        else hi = mid - 1;
    }
    work[g] = make_uint2((uint32_t)lo, (uint32_t)(j - p[lo]));
}

// Up to four regions (16-byte multiples) zeroed in one launch: the per-run flags and state.
struct ZeroRegions {
    void *p[4];
    uint32_t n16[4];  // 16-byte units
};
__global__ __launch_bounds__(256) void zs_zero_kernel(ZeroRegions z) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
#pragma unroll
    for (int r = 0; r < 4; r++)
        for (uint32_t i = g; i < z.n16[r]; i += stride) ((uint4 *)z.p[r])[i] = make_uint4(0, 0, 0, 0);
}

// ------------------------------------------------------------------ K0
__global__ void zs_clear_kernel(const StreamDesc *sd, const uint2 *work) {
    uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    int64_t beg = (int64_t)w.y * 65536, end = beg + 65536;
    if (end > s.out_cap) end = s.out_cap;
    uint8_t *o = s.out;
    // 16-byte stores on the aligned interior, bytes at the edges
    int64_t a = ((uintptr_t)(o + beg) + 15) & ~(uintptr_t)15;
    int64_t ab = a - (int64_t)(uintptr_t)o;
    if (ab > end) ab = end;
    for (int64_t i = beg + threadIdx.x; i < ab; i += blockDim.x) o[i] = 0;
    int64_t nvec = (end - ab) / 16;
    uint4 *v = (uint4 *)(o + ab);
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) v[i] = make_uint4(0, 0, 0, 0);
    for (int64_t i = ab + nvec * 16 + threadIdx.x; i < end; i += blockDim.x) o[i] = 0;
}

// DS_MSKOR_RTN_B32: word = (word & ~mask) | bits, returns the old word (an atomic exchange of part of a dword).
__device__ __forceinline__ uint32_t lds_mskor_rtn(uint32_t *lds_word, uint32_t mask, uint32_t bits) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds_word;
    uint32_t old;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(mask), "v"(bits) : "memory");
    return old;
}
// Self-test of the property K1 builds on: the 64 lanes of one DS_MSKOR_RTN_B32 are applied in lane order, whether they
// hit the same half of a word, the other half, or other words.  ok[0] = 1 when every lane saw its predecessor.
__global__ __launch_bounds__(64) void zs_lds_order_kernel(int *ok) {
    __shared__ uint32_t w[64];
    const int lane = threadIdx.x;
    w[lane] = 0xFFFFFFFFu;
    __syncthreads();
    bool good = true;
    for (int pat = 0; pat < 4; pat++) {
        // pattern 0: all lanes one half-word; 1: alternate halves of one word; 2: groups of 4 lanes per half-word;
        // 3: lane-dependent scatter over 8 half-words
        const uint32_t slot = pat == 0 ? 0u : pat == 1 ? (uint32_t)(lane & 1) : pat == 2 ? (uint32_t)(lane >> 2) : (uint32_t)((lane * 5) & 7);
        const uint32_t sh = (slot & 1u) * 16u, val = (uint32_t)(pat * 64 + lane + 1);
        const uint32_t oldw = lds_mskor_rtn(&w[8 * pat + (slot >> 1)], 0xFFFFu << sh, val << sh);
        const uint32_t got = (oldw >> sh) & 0xFFFFu;
        // expected: the value of the nearest lower lane with the same slot, else the initial 0xFFFF
        uint32_t want = 0xFFFFu;
        for (int l = 0; l < lane; l++) {
            const uint32_t sl = pat == 0 ? 0u : pat == 1 ? (uint32_t)(l & 1) : pat == 2 ? (uint32_t)(l >> 2) : (uint32_t)((l * 5) & 7);
            if (sl == slot) want = (uint32_t)(pat * 64 + l + 1);
        }
        good = good && got == want;
    }
    // the returning add K1 ranks positions with: lanes sharing a counter must get consecutive values in lane order
    __shared__ uint32_t ctr[4];
    if (lane < 4) ctr[lane] = 0;
    __syncthreads();
    const uint32_t rank = atomicAdd(&ctr[lane & 3], 1u);
    good = good && rank == (uint32_t)(lane >> 2);
    const uint64_t all = __ballot(good);
    if (lane == 0) ok[0] = all == ~0ull ? 1 : 0;
}

// ------------------------------------------------------------------ K1
// InsertString replayed in absolute coordinates: link[q] = distance from q to the previous position of q's
// bucket (0 if none within kMaxDist).  The replay is sequential only within a bucket, so a 1024-thread workgroup
// takes its span 16 Ki positions at a time and splits every tile by bucket class (h & 15): phase 1 hashes the
// tile and ranks every position inside its (wave, class) cell with ballots, phase 2 turns the 16 x 16 cell counts
// into offsets, phase 3 scatters (h, index) into 16 position-ordered class lists in LDS, and in phase 4 wave k
// replays list k against the shared 32 Ki-entry head table, 64 entries per step, with one returning masked-OR per
// lane (phase 4 below).  The head table lives in LDS for the whole span (entries are positions relative to a base that
// moves with the tile, re-based by 16 Ki per tile like SlideHash), after 32 Ki positions of warm-up before the span.
constexpr int kLkTile = 16384;
constexpr int kLkWarm = 32768;  // >= kMaxDist, multiple of kLkTile
constexpr int kLkLds = 2 * kHashSize + 4 * kLkTile + 4096 + 2 * 16 * 16 * 2 + 64 * 4;
__global__ __launch_bounds__(1024) void zs_links_kernel(const StreamDesc *sd, const uint2 *work, uint16_t *link,
                                                        const uint32_t *crc_tab_g, int hash_variant, int span) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint16_t *head = (uint16_t *)smem;                         // kHashSize, 0xFFFF = empty
    uint32_t *list = (uint32_t *)(smem + 2 * kHashSize);       // kLkTile: (h << 16) | index in tile, grouped by class
    uint32_t *tab = list + kLkTile;                            // 1024
    uint16_t *coff = (uint16_t *)(tab + 1024) + 256;           // [class][wave] start of the cell in `list`
    uint32_t *cstart = (uint32_t *)(coff + 256);               // [17] start of each class list
    __shared__ uint32_t wsum4[4];
    __shared__ uint32_t cell[256];  // [class][wave] positions counted so far in the tile
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const gcbytes in = as_global(s.in);
    const int64_t qend = (int64_t)s.n - 5;  // positions q with q + 5 < n are inserted
    const int64_t span0 = (int64_t)w.y * span;
    if (span0 >= qend) return;
    int64_t span_end = span0 + span;
    if (span_end > qend) span_end = qend;
    int64_t w0 = span0 - kLkWarm;
    if (w0 < 0) w0 = 0;
    for (int i = tid; i < kHashSize / 2; i += 1024) ((uint32_t *)head)[i] = 0xFFFFFFFFu;
    if (tid < 256) cell[tid] = 0;
    load_crc_tab(tab, crc_tab_g);
    __syncthreads();
    uint16_t *lk = link + s.pos_off;
    constexpr uint32_t kRel0 = 32768;  // position t0 + i is stored as kRel0 + i
    for (int64_t t0 = w0; t0 < span_end; t0 += kLkTile) {
        if (t0 != w0) {
            // re-base the table by one tile; entries further back than 32 Ki positions drop out
            for (int i = tid; i < kHashSize / 2; i += 1024) {
                const uint32_t v = ((uint32_t *)head)[i];
                uint32_t lo = v & 0xFFFFu, hi = v >> 16;
                lo = (lo == 0xFFFFu || lo < (uint32_t)kLkTile) ? 0xFFFFu : lo - kLkTile;
                hi = (hi == 0xFFFFu || hi < (uint32_t)kLkTile) ? 0xFFFFu : hi - kLkTile;
                ((uint32_t *)head)[i] = lo | (hi << 16);
            }
        }
        int64_t tend = t0 + kLkTile;
        if (tend > span_end) tend = span_end;
        const int tlen = (int)(tend - t0);
        // ---- phase 1: buckets, and the rank of every position inside its (wave, class) cell: one returning LDS add per
        //      position on the cell's counter (lane order again: ranks follow the positions)
        uint32_t hh[16];  // bucket, 0xFFFFFFFF when the position is past the end
        uint16_t rk[16];
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int idx = wave * 1024 + g * 64 + lane;
            uint32_t h = 0xFFFFFFFFu, r = 0;
            if (idx < tlen) {
                h = dev_bucket(tab, *(gcu32u)(in + t0 + idx + 2), hash_variant);
                r = atomicAdd(&cell[(h & 15u) * 16 + wave], 1u);
            }
            hh[g] = h;
            rk[g] = (uint16_t)r;
        }
        __syncthreads();
        // ---- phase 2: cell offsets = exclusive scan of the 256 counts in class-major order
        if (tid < 256) {
            const uint32_t cnt = cell[tid];
            cell[tid] = 0;  // for the next tile
            uint32_t inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t x = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= o) inc += x;
            }
            if (lane == 63) wsum4[wave] = inc;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k < wave) before += wsum4[k];
            const uint32_t off = before + inc - cnt;
            coff[tid] = (uint16_t)off;
            if ((tid & 15) == 0) cstart[tid >> 4] = off;
            if (tid == 255) cstart[16] = off + cnt;
        } else {
            __syncthreads();
        }
        __syncthreads();
        // ---- phase 3: scatter into the class lists
#pragma unroll
        for (int g = 0; g < 16; g++) {
            if (hh[g] != 0xFFFFFFFFu) {
                const uint32_t h = hh[g];
                list[(uint32_t)coff[(h & 15u) * 16 + wave] + rk[g]] = (h << 16) | (uint32_t)(wave * 1024 + g * 64 + lane);
            }
        }
        __syncthreads();
        // ---- phase 4: wave k replays class k, 64 entries per step.  One DS_MSKOR_RTN_B32 per step writes the lane's
        //      position into its bucket's 16-bit half of the head word and returns what was there: the LDS applies the
        //      lanes of one instruction in lane order, so a lane gets the position left by the nearest lower lane of the
        //      same bucket, or the head from earlier steps -- exactly InsertString's `prev[str] = head[h]; head[h] = str`
        //      in stream order, with no separate pass for buckets hit several times in one step.  (zs_ctx_create checks
        //      the lane order on the device with zs_lds_order_kernel.)
        {
            const uint32_t cb = cstart[wave], ce = cstart[wave + 1];
            const bool store = t0 >= span0;
            for (uint32_t i0 = cb; i0 < ce; i0 += 64) {
                const uint32_t i = i0 + (uint32_t)lane;
                if (i < ce) {
                    const uint32_t e = list[i];
                    const uint32_t h = e >> 16, idx = e & 0x3FFFu;
                    const uint32_t rel = kRel0 + idx, sh = (h & 1u) * 16u;
                    const uint32_t oldw = lds_mskor_rtn((uint32_t *)head + (h >> 1), 0xFFFFu << sh, rel << sh);
                    const uint32_t prevrel = (oldw >> sh) & 0xFFFFu;
                    if (store) {
                        uint32_t d = prevrel != 0xFFFFu ? rel - prevrel : 0;
                        if (d > (uint32_t)kMaxDist) d = 0;
                        lk[t0 + idx] = (uint16_t)d;
                    }
                }
            }
        }
        __syncthreads();
    }
    // the positions behind the last inserted one have no link: the stream's last span writes those zeros (the array is not
    // cleared as a whole: every other entry is written above)
    if (span_end == qend) {
        const int64_t n_pad = ((int64_t)s.n + 64 + 63) & ~63LL;
        for (int64_t q = qend + tid; q < n_pad; q += 1024) lk[q] = 0;
    }
}

// Every lane owns one walk at a time, in one of four states:
//   1 stepping   at a candidate that has not been looked at yet
//   2 comparing  the candidate passed the 4-byte prefilter and needs a real compare (cl bytes done so far)
//   3 finished   result to be stored, then the lane pulls the next position from the wave cursor
//   0 idle       the cursor is exhausted
// The kernel is bound by vector-instruction issue (a wave64 instruction holds its SIMD for 4 cycles), and a wave
// pays for every path any of its lanes takes.  The common step (read the link and 4 prefilter bytes of the
// candidate, move on) is ~20 instructions without branches; the compare and finish paths are several times
// that, and with 64 walks per wave some lane needs one of them in nearly every round.  So the stepping phase
// keeps going with the lanes still in state 1 and lets the others wait until half as many lanes wait as step
// (nwait * kWaitNum >= nact * kWaitDen), and only then runs the compare and finish phases once for all of them.
#ifndef ZS_WQ
#define ZS_WQ 2
#endif
#ifndef ZS_WD
#define ZS_WD 1
#endif
#ifndef ZS_RUN
#define ZS_RUN 2
#endif
#ifndef ZS_STEP_UNROLL
#define ZS_STEP_UNROLL 4
#endif
// The walk of one tile by one wave's lanes (zs_match_kernel below).  RUNS: positions are handed out in runs of ZS_RUN
// per lane instead of one at a time, and the match found at p bounds what the same distance gives at p + 1 (the hint).
template <bool RUNS>
__device__ __forceinline__ void match_walk(const uint8_t *wb, const uint16_t *wl, uint2 *om, int64_t lo, int wendi, int *wg_cursor,
                                           int K, int K4, int nice, int lane) {
    constexpr int kWaitNum = ZS_WQ, kWaitDen = ZS_WD;
    constexpr int kRun = RUNS ? ZS_RUN : 1;
    int st = 3, p = -1, c = 8, best = 2, bdist = 0, n_eval = 0, cl = 0;
    uint32_t snap = 0;       // record for the K>>2 budget once it is known to differ from the final one, else 0
    int snapped = 0;
    uint32_t scan_end = 0, mask = 0;  // scan_end = bytes p+best-3 .. p+best; mask drops the byte before p when best == 2
    uint32_t fmask = 0;               // the candidate's first bytes must match too: 3 of them while best == 2, then 4
    uint32_t sc0 = 0, sc1 = 0;        // bytes p .. p+7 (the first 8 bytes of every compare)
    int run_next = 0, run_end = 0;    // the lane's run of positions: next one to take, end
    int hint_l = 0, hint_d = 0;       // match found at the position just before p (0: none, or p does not follow it)
    // after a finished compare of `len` bytes against candidate c: take the improvement, count the candidate, move on
    auto after_compare = [&](int len) {
        len = len > kMaxMatch ? kMaxMatch : len;
        const int better = len > best;
        // the record for budget K>>2 is the state after candidate number K>>2: an improvement by a later candidate
        // freezes the record as it was
        const int freeze = better & (n_eval >= K4) & !snapped;
        snap = freeze ? (best >= kMinMatch ? pack_match(best, bdist) : kNoMatch) : snap;
        snapped |= freeze;
        best = better ? len : best;
        bdist = better ? p - c : bdist;
        if (better) scan_end = lds0_u32(p + len - 3);
        mask = better ? 0xFFFFFFFFu : mask;
        fmask = better ? 0xFFFFFFFFu : fmask;
        const int nice_hit = better & (len >= nice);
        cl = 0;
        n_eval++;
        const int nc = c - lds0_link(c);
        const int stop = nice_hit | (n_eval >= K) | (p - nc >= kMaxDist);
        c = stop ? c : nc;
        st = stop ? 3 : 1;
    };
    for (;;) {
        // ---- finish phase: store results, take new positions (written with selects: every lane runs every line).
        // A lane takes runs of kRun consecutive positions: one counter update per run, and the match found at p bounds
        // what the same distance gives at p + 1 from below (hint_l - 1 equal bytes are known before any compare) -- the
        // long matches of image rows and runs are then compared over their last bytes only.
        if (st == 3 && p >= 0) {
            const uint32_t rec = best >= kMinMatch ? pack_match(best, bdist) : kNoMatch;
            om[(int64_t)p + lo] = make_uint2(rec | (sc0 << 24), snapped ? snap : rec);  // sc0's low byte is the byte at p
            if constexpr (RUNS) hint_l = best >= kMinMatch ? best : 0, hint_d = bdist;
        }
        p = st == 3 ? -1 : p;
        if constexpr (!RUNS) {
            // one position per lane and visit, neighbouring lanes taking neighbouring positions
            const uint64_t need = __ballot(st == 3);
            if (need) {
                int base = 0;
                if (lane == 0) base = atomicAdd(wg_cursor, (int)__builtin_popcountll(need));
                base = __builtin_amdgcn_readfirstlane(base);
                const int mine = base + __builtin_popcountll(need & lanemask_lt());
                const bool take = st == 3 && mine < wendi, dry = st == 3 && mine >= wendi;
                const int q = take ? mine : 8;  // lanes that take nothing read an in-range dummy
                const int l = lds0_link(q);
                const uint64_t first8 = lds0_u64(q);
                const bool has = l != 0xFFFF;  // link distances are already <= kMaxDist
                // a position without a usable link is done at once; its lane takes its next one at the next visit (an
                // immediate second pull cost more than the idle lane)
                if (take && !has) om[(int64_t)mine + lo] = make_uint2((uint32_t)first8 << 24, kNoMatch);
                p = take && has ? mine : p;
                c = take ? mine - (has ? l : 0) : c;
                st = take ? (has ? 1 : 3) : (dry ? 0 : st);
                best = take ? 2 : best, bdist = take ? 0 : bdist, n_eval = take ? 0 : n_eval, snapped = take ? 0 : snapped;
                cl = take ? 0 : cl;
                scan_end = take ? (uint32_t)first8 << 8 : scan_end;  // bytes p-1 .. p+2; the mask drops the byte before p
                mask = take ? 0xFFFFFF00u : mask;
                fmask = take ? 0x00FFFFFFu : fmask;
                sc0 = take ? (uint32_t)first8 : sc0, sc1 = take ? (uint32_t)(first8 >> 32) : sc1;
            }
        } else if (__ballot(st == 3)) {
            const bool cont = st == 3 && run_next < run_end;  // the next position of the lane's own run
            const uint64_t need = __ballot(st == 3 && !cont);
            int base = 0;
            if (need) {
                if (lane == 0) base = atomicAdd(wg_cursor, kRun * (int)__builtin_popcountll(need));
                base = __builtin_amdgcn_readfirstlane(base);
            }
            const int fresh = base + kRun * (int)__builtin_popcountll(need & lanemask_lt());
            const int mine = cont ? run_next : fresh;
            const bool take = st == 3 && mine < wendi, dry = st == 3 && mine >= wendi;
            if (st == 3 && !cont) run_end = fresh + kRun < wendi ? fresh + kRun : wendi, hint_l = 0;
            run_next = st == 3 ? mine + 1 : run_next;
            const int q = take ? mine : 8;  // lanes that take nothing read an in-range dummy
            const int l = lds0_link(q);
            const uint64_t first8 = lds0_u64(q);
            const bool has = l != 0xFFFF;  // link distances are already <= kMaxDist
            // a position without a usable link is done at once; its lane takes its next one at the next visit
            if (take && !has) om[(int64_t)mine + lo] = make_uint2((uint32_t)first8 << 24, kNoMatch);
            hint_l = take && !has ? 0 : hint_l;
            p = take && has ? mine : p;
            c = take ? mine - (has ? l : 0) : c;
            st = take ? (has ? 1 : 3) : (dry ? 0 : st);
            best = take ? 2 : best, bdist = take ? 0 : bdist, n_eval = take ? 0 : n_eval, snapped = take ? 0 : snapped;
            cl = take ? 0 : cl;
            scan_end = take ? (uint32_t)first8 << 8 : scan_end;  // bytes p-1 .. p+2; the mask drops the byte before p
            mask = take ? 0xFFFFFF00u : mask;
            fmask = take ? 0x00FFFFFFu : fmask;
            sc0 = take ? (uint32_t)first8 : sc0, sc1 = take ? (uint32_t)(first8 >> 32) : sc1;
        }
        if (!__ballot(st != 0)) break;
        // ---- stepping phase
        for (;;) {
            const int nact = __builtin_popcountll(__ballot(st == 1));
            const int nwait = __builtin_popcountll(__ballot(st >= 2));
            if (nact == 0 || nwait * kWaitNum >= nact * kWaitDen) break;
#pragma unroll
            for (int u = 0; u < ZS_STEP_UNROLL; u++) {  // steps per look at the lane counts
                // Only the lanes in state 1 run the step: the others' stale candidates would take part in the LDS bank conflicts
                // (35 LDS cycles per wave-step with all 64 lanes reading, SQ_LDS_IDX_ACTIVE: 2.4 of the kernel's 3.0 ms; masked,
                // 2.75 ms).  From here neither fewer vector instructions per step (lanes leaving through the execution mask
                // instead of selects: -5 of 21) nor fewer LDS reads (none for `e` while best <= 3) moved the time: DESIGN.md.
                if (st == 1) {
                    const int l = lds0_link(c);
                    // a candidate can only beat `best` if bytes [best-3 .. best] match too (bytes [0 .. 2] when best == 2) -- and
                    // its first bytes: the chain is keyed on bytes 2 .. 5, so most candidates that agree with the scan around
                    // `best` differ from it in bytes 0 and 1, and each of those would cost a visit to the compare phase
                    const uint32_t e = lds0_u32(c + best - 3), f = lds0_u32(c);
                    const int pass = (((e ^ scan_end) & mask) | ((f ^ sc0) & fmask)) == 0;
                    // leave the candidate: count it and follow its link; `cur_match > limit` is distance < kMaxDist
                    const int ne = n_eval + 1, nc = c - l;
                    const int stop = (ne >= K) | (p - nc >= kMaxDist);
                    n_eval = pass ? n_eval : ne;
                    c = (pass | stop) ? c : nc;
                    st = pass ? 2 : (stop ? 3 : 1);
                }
            }
        }
        // ---- compare phase.  First 8 bytes of every compare against the cached bytes of p (most end here) ...
        if (st == 2 && cl == 0) {
            if (RUNS && hint_l > 16 && p - c == hint_d) {
                cl = (hint_l - 1) & ~7;  // the same distance matched hint_l bytes one position earlier
            } else {
                const uint64_t x = lds0_u64(c) ^ ((uint64_t)sc0 | ((uint64_t)sc1 << 32));
                if (x) after_compare((int)(__builtin_ctzll(x) >> 3));
                else cl = 8;
            }
        }
        // ... then up to 32 more per visit for the lanes inside a long match (skipped by the wave when there is none)
        for (int r = 0; r < 4 && __ballot(st == 2 && cl != 0); r++) {
            if (st == 2 && cl != 0) {
                const uint64_t x = lds0_u64(p + cl) ^ lds0_u64(c + cl);
                if (x) {
                    after_compare(cl + (int)(__builtin_ctzll(x) >> 3));
                } else {
                    cl += 8;
                    if (cl >= kMaxMatch) after_compare(kMaxMatch);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ K1r: a resumed run's chains
// A run that takes a stream over in the middle (StreamDesc::resume) finds the chains as the engine before it left them in
// `persist`: the last 32 Ki positions below p0 with everything that happened to them -- the equal-bucket events' cuts
// (Deflate.cs:1009-1012 inserting strstart + 1 ahead of strstart), the last positions in front of a flush, which went in
// under hashes that read window bytes behind the data or not at all (Deflate.Slow.cs:58,121-129), the heads a FullFlush
// forgot (Deflate.cs:596-604) -- none of which the data alone tells K1.  So below p0 the links are the engine's prev[], and
// a position from p0 on whose bucket has no member in [p0, q) gets the engine's head of the bucket: for the walks of the
// run the chains are then the reference's own.  Workgroup b takes positions p0 - 32768 + 1024 b ...; 64 workgroups.
__global__ __launch_bounds__(1024) void zs_import_chains_kernel(const StreamDesc *sd, int stream_idx, uint16_t *link, const uint32_t *crc_tab_g,
                                                                int hash_variant, int64_t p0) {
    __shared__ uint32_t tab[1024];
    const StreamDesc s = sd[stream_idx];
    const LitPersist *ps = s.persist;
    load_crc_tab(tab, crc_tab_g);
    __syncthreads();
    uint16_t *lk = link + s.pos_off;
    const int64_t wbase = ps->base - s.persist_off;  // buffer position of window[0]
    const int64_t q = p0 - kWSize + (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (q < wbase || q < 0) return;
    const int idx = (int)(q - wbase);  // (behind p0 it may lie beyond the window the engine left: a distance is all it is used for)
    if (q < p0) {
        // (an entry above its own index: the position ahead of an equal-bucket event's loop-top -- the walk would go round
        // between the two until its budget ends and find nothing new: the chain ends there)
        const int pv = ps->prev[idx & kWMask];
        int d = (pv != 0 && pv < idx) ? idx - pv : 0;
        if (d > kMaxDist) d = 0;
        lk[q] = (uint16_t)d;
        return;
    }
    if (q + 5 >= (int64_t)s.n) return;
    const int have = lk[q];
    if (have != 0 && q - have >= p0) return;  // the bucket's latest member is one of the run's own positions
    const uint32_t h = dev_bucket(tab, g_u32_bytes(as_global(s.in) + q + 2), hash_variant);
    const int hv = ps->head[h];
    int d = hv != 0 ? idx - hv : 0;
    if (d < 0 || d > kMaxDist) d = 0;
    lk[q] = (uint16_t)d;
}

// Can the chains of the suspended engine be written as links at all?  A link is a distance back; the engine's prev[] has an
// entry that points *forward* where a read inserted strstart + 1 ahead of strstart into one bucket (prev[s] = s + 1).  When
// s + 1 is inserted again at its own loop-top the two close a cycle, which a walk leaves only by running out of budget: a
// cut, and a link of 0 says that.  But a Write of one or two bytes behind a flush never reaches that second insert
// (lookahead < MIN_MATCH), and the walk goes on from s + 1 into the older chain -- a path links cannot hold: such a stream
// (seen once in 12 000 random ones) stays with the literal engine.  flag[0] = 1 then.
__global__ __launch_bounds__(1024) void zs_resume_check_kernel(const LitPersist *ps, int64_t p0_abs, int *flag) {
    const int64_t q = p0_abs - kWSize + (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (q < ps->base || q < 0 || q >= p0_abs) return;
    const int idx = (int)(q - ps->base);
    const int pv = ps->prev[idx & kWMask];
    if (pv > idx && pv < kWindowSize && ps->prev[pv & kWMask] != idx && pv - idx < kWSize) flag[0] = 1;
}

// ------------------------------------------------------------------ K2
// 1024 threads per 16 Ki-position tile; the tile's 48 KiB of input and 96 KiB
// of links are staged in LDS once, then every lane walks hash chains for one
// position at a time, pulling the next position from a per-wave cursor as soon
// as its walk ends (lanes of a wave have very different chain lengths).
// Per main-loop iteration a lane does one unit of work: test a candidate and
// compare its first 8 bytes, or compare 8 more bytes of a long match.
__global__ __launch_bounds__(1024) void zs_match_kernel(const StreamDesc *sd, const uint2 *work, const uint16_t *link,
                                                        uint2 *mm, LevelCfg lv, int strategy) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *wb = smem;
    uint16_t *wl = (uint16_t *)(smem + kMatchLdsBytes);
    uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int64_t t0 = (int64_t)w.y * kMatchTile;
    const int64_t n = s.n;
    if (t0 > s.body_end) return;
    // the two counters live behind the tile in the dynamic allocation: with no static LDS the tile starts at LDS address 0
    // and the byte / link addresses of the walk need no base added (two vector instructions per candidate step)
    int &wg_same = *(int *)(smem + kMatchLds), &wg_cursor = *(int *)(smem + kMatchLds + 4);
    if (threadIdx.x == 0) wg_same = 0;
    __syncthreads();
    const int64_t lo = t0 - kMatchBack;
    const gcbytes in = as_global(s.in);
    // ---- stage bytes (dword granularity, zero outside [0, n)) ----
    {
        // 16 bytes per lane (lo is a multiple of 16; the caller's buffer and the link array are 16-byte aligned
        // in the common case), scalar fallback at the edges of the stream
        const bool aligned = (((uintptr_t)in) & 15) == 0;
        for (int i = threadIdx.x; i < kMatchLdsBytes / 16; i += 1024) {
            int64_t a = lo + (int64_t)i * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a >= 0 && a + 15 < n && aligned) {
                const u32x4 t = *(gcu32x4)(in + a);
                v = make_uint4(t[0], t[1], t[2], t[3]);
            } else if (a + 15 >= 0 && a < n) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; k++) {
                    int64_t b = a + k;
                    if (b >= 0 && b < n) t[k >> 2] |= (uint32_t)in[b] << (8 * (k & 3));
                }
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            ((uint4 *)wb)[i] = v;
        }
        const uint16_t *lk = link + s.pos_off;  // pos_off is a multiple of 64 and the array 16-byte aligned
        int n_same = 0;
        for (int i = threadIdx.x; i < kMatchLdsLinks / 8; i += 1024) {
            int64_t a = lo + (int64_t)i * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a >= 0 && a + 7 < n) {
                v = *(const uint4 *)(lk + a);
            } else if (a + 7 >= 0 && a < n) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (int k = 0; k < 8; k++) {
                    int64_t b = a + k;
                    if (b >= 0 && b < n) t[k >> 1] |= (uint32_t)lk[b] << (16 * (k & 1));
                }
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            // eight positions in a row whose previous occurrence lies at one and the same distance: the inside of a long
            // match (image rows, runs); counted to choose how positions are handed out (below)
            n_same += v.x != 0 && v.x == v.y && v.y == v.z && v.z == v.w && (v.x >> 16) == (v.x & 0xFFFFu);
            // LDS form of a link: 0xFFFF = none (a step over it lands beyond kMaxDist, so the walk needs no
            // separate test); a link onto position 0 is none too (Longest_match never visits position 0)
            uint32_t t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t nz = (((t[k] & 0x7FFF7FFFu) + 0x7FFF7FFFu) | t[k]) & 0x80008000u;  // bit 15 of each non-zero half
                t[k] |= ((nz ^ 0x80008000u) >> 15) * 0xFFFFu;
            }
            if (a <= kMaxDist && a + 7 >= 1) {
                for (int k = 0; k < 8; k++) {
                    const uint32_t d = (t[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                    if ((int64_t)d == a + k) t[k >> 1] |= 0xFFFFu << (16 * (k & 1));
                }
            }
            ((uint4 *)wl)[i] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        if (n_same) atomicAdd(&wg_same, n_same);
    }
    __syncthreads();

    // ---- walk ----
    int64_t pbeg = t0 < 1 ? 1 : t0;
    int64_t pend = t0 + kMatchTile;
    if (pend > (int64_t)s.body_end + 1) pend = (int64_t)s.body_end + 1;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    // one position cursor for the workgroup (an LDS counter): waves that run out of long walks early keep pulling, so
    // the idle tail is that of the tile, not of 16 separate ranges
    if (threadIdx.x == 0) wg_cursor = (int)(pbeg - lo);
    __syncthreads();
    const int wendi = (int)(pend - lo);  // LDS-relative end of the tile's positions
    if (t0 == 0 && threadIdx.x == 0) mm[s.pos_off] = make_uint2((uint32_t)wb[0 - lo] << 24, kNoMatch);  // position 0 is never searched
    uint2 *om = mm + s.pos_off;
    const int K = lv.chain, K4 = lv.chain >> 2, nice = lv.nice;

    // Positions are handed out one at a time -- neighbouring lanes then walk neighbouring positions, whose candidates are
    // neighbours too: their LDS reads fall into the same words -- unless the tile is mostly the inside of long matches
    // (image rows, runs): then in runs per lane, for the sake of the hint.  Two instances of the walk, so that a text tile
    // pays nothing for the other kind.
    if (wg_same * 4 > kMatchLdsLinks / 8) match_walk<true>(wb, wl, om, lo, wendi, &wg_cursor, K, K4, nice, lane);
    else match_walk<false>(wb, wl, om, lo, wendi, &wg_cursor, K, K4, nice, lane);
}

// ------------------------------------------------------------------ parse-segment tables (StreamDesc)
__device__ __forceinline__ int seg_first(const StreamDesc &s, int seg) { return seg < s.nsegs ? s.seg_c0[seg] : s.nchunks; }
__device__ __forceinline__ int seg_of(const StreamDesc &s, int c) {  // last segment whose first chunk is <= c
    int lo = 0, hi = s.nsegs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s.seg_c0[mid] <= c) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// chunk c of stream s as the walkers of zs_core.h see it (geometry and read-event cluster from the host's tables)
__device__ __forceinline__ ChunkCtx chunk_ctx(const StreamDesc &s, int c) {
    ChunkCtx cx;
    int h;
    if (s.grid_chunks) {
        // one Write: chunk c >= 1 begins at 2048 c - 261, and the segments are those of the window ends (65536 + 32768 k
        // - 261 = the beginning of chunk 32 + 16 k); the host has checked both against its tables (zs_engine.hip) -- a
        // workgroup's first loads would otherwise be a round trip for these three values before it can ask for anything else
        cx.cs = c ? (int64_t)c * kChunk - (kMinLookahead - 1) : 0;
        cx.ce = c + 1 < s.nchunks ? (int64_t)(c + 1) * kChunk - (kMinLookahead - 1) : (int64_t)s.body_end + 1;
        h = (c >= 32 && ((c - 32) & 15) == 0) ? ((c - 32) >> 4) + 2 : 0;
    } else {
        cx.cs = s.cstart[c], cx.ce = s.cstart[c + 1];
        h = s.head[c];
    }
    if (cx.ce > (int64_t)s.body_end + 1) cx.ce = (int64_t)s.body_end + 1;
    cx.cl = nullptr, cx.m = 0, cx.S = 0, cx.after = 0;
    if (h) {
        const int k = h - 1;
        const int o = s.seg_cl[k];
        cx.cl = s.cl + o, cx.m = s.seg_cl[k + 1] - o, cx.S = s.seg_S[k], cx.after = s.seg_after[k];
    }
    return cx;
}
__device__ __forceinline__ int chunk_of(const StreamDesc &s, int64_t p) {  // last chunk whose first position is <= p
    // a single Write's chunks are on the grid 2048 c - 261; otherwise from a guess by proportion, a few steps either way (every
    // step is a dependent load from the table: a bisection over a long stream's table was 10 us, and the resolve kernel asks
    // twice per cut)
    int c;
    if (s.grid_chunks) {
        c = (int)((p + (kMinLookahead - 1)) >> kChunkBits);
        return c < s.nchunks ? c : s.nchunks - 1;
    }
    c = (int)((p * (int64_t)s.nchunks) / ((int64_t)s.body_end + 1));
    c = c < 0 ? 0 : c >= s.nchunks ? s.nchunks - 1 : c;
    while (c > 0 && (int64_t)s.cstart[c] > p) c--;
    while (c + 1 < s.nchunks && (int64_t)s.cstart[c + 1] <= p) c++;
    return c;
}

// ------------------------------------------------------------------ K3 / K4 / K5 accessors
struct GlobalAcc {
    gcbytes in;
    const uint2 *mm;  // already offset to the stream's position 0
    const uint32_t *tab;
    int strategy, hash_variant;
    const uint16_t *lk;  // the stream's links
    __device__ int link(int64_t p) const { return (int)lk[p]; }
    __device__ uint32_t flt(uint32_t m) const {
        m &= kRecMask;
        return m ? filter_match(match_len(m), match_dist(m), strategy) : kNoMatch;
    }
    __device__ uint32_t mK(int64_t p) const { return flt(mm[p].x); }
    __device__ uint32_t mK4(int64_t p) const { return flt(mm[p].y); }
    __device__ uint8_t byte(int64_t p) const { return in[p]; }
    __device__ uint32_t bucket(int64_t p) const { return dev_bucket(tab, g_u32_bytes(in + p + 2), hash_variant); }
    __device__ int run1(int64_t p) const {  // (p is a loop-top of the body: 258 bytes and more behind it are input)
        int len = 0;
        while (len < kMaxMatch) {
            const uint64_t x = *(gcu64u)(in + p + len) ^ *(gcu64u)(in + p - 1 + len);
            if (x) {
                len += (int)(__builtin_ctzll(x) >> 3);
                break;
            }
            len += 8;
        }
        return len < kMaxMatch ? len : kMaxMatch;
    }
};
// matches staged (already filtered) in LDS for one chunk: index p - (cs - 1)
struct LdsAcc {
    gcbytes in;
    const uint32_t *fk, *fk4;
    int64_t org;  // cs - 1
    const uint32_t *tab;
    int hash_variant;
    const uint16_t *lk;               // the stream's links
    const uint8_t *lbytes = nullptr;  // optional: input bytes [org, org + kChunk + 1) staged in LDS
    __device__ int link(int64_t p) const { return (int)lk[p]; }
    __device__ uint32_t mK(int64_t p) const { return fk[p - org]; }
    __device__ uint32_t mK4(int64_t p) const { return fk4[p - org]; }
    __device__ uint8_t byte(int64_t p) const { return lbytes ? lbytes[p - org] : in[p]; }
    __device__ uint32_t bucket(int64_t p) const { return dev_bucket(tab, g_u32_bytes(in + p + 2), hash_variant); }
    __device__ int run1(int64_t p) const {  // (p is a loop-top of the body: 258 bytes and more behind it are input)
        int len = 0;
        while (len < kMaxMatch) {
            const uint64_t x = *(gcu64u)(in + p + len) ^ *(gcu64u)(in + p - 1 + len);
            if (x) {
                len += (int)(__builtin_ctzll(x) >> 3);
                break;
            }
            len += 8;
        }
        return len < kMaxMatch ? len : kMaxMatch;
    }
};

// Returns the largest distance among the thread's raw records of the chunk's own positions (K3 keeps the chunk's maximum for
// the resolve kernel's repairs).
__device__ __forceinline__ uint32_t stage_chunk_matches(const StreamDesc &s, int64_t cs, const uint2 *mm, int strategy, uint32_t *fk,
                                                        uint32_t *fk4) {
    const uint2 *a = mm + s.pos_off;
    int64_t org = cs - 1;
    uint32_t far = 0;
    const uint32_t klm = strategy == kFiltered ? 2u : 0u, kdm = strategy == kFiltered ? 0u : (uint32_t)kTooFar;
    for (int i = threadIdx.x; i < kChunk + 1; i += blockDim.x) {
        int64_t p = org + i;
        uint32_t x = 0, y = 0;
        if (p >= 1 && p <= s.body_end) {
            const uint2 v = a[p];
            x = v.x & kRecMask, y = v.y;
            if (i >= 1) {
                const uint32_t dx = (uint32_t)match_dist(x), dy = (uint32_t)match_dist(y);
                far = far > dx ? far : dx;
                far = far > dy ? far : dy;
            }
            // filter_match (zs_core.h) on a record: gone if len - 3 <= klm and dist > kdm (TOO_FAR: 3 / 4096; Filtered: <= 5 / any)
            x = (((x >> 16) <= klm) & ((x & 0xFFFFu) > kdm)) ? 0u : x;
            y = (((y >> 16) <= klm) & ((y & 0xFFFFu) > kdm)) ? 0u : y;
        }
        fk[i] = x;
        fk4[i] = y;
    }
    return far;
}

// ------------------------------------------------------------------ K3
// Transfer map of one chunk for all 260 entry slots.  512 threads: one lazy_step per table node (3 rows x 2048
// positions: R, L-or-XK, XK4 -- zs_core.h node_step3) builds the 1-step table in LDS, in-place jumping passes turn it
// into node -> (exit slot, symbols), then one lane per slot reads its entry (the refill-rule positions of
// segment-first chunks are stepped explicitly).
// The transfer map of one chunk by the threads of a workgroup (K3's whole job; the resolve kernel calls it again for the few
// chunks whose records a repair has changed).  fk, fk4: kChunk + 1 words each; tbl: kNodeExit3 words; tab: the CRC tables
// (read for event chunks only; the caller has loaded them); far_word: a zeroed LDS word or nullptr.
template <int NT>  // threads of the workgroup (compile-time: the loops below are K3's whole time)
__device__ __forceinline__ void chunkmap_compute(const StreamDesc &s, int c, const uint2 *mm, const uint16_t *link, uint32_t *maps, LevelCfg lv, int strategy, int hash_variant,
                                                 uint32_t *fk, uint32_t *fk4, uint32_t *tbl, uint32_t *tab, uint32_t *far_word, uint16_t *chunk_far) {
    constexpr int nt = NT;
    const ChunkCtx cx = chunk_ctx(s, c);
    uint32_t far = stage_chunk_matches(s, cx.cs, mm, strategy, fk, fk4);
    // the largest match distance recorded in the chunk: a cut at e can only have been seen through from a chunk whose
    // largest distance reaches back to e (zs_resolve_kernel's repair scans no others)
    if (far_word) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t t = (uint32_t)__shfl_xor((int)far, o);
            far = far > t ? far : t;
        }
        if (threadIdx.x == 0) *far_word = 0;
    }
    __syncthreads();
    if (far_word && lane_id() == 0 && far) atomicMax(far_word, far);  // read behind the barriers of the passes below
    const int64_t ce = cx.ce;
    const int len = (int)(ce - cx.cs);
    LdsAcc acc{as_global(s.in), fk, fk4, cx.cs - 1, tab, hash_variant, link + s.pos_off};
    // node_step3_all (zs_core.h: three lazy_steps and their node ids) written out with selects: through the shared code it
    // was half of this kernel's instructions, most of them execution-mask pairs around a move or two (the same finding as
    // in the symbol kernel; the shared form stays the specification and what the CPU model runs)
    static_assert(kR == 0 && kL == 1 && kXK == 2 && kXK4 == 3 && kNoMatch == 0, "node arithmetic below");
    const int lazy = lv.lazy, good = lv.good;
    auto node_at = [len](int kind, int rel, uint32_t cnt) -> uint32_t {  // node_pack(node_of3(kind, cs + rel, cs, ce), cnt)
        const int inside = ((kind + 1) >> 1) * kChunk + rel, outside = kNodeExit3 + (kind == kR ? rel - len : 256 + kind);
        return (uint32_t)(rel < len ? inside : outside) | (cnt << 14);
    };
    auto x_step = [&](uint32_t pend, uint32_t cK, uint32_t cK4, int rel) -> uint32_t {  // lazy_step at an XK / XK4 loop-top
        const int m = pend ? (int)(pend >> 16) + 3 : 2;
        const bool use4 = m >= good;
        const uint32_t curv = use4 ? cK4 : cK;
        const int mv = curv ? (int)(curv >> 16) + 3 : 2;
        const bool defer = (m < lazy) & (mv > m);
        return node_at(defer ? (use4 ? (int)kXK4 : (int)kXK) : (int)kR, defer ? rel + 1 : rel - 1 + m, 1u);
    };
    for (int off = threadIdx.x; off < len; off += nt) {
        const uint32_t before = fk[off], before4 = fk4[off];
        uint32_t cK = fk[off + 1], cK4 = fk4[off + 1];
        if (cx.cs + off == 0) cK = cK4 = kNoMatch;
        const int plain_kind = cK ? (int)kXK : (int)kL;
        const uint32_t r0 = node_at(plain_kind, off + 1, 0u);
        const uint32_t r1x = x_step(before, cK, cK4, off), r1l = node_at(plain_kind, off + 1, 1u);
        const uint32_t r2 = x_step(before4, cK, cK4, off);
        tbl[off] = r0, tbl[kChunk + off] = before ? r1x : r1l, tbl[2 * kChunk + off] = r2;
    }
    __syncthreads();
    // two dependent lookups per pass: after pass r every entry jumps >= 3^r steps or reaches its exit (a step advances
    // at least one position, a chunk has 2048: 3^7 = 2187 would finish every entry).  The passes stop earlier: the 260
    // slot lookups below follow unfinished entries (<= 2048 / 3^r lookups each), which costs less than the passes saved.
#ifndef ZS_JUMP_PASSES
#define ZS_JUMP_PASSES 3
#endif
    for (int r = 0; r < ZS_JUMP_PASSES; r++) {
        // four nodes per thread at a time, their lookups issued together (the LDS round trips overlap); a node past the
        // end of the chunk or already at its exit looks itself up and stays as it is
        for (int i0 = threadIdx.x; i0 < 3 * kChunk; i0 += 4 * nt) {
            uint32_t v[4], w[4];
            bool live[4], have[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * nt;
                have[k] = (3 * kChunk) % (4 * NT) == 0 || i < 3 * kChunk;
                v[k] = tbl[have[k] ? i : 0];
                live[k] = have[k] && (i & (kChunk - 1)) < len && node_succ(v[k]) < kNodeExit3;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = tbl[live[k] ? node_succ(v[k]) : 0];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (live[k]) v[k] = node_jump(v[k], w[k]);
                live[k] = live[k] && node_succ(v[k]) < kNodeExit3;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = tbl[live[k] ? node_succ(v[k]) : 0];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (live[k]) v[k] = node_jump(v[k], w[k]);
                if (have[k] && ((i0 + k * nt) & (kChunk - 1)) < len) tbl[i0 + k * nt] = v[k];
            }
        }
        __syncthreads();
    }
    const int slot = threadIdx.x;
    if (slot < kSlots) {
        uint32_t out = 0;
        if (slot_valid(cx, slot, s.body_end)) out = chunk_exit_by_table3(acc, tbl, cx, slot, lv, strategy);
        maps[((int64_t)s.chunk_off + c) * kSlots + slot] = out;
    }
    if (far_word && threadIdx.x == NT - 1) chunk_far[s.chunk_off + c] = (uint16_t)*far_word;
}

__global__ __launch_bounds__(512) void zs_chunkmap_kernel(const StreamDesc *sd, const uint2 *work, const uint2 *mm, const uint16_t *link,
                                                          uint32_t *maps, const uint32_t *crc_tab_g,
                                                          LevelCfg lv, int strategy, int hash_variant, uint16_t *chunk_far,
                                                          uint8_t *only_stale = nullptr, const StreamState *st = nullptr) {
    __shared__ uint32_t fk[kChunk + 1], fk4[kChunk + 1];
    __shared__ uint32_t tbl[kNodeExit3];
    __shared__ uint32_t tab[1024];
    __shared__ uint32_t sh_far;
    uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int c = (int)w.y;
    // (batched cut rounds: only the chunks whose records a restore or a repair has changed, of the streams in the rounds)
    if (only_stale && (st[w.x].deferred != 1 || st[w.x].cuts_same || !only_stale[s.chunk_off + c])) return;
    if (chunk_ctx(s, c).m != 0) load_crc_tab(tab, crc_tab_g);  // (chunkmap_compute's first barrier comes before the table's first use)
    chunkmap_compute<512>(s, c, mm, link, maps, lv, strategy, hash_variant, fk, fk4, tbl, tab, &sh_far, chunk_far);
    if (only_stale && threadIdx.x == 0) only_stale[s.chunk_off + c] = 0;
}

// ------------------------------------------------------------------ K3b
// Compose the chunk maps of one parse segment (16 chunks; 32 for segment 0) for every
// entry slot: 260 lanes, one dependent lookup per chunk.
__global__ __launch_bounds__(320) void zs_segmap_kernel(const StreamDesc *sd, const uint2 *work, const uint32_t *maps,
                                                        uint2 *segmap) {
    uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int seg = (int)w.y;
    int slot = threadIdx.x;
    if (slot >= kSlots) return;
    const int c0 = seg_first(s, seg), c1 = seg_first(s, seg + 1);
    const int entry_slot = slot;
    uint32_t total = 0, flags = 0;
    for (int c = c0; c < c1; c++) {
        uint32_t v = maps[((int64_t)s.chunk_off + c) * kSlots + slot];
        if (c == c0) flags = v & (kMapEqualBit | kMapPoisonBit);
        slot = map_exit(v);
        total += (uint32_t)map_count(v);
    }
    segmap[((int64_t)s.seg_off + seg) * kSlots + entry_slot] = make_uint2((uint32_t)slot | flags, total);
}

// ------------------------------------------------------------------ K3c
// The maps of kSupSegs consecutive parse segments composed for every entry slot, all groups of all streams at once: exit
// slot, symbols, and bit 15 when the path from that slot meets something the resolve kernel has to look at itself (a
// refill whose loop-top and successor share a bucket, a stale segment).  The resolve kernel then follows a stream through
// 1/16 of the rows, and its per-segment results are filled in by one thread per group (K4's first part).
__device__ __forceinline__ uint32_t seg_row_meta(const StreamDesc &s, int seg, const uint8_t *seg_stale) {
    // bit 0 = the segment starts with a cluster of read events, bit 1 = it holds stale chunks, bits 2.. = largest entry
    // offset that is still a loop-top of the body (only the last segment limits it)
    const int64_t cs = s.cstart[s.seg_c0[seg]];
    int64_t lim = (int64_t)s.body_end - cs;
    lim = lim > 511 ? 511 : lim;
    uint32_t m = (s.head[s.seg_c0[seg]] != 0 && lim >= 0) ? 1u : 0u;
    if (seg_stale[s.seg_off + seg]) m |= 2u;
    return m | ((uint32_t)(lim < 0 ? 0 : lim) << 2);
}
__global__ __launch_bounds__(320) void zs_supmap_kernel(const StreamDesc *sd, const uint2 *work, const uint2 *segmap,
                                                        const uint8_t *seg_stale, uint2 *supmap) {
    const uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int g = (int)w.y;
    if ((int)threadIdx.x >= kSlots) return;
    int cur = threadIdx.x;
    uint32_t cnt = 0, flag = 0, hard = 0;  // hard: attention a dry pass needs too (it takes equal-bucket events in its stride)
    for (int r = 0; r < kSupSegs; r++) {
        const int seg = g * kSupSegs + r;
        if (seg >= s.nsegs) break;
        const uint2 v = segmap[((int64_t)s.seg_off + seg) * kSlots + cur];
        const uint32_t m = seg_row_meta(s, seg, seg_stale);
        const bool fires = (m & 1u) && (uint32_t)(cur <= 256 ? cur : 0) <= (m >> 2);
        flag |= (m & 2u) | ((fires && (v.x & (kMapEqualBit | kMapPoisonBit))) ? 1u : 0u);
        hard |= (m & 2u) | ((fires && (v.x & kMapPoisonBit)) ? 1u : 0u);
        cur = (int)(v.x & 0x1FF);
        cnt += v.y;
    }
    supmap[((int64_t)s.sup_off + g) * kSlots + threadIdx.x] = make_uint2((uint32_t)cur | (flag ? 0x8000u : 0u) | (hard ? 0x4000u : 0u), cnt);
}

constexpr int kSegBatch = 64;
constexpr int kSegGroup = 8;  // rows composed in parallel before the sequential walk
constexpr int kResolveLds = kSegBatch * kSlots * 8 + 4096 + kSegBatch * 12 + (kSegBatch / kSegGroup) * kSlots * 8;
// ------------------------------------------------------------------ K4
// One workgroup per stream.  Thread 0 follows the true parse path through the segment
// maps (one dependent lookup per 32 Ki positions).  At a refill loop-top s_k whose bucket
// equals that of s_k + 1 the reference leaves prev[s_k] = s_k + 1 (Deflate.cs:1010-1013,
// 866-877), which hides everything older than s_k from later chain walks through that
// bucket: the workgroup cuts link[s_k], re-walks the positions whose recorded winner lies
// behind the cut, and marks the chunks whose matches changed; segments holding such
// chunks are then followed chunk by chunk, stale chunks by a direct walk.
// ------------------------------------------------------------------ repair of a cut (K4, and zs_repair_kernel over the chip)
// The reference leaves prev[e] = e + 1 at an equal-bucket refill loop-top e: everything older than e is hidden from later
// walks through that bucket.  link[e] is cut (by the caller); the positions of (from, to_all] whose recorded winner lies
// behind the cut are walked again and their chunks marked stale.  Two passes.  First every thread looks at its U positions:
// is the position in the cut's bucket and does its recorded winner lie behind the cut?  Straight-line code over loads that
// do not depend on each other, and the bucket test is "the same four bytes as at e" before it is a hash: on zeros or runs
// every refill is such a cut and every position behind it is in its bucket (one dependent load and one CRC per position
// were 42 us per cut, 86 ms for 64 MiB of zeros).  Only if some position has to be walked again do the window's bytes and
// links go into LDS (`lds`: up to 98 KiB) and the walks run from there (a run's long matches out of global memory: 0.25 ms
// per cut).  Returns 0: nothing to walk again, LDS untouched; 1: there is, and scan_only; 2: walked and written.
struct RepairArgs {
    const StreamDesc *s;
    uint2 *a;         // the stream's match records
    uint16_t *lk;     // the stream's links
    const uint32_t *tab;
    uint8_t *lds;
    uint8_t *stale, *seg_stale;
    const uint16_t *chunk_far;
    int nch;
    LevelCfg lv;
    int hash_variant;
    // batched cut rounds: the stream's cuts (ascending) and their buckets, this cut's index: a position with a later cut of
    // its bucket in front of it is that cut's business
    const int32_t *cl_pos = nullptr;
    const uint32_t *cl_bkt = nullptr;
    int cl_i = 0, cl_n = 0;
};
template <int NT, int U>
__device__ __forceinline__ int repair_cut(const RepairArgs &r, int64_t e, int64_t from, int64_t to_all, bool scan_only, int *sh_to,
                                          int *list, int *nlist, unsigned int *mark, int part = 0, int nparts = 1) {
    const StreamDesc &s = *r.s;
    uint2 *a = r.a;
    uint16_t *lk = r.lk;
    const uint32_t *tab = r.tab;
    const LevelCfg lv = r.lv;
    const int hash_variant = r.hash_variant, nch = r.nch;
    // a position can have seen through the cut only if its chunk's largest recorded distance reaches back to e (K3):
    // the scan ends with the last such chunk -- on zeros and the like right behind the cut
    if (threadIdx.x == 0) {
        *sh_to = (int)from;
        if (nlist) *nlist = 0;
    }
    if (mark && threadIdx.x < 24) mark[threadIdx.x] = 0;
    const int c0r = chunk_of(s, from + 1);  // the chunks a repair can touch: c0r .. (18 of them on the single-Write grid)
    __syncthreads();
    {
        const int c1 = chunk_of(s, to_all);
        for (int cc = c0r + (int)threadIdx.x; cc <= c1; cc += NT) {
            int64_t lo = s.cstart[cc], hi = (int64_t)s.cstart[cc + 1] - 1;
            if (lo < from + 1) lo = from + 1;
            if (hi > to_all) hi = to_all;
            if ((int64_t)r.chunk_far[s.chunk_off + cc] > lo - e) atomicMax(sh_to, (int)hi);
        }
    }
    __syncthreads();
    const int64_t to = *sh_to;
    if (to <= from) return 0;
    const gcbytes in = as_global(s.in);
    const int64_t nn = s.n;
    // e + 5 < n: e is a loop-top of the body.  Byte loads: e is uniform in zs_repair_kernel, where the compiler made the
    // unaligned dword load a scalar one, which reads from the address rounded down to 4 (the wrong bucket)
    const uint32_t vB = g_u32_bytes(in + e + 2);
    const uint32_t B = dev_bucket(tab, vB, hash_variant);
    // (positions dealt round the workgroup one by one: those to walk again cluster behind the cut.  Groups of 4 consecutive
    // positions per thread, so that a match walked again at p could bound the one at p + 1 from below, were slower.)
    uint32_t todo = 0;
    // (workgroup `part` of `nparts` takes every nparts-th position: the positions to walk again cluster behind the cut)
    auto pos_of = [&](int u) { return from + 1 + part + ((int64_t)u * NT + threadIdx.x) * nparts; };
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t p = pos_of(u);
        const bool in_range = p <= to;
        const int64_t pc = in_range ? p : to;
        const uint2 rec = a[pc];
        const uint32_t v = *(gcu32u)(in + pc + 2);
        const uint32_t x = rec.x & kRecMask, y = rec.y;
        const bool dirty = (x && pc - match_dist(x) < e) || (y && pc - match_dist(y) < e);
        bool inb = v == vB;
        if (in_range && dirty && !inb) inb = dev_bucket(tab, v, hash_variant) == B;
        if (in_range && dirty && inb && r.cl_pos)
            for (int j = r.cl_i + 1; j < r.cl_n; j++) {  // (slots in stream order, -1: no cut)
                const int64_t ej = r.cl_pos[j];
                if (ej < 0) continue;
                if (ej >= pc) break;
                if (r.cl_bkt[j] == B) inb = false;  // a later cut of the bucket hides this one from the position
            }
        todo |= (in_range && dirty && inb) ? 1u << u : 0u;
    }
    if (!__syncthreads_or(todo != 0)) return 0;  // nothing behind the cut was seen through it
    if (scan_only) return 1;
    // index = position - o with o = (e - 1) rounded down to 16: the cut position has an index >= 1, so that walk_matches'
    // "never position 0" holds as it stands, and both arrays move 16 bytes per lane and step
    uint8_t *rb = r.lds;
    const int64_t o = (e - 1) & ~15LL;
    const int ie = (int)(e - o);
    const int nby = ((int)(to - o) + 1 + 272 + 15) & ~15;
    uint16_t *rl = (uint16_t *)(rb + nby);
    if ((((uintptr_t)in) & 15) == 0) {
        for (int i = threadIdx.x * 16; i < nby; i += NT * 16) {
            const int64_t q = o + i;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q + 16 <= nn) {
                const u32x4 t = *(gcu32x4)(in + q);
                v = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                uint32_t t[4] = {0, 0, 0, 0};
                for (int k = 0; k < 16; k++)
                    if (q + k < nn) t[k >> 2] |= (uint32_t)in[q + k] << (8 * (k & 3));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            *(uint4 *)(rb + i) = v;
        }
    } else {
        for (int i = threadIdx.x * 4; i < nby; i += NT * 4) {
            const int64_t q = o + i;
            uint32_t v = 0;
            if (q + 4 <= nn) v = *(gcu32u)(in + q);
            else
                for (int k = 0; k < 4; k++)
                    if (q + k < nn) v |= (uint32_t)in[q + k] << (8 * (k & 3));
            *(uint32_t *)(rb + i) = v;
        }
    }
    const int nlk = ((int)(to - o) + 1 + 7) & ~7;  // links of [o, to], 8 per lane and step (the array has room behind a stream's end)
    for (int i = threadIdx.x * 8; i < nlk; i += NT * 8) *(uint4 *)(rl + i) = *(const uint4 *)(lk + o + i);
    __syncthreads();
    for (int i = threadIdx.x; i <= ie; i += NT) rl[i] = 0;  // link[e] is cut; nothing below it is reached
    __syncthreads();
    auto lkf = [rl](int64_t q) { return (int)rl[q]; };  // positions relative to o from here on
    auto lcp = [rb](int64_t u, int64_t v) {
        int len = 0;
        while (len < kMaxMatch) {
            const uint64_t x = lds_u64(rb, (int)u + len) ^ lds_u64(rb, (int)v + len);
            if (x) {
                len += (int)(__builtin_ctzll(x) >> 3);
                break;
            }
            len += 8;
        }
        return len < kMaxMatch ? len : kMaxMatch;
    };
    while (todo) {
        const int u = __builtin_ctz(todo);
        todo &= todo - 1;
        const int64_t p = pos_of(u);
        const uint2 old = a[p];
        const uint32_t x = old.x & kRecMask, y = old.y;
        uint32_t nx, ny;
        {
            // walk_matches (zs_core.h) with Longest_match's own shortcut in front of every compare: a candidate whose
            // bytes [best - 1, best] or [0, 1] differ from the scan's cannot beat `best` (Deflate.cs:1072-1078)
            nx = ny = kNoMatch;
            const int64_t q = p - o;
            int l = lkf(q);
            int64_t c = q - l;
            if (l && c >= 1 && q - c <= kMaxDist) {
                int best = 2, bdist = 0, n_eval = 0;
                const int k4 = lv.chain >> 2;
                bool snap = false;
                const uint32_t s01 = lds_u32(rb, (int)q) & 0xFFFFu;
                uint32_t send = lds_u32(rb, (int)q + best - 1) & 0xFFFFu;
                for (;;) {
                    n_eval++;
                    bool nice_exit = false;
                    if ((lds_u32(rb, (int)c + best - 1) & 0xFFFFu) == send && (lds_u32(rb, (int)c) & 0xFFFFu) == s01) {
                        const int len = lcp(q, c);
                        if (len > best) {
                            best = len, bdist = (int)(q - c);
                            if (len >= lv.nice) nice_exit = true;
                            send = lds_u32(rb, (int)q + best - 1) & 0xFFFFu;
                        }
                    }
                    if (!snap && (n_eval == k4 || nice_exit)) ny = best >= kMinMatch ? pack_match(best, bdist) : kNoMatch, snap = true;
                    if (nice_exit || n_eval == lv.chain) break;
                    l = lkf(c);
                    if (!l) break;
                    c -= l;
                    if (c < 1 || q - c >= kMaxDist) break;
                }
                nx = best >= kMinMatch ? pack_match(best, bdist) : kNoMatch;
                if (!snap) ny = nx;
            }
        }
        if (nx != x || ny != y) {
            a[p] = make_uint2(nx | (old.x & ~kRecMask), ny);
            int cp = chunk_of(s, p);
            // the chunk's map (and its successor's, whose pending-match row starts with p) no longer holds: listed once
            for (int k = 0; k < 2; k++) {
                const int cc = cp + k;
                if (k == 1 && !(p + 1 == (int64_t)s.cstart[cp + 1] && cp + 1 < nch)) break;
                r.seg_stale[s.seg_off + seg_of(s, cc)] = 1;
                r.stale[s.chunk_off + cc] = 1;
                if (mark && cc - c0r >= 0 && cc - c0r < 24 && atomicExch(&mark[cc - c0r], 1u) == 0) {
                    const int at = atomicAdd(nlist, 1);
                    if (at < 24) list[at] = cc;
                }
            }
        }
    }
    __threadfence();
    __syncthreads();
    return 2;
}

__global__ __launch_bounds__(1024) void zs_resolve_kernel(const StreamDesc *sd, StreamState *st, uint16_t *link, uint2 *mm,
                                                         uint32_t *maps, const uint2 *segmap,
                                                         uint16_t *seg_entry, uint32_t *seg_symbase, uint8_t *stale,
                                                         uint8_t *seg_stale, const uint32_t *crc_tab_g, LevelCfg lv,
                                                         int strategy, int hash_variant, int seg_limit, int mm_limit,
                                                         const uint2 *supmap, const uint16_t *chunk_far, int defer_mode_in,
                                                         int32_t *cut_pos, uint32_t *cut_bkt, int cut_stride, int cut_iter) {
    // defer_mode: 0 every cut repaired here; 1 the same until the budgets are spent, then the stream is given up where it stands
    // (StreamState::deferred = 1) for the batched cut rounds; 3 a dry pass of those rounds: the walk goes on from where the
    // stream was given up, the cuts on its way are collected, nothing is repaired
    const int defer_mode = defer_mode_in & 0xFF;
    const bool dry = defer_mode == 3;
    const uint32_t attn = dry ? 0x4000u : 0x8000u;  // which flag of a composed row stops the walk (zs_supmap_kernel)
    const int cut_budget = (defer_mode_in >> 16) & 0xFF;  // (ZS_FORCE_ROUNDS: 0)
    const bool dbg = (defer_mode_in & 0x100) != 0;  // ZS_DEBUG_CUTS: the cuts as they are applied
    // > 64 KiB of LDS: dynamic allocation, carved by hand
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint2 *rows = (uint2 *)smem;                                  // segment-map rows of the current batch (130 KiB)
    uint32_t *tab = (uint32_t *)(smem + kSegBatch * kSlots * 8);   // 1024
    uint32_t *out_base = tab + 1024;                              // per segment of the batch: first-symbol index,
    uint16_t *out_slot = (uint16_t *)(out_base + kSegBatch);      // entry slot,
    uint32_t *row_meta = (uint32_t *)(out_slot + kSegBatch);      // flags and entry limit per row
    uint2 *gmap = (uint2 *)(row_meta + kSegBatch);                // per group of kSegGroup rows: composed map of every slot
    __shared__ int g_slot[kSegBatch / kSegGroup], g_kf[kSegBatch / kSegGroup], g_ks[kSegBatch / kSegGroup], g_fast[kSegBatch / kSegGroup];
    __shared__ uint32_t g_base[kSegBatch / kSegGroup];
    __shared__ int sh_kf, sh_ks;
    __shared__ int sh_seg, sh_slot, sh_scan, sh_kfired, sh_kslot;
    __shared__ int sh_defer, sh_nexp, sh_ncut_any, sh_cutidx, sh_poison, sh_cut_e, sh_nc, sh_diff;
    __shared__ int row_cs[kSegBatch];  // per staged row: the first position of the segment's first chunk, bit 31: its cluster is one boundary
    __shared__ int b_seg0, b_nrow, b_groups;  // the batch of rows in LDS: first segment, rows, whether its composed groups may be used
    __shared__ uint32_t sh_total;
    const StreamDesc s = sd[blockIdx.x];
    StreamState &ss = st[blockIdx.x];
#ifdef ZS_FV_PROF
    const long long k4_t0 = wall_clock64();
#endif
    load_crc_tab(tab, crc_tab_g);
    // seg_limit / mm_limit (one long stream run part by part, else "everything"): segments below seg_limit have their maps,
    // positions up to mm_limit their match records; the kernel goes on from where the launch before stopped
    if (threadIdx.x == 0)
        sh_seg = ss.r_seg, sh_slot = ss.r_slot, sh_total = ss.r_total, sh_kfired = ss.r_kfired, sh_kslot = ss.r_kslot, sh_scan = ss.r_scan,
        ss.r_scan = dry ? ss.r_scan : 0, b_seg0 = 0, b_nrow = 0, b_groups = 0, sh_defer = 0, sh_nexp = 0, sh_ncut_any = 0, sh_cutidx = ss.r_cutidx, sh_poison = 0,
        sh_cut_e = -1, sh_nc = 0, sh_diff = 0x7FFFFFFF;
    __syncthreads();
    if (dry && ss.deferred != 1) return;  // only the streams that were given up take part in the rounds
    // a run that took the stream over in the middle (StreamDesc::resume) starts in the node and with the symbols it was handed
    if (s.resume && threadIdx.x == 0 && sh_seg == 0 && sh_slot == 0 && sh_total == 0) sh_slot = s.start_slot, sh_total = s.start_syms;
    __syncthreads();
    if (s.body_end < 0) {
        if (threadIdx.x == 0) ss.tail_p = 0, ss.tail_kind = kR, ss.tail_pend = 0, ss.k_done = 0, ss.preins = -1, ss.body_syms = 0;
        return;
    }
    uint16_t *lk = link + s.pos_off;
    uint2 *a = mm + s.pos_off;
    GlobalAcc acc{as_global(s.in), a, tab, strategy, hash_variant, lk};
    const int nseg = s.nsegs < seg_limit ? s.nsegs : seg_limit, nch = s.nchunks;
    const int64_t mm_end = (int64_t)s.body_end < (int64_t)mm_limit ? (int64_t)s.body_end : (int64_t)mm_limit;
    __shared__ int rp_to, rp_nlist, rp_list[24];
    __shared__ unsigned int rp_mark[24];
    // the cut's repair (repair_cut above) with the staged rows' room as its LDS; then, when the stream is resolved in one
    // launch (all match records exist), the maps of the chunks it changed again, by the whole workgroup (K3's routine): a
    // stale chunk is otherwise walked by one thread out of global memory wherever its map is wanted -- 1.5 ms each here, as
    // much again in K4b
    auto repair = [&](int64_t e, int64_t from, int64_t to_all, bool scan_only) {
        RepairArgs ra{&s, a, lk, tab, (uint8_t *)rows, stale, seg_stale, chunk_far, nch, lv, hash_variant};
        const int rc = repair_cut<1024, 32>(ra, e, from, to_all, scan_only, &rp_to, rp_list, &rp_nlist, rp_mark);
        if (rc != 2) return rc;
        if (supmap != nullptr && rp_nlist <= 24) {
            uint32_t *c_fk = (uint32_t *)rows, *c_fk4 = c_fk + kChunk + 1, *c_tbl = c_fk4 + kChunk + 1;
            const int nl = rp_nlist;
            for (int k = 0; k < nl; k++) {
                const int cc = rp_list[k];
                __syncthreads();
                chunkmap_compute<1024>(s, cc, mm, link, maps, lv, strategy, hash_variant, c_fk, c_fk4, c_tbl, tab, nullptr, nullptr);
                __syncthreads();
                if (threadIdx.x == 0) stale[s.chunk_off + cc] = 0;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) b_nrow = 0;  // the rows are gone: the next batch is staged afresh
        return rc;
    };
    // (The composed rows of kSupSegs segments each -- zs_supmap_kernel -- carry the walk wherever nothing on the path needs
    // attention: "fast stretches" inside the loop below.)
    const bool can_sup = supmap != nullptr && seg_limit >= s.nsegs && (int64_t)mm_limit >= (int64_t)s.body_end && s.nsegs > kSupSegs;
    __shared__ int fs_try, fs_n;
    __shared__ unsigned long long fp_kf;
    // cuts of an earlier launch whose repair had to stop where the match records ended
    {
        const int nc = ss.r_ncut;
        int keep = 0;
        for (int i = 0; i < nc; i++) {
            const int64_t e = ss.r_cut_e[i], from = ss.r_cut_done[i];
            int64_t full = e + kMaxDist;
            if (full > s.body_end) full = s.body_end;
            const int64_t to = full < mm_end ? full : mm_end;
            if (to > from) repair(e, from, to, false);
            __syncthreads();
            if (threadIdx.x == 0 && to < full) ss.r_cut_e[keep] = (int32_t)e, ss.r_cut_done[keep] = (int32_t)to;
            if (to < full) keep++;
        }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) ss.r_ncut = keep;
    }
    // The events of the cluster at the head of segment sh_seg, on the path from slot sh_slot, whose loop-top shares its bucket
    // with the next position: the reference leaves prev[e] = e + 1 there (see above).  In stream order: thread 0 finds the
    // next one (a repair changes records, and with them the path to the events behind it), link[e] is cut, the positions that
    // saw through the cut are walked again.  sh_cutidx counts the cuts applied (a round leaves the kernel in the middle).
    struct NthEqualEv {
        int want, seen;
        int64_t pos;
        __device__ void operator()(int64_t p, bool eq) {
            if (!eq) return;
            if (seen == want) pos = p;
            seen++;
        }
    };
    auto process_cuts = [&]() {
        const ChunkCtx cx = chunk_ctx(s, s.seg_c0[sh_seg]);
        for (;;) {
            if (threadIdx.x == 0) {
                if (cx.m == 1) {
                    // one boundary, one event: the entry loop-top (the flag that brought the walk here says its buckets are equal)
                    sh_cut_e = sh_cutidx == 0 ? (int)(sh_slot <= 256 ? cx.cs + sh_slot : cx.cs) : -1;
                } else {
                    NthEqualEv ne{sh_cutidx, 0, -1};
                    NullSink nsk;
                    int kind, ns;
                    int64_t pp;
                    uint32_t fl;
                    chunk_special_prefix(acc, nsk, cx, sh_slot, lv, strategy, kind, pp, ns, fl, ne);
                    sh_cut_e = (int)ne.pos;
                    if (fl & kMapPoisonBit) sh_poison = 1;
                }
            }
            __syncthreads();
            const int64_t e = sh_cut_e;
            if (dbg && threadIdx.x == 0) printf("zs resolve: segment %d slot %d cut #%d at %ld (defer mode %d)\n", sh_seg, sh_slot, sh_cutidx, (long)e, defer_mode);
            if (e < 0 || sh_poison) break;
            if (threadIdx.x == 0) lk[e] = 0, sh_cutidx++;
            __threadfence_block();
            __syncthreads();
            int64_t full = e + kMaxDist;
            if (full > s.body_end) full = s.body_end;
            const int64_t to = full < mm_end ? full : mm_end;
            {
                const int rc = repair(e, e, to, false);
                if (threadIdx.x == 0 && to < full) {  // the rest once the records beyond exist
                    const int k = ss.r_ncut < 8 ? ss.r_ncut++ : 7;
                    ss.r_cut_e[k] = (int32_t)e, ss.r_cut_done[k] = (int32_t)to;
                }
                // not this CU's job any longer: the batched cut rounds take the stream from here (zs_engine.hip)
                if (defer_mode == 1 && threadIdx.x == 0 && ((rc == 2 && ++sh_nexp > (cut_budget < kDeferBudget ? cut_budget : kDeferBudget)) || ++sh_ncut_any > cut_budget)) sh_defer = 1;
            }
            __threadfence_block();
            __syncthreads();
            if (sh_defer) break;
        }
        __syncthreads();
        if (threadIdx.x == 0 && !sh_defer) sh_cutidx = 0;
        __syncthreads();
    };
    // A dry pass: the equal-bucket events on the path go into the stream's cut slots of this pass -- every segment has a slot
    // per boundary of its cluster (slot of segment k's first: seg_cl[k] - seg_cl[0] + k), -1 = no cut: whoever fills in a
    // segment's entry (thread 0 row by row, a lane per group or per composed row) writes them, so the walk itself stops for
    // nothing but stale rows.  A cluster of one boundary has its one event at the entry loop-top.
    int32_t *cl_out = cut_pos ? cut_pos + (size_t)(cut_iter & 1) * (size_t)cut_stride + s.cut_off : nullptr;
    const int cl_cap = s.nsegs > 0 ? s.seg_cl[s.nsegs] - s.seg_cl[0] + s.nsegs : 0;
    if (dry) {
        for (int i = threadIdx.x; i < cl_cap; i += blockDim.x) cl_out[i] = -1;
        __threadfence_block();
        __syncthreads();
    }
    auto collect_cuts = [&](int seg, int slot, int skip, int cs_single) {  // cs_single: row_cs of the segment, or 0
        const int k0 = s.seg_cl[seg];
        int32_t *dst = cl_out + (k0 - s.seg_cl[0]) + seg;
        if (cs_single < 0) {  // one boundary, one event: the entry loop-top
            const int64_t cs = cs_single & 0x7FFFFFFF;
            if (skip == 0) dst[0] = (int32_t)(slot <= 256 ? cs + slot : cs);
            return;
        }
        const ChunkCtx cx = chunk_ctx(s, seg_first(s, seg));
        if (cx.m == 1) {
            if (skip == 0) dst[0] = (int32_t)(slot <= 256 ? cx.cs + slot : cx.cs);
            return;
        }
        struct SlotEqualEv {
            int32_t *dst;
            int skip, seen, n, cap;
            __device__ void operator()(int64_t p, bool eq) {
                if (eq && seen++ >= skip && n < cap) dst[n++] = (int32_t)p;
            }
        } ae{dst, skip, 0, 0, cx.m};
        NullSink nsk;
        int kind, ns;
        int64_t pp;
        uint32_t fl;
        chunk_special_prefix(acc, nsk, cx, slot, lv, strategy, kind, pp, ns, fl, ae);
        if (fl & kMapPoisonBit) sh_poison = 1;
    };
    // the kernel was left in the middle of a cluster's cuts (a stream given up there): the rest of them first
    if (sh_scan && sh_seg < nseg) {
        if (dry) {
            if (threadIdx.x == 0) collect_cuts(sh_seg, sh_slot, sh_cutidx, 0);
            __syncthreads();
        } else {
            process_cuts();
        }
    }
#ifdef ZS_FV_PROF
    long long kp[6] = {0, 0, 0, 0, 0, 0}, kt = wall_clock64();
    const long long k4_t1 = kt;
    int kiter = 0;
#define K4_PF(i) { const long long now_ = wall_clock64(); kp[i] += now_ - kt; kt = now_; }
#else
#define K4_PF(i)
#endif
    // The rows a fast stretch has walked wait here until their segments' entries are filled in (one thread per row follows
    // its sixteen segment maps: sixteen dependent loads, ~30 us whatever the number of rows, so the stretches of a stream
    // share one such pass; it runs before anything looks at the results: the long way below, the cuts, the kernel's end)
    constexpr int kPendRows = 512;
    __shared__ uint16_t pf_slot[kPendRows];
    __shared__ uint32_t pf_base[kPendRows];
    __shared__ int pf_g0, pf_n;
    if (threadIdx.x == 0) pf_g0 = 0, pf_n = 0, fp_kf = 0;
    __syncthreads();
    auto flush_fill = [&]() {  // every thread of the workgroup
        const int n_p = pf_n, g_p = pf_g0;
        if (n_p == 0) return;
        for (int t = threadIdx.x; t < n_p; t += blockDim.x) {
            const int g = g_p + t;
            int cur = pf_slot[t];
            uint32_t total = pf_base[t];
            unsigned long long kf = 0;
            // (the segments' flags come from three levels of tables: all sixteen asked for before the chain of map lookups)
            uint32_t metas[kSupSegs];
#pragma unroll
            for (int r = 0; r < kSupSegs; r++) {
                const int seg = g * kSupSegs + r;
                metas[r] = seg < s.nsegs ? seg_row_meta(s, seg, seg_stale) : 0u;
            }
#pragma unroll
            for (int r = 0; r < kSupSegs; r++) {
                const int seg = g * kSupSegs + r;
                if (seg < s.nsegs) {  // (no break: the loop is unrolled, metas[] stays in registers)
                    const uint2 v = segmap[((int64_t)s.seg_off + seg) * kSlots + cur];
                    const uint32_t m = metas[r];
                    if ((m & 1u) && (uint32_t)(cur <= 256 ? cur : 0) <= (m >> 2)) {
                        kf = ((unsigned long long)(seg + 1) << 16) | (unsigned)cur;
                        if (dry && (v.x & kMapEqualBit) && strategy != kHuffmanOnly) collect_cuts(seg, cur, 0, 0);
                    }
                    seg_entry[s.seg_off + seg] = (uint16_t)cur;
                    seg_symbase[s.seg_off + seg] = total;
                    cur = (int)(v.x & 0x1FF);
                    total += v.y;
                }
            }
            if (kf) atomicMax(&fp_kf, kf);  // the last segment whose events fired
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (fp_kf) sh_kfired = (int)(fp_kf >> 16) - 1, sh_kslot = (int)(fp_kf & 0xFFFF);
            pf_n = 0, fp_kf = 0;
        }
        __syncthreads();
    };
    for (;;) {
        const bool k4_done = sh_seg >= nseg || sh_defer || sh_poison;  // done, or the rest of a cluster's cuts stopped again
#ifdef ZS_FV_PROF
        kiter++;
        kt = wall_clock64();
#endif
        // ---- a fast stretch: at a multiple of kSupSegs segments the walk takes whole composed rows (zs_supmap_kernel: exit
        //      slot, symbols and "something on the path from this slot needs attention" for kSupSegs segments at once) for as
        //      long as neither the path nor a repair has touched them: 64 rows staged at a time, thread 0 follows the path
        //      through them -- 1/16 of the dependent lookups and of the bytes this one CU has to pull -- and one thread per
        //      row then fills in its segments' entries and symbol bases from the segment maps.  A flagged or stale row is
        //      walked the long way below, and the stretches go on behind it. ----
        bool went_fast = false;
        if (!k4_done && can_sup && (sh_seg % kSupSegs) == 0 && !sh_scan) {
            const int g0 = sh_seg / kSupSegs, nsup = (s.nsegs + kSupSegs - 1) / kSupSegs;
            int nrow = nsup - g0;
            if (nrow > kSegBatch) nrow = kSegBatch;
            if (threadIdx.x == 0) fs_try = !(supmap[((int64_t)s.sup_off + g0) * kSlots + sh_slot].x & attn), fs_n = 0;
            __syncthreads();
            K4_PF(0);
            if (fs_try) {  // uniform over the workgroup
                const uint4 *src = (const uint4 *)(supmap + ((int64_t)s.sup_off + g0) * kSlots);
                for (int i = threadIdx.x; i < nrow * kSlots / 2; i += blockDim.x) ((uint4 *)rows)[i] = src[i];
                if ((int)threadIdx.x < nrow) {  // has a repair touched the row since it was composed?
                    uint32_t st_any = 0;
                    for (int r = 0; r < kSupSegs; r++) {
                        const int seg = (g0 + (int)threadIdx.x) * kSupSegs + r;
                        if (seg < s.nsegs) st_any |= seg_stale[s.seg_off + seg];
                    }
                    row_meta[threadIdx.x] = st_any;
                }
                __syncthreads();
                K4_PF(1);
                if (threadIdx.x == 0) {
                    int slot = sh_slot, i = 0;
                    uint32_t total = sh_total;
                    if (pf_n == 0) pf_g0 = g0;
                    for (; i < nrow; i++) {
                        const uint2 e = rows[i * kSlots + slot];
                        if ((e.x & attn) || row_meta[i]) break;
                        pf_slot[pf_n + i] = (uint16_t)slot, pf_base[pf_n + i] = total;
                        slot = (int)(e.x & 0x1FF), total += e.y;
                    }
                    sh_slot = slot, sh_total = total, fs_n = i, b_nrow = 0;
                    pf_n += i;
                    if (i > 0) {
                        const int seg = (g0 + i) * kSupSegs;
                        sh_seg = seg > s.nsegs ? s.nsegs : seg;
                    }
                }
                __syncthreads();
                K4_PF(2);
                went_fast = fs_n > 0;
                // (the rows' segments are filled in later, those of several stretches at once: flush_fill)
                if (fs_n == nrow && pf_n + kSegBatch <= kPendRows && sh_seg < nseg) continue;
            }
        }
        flush_fill();
        K4_PF(5);
        if (k4_done) break;
        if (went_fast) continue;
        // ---- stage the next kSegBatch segment-map rows (coalesced), then thread 0 follows the path through them
        //      (one LDS lookup per segment; its per-segment results go out coalesced afterwards); it stops early
        //      when a refill needs the whole workgroup ----
        // (after a repair the walk goes on inside the batch that is staged already -- data whose every refill is an equal-bucket
        // one, zeros or runs, would otherwise stage 130 KiB per segment: the rows keep, their flags are read again because the
        // repair may have marked segments stale, and the composed groups are not used for the rest of the batch)
        const bool resume = sh_scan && b_nrow > 0 && sh_seg >= b_seg0 && sh_seg < b_seg0 + b_nrow;
        const int seg0 = resume ? b_seg0 : sh_seg;
        int nrow = nseg - seg0;
        // (a batch ends on a multiple of kSupSegs segments, where a fast stretch can take over)
        const int batch_max = resume ? b_nrow : kSegBatch - (can_sup ? seg0 % kSupSegs : 0);
        if (nrow > batch_max) nrow = batch_max;
        if (!resume) {
            const uint4 *src = (const uint4 *)(segmap + ((int64_t)s.seg_off + seg0) * kSlots);  // kSlots is even: 16-byte aligned
            for (int i = threadIdx.x; i < nrow * kSlots / 2; i += blockDim.x) ((uint4 *)rows)[i] = src[i];
        }
        if (threadIdx.x < nrow) {
            const int sg = seg0 + (int)threadIdx.x, c0s = s.seg_c0[sg];
            row_meta[threadIdx.x] = seg_row_meta(s, sg, seg_stale);
            row_cs[threadIdx.x] = s.cstart[c0s] | ((s.head[c0s] != 0 && s.seg_cl[sg + 1] - s.seg_cl[sg] == 1) ? (int)0x80000000 : 0);
        }
        __syncthreads();
        if (threadIdx.x == 0) b_seg0 = seg0, b_nrow = nrow, b_groups = resume ? 0 : 1;
        K4_PF(0);
        // ---- compose every group of kSegGroup rows for all slots in parallel: exit slot, symbols, and whether the
        //      path from that slot meets anything the sequential walk must look at (an equal-bucket refill, a stale row)
        const int ngroup = resume ? 0 : (nrow + kSegGroup - 1) / kSegGroup;
        for (int t = threadIdx.x; t < ngroup * kSlots; t += blockDim.x) {
            const int g = t / kSlots;
            int cur = t - g * kSlots;
            uint32_t cnt = 0, flag = 0, hard = 0;
            for (int r = 0; r < kSegGroup; r++) {
                const int i = g * kSegGroup + r;
                if (i >= nrow) break;
                const uint2 v = rows[i * kSlots + cur];
                const uint32_t m = row_meta[i];
                const bool fires = (m & 1u) && (uint32_t)(cur <= 256 ? cur : 0) <= (m >> 2);
                flag |= (m & 2u) | ((fires && (v.x & (kMapEqualBit | kMapPoisonBit))) ? 1u : 0u);
                hard |= (m & 2u) | ((fires && (v.x & kMapPoisonBit)) ? 1u : 0u);
                cur = (int)(v.x & 0x1FF);
                cnt += v.y;
            }
            gmap[t] = make_uint2((uint32_t)cur | (flag ? 0x8000u : 0u) | (hard ? 0x4000u : 0u), cnt);
        }
        __syncthreads();
        if (threadIdx.x < kSegBatch / kSegGroup) g_fast[threadIdx.x] = 0;
        __syncthreads();
        K4_PF(1);
        if (threadIdx.x == 0) {
            int seg = resume ? sh_seg : seg0, slot = sh_slot;
            uint32_t total = sh_total;
            bool scanned = sh_scan != 0;  // the pending segment's cut has just been applied
            const bool groups = b_groups != 0;
            int kf = -1, kslot = 0;       // last refill that fired among the rows walked one by one
            const bool cuts = strategy != kHuffmanOnly;
            bool stop = false;
            while (seg < seg0 + nrow) {
                const int i = seg - seg0;
                if (groups && !scanned && i % kSegGroup == 0) {
                    // a whole group at once when nothing on the path from `slot` needs attention; its per-row results are
                    // filled in afterwards, one lane per group
                    const int g = i / kSegGroup;
                    const uint2 e = gmap[g * kSlots + slot];
                    if (!(e.x & attn)) {
                        g_slot[g] = slot, g_base[g] = total, g_fast[g] = 1;
                        slot = (int)(e.x & 0x1FF);
                        total += e.y;
                        seg += kSegGroup;
                        if (seg > seg0 + nrow) seg = seg0 + nrow;
                        continue;
                    }
                }
                const uint2 v = rows[i * kSlots + slot];
                const uint32_t m = row_meta[i];
                if ((m & 1u) && (uint32_t)(slot <= 256 ? slot : 0) <= (m >> 2)) {
                    kf = seg, kslot = slot;
                    // the flags of the path through the cluster: the row's, unless the segment has gone stale since the row was
                    // made -- the events behind a cluster's first one are where the records of the positions before them put
                    // them, and a repair may have changed those: then from what the first chunk holds now
                    uint32_t vx = v.x;
                    if (m & 2u) {
                        const int c0s = seg_first(s, seg);
                        if (stale[s.chunk_off + c0s]) {
                            NullSink ns0;
                            int ex0, cnt0;
                            uint32_t fl0;
                            walk_chunk(acc, ns0, chunk_ctx(s, c0s), slot, lv, strategy, ex0, cnt0, &fl0);
                            vx = fl0;
                        } else {
                            vx = maps[((int64_t)s.chunk_off + c0s) * kSlots + slot];
                        }
                    }
                    if (vx & kMapPoisonBit) {  // not a stream for the bulk path (zs_core.h kMapPoisonBit)
                        sh_poison = 1;
                        break;
                    }
                    if ((vx & kMapEqualBit) && cuts && !scanned) {
                        if (!dry) {
                            stop = true;
                            break;
                        }
                        collect_cuts(seg, slot, 0, row_cs[i]);
                        if (sh_poison) break;
                    }
                }
                scanned = false;
                out_slot[i] = (uint16_t)slot;
                out_base[i] = total;
                if (m & 2u) {
                    const int c0 = seg_first(s, seg), c1 = seg_first(s, seg + 1);
                    for (int cc = c0; cc < c1; cc++) {
                        int ex, cnt;
                        if (stale[s.chunk_off + cc]) {
                            NullSink ns;
                            uint32_t fl;
                            walk_chunk(acc, ns, chunk_ctx(s, cc), slot, lv, strategy, ex, cnt, &fl);
                            if (fl & kMapPoisonBit) sh_poison = 1;
                        } else {
                            uint32_t mp = maps[((int64_t)s.chunk_off + cc) * kSlots + slot];
                            ex = map_exit(mp), cnt = map_count(mp);
                        }
                        slot = ex;
                        total += (uint32_t)cnt;
                    }
                } else {
                    slot = (int)(v.x & 0x1FF);
                    total += v.y;
                }
                seg++;
            }
            sh_kf = kf, sh_ks = kslot;
            sh_scan = stop ? 1 : 0;
            sh_seg = seg, sh_slot = slot, sh_total = total;
        }
        __syncthreads();
        K4_PF(2);
        if (threadIdx.x < kSegBatch / kSegGroup && g_fast[threadIdx.x]) {
            const int g = threadIdx.x;
            int cur = g_slot[g], kf = -1, ks = 0;
            uint32_t total = g_base[g];
            for (int r = 0; r < kSegGroup; r++) {
                const int i = g * kSegGroup + r;
                if (i >= nrow) break;
                const uint2 v = rows[i * kSlots + cur];
                const uint32_t m = row_meta[i];
                if ((m & 1u) && (uint32_t)(cur <= 256 ? cur : 0) <= (m >> 2)) {
                    kf = seg0 + i, ks = cur;
                    if (dry && (v.x & kMapEqualBit) && strategy != kHuffmanOnly) collect_cuts(seg0 + i, cur, 0, row_cs[i]);
                }
                out_slot[i] = (uint16_t)cur;
                out_base[i] = total;
                cur = (int)(v.x & 0x1FF);
                total += v.y;
            }
            g_kf[g] = kf, g_ks[g] = ks;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            // the last refill that fired in this batch: among the rows walked one by one or inside a group
            int kf = sh_kf, ks = sh_ks;
            for (int g = 0; g < kSegBatch / kSegGroup; g++)
                if (g_fast[g] && g_kf[g] > kf) kf = g_kf[g], ks = g_ks[g];
            if (kf >= 0) sh_kfired = kf, sh_kslot = ks;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < sh_seg - seg0; i += blockDim.x) {
            seg_entry[s.seg_off + seg0 + i] = out_slot[i];
            seg_symbase[s.seg_off + seg0 + i] = out_base[i];
        }
        K4_PF(3);
        if (sh_poison) break;
        if (!sh_scan) {
            if (sh_seg >= nseg) break;
            continue;
        }
        // ---- equal-bucket events in the cluster at the head of segment sh_seg: cuts and repairs ----
        process_cuts();
        __syncthreads();
        K4_PF(4);
        if (sh_defer || sh_poison) break;
    }
#ifdef ZS_FV_PROF
    const long long k4_t2 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0 && kiter > 4)
        printf("K4PROF iterations=%d ticks: stage+meta=%lld compose=%lld walk=%lld fill+flush=%lld repair=%lld\n", kiter, kp[0], kp[1], kp[2], kp[3], kp[4]);
#endif
    if (threadIdx.x == 0 && !dry) ss.r_seg = sh_seg, ss.r_slot = sh_slot, ss.r_total = sh_total, ss.r_kfired = sh_kfired, ss.r_kslot = sh_kslot, ss.r_cutidx = sh_cutidx;
    if (dry && !sh_poison) {
        // this pass's cuts against the pass before: the same cuts mean the records they were repaired for are the ones this
        // pass walked -- the stream is resolved; else the first cut that differs says from where to repair again
        const int cur = cut_iter & 1;
        const int32_t *a = cut_pos + (size_t)cur * (size_t)cut_stride + s.cut_off, *b = cut_pos + (size_t)(cur ^ 1) * (size_t)cut_stride + s.cut_off;
        uint32_t *cbw = cut_bkt + (size_t)cur * (size_t)cut_stride + s.cut_off;
        __threadfence_block();
        __syncthreads();
        for (int i = threadIdx.x; i < cl_cap; i += blockDim.x) {
            const int32_t e = a[i];
            if (e >= 0) cbw[i] = acc.bucket(e);  // the cuts' buckets, for the repairs (a later cut of a bucket hides an earlier one)
            if (e != b[i]) atomicMin(&sh_diff, i);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int di = sh_diff < cl_cap ? sh_diff : cl_cap;
            int64_t dp = 0x7FFFFFFF;
            if (di < cl_cap) {
                if (a[di] >= 0) dp = a[di];
                if (b[di] >= 0 && b[di] < dp) dp = b[di];
            }
            ss.nc[cur] = cl_cap, ss.cuts_same = di >= cl_cap ? 1 : 0, ss.cut_diff_idx = di, ss.cut_diff_pos = (int32_t)dp;
        }
    }
    if (sh_poison) {
        if (threadIdx.x == 0) ss.poison = 1, ss.deferred = 3;  // the kernels behind skip the stream; the host runs it on the literal engine
        return;
    }
    if (sh_defer) {
        if (threadIdx.x == 0) ss.deferred = sh_defer, ss.r_scan = 1;
        return;
    }
    if (sh_seg < s.nsegs) return;  // more segments to come in a later launch
    if (threadIdx.x == 0) {
        int slot = sh_slot;
        int64_t ce = (int64_t)s.body_end + 1;
        int kind = slot <= 256 ? kR : slot - 256;
        int64_t p = slot <= 256 ? ce + slot : ce;
        ss.tail_p = (int32_t)p;
        ss.tail_kind = kind;
        ss.tail_pend = kind == kXK ? acc.mK(p - 1) : kind == kXK4 ? acc.mK4(p - 1) : 0;
        ss.k_done = sh_kfired;
        // the position the last event pre-inserted: the last event loop-top of the last cluster that fired, + 1
        struct LastEv {
            int64_t pos;
            __device__ void operator()(int64_t q, bool) { pos = q; }
        } lev{-1};
        if (s.nsegs > 0) {
            NullSink nsk;
            int k2, n2;
            int64_t p2;
            uint32_t f2;
            chunk_special_prefix(acc, nsk, chunk_ctx(s, s.seg_c0[sh_kfired]), sh_kslot, lv, strategy, k2, p2, n2, f2, lev);
        }
        ss.preins = lev.pos >= 0 ? (int32_t)lev.pos + 1 : -1;
        ss.body_syms = sh_total;
#ifdef ZS_FV_PROF
        const long long k4_t3 = wall_clock64();
        if (blockIdx.x == 0) printf("K4PROF ticks: prologue=%lld loop=%lld epilogue=%lld segs=%d; fast stretches: try=%lld stage=%lld walk=%lld fill=%lld\n", k4_t1 - k4_t0, k4_t2 - k4_t1, k4_t3 - k4_t2, s.nsegs, kp[0], kp[1], kp[2], kp[5]);
#endif
    }
}

// ------------------------------------------------------------------ K4b
// One thread per parse segment: entry slot and first-symbol index of each of its chunks.
// ------------------------------------------------------------------ a round of repairs over the chip
// For the streams the resolve kernel has stopped at a cut (deferred == 2): the positions behind the cut walked again, 1 Ki
// positions per workgroup (repair_cut), then the maps of the chunks that changed (K3's routine), then the flags reset;
// the host launches the resolve kernel again.
constexpr int kRepairLds = ((kMaxDist + 16 + 273 + 15) & ~15) + 2 * (kMaxDist + 32) + 64;
// The records of a stream in the batched cut rounds, from the position of the first cut that changed on, as they were when
// the rounds began (the copy `bak`): the pass's cuts are applied to records that have seen no cut of an earlier pass.  A
// record that changes marks its chunk (and the next one when it is the chunk's last: its pending-match row) and segment.
// ... and gets back the largest match distance its chunk had then (chunk_far, which the repairs' scans go by: the map pass of the
// round before has lowered it to what that round's repairs left).
__global__ __launch_bounds__(256) void zs_cut_restore_kernel(const StreamDesc *sd, const StreamState *st, uint2 *mm, const uint2 *bak, uint8_t *stale,
                                                             uint8_t *seg_stale, uint16_t *chunk_far, const uint16_t *far_bak, int n_streams) {
    for (int si = (int)blockIdx.y; si < n_streams; si += (int)gridDim.y) {  // (a grid's y and z end at 65 535: more streams than that take turns)
        const StreamDesc s = sd[si];
        const StreamState &ss = st[si];
        if (ss.deferred != 1 || ss.cuts_same) continue;
        const int64_t from = (int64_t)ss.cut_diff_pos + 1, to = s.body_end;
        for (int64_t p = from + (int64_t)blockIdx.x * 256 + threadIdx.x; p <= to; p += (int64_t)gridDim.x * 256) {
            const uint2 want = bak[s.pos_off + p], have = mm[s.pos_off + p];
            if (want.x == have.x && want.y == have.y) continue;
            mm[s.pos_off + p] = want;
            const int cp = chunk_of(s, p);
            stale[s.chunk_off + cp] = 1, seg_stale[s.seg_off + seg_of(s, cp)] = 1;
            chunk_far[s.chunk_off + cp] = far_bak[s.chunk_off + cp];
            if (p + 1 == (int64_t)s.cstart[cp + 1] && cp + 1 < s.nchunks) stale[s.chunk_off + cp + 1] = 1, seg_stale[s.seg_off + seg_of(s, cp + 1)] = 1;
        }
    }
}
// One pass's cuts applied over the chip: workgroup (part, cut, stream) takes every nparts-th position behind its cut
// (repair_cut: the scan, and where something has to be walked again the window behind the cut in LDS).  Cuts whose reach
// ends before the first cut that changed have their repairs in the records already.
template <int NT, int U>
__global__ __launch_bounds__(NT) void zs_cuts_repair_kernel(const StreamDesc *sd, const StreamState *st, uint16_t *link, uint2 *mm, uint8_t *stale,
                                                             uint8_t *seg_stale, const uint16_t *chunk_far, const uint32_t *crc_tab_g, LevelCfg lv,
                                                             int hash_variant, const int32_t *cut_pos, const uint32_t *cut_bkt, int cut_stride, int cut_iter,
                                                             int n_streams) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ uint32_t tab[1024];
    __shared__ int sh_to;
    load_crc_tab(tab, crc_tab_g);
    __syncthreads();
    // A grid's y and z end at 65 535, and the slots of a stream are one per data end plus one per parse segment -- a GiB in one
    // Write, or 64 MiB in 1000-byte Writes, has more: the workgroups take the slots (and the streams) in turns.
    for (int si = (int)blockIdx.z; si < n_streams; si += (int)gridDim.z) {
        const StreamDesc s = sd[si];
        const StreamState &ss = st[si];
        if (ss.deferred != 1 || ss.cuts_same) continue;
        const int cur = cut_iter & 1, nc = ss.nc[cur];
        const int32_t *cl = cut_pos + (size_t)cur * (size_t)cut_stride + s.cut_off;
        const uint32_t *cb = cut_bkt + (size_t)cur * (size_t)cut_stride + s.cut_off;
        const int64_t dp = ss.cut_diff_pos;
        for (int j = (int)blockIdx.y; j < nc; j += (int)gridDim.y) {
            const int64_t e = cl[j];
            if (e < 0) continue;  // no cut in this slot
            int64_t full = e + kMaxDist;
            if (full > s.body_end) full = s.body_end;
            const int64_t from = e > dp ? e : dp;  // the records up to the first changed cut were not restored
            if (from >= full) continue;
            RepairArgs ra{&s, mm + s.pos_off, link + s.pos_off, tab, smem, stale, seg_stale, chunk_far, s.nchunks, lv, hash_variant, cl, cb, j, nc};
            repair_cut<NT, U>(ra, e, from, full, false, &sh_to, nullptr, nullptr, nullptr, (int)blockIdx.x, (int)gridDim.x);
            __syncthreads();  // (the window in LDS and sh_to are the next slot's)
        }
    }
}
// The rounds are over for a stream whose last pass found the cuts of the pass before: the cuts go into the links (the tail
// engine restores prev[] from them, zs_lit_engine.h le_restore_prev) and the stream goes on down the pipeline.
__global__ __launch_bounds__(256) void zs_cuts_apply_kernel(const StreamDesc *sd, StreamState *st, uint16_t *link, const int32_t *cut_pos, int cut_stride,
                                                            int cut_iter) {
    const StreamDesc s = sd[blockIdx.x];
    StreamState &ss = st[blockIdx.x];
    if (ss.deferred != 1 || !ss.cuts_same) return;
    const int cur = cut_iter & 1, nc = ss.nc[cur];
    const int32_t *cl = cut_pos + (size_t)cur * (size_t)cut_stride + s.cut_off;
    for (int i = threadIdx.x; i < nc; i += 256)
        if (cl[i] >= 0) link[s.pos_off + cl[i]] = 0;
    __syncthreads();
    if (threadIdx.x == 0) ss.deferred = 0, ss.r_scan = 0;
}

__global__ __launch_bounds__(64) void zs_expand_kernel(const StreamDesc *sd, const StreamState *st, const uint2 *work, int nwork, const uint2 *mm,
                                                       const uint16_t *link, const uint32_t *maps, const uint16_t *seg_entry,
                                                       const uint32_t *seg_symbase, const uint8_t *stale, uint16_t *entry,
                                                       uint32_t *symbase, const uint32_t *crc_tab_g, LevelCfg lv, int strategy,
                                                       int hash_variant) {
    int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nwork) return;
    uint2 w = work[i];
    if (st[w.x].deferred) return;  // the resolve kernel gave the stream up: the batch is run again in rounds
    const StreamDesc s = sd[w.x];
    const int seg = (int)w.y;
    GlobalAcc acc{as_global(s.in), mm + s.pos_off, crc_tab_g, strategy, hash_variant, link + s.pos_off};
    int slot = seg_entry[s.seg_off + seg];
    uint32_t total = seg_symbase[s.seg_off + seg];
    const int c0 = seg_first(s, seg), c1 = seg_first(s, seg + 1);
    for (int c = c0; c < c1; c++) {
        entry[s.chunk_off + c] = (uint16_t)slot;
        symbase[s.chunk_off + c] = total;
        int ex, cnt;
        if (stale[s.chunk_off + c]) {
            NullSink ns;
            walk_chunk(acc, ns, chunk_ctx(s, c), slot, lv, strategy, ex, cnt);
        } else {
            uint32_t m = maps[((int64_t)s.chunk_off + c) * kSlots + slot];
            ex = map_exit(m), cnt = map_count(m);
        }
        slot = ex;
        total += (uint32_t)cnt;
    }
}

// ------------------------------------------------------------------ K5
// Symbols + block cuts along the true path, one lane per chunk: every lane walks its chunk from its entry node
// straight out of HBM (a lane's reads run along consecutive positions, so each 128-byte line of `mm` serves 16
// steps out of L1/L2) and writes its symbols in place.  All chunks of the batch walk at once, so the ~1000
// dependent steps of a chunk are paid once per launch.  (An earlier form built a jump table per chunk in LDS and
// expanded checkpoint intervals with one workgroup per chunk: 2.5x slower.)
struct GlobalSymSink {
    uint32_t *out;   // the chunk's first symbol
    uint32_t base;   // its stream-global index
    int32_t *blk_end, *blk_top;
    uint32_t next_cut;  // index (in the chunk) of the next symbol that completes a block
    __device__ GlobalSymSink(uint32_t *o, uint32_t b, int32_t *be, int32_t *bt)
        : out(o), base(b), blk_end(be), blk_top(bt), next_cut((b / kBlockSyms + 1) * kBlockSyms - 1 - b) {}
    __device__ void operator()(int i, uint32_t sym, int64_t end, int64_t top) {
        out[i] = sym;
        if ((uint32_t)i == next_cut) {
            const uint32_t bi = (base + (uint32_t)i) / kBlockSyms;
            blk_end[bi] = (int32_t)end;
            blk_top[bi] = (int32_t)top;
            next_cut += kBlockSyms;
        }
    }
};
// One lane per chunk, ~1300 dependent steps, two walking waves per CU.  What a step costs is first of all the latency of
// its own instructions (a lone wave issues a dependent one every 4-10 cycles): the step is written out here without
// branches and in 32-bit positions -- lazy_step (zs_core.h, which K3 / K4 and the CPU model use) is the specification,
// tests/test_gpu_parity.py the check -- and only then did the memory side show: the records come through LDS, staged by
// feeding waves, and no step waits for a store's acknowledgement (DESIGN.md section 6 has the order of the findings).
constexpr uint32_t kK5Idle = 0xFFFFFFFEu, kK5Done = 0xFFFFFFFFu;
// Lines (16 records, 128 bytes) per walking lane in LDS; slot r of lane w: r * 8192 + w * 128.  The kernel runs with four
// (eight: the feeding sweep twice as long, slower; three: the same as four).
template <int R>
__device__ __forceinline__ uint32_t k5_slot(uint32_t line) {  // line mod R (lines of a chunk: < 512)
    if constexpr (R == 4) return line & 3u;
    else return line - (uint32_t)R * ((line * (uint32_t)(512 / R + 1)) >> 9);
}
static_assert(512 / 3 + 1 == 171, "k5_slot<3>: floor(L * 171 / 512) = floor(L / 3) for L < 512");
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
__device__ __forceinline__ void k5_publish(lds_u32p slot, uint32_t v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(uintptr_t)slot), "v"(v) : "memory");
}
// (reads the compiler knows about: it places the waits; volatile keeps them in program order among themselves)
__device__ __forceinline__ uint32_t k5_peek(lds_u32p slot) { return *(volatile __attribute__((address_space(3))) uint32_t *)slot; }
__device__ __forceinline__ uint64_t k5_peek64(uint32_t addr) { return *(volatile __attribute__((address_space(3))) uint64_t *)(uintptr_t)addr; }
template <int kK5Ring>
__global__ __launch_bounds__(kK5Threads) void zs_emit_syms_lane_kernel(const StreamDesc *sd, const StreamState *st, const uint2 *work, int nwork, const uint2 *mm,
                                                               const uint16_t *link, const uint16_t *entry, const uint32_t *symbase, uint32_t *syms,
                                                               int32_t *blk_end, int32_t *blk_top, const uint32_t *crc_tab_g,
                                                               LevelCfg lv, int strategy, int hash_variant, int ahead) {
    // Wave 0 walks (one lane per chunk), wave 1 feeds it.  A lane's records come 8 bytes a step out of its own 128-byte
    // lines: as loads of the walking wave they are 64 different lines per instruction (the texture path takes them one
    // lane at a time), two instructions a step, and a wave's loads return in order, so it cannot ask ahead for itself.
    // The feeding wave reads every line once, 8 lanes x 16 bytes, straight into LDS (LDS-DMA: no register is named, so
    // nothing has to wait for the data), kK5Ring lines ahead of where each walking lane says it is; the walking lanes take
    // their records from LDS and fall back to a load of their own where a jump (a long match) outran the ring.
    // Lines are counted from the line of the chunk's first loop-top.  Slot of line L: L mod kK5Ring; line L + kK5Ring
    // goes into it only when the lane has said it is past L, and the lane reads nothing below the line it has announced.
    __shared__ uint4 sh_ring_[kK5Ring * 64 * 128 / 16];
    __shared__ uint32_t sh_line_[64], sh_last_[64], sh_base_[64], sh_filled_[64];
    const int lane = (int)(threadIdx.x & 63);
    const lds_u32p sh_line = (lds_u32p)sh_line_ + lane, sh_last = (lds_u32p)sh_last_ + lane, sh_base = (lds_u32p)sh_base_ + lane,
                   sh_filled = (lds_u32p)sh_filled_ + lane;
    if (threadIdx.x < 64) {
        k5_publish(sh_line, kK5Idle);
        k5_publish(sh_filled, (uint32_t)-1);
    }
    __syncthreads();
    if (threadIdx.x >= 64) {
        if (ahead <= 0) return;
        // feeding wave f serves the walking lanes 16 f .. 16 f + 15: two groups of 8, each lane one 16-byte piece of a line
        constexpr int kG = 8 / kK5Feeders;
        const int piece = lane & 7, wsub = lane >> 3;
        const int g0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) - 1) * kG;
        int32_t nf[kG];  // per group of 8 walking lanes: the next line this lane's walker has not been given yet
#pragma unroll
        for (int g = 0; g < kG; g++) nf[g] = 0;
        for (;;) {
            uint32_t c[kG], la[kG], ba[kG];
#pragma unroll
            for (int g = 0; g < kG; g++) {
                c[g] = k5_peek((lds_u32p)sh_line_ + 8 * (g0 + g) + wsub);
                la[g] = k5_peek((lds_u32p)sh_last_ + 8 * (g0 + g) + wsub);
                ba[g] = k5_peek((lds_u32p)sh_base_ + 8 * (g0 + g) + wsub);
            }
            bool busy = false, did = false;
            int32_t pub[kG];
#pragma unroll
            for (int g = 0; g < kG; g++) {
                busy |= c[g] != kK5Done;
                pub[g] = -1;
                if (c[g] < kK5Idle) {
                    const int32_t cl = (int32_t)c[g], lo = nf[g] > cl ? nf[g] : cl;
                    int32_t hi = cl + kK5Ring - 1;
                    hi = hi > (int32_t)la[g] ? (int32_t)la[g] : hi;
                    const int lo_slot = (int)k5_slot<kK5Ring>((uint32_t)lo);
#pragma unroll
                    for (int r = 0; r < kK5Ring; r++) {
                        const int32_t L = lo + (r >= lo_slot ? r - lo_slot : r + kK5Ring - lo_slot);  // the line of [lo, lo + R) in slot r
                        if (L <= hi) {
                            const uint2 *src = mm + (((size_t)ba[g] + (size_t)L) << 4) + 2 * piece;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                                             (__attribute__((address_space(3))) void *)((__attribute__((address_space(3))) uint8_t *)sh_ring_ + r * 8192 + (g0 + g) * 1024),
                                                             16, 0, 0);
                            did = true;
                        }
                    }
                    if (lo <= hi) {
                        nf[g] = hi + 1;
                        pub[g] = hi;
                    }
                }
            }
            if (__ballot(busy) == 0) break;
            if (__ballot(did) != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int g = 0; g < kG; g++)
                    if (pub[g] >= 0 && piece == 0) k5_publish((lds_u32p)sh_filled_ + 8 * (g0 + g) + wsub, (uint32_t)pub[g]);
            } else {
                __builtin_amdgcn_s_sleep(2);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS must not be somebody else's when data arrives
        return;
    }
    struct Done {
        lds_u32p slot;
        __device__ ~Done() { k5_publish(slot, kK5Done); }
    } done{sh_line};
    const int i = blockIdx.x * 64 + lane;
    if (i >= nwork) return;
    const uint2 w = work[i];
    if (st[w.x].deferred) return;
    const StreamDesc s = sd[w.x];
    const int c = (int)w.y;
    GlobalAcc acc{as_global(s.in), mm + s.pos_off, crc_tab_g, strategy, hash_variant, link + s.pos_off};
    const uint32_t base = symbase[s.chunk_off + c];
    GlobalSymSink sink(syms + s.sym_off + base, base, blk_end + s.blk_off, blk_top + s.blk_off);
    // the refill-rule prefix (first chunks of segments) by the shared code, then plain automaton steps
    int kind, ns;
    int64_t p;
    uint32_t pflags;
    NullEv nev;
    const ChunkCtx cx = chunk_ctx(s, c);
    chunk_special_prefix(acc, sink, cx, (int)entry[s.chunk_off + c], lv, strategy, kind, p, ns, pflags, nev);
    const int64_t ce = cx.ce;
    if (p >= ce) return;
    // A step's successor is p+1 or, when the pending match is emitted, p-1+len(pend): both records (the literal byte a
    // step at p+1 may emit rides in bits 24..31 of the first) are requested before the step is worked out.
    const uint2 *a = acc.mm;
    const gcbytes gin = acc.in;
    uint32_t pend = kNoMatch;
    if (kind == kXK) pend = acc.mK(p - 1);
    else if (kind == kXK4) pend = acc.mK4(p - 1);
    uint8_t lit = p >= 1 ? gin[p - 1] : 0;
    const bool rec_lits = strategy != kHuffmanOnly;  // HuffmanOnly has no match pass: its records are zero-filled
    uint2 cur = a[p];
    if (p == 0) cur.x &= ~kRecMask, cur.y = 0;  // no search at position 0
    // filter_match on a record: gone if len - 3 <= klm and dist > kdm (TOO_FAR: 3 / 4096; Filtered: <= 5 / any)
    static_assert(kNoMatch == 0 && kMinMatch == 3, "record arithmetic below");
    const uint32_t klm = strategy == kFiltered ? 2u : 0u, kdm = strategy == kFiltered ? 0u : (uint32_t)kTooFar;
    const int q_end = (int)ce, q_last = s.n - 1, lazy = lv.lazy, good = lv.good;
    int q = (int)p;
    // line 0 = the line of the first loop-top's record; q0 = the position its first record would have
    const int64_t gi0 = s.pos_off + q;
    const int q0 = q - (int)(gi0 & 15);
    const uint32_t ring_lane = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)sh_ring_ + (uint32_t)lane * 128u;
    k5_publish(sh_base, (uint32_t)(gi0 >> 4));
    k5_publish(sh_last, (uint32_t)(q_last - q0) >> 4);
    // the loop in two copies: with one, the literal's register is the target of a load in one case and of a shift in the
    // other, and the compiler guards the shift with a wait for everything in flight -- the loads just issued
    auto steps = [&](auto rl_tag, auto ring_tag) {
        constexpr bool kRecLits = decltype(rl_tag)::value, kRing = decltype(ring_tag)::value;
        // a step's symbol is stored at the top of the next step, ahead of that step's loads (the wave's memory operations
        // complete in order: behind the loads, its acknowledgement would be waited for with them in the same step)
        uint32_t dsym = 0;
        int dns = -1, dend = 0, dtop = 0;
#ifdef ZS_FV_PROF
        long long k5t0 = wall_clock64();
        int k5steps = 0, k5miss = 0, k5lanemiss = 0, k5mysteps = 0;
#endif
        while (q < q_end) {
            if (dns >= 0) sink(dns, dsym, dend, dtop);
            const int m = pend ? (int)(pend >> 16) + 3 : 2;
            int qa = q + 1, qb = q - 1 + m;
            qa = qa > q_last ? q_last : qa;
            qb = qb > q_last ? q_last : qb;
            qb = pend ? qb : qa;
            uint2 na, nb;
            int32_t filled = 0;
            if constexpr (kRing) {
                const uint32_t ua = (uint32_t)(qa - q0), ub = (uint32_t)(qb - q0);
                k5_publish(sh_line, (uint32_t)(q - q0) >> 4);
                filled = (int32_t)k5_peek(sh_filled);  // read ahead of the records: what it vouches for is in them
                const uint64_t ra = k5_peek64(ring_lane + (k5_slot<kK5Ring>(ua >> 4) << 13) + ((ua & 15u) << 3));
                const uint64_t rb = k5_peek64(ring_lane + (k5_slot<kK5Ring>(ub >> 4) << 13) + ((ub & 15u) << 3));
                na = make_uint2((uint32_t)ra, (uint32_t)(ra >> 32));
                nb = make_uint2((uint32_t)rb, (uint32_t)(rb >> 32));
            } else {
                na = a[qa];
                nb = a[qb];
            }
            uint8_t nlit;
            if constexpr (kRecLits) nlit = (uint8_t)(cur.x >> 24);
            else nlit = gin[q];
            uint32_t cK = cur.x & kRecMask, cK4 = cur.y & kRecMask;
            cK = (((cK >> 16) <= klm) & ((cK & 0xFFFFu) > kdm)) ? 0u : cK;
            cK4 = (((cK4 >> 16) <= klm) & ((cK4 & 0xFFFFu) > kdm)) ? 0u : cK4;
            const bool is_x = kind >= kXK, use4 = m >= good;
            const uint32_t curv = use4 ? cK4 : cK;
            const int mv = curv ? (int)(curv >> 16) + 3 : 2;
            const bool emit_match = is_x & !((m < lazy) & (mv > m));
            const bool emit = is_x | (kind == kL);
            const int nkind = is_x ? (emit_match ? (int)kR : use4 ? (int)kXK4 : (int)kXK) : cK ? (int)kXK : (int)kL;
            const int npos = emit_match ? q - 1 + m : q + 1;
            dsym = emit_match ? ((pend & 0xFFFFu) << 16) | (pend >> 16) : (uint32_t)lit;
            dns = emit ? ns : -1;
            dend = emit_match ? npos : q;
            dtop = q;
            ns += emit ? 1 : 0;
            pend = nkind == kXK ? cK : nkind == kXK4 ? cK4 : 0u;
            kind = nkind;
            if constexpr (kRing) {
                const int qn = npos == q + 1 ? qa : qb;
                cur = npos == q + 1 ? na : nb;
#ifdef ZS_FV_PROF
                k5steps++, k5mysteps++;
                k5miss += __ballot((int32_t)((uint32_t)(qn - q0) >> 4) > filled) != 0;
                k5lanemiss += (int32_t)((uint32_t)(qn - q0) >> 4) > filled;
#endif
                if ((int32_t)((uint32_t)(qn - q0) >> 4) > filled) {  // a jump past what the ring holds
                    // (load and wait in one piece the compiler does not look into: a load of its own would make it wait,
                    // in every step, for everything in flight -- the acknowledgement of the store at the step's top)
                    uint64_t rec;
                    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(rec) : "v"(a + qn) : "memory");
                    cur = make_uint2((uint32_t)rec, (uint32_t)(rec >> 32));
                }
            } else {
                cur = npos == q + 1 ? na : nb;
            }
            asm volatile("" ::"v"(cur.x), "v"(cur.y));  // the wait for this step's loads stays in this step
            lit = nlit;
            q = npos;
        }
        if (dns >= 0) sink(dns, dsym, dend, dtop);
#ifdef ZS_FV_PROF
        if (kRing && (blockIdx.x & 63) == 7 && (lane == 0 || lane == 37))
            printf("K5PROF block %d lane %d: my steps %d, my misses %d; steps with a miss in the wave %d; ticks (100 MHz) %lld\n", (int)blockIdx.x, lane,
                   k5mysteps, k5lanemiss, k5miss, wall_clock64() - k5t0);
#endif
    };
    if (!rec_lits) steps(std::false_type{}, std::false_type{});
    else if (ahead > 0) steps(std::true_type{}, std::true_type{});
    else steps(std::true_type{}, std::false_type{});
}

// ------------------------------------------------------------------ K5b
// The block cuts recorded by K5 become BlockRec entries (the blocks that end inside the bulk parse).
__global__ __launch_bounds__(256) void zs_body_blocks_kernel(const StreamDesc *sd, const StreamState *st, const int32_t *blk_end,
                                                             const int32_t *blk_top, BlockRec *blocks, uint32_t *syms) {
    const StreamDesc s = sd[blockIdx.x];
    if (s.fast_runs > 0 || st[blockIdx.x].deferred) return;
    if (s.resume)  // the symbols of the block the run took over in the middle: in front of the run's own
        for (uint32_t i = threadIdx.x; i < s.start_syms; i += blockDim.x) syms[s.sym_off + i] = s.persist->syms[i];
    const int nb_body = (int)(st[blockIdx.x].body_syms / kBlockSyms);
    BlockRec *blk = blocks + s.blk_off;
    // start = end of the previous block; stored blocks are allowed only while blockStart has not slid out of the
    // window (Deflate.cs:953)
    for (int i = threadIdx.x; i < nb_body; i += blockDim.x) {
        int64_t start = i ? blk_end[s.blk_off + i - 1] : (s.resume ? s.start_block : 0);
        int64_t end = blk_end[s.blk_off + i];
        BlockRec r;
        r.start = start;
        r.sym_start = (int64_t)i * kBlockSyms;
        r.stored_len = (int32_t)(end - start);
        r.nsyms = kBlockSyms;
        r.can_store = start >= s.base0 + (int64_t)kWSize * (s.rle_end >= 0 ? rle_refills_fired_at(blk_top[s.blk_off + i], s.kl)
                                                                            : refills_fired_at(blk_top[s.blk_off + i] - s.base0, s.kl));
        r.eof = 0;
        blk[i] = r;
    }
    // Writes that began inside the body (every Write is a Deflate call of its own, and under a flush mode the chunk accounting
    // counts them: zs_core.h FlushAcct): Write w begins at the first loop-top that finds its predecessor's data short
    // (>= end - 261), so the blocks flushed before it are those whose last symbol came at a loop-top below that.  The Writes
    // the tail engine enters are its own to count.
    if (s.wr_blk && s.wr_end && s.n_wr > 1) {
        const StreamState &ss = st[blockIdx.x];
        const int64_t read_to = s.nsegs > 0 ? (int64_t)s.seg_after[ss.k_done] : 0;  // data end where the tail engine takes over
        for (int w = 1 + (int)threadIdx.x; w < s.n_wr; w += blockDim.x) {
            if (s.wr_end[w - 1] >= read_to) continue;
            const int64_t bound = s.wr_end[w - 1] - (kMinLookahead - 1);
            int lo = 0, hi = nb_body;  // first block whose last loop-top is not below the bound
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((int64_t)blk_top[s.blk_off + mid] < bound) lo = mid + 1;
                else hi = mid;
            }
            s.wr_blk[w] = lo;
        }
    }
    // the first block flushed by the tail engine (which ran beside K5) starts where the last finished block ends
    if (threadIdx.x == 0 && nb_body > 0 && s.final_run) {
        BlockRec r = blk[nb_body];
        if (r.can_store < 0) {
            const int64_t base = (int64_t)(-r.can_store - 1) * kWSize, start = blk_end[s.blk_off + nb_body - 1];
            r.start = start;
            r.stored_len -= (int32_t)start;
            r.can_store = start >= base;
            blk[nb_body] = r;
        }
    }
}

// ------------------------------------------------------------------ K6
// One wave per stream; all lanes run the engine uniformly.  Also turns the
// block cuts recorded by K5 into BlockRec entries.
// (Tried for batches, where this kernel and the symbol kernel are launched side by side and mostly take the CUs in turns:
// 112 registers instead of 115 -- `amdgpu_num_vgpr(56)`, the attribute counts in pairs on this target -- so that four waves
// per SIMD leave room for a wave of the symbol kernel, and a three-line ring there so that its 25 KiB fit beside this
// kernel's 133.  Nothing moved (4096 x 32 KiB: 13.6 against 13.4 ms) and the four spilled registers cost a lone stream's
// tail 0.04 ms; not kept.)
__global__ __launch_bounds__(1024) void zs_tail_kernel(const StreamDesc *sd, StreamState *st, const uint16_t *link, uint32_t *syms,
                                                     const int32_t *blk_end, const int32_t *blk_top, BlockRec *blocks,
                                                     uint8_t *scratch, const uint32_t *crc_tab_g, LevelCfg lv, int strategy,
                                                     int hash_variant, int level) {
#ifdef ZS_FV_PROF
    const long long tk_start = wall_clock64();
    long long pf_restore[4] = {0, 0, 0, 0};
#endif
    const StreamDesc s = sd[blockIdx.x];
    if (s.fast_runs > 0) return;  // handled by zs_fast_run_kernel / zs_fast_stitch_kernel
    StreamState &ss = st[blockIdx.x];
    if (ss.deferred) return;  // given up by the resolve kernel: the batch is run again in rounds
    const int tid = threadIdx.x, nth = blockDim.x;  // all threads restore; wave 0 then runs the engine
    if (s.plan_nblk > 0) {
        // level 0: the stored blocks were planned on the host from the sizes; nothing to parse
        for (int i = tid; i < s.plan_nblk; i += nth) blocks[s.blk_off + i] = s.plan_blk[i];
        if (tid == 0) ss.nsyms = 0, ss.nblocks = s.plan_nblk;
        return;
    }
    uint8_t *sc = scratch + (int64_t)blockIdx.x * kScratchBytes;
    BlockRec *blk = blocks + s.blk_off;
    const uint32_t body_syms = ss.body_syms;
    const int nb_body = (int)(body_syms / kBlockSyms);  // their BlockRecs come from zs_body_blocks_kernel
    // window and prev live in LDS (129 KiB: the engine is latency-bound on them), head in HBM scratch
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    LitEngine e;
    le_defaults(e);
    e.window = smem;
    e.head = (uint16_t *)(sc + kScratchHead);
    e.prev = (uint16_t *)(smem + kScratchHead);
    uint32_t *head32 = (uint32_t *)(sc + kScratchHead32);
    uint32_t *tabl = (uint32_t *)(smem + kScratchHead + 2 * kWSize);  // the hash of every insert goes through these
    load_crc_tab(tabl, crc_tab_g);
    e.crc_tab = tabl;
    e.data = s.in;
    e.n = s.n;
    e.wr_end = s.wr_end;  // set for several Writes and for any Write under a flush mode
    e.n_wr = s.n_wr;
    e.cur_wr = 0;
    e.wr_flush = s.wr_flush;
    e.wr_blk = s.wr_blk;
    e.lv = lv;
    e.strategy = strategy;
    e.hash_variant = hash_variant;
    e.syms = syms + s.sym_off;
    e.nsyms = body_syms;
    e.blocks = blk;
    e.nblocks = nb_body;
    // this kernel runs beside K5, which records where the last finished block ends: the first block flushed here
    // gets its start from zs_body_blocks_kernel afterwards
    e.block_start_abs = 0;
    e.defer_start = nb_body > 0;
    if ((!s.final_run || s.resume) && nb_body > 0) {
        // a run that is not the stream's end runs behind K5 (the pending block's symbols and start must be final when the
        // engine is left for the next run): the last finished block's end is known
        e.block_start_abs = blk_end[s.blk_off + nb_body - 1];
        e.defer_start = 0;
    }
    if (s.resume && nb_body == 0) e.block_start_abs = s.start_block;  // the block the run took over is still in progress
    e.block_sym_start = (int64_t)nb_body * kBlockSyms;
    e.block_syms = level == 0 ? (kLitBufsize / 2) - 1 : kBlockSyms;
    const uint16_t *lk = link + s.pos_off;
    const int64_t p = ss.tail_p;
    e.final_run = s.final_run;
    LitPersist *ps = s.persist;
    bool use_rec = false, no_head = false;
    if (s.cont) {
        // a later run of an incremental stream: the engine as the run before left it.  The input buffer starts at stream
        // position abs_off: the last 64 KiB the engine has already read (a stored block is copied from there, like the
        // reference copies it from its window) and then the new bytes; the symbols of the block in progress go first
        for (int i = tid * 16; i < kWindowSize + 512; i += nth * 16) *(uint4 *)(smem + i) = *(const uint4 *)(ps->window + i);
        for (int i = tid * 8; i < kWSize; i += nth * 8) *(uint4 *)(e.prev + i) = *(const uint4 *)(ps->prev + i);
        for (int i = tid; i < ps->pending_syms; i += nth) e.syms[i] = ps->syms[i];
        e.head = ps->head;
        e.data = s.in - s.abs_off;
        e.n = s.abs_off + (int64_t)s.n;
        e.base = ps->base, e.avail_end = ps->avail_end, e.block_start_abs = ps->block_start_abs;
        e.strstart = ps->strstart, e.lookahead = ps->lookahead, e.match_length = ps->match_length, e.match_start = ps->match_start;
        e.match_available = ps->match_available, e.prev_length = ps->prev_length, e.prev_match = ps->prev_match;
        e.nsyms = ps->pending_syms, e.block_sym_start = 0, e.nblocks = 0, e.defer_start = 0;
        e.stop_abs = s.stop_abs;
        __syncthreads();
    } else {
        // ss.k_done is the parse segment of the last read event that fired before p
        const bool have_seg = s.nsegs > 0;
#ifdef ZS_FV_PROF
        long long tk0 = wall_clock64(), tk1 = 0, tk2 = 0, tk3 = 0;
#endif
        int64_t base_in = have_seg ? s.seg_base[ss.k_done] : 0;
        const int64_t after_in = have_seg ? s.seg_after[ss.k_done] : 0;
        // A tail that starts at or behind the slide point (Deflate.cs:979: strstart >= 2 * WSIZE - MIN_LOOKAHEAD; every stream
        // whose length is a multiple of 32 KiB) slides the window in its first pass through the loop and reads nothing: the
        // engine is restored in the slid state right away (le_tail_preslide), where the route without hash heads is open
        const bool preslid = lv.func == 2 && strategy != kRle && strategy != kHuffmanOnly && s.fv_end < 0 &&
                             le_tail_preslide(e, p, base_in, after_in, ss.preins);
        if (preslid) base_in += kWSize;
        le_restore(e, p, base_in, after_in, ss.tail_kind, ss.tail_pend, lk, ss.preins, tid, nth, preslid);
        // A resumed run that has not slid the window the engine before it left (a short run behind a flush): what lies behind
        // the data in it is what that engine had there, not what a window filled by this run alone would hold
        const int64_t img_base = preslid ? e.base - kWSize : e.base;
        const bool same_window = s.resume && ps && ps->base - s.persist_off == img_base;
        if (same_window) {
            const int64_t valid = e.avail_end - img_base;
            for (int w = tid; w < kWindowSize + 512; w += nth) {
                const int wi = preslid && w < kWSize ? w + kWSize : w;
                if (wi >= valid) e.window[w] = ps->window[wi];
            }
        }
        // a slow level, one Write, everything read, no pre-insert pending: the tail's searches are done ahead (below) and the
        // engine runs without the hash heads (LitEngine::no_head) -- the table is not built
        // (the run's last Write, which ends the stream or closes under a flush mode: le_tail_reads_done)
        use_rec = lv.func == 2 && strategy != kRle && strategy != kHuffmanOnly && s.fv_end < 0 && le_tail_reads_done(e) && ss.preins < p &&
                  le_tail_record_end(e) > p && le_tail_record_end(e) - p <= kTailRecMax;
        // (... only where the stream ends: a run that goes on leaves the heads for the next one; and not while the window still
        // holds positions of the engine before a resumed run: their buckets are not the data's)
        no_head = use_rec && s.final_run && !e.wr_end && le_no_head_ok(e) && !(s.resume && p - (kWSize - 1) < s.start_pos);
        if (!no_head) {
            // (a resumed run: the heads as the engine before it left them, in this window's indices -- the latest member of
            // a bucket is one of the run's own positions or that; its chains and a FullFlush's forgetting are not the data's)
            const int64_t shift = s.resume && ps ? e.base - (ps->base - s.persist_off) : 0;
            for (int i = tid; i < kHashSize; i += nth) {
                int64_t hv = s.resume && ps ? (int64_t)ps->head[i] : 0;
                hv = hv != 0 && hv - shift > 0 ? hv - shift : 0;
                head32[i] = hv ? (uint32_t)hv + 1u : 0u;
            }
        }
        __syncthreads();
#ifdef ZS_FV_PROF
        tk1 = wall_clock64();
#endif
        if (e.avail_end > 0) {
            int64_t lo = p - (kWSize - 1);
            if (lo < e.base) lo = e.base;
            if (lo < 0) lo = 0;
            int64_t hi = p;
            if (hi > (int64_t)s.n - 5) hi = (int64_t)s.n - 5;
            const uint32_t *gb = s.ins_bits;
            auto insf = [gb](int64_t c) { return ((gb[c >> 5] >> (c & 31)) & 1u) != 0; };
            for (int64_t q = lo + tid; q < hi; q += nth) {
                if (s.fv_end >= 0) {  // DeflateFast: only what was inserted is in the chains
                    if (!insf(q)) continue;
                    le_restore_prev_ins(e, q, lk, insf);
                } else {
                    le_restore_prev(e, q, lk);
                }
                if (!no_head && q >= s.start_pos) atomicMax(&head32[le_bucket(e, q)], (uint32_t)(q - e.base) + 1u);
            }
            __syncthreads();
#ifdef ZS_FV_PROF
            tk2 = wall_clock64();
#endif
            if (!no_head)
                for (int i = tid; i < kHashSize; i += nth) e.head[i] = (uint16_t)(head32[i] ? head32[i] - 1 : 0);
        }
        __syncthreads();
#ifdef ZS_FV_PROF
        tk3 = wall_clock64();
        pf_restore[0] = tk0 - tk_start, pf_restore[1] = tk1 - tk0, pf_restore[2] = tk2 - tk1, pf_restore[3] = tk3 - tk2;
#endif
    }
    // ---- the searches of the tail's loop-tops ahead of its parse, one position per thread (le_tail_record,
    //      zs_lit_engine.h): a slow level, one Write, everything read, no pre-insert pending.  The engine then looks its
    //      matches up; only the last max_lazy positions are searched by the engine itself.
    __shared__ uint32_t pre_tab[2 * kTailRecMax];
    __shared__ uint32_t sh_tail_head[3];
    {
        const int64_t hi = le_tail_record_end(e);
        if (use_rec) {  // uniform over the workgroup
            // what the engine's inserts would write (and return: LitEngine::no_head), up to the last position K1 has a link for
            for (int64_t q = p + tid; q < (no_head ? e.n - 5 : hi); q += nth) le_restore_prev(e, q, lk);
            // ... and for the three positions behind: the nearest position below each with its bucket (le_tail_head_bucket)
            if (tid < 3) sh_tail_head[tid] = 0;
            __syncthreads();
            if (no_head) {
                const uint32_t hb0 = le_tail_head_bucket(e, 0), hb1 = le_tail_head_bucket(e, 1), hb2 = le_tail_head_bucket(e, 2);
                int64_t lo = p - (kWSize - 1);
                if (lo < e.base) lo = e.base;
                if (lo < 1) lo = 1;  // position 0 is never a candidate
                for (int64_t q = lo + tid; q < e.n - 3; q += nth) {
                    const uint32_t h = le_bucket(e, q);
                    if (h == hb0 && q < e.n - 5) atomicMax(&sh_tail_head[0], (uint32_t)(q - e.base));
                    if (h == hb1 && q < e.n - 4) atomicMax(&sh_tail_head[1], (uint32_t)(q - e.base));
                    if (h == hb2 && q < e.n - 3) atomicMax(&sh_tail_head[2], (uint32_t)(q - e.base));
                }
            }
            __syncthreads();
            e.no_head = no_head ? 1 : 0;
            e.tail_head[0] = (int)sh_tail_head[0], e.tail_head[1] = (int)sh_tail_head[1], e.tail_head[2] = (int)sh_tail_head[2];
            for (int64_t q = p + tid; q < hi; q += nth) {
                pre_tab[2 * (q - p)] = le_tail_record(e, (int)(q - e.base), (int)(e.n - q), lv.chain);
                pre_tab[2 * (q - p) + 1] = le_tail_record(e, (int)(q - e.base), (int)(e.n - q), lv.chain >> 2);
            }
            e.pre_rec = pre_tab, e.pre_lo = p, e.pre_hi = hi;
            __syncthreads();
        }
    }
#ifdef ZS_FV_PROF
    const long long tk_pre = wall_clock64() - tk_start;
#endif
    if (tid >= 64) return;  // the engine is one wave, every lane running the same scalar code
#ifdef ZS_FV_PROF
    long long te0 = wall_clock64();
#endif
    if (!s.cont && e.avail_end > 0) {
        if (s.fv_end >= 0) {  // DeflateFast: the pending pre-insert finds the nearest inserted position of its bucket
            const uint32_t *gb = s.ins_bits;
            le_restore_finish(e, p, lk, ss.preins, [gb](int64_t c) { return ((gb[c >> 5] >> (c & 31)) & 1u) != 0; });
        } else {
            le_restore_finish(e, p, lk, ss.preins);
        }
    }
    __syncthreads();
    le_run(e, level, tid, 64);
#ifdef ZS_FV_PROF
    if (tid == 0 && blockIdx.x == 0)
        printf("TAILPROF setup=%lld window+clear=%lld prev+head=%lld convert=%lld; before the engine (with the searches ahead) %lld; kernel so far %lld; engine ticks=%lld syms=%lld: refill=%lld insert=%lld match=%lld tally=%lld flush=%lld\n", pf_restore[0], pf_restore[1], pf_restore[2], pf_restore[3], tk_pre, wall_clock64() - tk_start, wall_clock64() - te0,
               (long long)(e.nsyms - body_syms), e.pf[0], e.pf[1], e.pf[2], e.pf[3], e.pf[4]);
#endif
    if (tid == 0 && e.stopped && e.wr_blk)  // the Writes the engine did not get to: no block of this run lies behind their start
        for (int w = e.cur_wr + 1; w < e.n_wr; w++) e.wr_blk[w] = 0x7FFFFFFF;
    if (tid == 0) {
        ss.nsyms = (uint32_t)e.nsyms;
        ss.nblocks = e.nblocks;
    }
    if (!ps || s.final_run) return;
    // ---- the stream goes on: the wave leaves the engine for the next run (its lanes all hold the same scalars)
    __syncthreads();
    for (int i = tid * 16; i < kWindowSize + 512; i += 64 * 16) *(uint4 *)(ps->window + i) = *(const uint4 *)(smem + i);
    for (int i = tid * 8; i < kWSize; i += 64 * 8) *(uint4 *)(ps->prev + i) = *(const uint4 *)(e.prev + i);
    if (e.head != ps->head)
        for (int i = tid * 8; i < kHashSize; i += 64 * 8) *(uint4 *)(ps->head + i) = *(const uint4 *)(e.head + i);
    const int pending = (int)(e.nsyms - e.block_sym_start);
    for (int i = tid; i < pending; i += 64) ps->syms[i] = e.syms[e.block_sym_start + i];
    if (tid == 0) {
        // (a resumed run works in buffer positions: the state goes back into stream positions)
        const int64_t po = s.resume ? s.persist_off : 0;
        ps->base = e.base + po, ps->avail_end = e.avail_end + po, ps->block_start_abs = e.block_start_abs + po;
        ps->strstart = e.strstart, ps->lookahead = e.lookahead, ps->match_length = e.match_length, ps->match_start = e.match_start;
        ps->match_available = e.match_available, ps->prev_length = e.prev_length, ps->prev_match = e.prev_match;
        ps->pending_syms = pending;
        ps->stopped = e.stopped, ps->good_prev = e.prev_length >= lv.good ? 1 : 0;
    }
}

constexpr int kTailLds = (int)kScratchHead + 2 * kWSize + 4096;  // window, prev, CRC tables
constexpr int kFastRunLds = kTailLds + kHeadCacheLds;          // ... and the cache in front of the run's hash heads (zs_lit_engine.h)
static_assert(kFastRunLds + 1024 <= 160 * 1024, "a run's LDS");

// ------------------------------------------------------------------ KF: DeflateFast by speculative chunk runs
// One workgroup per run (1024 threads restore the engine state, wave 0 runs Deflate.Fast.cs:20-128 literally).
__global__ __launch_bounds__(1024) void zs_fast_run_kernel(const StreamDesc *sd, const uint2 *work, const uint16_t *link,
                                                           uint32_t *run_syms, uint32_t *run_bits, uint8_t *run_scratch,
                                                           FastRunOut *outs, const uint32_t *crc_tab_g, LevelCfg lv, int strategy,
                                                           int hash_variant) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ BlockRec dummy_blk[2];
    const uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int j = (int)w.y;
    const int64_t run = (int64_t)s.run_off + j;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int64_t cj = (int64_t)j * s.run_chunk;
    const bool last = j + 1 == s.fast_runs;
    int64_t x = cj - kFastWarm;
    if (x < 0) x = 0;
    uint8_t *sc = run_scratch + run * kFastRunScratch;
    uint32_t *bits = run_bits + run * kFastRunBitWords;
    for (int64_t i = tid; i < kFastRunBitWords * (s.fast_runs == 1 ? (int64_t)s.run_slots : 1); i += nth) bits[i] = 0;
    LitEngine e;
    le_defaults(e);
    e.window = smem;
    e.prev = (uint16_t *)(smem + kScratchHead);
    e.head = (uint16_t *)sc;
    uint32_t *head32 = (uint32_t *)(sc + 2 * kHashSize);
    uint32_t *tabl = (uint32_t *)(smem + kScratchHead + 2 * kWSize);
    load_crc_tab(tabl, crc_tab_g);
    e.crc_tab = tabl;
    e.hc_val = (uint16_t *)(smem + kTailLds), e.hc_nib = (uint32_t *)(smem + kTailLds + 2 * kHeadCache), e.hc_pres = e.hc_nib + kHeadCache / 8;
    e.hc_claim = e.hc_pres + kHashSize / 32;
    for (int i = tid; i < kHeadCache / 32; i += nth) e.hc_claim[i] = 0;
    e.data = s.in;
    e.n = s.n;
    e.lv = lv;
    e.strategy = strategy;
    e.hash_variant = hash_variant;
    e.syms = run_syms + run * kFastRunSyms;
    e.blocks = dummy_blk;
    e.no_blocks = 1;
    e.stop_abs = last ? -1 : cj + s.run_chunk;
    e.mark_abs = cj;
    e.ins_bits = bits;
    e.ins_base = x;
    e.ev_log = outs[run].ev;
    const uint16_t *lk = link + s.pos_off;
    const int k_done = x == 0 ? 0 : refills_fired_at(x, s.kl);
    le_restore(e, x, (int64_t)kWSize * k_done, read_end_before(k_done + 1), kR, 0, lk, -1, tid, nth);
    for (int i = tid; i < kHashSize; i += nth) head32[i] = 0;
    __syncthreads();
    if (e.avail_end > 0) {
        int64_t lo = x - (kWSize - 1);
        if (lo < e.base) lo = e.base;
        if (lo < 0) lo = 0;
        for (int64_t q = lo + tid; q < x; q += nth) {
            le_restore_prev(e, q, lk);
            atomicMax(&head32[le_bucket(e, q)], (uint32_t)(q - e.base) + 1u);
        }
        __syncthreads();
        for (int i = tid; i < kHashSize; i += nth) e.head[i] = (uint16_t)(head32[i] ? head32[i] - 1 : 0);
    }
    __syncthreads();
    // the cache in front of the table (zs_lit_engine.h): which buckets have a head at all, and in every slot the most recent
    // of the four buckets that share it
    for (int w = tid; w < kHashSize / 32; w += nth) {
        uint32_t bits_w = 0;
        for (int k = 0; k < 32; k++) bits_w |= (uint32_t)(e.head[32 * w + k] != 0) << k;
        e.hc_pres[w] = bits_w;
    }
    for (int w = tid; w < kHeadCache / 8; w += nth) {
        uint32_t nw = 0;
        for (int k = 0; k < 8; k++) {
            const int i = 8 * w + k;
            int best_v = 0, best_t = 0;
            for (int t = 0; t < 4; t++) {
                const int v = e.head[t * kHeadCache + i];
                if (v > best_v) best_v = v, best_t = t;
            }
            e.hc_val[i] = (uint16_t)best_v;
            if (best_v) nw |= (8u | (uint32_t)best_t) << (4 * k);
        }
        e.hc_nib[w] = nw;
    }
    __syncthreads();
#ifdef ZS_FV_PROF
    const long long tk_pre = wall_clock64();
#endif
    if (tid >= 64) return;
    le_run_fast_hot(e, tid);
    le_flush_ins(e);
#ifdef ZS_FV_PROF
    if (tid == 0 && (blockIdx.x == 3 || blockIdx.x == 100))
        printf("FASTRUN %d: engine ticks(100MHz)=%lld syms=%lld: refill+checks=%lld insert=%lld match=%lld tally+advance=%lld flush=%lld\n", (int)blockIdx.x,
               wall_clock64() - tk_pre, (long long)e.nsyms, e.pf[0], e.pf[1], e.pf[2], e.pf[3], e.pf[4]);
#endif
    if (tid == 0) {
        FastRunOut &o = outs[run];
        o.mark_pos = e.mark_pos, o.mark_nsyms = e.mark_nsyms;
        o.end_pos = (e.stop_abs >= 0 && e.lookahead > 0) ? e.base + e.strstart : (int64_t)s.n;
        if (e.stop_abs >= 0 && e.base + e.strstart < e.stop_abs) o.end_pos = -1;  // ran out of input before its stop: not a valid hand-over
        o.nsyms = e.nsyms;
        o.final_base = e.base;
        o.sym_dst = 0;
        o.ok = 1, o.n_ev = e.n_ev;
        o.n_match = e.n_match;
        if (j == 0) o.mark_pos = 0, o.mark_nsyms = 0;
    }
}

// Is a stream worth the speculative runs?  They verify on data whose parse falls back into step whatever came before it --
// periodic data: image rows, runs, tables -- and never on text, where the attempt costs as much as the sweeps that then do
// the work (8 MiB of text: 140 ms thrown away in front of 190).  On periodic data the previous position of a bucket lies at
// one and the same distance for position after position: the share of positions whose link equals their neighbour's is 0.97
// for image rows, 0.80 for ptt5, 0.61 for kennedy.xls, 0.2-0.33 for text.  64 windows of 256 positions are looked at.
__global__ __launch_bounds__(256) void zs_fast_probe_kernel(const StreamDesc *sd, const uint16_t *link, int32_t *not_periodic) {
    __shared__ int same;
    const StreamDesc s = sd[blockIdx.x];
    if (s.fast_runs <= 0) {
        if (threadIdx.x == 0) not_periodic[blockIdx.x] = 0;
        return;
    }
    if (threadIdx.x == 0) same = 0;
    __syncthreads();
    const uint16_t *lk = link + s.pos_off;
    const int64_t span = (int64_t)s.n - 8 - kWSize;
    int mine = 0;
    for (int w = 0; w < 64; w++) {
        const int64_t p = kWSize + span * w / 64 + threadIdx.x;
        if (p + 8 < s.n) mine += lk[p] != 0 && lk[p] == lk[p - 1];
    }
    atomicAdd(&same, mine);
    __syncthreads();
    // (below 4 MiB a failed attempt is not small beside the sweeps it falls back to: only what is nearly all period -- image rows,
    // runs, a short period: 0.97 and more; ptt5's 0.80 stays with the sweeps)
    if (threadIdx.x == 0) not_periodic[blockIdx.x] = s.n >= kFastMinInput ? same * 4 < 64 * 256 * 3 : same * 16 < 64 * 256 * 15;
}

// run j (j >= 1) is exact iff it entered its chunk where run j-1 stopped, with the same strings inserted in the
// 32 KiB before that loop-top (chain walks never look further back)
__global__ __launch_bounds__(256) void zs_fast_verify_kernel(const StreamDesc *sd, const uint2 *work, const uint32_t *run_bits,
                                                             FastRunOut *outs, int32_t *stream_fail) {
    __shared__ int bad;
    const uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int j = (int)w.y;
    if (j == 0) return;
    const int64_t run = (int64_t)s.run_off + j;
    const FastRunOut a = outs[run - 1], b = outs[run];
    if (threadIdx.x == 0) bad = (b.mark_pos < 0 || b.mark_pos != a.end_pos);
    __syncthreads();
    if (!bad) {
        const int64_t xa = (j - 1) * (int64_t)s.run_chunk - kFastWarm < 0 ? 0 : (j - 1) * (int64_t)s.run_chunk - kFastWarm;
        const int64_t xb = (int64_t)j * s.run_chunk - kFastWarm < 0 ? 0 : (int64_t)j * s.run_chunk - kFastWarm;
        const uint32_t *ba = run_bits + (run - 1) * kFastRunBitWords, *bb = run_bits + run * kFastRunBitWords;
        int64_t lo = b.mark_pos - kWSize;
        if (lo < 0) lo = 0;
        int mism = 0;
        for (int64_t p = lo + threadIdx.x; p < b.mark_pos; p += 256) {
            const int64_t ia = p - xa, ib = p - xb;
            const uint32_t va = (ba[ia >> 5] >> (ia & 31)) & 1u;
            const uint32_t vb = ib >= 0 ? (bb[ib >> 5] >> (ib & 31)) & 1u : va;  // before run j's start: restored as inserted; require run j-1 agrees
            const uint32_t want = ib >= 0 ? vb : 1u;
            mism |= (va != want);
#ifdef ZS_FV_PROF
            if (va != want && j == 1) printf("VERIFY run %d: position %lld run before has %u, this run %u (mark %lld)\n", j, (long long)p, va, want, (long long)b.mark_pos);
#endif
        }
        if (mism) bad = 1;
        // refills inside the compared window must have happened at the same loop-tops (the string they pre-insert
        // and the prev[] 2-cycle they can leave are not visible in the bitmap)
        if (threadIdx.x == 0) {
            for (int u = 0; u < a.n_ev; u++) {
                if (a.ev[u] < lo - 1 || a.ev[u] > b.mark_pos) continue;
                bool found = false;
                for (int v = 0; v < b.n_ev; v++) found |= b.ev[v] == a.ev[u];
                if (!found) bad = 1;
#ifdef ZS_FV_PROF
                if (!found && j == 1) printf("VERIFY run %d: event of the run before at %lld not in this run\n", j, (long long)a.ev[u]);
#endif
            }
            for (int v = 0; v < b.n_ev; v++) {
                if (b.ev[v] < lo - 1 || b.ev[v] > b.mark_pos) continue;
                bool found = false;
                for (int u = 0; u < a.n_ev; u++) found |= a.ev[u] == b.ev[v];
                if (!found) bad = 1;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && bad) {
        outs[run].ok = 0;
        stream_fail[w.x] = 1;
    }
}

// per stream: where each run's own symbols go; block cuts every kBlockSyms symbols.  One workgroup per stream, a prefix sum over
// its runs (one thread walking 256 runs was 0.06 ms of sparse64's 3.4).
__global__ __launch_bounds__(256) void zs_fast_plan_kernel(const StreamDesc *sd, StreamState *st, FastRunOut *outs, int nstreams) {
    __shared__ long long part[256];
    __shared__ long long carry;
    const int si = blockIdx.x, tid = threadIdx.x;
    if (si >= nstreams) return;
    const StreamDesc s = sd[si];
    if (s.fast_runs <= 0) return;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < s.fast_runs; j0 += 256) {
        const int j = j0 + tid;
        const long long own = j < s.fast_runs ? (long long)(outs[s.run_off + j].nsyms - outs[s.run_off + j].mark_nsyms) : 0;
        part[tid] = own;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {  // inclusive prefix sum
            const long long x = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += x;
            __syncthreads();
        }
        if (j < s.fast_runs) outs[s.run_off + j].sym_dst = carry + part[tid] - own;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) {
        const long long total = carry;
        st[si].nsyms = (uint32_t)total;
        st[si].nblocks = (int32_t)(total / kBlockSyms) + 1;
        // a stream whose symbol count is a multiple of kBlockSyms ends with an empty last block, like the reference
    }
}

// one workgroup per run: copy the run's own symbols into place; the byte positions of the block cuts that fall inside the run
// from the symbols' own lengths (a cut every kBlockSyms symbols: at most a few per run, none in most -- one thread walking
// all of a run's symbols for them was 0.5 ms of sparse64's 7)
__global__ __launch_bounds__(256) void zs_fast_stitch_kernel(const StreamDesc *sd, const uint2 *work, const uint32_t *run_syms,
                                                             const FastRunOut *outs, uint32_t *syms, int32_t *blk_end,
                                                             int32_t *blk_top) {
    __shared__ long long part[256];
    const uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    const int64_t run = (int64_t)s.run_off + w.y;
    const FastRunOut o = outs[run];
    const uint32_t *src = run_syms + run * kFastRunSyms + o.mark_nsyms;
    uint32_t *dst = syms + s.sym_off + o.sym_dst;
    const int64_t cnt = o.nsyms - o.mark_nsyms;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < cnt; i += 256) dst[i] = src[i];
    // symbol i of the run is the stream's symbol sym_dst + i; the first one that ends a block:
    int64_t i_cut = (kBlockSyms - 1) - o.sym_dst % kBlockSyms, i_done = 0, pos = o.mark_pos;
    for (; i_cut < cnt; i_cut += kBlockSyms) {
        long long sum = 0;
        for (int64_t i = i_done + tid; i < i_cut; i += 256) {
            const uint32_t v = src[i];
            sum += (v >> 16) ? (long long)(v & 0xFFFF) + 3 : 1;
        }
        part[tid] = sum;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (tid < k) part[tid] += part[tid + k];
            __syncthreads();
        }
        const int64_t start = pos + part[0];
        __syncthreads();
        const uint32_t v = src[i_cut];
        pos = start + ((v >> 16) ? (int64_t)(v & 0xFFFF) + 3 : 1);
        if (tid == 0) {
            const int64_t g = o.sym_dst + i_cut;
            blk_end[s.blk_off + g / kBlockSyms] = (int32_t)pos;
            blk_top[s.blk_off + g / kBlockSyms] = (int32_t)start;  // DeflateFast flushes at the loop-top that emitted the symbol
        }
        i_done = i_cut + 1;
    }
}

// per stream: block records from the cuts (Flush_block_only, Deflate.cs:951-956)
__global__ __launch_bounds__(64) void zs_fast_blocks_kernel(const StreamDesc *sd, const StreamState *st, const FastRunOut *outs,
                                                            const int32_t *blk_end, const int32_t *blk_top, BlockRec *blocks,
                                                            int nstreams) {
    const int si = blockIdx.x;
    const StreamDesc s = sd[si];
    if (s.fast_runs <= 0) return;
    const int nb = st[si].nblocks;
    const int64_t total = st[si].nsyms;
    for (int b = threadIdx.x; b < nb; b += 64) {
        BlockRec r;
        r.start = b ? blk_end[s.blk_off + b - 1] : 0;
        r.sym_start = (int64_t)b * kBlockSyms;
        if (b + 1 < nb) {
            r.stored_len = (int32_t)(blk_end[s.blk_off + b] - r.start);
            r.nsyms = kBlockSyms;
            r.can_store = r.start >= (int64_t)kWSize * refills_fired_at(blk_top[s.blk_off + b], s.kl);
            r.eof = 0;
        } else {
            r.stored_len = (int32_t)((int64_t)s.n - r.start);
            r.nsyms = (int32_t)(total - r.sym_start);
            r.can_store = r.start >= outs[s.run_off + s.fast_runs - 1].final_base;
            r.eof = 1;
        }
        blocks[s.blk_off + b] = r;
    }
}

// ------------------------------------------------------------------ KS: DeflateFast (levels 1-3) as window-wide sweeps of a workgroup
// (round 3's form -- one wave taking the stream through windows of 64 positions over per-position candidate lists, zs_fast_vec.h,
// which the CPU model still runs as mode "fvec" -- did 9 MB/s on one stream and moved 76x the algorithmic bytes through HBM)
#include "zs_fast_sweep.hip"
#include "zs_rle.hip"

// Build_tree (Trees.cs:404-501) by one wave.  The priority queue is sifted by lane 0 exactly as the reference does
// (zs_core.h build_tree: tie-breaking decides the tree); what surrounds it is data-parallel and costs as much as the
// queue on blocks with a full alphabet (binary or image data: ~260 of 286 symbols in use):
//   * the queue's initial contents: ballot-compaction of the symbols in use;
//   * Gen_bitlen (Trees.cs:999-1109): a node's length is its depth -- pointer jumping over (parent, distance) words
//     instead of one dependent lookup per node; any depth beyond max_length sends the tree to the reference's
//     overflow repair (the sequential gen_bitlen, untouched);
//   * Gen_codes (Trees.cs:1123-1151): `next_code[len]++` in symbol order is a returning LDS add per symbol (the LDS
//     applies the lanes of one instruction in lane order, see K1).
// `pd`: kHeapSize words, `nc`: 16 words of LDS scratch.  Every lane returns max_code.
__device__ int build_tree_wave(TreeWork &w, uint32_t *hk, uint32_t *pd, uint32_t *nc, CtData *tree, const TreeDesc &d, int lane) {
    int len = 0, max_code = -1;
    for (int base = 0; base < d.elems; base += 64) {
        const int n = base + lane;
        const uint32_t f = n < d.elems ? tree[n].fc : 0;
        const uint64_t nz = __ballot(f != 0);
        if (f) hk[len + 1 + __builtin_popcountll(nz & lanemask_lt())] = hk_pack(f, 0, (uint32_t)n);
        else if (n < d.elems) tree[n].dl = 0;
        if (nz) max_code = base + 63 - __builtin_clzll(nz);
        len += __builtin_popcountll(nz);
    }
    while (len < 2) {  // force at least two codes of non-zero frequency
        const int node = max_code < 2 ? ++max_code : 0;
        len++;
        if (lane == 0) {
            hk[len] = hk_pack(1, 0, (uint32_t)node);
            tree[node].fc = 1;
            w.opt_len--;
            if (d.which != 2) w.static_len -= desc_static_len(d, node);
        }
    }
    __threadfence_block();
    if (lane == 0) {
        w.heap_max = kHeapSize;
        int hl = len;
        for (int n = hl / 2; n >= 1; n--) pqdownheap_packed(hk, hl, n);
        int node = d.elems;
        do {
            const uint32_t n = hk[1];
            hk[1] = hk[hl--];
            pqdownheap_packed(hk, hl, 1);
            const uint32_t m = hk[1];
            w.heap[--w.heap_max] = (uint16_t)(n & 1023u);
            w.heap[--w.heap_max] = (uint16_t)(m & 1023u);
            const uint32_t f = (n >> 16) + (m >> 16), dn = (n >> 10) & 63u, dm = (m >> 10) & 63u;
            tree[node].fc = (uint16_t)f;
            tree[n & 1023u].dl = tree[m & 1023u].dl = (uint16_t)node;
            hk[1] = hk_pack(f, (dn >= dm ? dn : dm) + 1, (uint32_t)node);
            node++;
            pqdownheap_packed(hk, hl, 1);
        } while (hl >= 2);
        w.heap[--w.heap_max] = (uint16_t)(hk[1] & 1023u);
        w.heap_len = hl;
    }
    __threadfence_block();
    // ---- lengths = depths
    const int hmax = w.heap_max, root = w.heap[hmax];
    for (int h = hmax + lane; h < kHeapSize; h += 64) {
        const int n = w.heap[h];
        pd[n] = h == hmax ? ((uint32_t)n << 8) : (((uint32_t)tree[n].dl << 8) | 1u);
    }
    __threadfence_block();
    for (int r = 0; r < 6; r++) {  // 2^6 >= any depth a tree over <= 65535 counts can have
        for (int h = hmax + lane; h < kHeapSize; h += 64) {
            const int n = w.heap[h];
            const uint32_t v = pd[n], pv = pd[v >> 8];
            pd[n] = (pv & ~0xFFu) | ((v & 0xFFu) + (pv & 0xFFu));  // one word: always a consistent (ancestor, distance) pair
        }
        __threadfence_block();
    }
    bool over = false;
    for (int h = hmax + lane; h < kHeapSize; h += 64) over |= (int)(pd[w.heap[h]] & 0xFFu) > d.max_length;
    (void)root;
    if (__ballot(over)) {
        if (lane == 0) gen_bitlen(w, tree, max_code, d);  // the reference's walk with its overflow repair
        __threadfence_block();
    } else {
        if (lane < 16) nc[lane] = 0;
        __threadfence_block();
        int opt = 0, stat = 0;
        for (int h = hmax + lane; h < kHeapSize; h += 64) {
            const int n = w.heap[h];
            const int bits = (int)(pd[n] & 0xFFu);
            tree[n].dl = (uint16_t)bits;
            if (n <= max_code) {
                atomicAdd(&nc[bits], 1u);
                const int xbits = desc_extra(d, n), f = tree[n].fc;
                opt += f * (bits + xbits);
                if (d.which != 2) stat += f * (desc_static_len(d, n) + xbits);
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) opt += __shfl_xor(opt, o), stat += __shfl_xor(stat, o);
        __threadfence_block();
        if (lane <= kMaxBits) w.bl_count[lane] = (uint16_t)nc[lane];
        if (lane == 0) w.opt_len += opt, w.static_len += stat;
        __threadfence_block();
    }
    // ---- codes
    if (lane == 0) {
        unsigned code = 0;
        nc[0] = 0;
        for (int bits = 1; bits <= kMaxBits; bits++) {
            code = (code + w.bl_count[bits - 1]) << 1;
            nc[bits] = code;
        }
    }
    __threadfence_block();
    for (int base = 0; base <= max_code; base += 64) {
        const int n = base + lane;
        const int l = n <= max_code ? (int)tree[n].dl : 0;
        if (l) {
            const uint32_t code = atomicAdd(&nc[l], 1u);  // lanes in order: the symbols of one length get consecutive codes
            tree[n].fc = (uint16_t)(__brev(code) >> (32 - l));
        }
    }
    __threadfence_block();
    return max_code;
}
// Tr_flush_block's tree phase (zs_core.h build_block_trees) with the wave form of Build_tree; lane 0's return value counts.
__device__ int build_block_trees_wave(TreeWork &w, uint32_t *hk, uint32_t *pd, uint32_t *nc, int stored_len, bool can_store,
                                      int strategy, int lane) {
    if (lane == 0) {
        w.opt_len = w.static_len = 0;
        for (int i = 0; i < kBlCodes; i++) w.bltree[i].fc = 0;
    }
    __threadfence_block();
    const TreeDesc ld = {0, kLCodes, kMaxBits, kLiterals + 1};
    const TreeDesc dd = {1, kDCodes, kMaxBits, 0};
    const TreeDesc bd = {2, kBlCodes, kMaxBlBits, 0};
    const int lmax = build_tree_wave(w, hk, pd, nc, w.ltree, ld, lane);
    const int dmax = build_tree_wave(w, hk, pd, nc, w.dtree, dd, lane);
    if (lane == 0) {
        w.l_max_code = lmax, w.d_max_code = dmax;
        scan_tree(w, w.ltree, lmax);
        scan_tree(w, w.dtree, dmax);
    }
    __threadfence_block();
    build_tree_wave(w, hk, pd, nc, w.bltree, bd, lane);
    int type = 2;
    if (lane == 0) {
        int mb;
        for (mb = kBlCodes - 1; mb >= 3; mb--)
            if (w.bltree[bl_order(mb)].dl != 0) break;
        w.max_blindex = mb;
        w.opt_len += 3 * (mb + 1) + 5 + 5 + 4;
        int opt_lenb = (w.opt_len + 3 + 7) >> 3;
        const int static_lenb = (w.static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
        if (stored_len + 4 <= opt_lenb && can_store) type = 0;
        else if (strategy == kFixed || static_lenb == opt_lenb) type = 1;
    }
    return type;
}

// ------------------------------------------------------------------ K6b
// The block list with the blocks that exist in front.  The list is laid out before the parse, with room for every stream's
// worst case: a 32 KiB stream has four entries and uses one.  Workgroups are handed to the XCDs, and inside an XCD to its
// shader engines, in order and round robin, so live entries at a stride of four all land on a quarter of the CUs: the tree
// and bit-emission kernels of 4096 x 32 KiB streams ran 3-4 x longer than those of the same bytes in long streams (and no
// faster with fewer dead entries per live one, as long as the stride stayed a power of two).  For batches the list is
// rewritten once the block counts are known: a prefix sum over the streams, then (stream, block) pairs for the live
// entries and a mark on the rest.
constexpr uint32_t kNoWork = 0xFFFFFFFFu;
__global__ __launch_bounds__(1024) void zs_live_scan_kernel(const StreamState *st, int n, int32_t *live_pre) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int32_t v = i < n && !st[i].deferred ? st[i].nblocks : 0;
        int32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int32_t before = carry;
        for (int k = 0; k < wave; k++) before += wsum[k];
        if (i < n) live_pre[i] = before + inc - v;
        __syncthreads();
        if (tid == 1023) carry = before + inc;
        __syncthreads();
    }
    if (tid == 0) live_pre[n] = carry;
}
__global__ __launch_bounds__(256) void zs_live_fill_kernel(const int32_t *live_pre, int n, uint32_t n_items, uint2 *work) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_items) return;
    if ((int32_t)g >= live_pre[n]) {
        work[g] = make_uint2(kNoWork, 0);
        return;
    }
    int lo = 0, hi = n - 1;  // the last stream whose first entry is <= g (streams without blocks share their successor's start)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (live_pre[mid] <= (int32_t)g) lo = mid;
        else hi = mid - 1;
    }
    work[g] = make_uint2((uint32_t)lo, g - (uint32_t)live_pre[lo]);
}

// ------------------------------------------------------------------ K7
// One workgroup per block: histogram the block's symbols (Tr_tally_*), then wave 0 replays Build_tree x3 exactly
// (build_tree_wave) and picks the block type.
__global__ __launch_bounds__(256) void zs_trees_kernel(const StreamDesc *sd, const StreamState *st, const uint2 *work,
                                                       const uint32_t *syms, const BlockRec *blocks, TreeWork *trees,
                                                       BlockInfo *info, int strategy, int level, int phase) {
    __shared__ TreeWork tw;
    __shared__ uint32_t hl[kLCodes], hd[kDCodes];
    __shared__ uint32_t hk[kHeapSize + 8];  // Build_tree's priority queue (the sift reads a few entries past the end)
    __shared__ uint32_t pd[kHeapSize + 1], nc[16];  // (ancestor, distance) words of the depth pass; per-length counters
    uint2 w = work[blockIdx.x];
    if (w.x == kNoWork || st[w.x].deferred) return;
    const StreamDesc s = sd[w.x];
    const int b = (int)w.y;
    // phase 0: the blocks that end inside the bulk parse; phase 1: the rest; 2: all
    const int nb_body = s.fast_runs > 0 ? 0 : (int)(st[w.x].body_syms / kBlockSyms);
    if (phase == 0 ? b >= nb_body : ((phase == 1 && b < nb_body) || b >= st[w.x].nblocks)) return;
    const BlockRec r = blocks[s.blk_off + b];
    if (level == 0 && r.nsyms == 0 && s.plan_nblk > 0) {
        // DeflateStored tallies nothing and level 0 skips the tree comparison (Trees.cs:601-620): stored while the block
        // start is in the window, else an empty static block (3 header bits + END_BLOCK)
        if (threadIdx.x == 0) {
            BlockInfo bi;
            bi.type = r.can_store ? 0 : 1;
            bi.bits = r.can_store ? 0 : 10;
            bi.bit_start = 0;
            info[s.blk_off + b] = bi;
        }
        return;
    }
    for (int i = threadIdx.x; i < kLCodes; i += blockDim.x) hl[i] = 0;
    for (int i = threadIdx.x; i < kDCodes; i += blockDim.x) hd[i] = 0;
    __syncthreads();
    const uint32_t *sy = syms + s.sym_off + r.sym_start;
    for (int i = threadIdx.x; i < r.nsyms; i += blockDim.x) {
        uint32_t v = sy[i];
        int dist = (int)(v >> 16), lc = (int)(v & 0xFFFF);
        if (dist == 0) atomicAdd(&hl[lc], 1u);
        else {
            atomicAdd(&hl[length_code(lc) + kLiterals + 1], 1u);
            atomicAdd(&hd[dist_code(dist - 1)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHeapSize; i += blockDim.x) {
        tw.ltree[i].fc = i < kLCodes ? (uint16_t)hl[i] : 0;
        tw.ltree[i].dl = 0;
    }
    for (int i = threadIdx.x; i < 2 * kDCodes + 1; i += blockDim.x) {
        tw.dtree[i].fc = i < kDCodes ? (uint16_t)hd[i] : 0;
        tw.dtree[i].dl = 0;
    }
    for (int i = threadIdx.x; i < 2 * kBlCodes + 1; i += blockDim.x) tw.bltree[i].fc = 0, tw.bltree[i].dl = 0;
    __syncthreads();
    if (threadIdx.x == 0) tw.ltree[kEndBlock].fc = 1;
    __syncthreads();
    if (threadIdx.x < 64) {
        int type = build_block_trees_wave(tw, hk, pd, nc, r.stored_len, r.can_store != 0, strategy, (int)threadIdx.x);
        if (threadIdx.x == 0) {
        // level 0 skips the tree comparison: opt_lenb = static_lenb = stored_len + 5, i.e. stored when the block
        // start is still in the window, else static trees (Trees.cs:601-620)
        if (level == 0) type = r.can_store ? 0 : 1;
        BlockInfo bi;
        bi.type = type;
        bi.bits = type == 1 ? 3 + tw.static_len : type == 2 ? 3 + tw.opt_len : 0;
        bi.bit_start = 0;
        info[s.blk_off + b] = bi;
        }
    }
    __syncthreads();
    uint32_t *dst = (uint32_t *)&trees[s.blk_off + b];
    const uint32_t *src = (const uint32_t *)&tw;
    for (int i = threadIdx.x; i < (int)(sizeof(TreeWork) / 4); i += blockDim.x) dst[i] = src[i];
}

// ------------------------------------------------------------------ K8
// One workgroup per stream: bit offset of every block (Send_bits is a pure concatenation; stored blocks and
// the last block align to a byte), zlib header, Adler-32 trailer, total length.  The scan over blocks is
// sequential (alignment depends on the absolute bit position) but runs out of LDS; the Adler pieces are
// combined by a tree.
// OR `nbits` (<= 57) of `v` into the zeroed output at bit position `pos`.
__device__ __forceinline__ void or_bits(uint8_t *out, int64_t pos, uint64_t v, int nbits) {
    if (nbits == 0) return;
    uint32_t *w = (uint32_t *)((uintptr_t)(out + (pos >> 3)) & ~(uintptr_t)3);
    int sh = (int)(pos - ((int64_t)((uint8_t *)w - out) << 3));  // 0..31
    uint64_t lo = v << sh;
    uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32);
    uint32_t w2 = sh ? (uint32_t)(v >> (64 - sh)) : 0;
    if (w0) atomicOr(w, w0);
    if (w1) atomicOr(w + 1, w1);
    if (w2) atomicOr(w + 2, w2);
}
struct OrBitsAt {  // FlushAcct's put: OR into the zeroed stream, never past the caller's capacity
    uint8_t *out;
    int64_t cap;
    int64_t bit_base;  // stream bit position of out[0] (0 unless the run continues an incremental stream)
    __device__ void operator()(int64_t pos, uint32_t v, int nbits) const {
        pos -= bit_base;
        if ((pos >> 3) + 12 <= cap) or_bits(out, pos, v, nbits);
    }
};
__global__ __launch_bounds__(256) void zs_offsets_kernel(const StreamDesc *sd, StreamState *st, const BlockRec *blocks,
                                                         BlockInfo *info, const TreeWork *trees, const uint32_t *adler_pieces,
                                                         int level, int nstreams) {
    __shared__ int32_t sh_type[1024], sh_bits[1024], sh_len[1024], sh_eof[1024];
    __shared__ int64_t sh_start[1024];
    __shared__ uint32_t ad_v[256];
    __shared__ uint64_t ad_len[256];
    __shared__ int64_t sh_pos, sh_wsum[4];
    __shared__ int sh_bad;
    // FlushMode Partial / Sync / Full: thread 0 replays Deflate.Compress's chunk accounting (zs_core.h, FlushAcct)
    FlushAcct fa;
    int fa_w = 0;
    const bool flushing = sd[blockIdx.x].wr_flush != nullptr;
    const int si = blockIdx.x;
    if (st[si].deferred) return;
    const StreamDesc s = sd[si];
    StreamState &ss = st[si];
    const int nb = ss.nblocks;
    LitPersist *ps = s.persist;
    // a run that continues an incremental stream starts where the run before stopped: `bit_base` is the stream bit position
    // of this run's out[0], whose low bits (the stream's last, incomplete byte) come with the descriptor
    const int64_t start_bits = s.cont_bits ? ps->fa.bits : 16;
    const int64_t bit_base = s.cont_bits ? (start_bits & ~7LL) : 0;
    if (threadIdx.x == 0) sh_pos = start_bits, sh_bad = 0;
    const OrBitsAt fa_put{s.out, s.out_cap, bit_base};
    if (flushing) {
        if (s.cont_bits) fa = ps->fa;
        else fa_init(fa, s.out_chunk, level, s.raw != 0);
        // the run's first Deflate call (of a stream: it delivers the header); a run that took the stream over in the middle of
        // a Write goes on inside the call the run before was in
        if (!s.mid_write) fa_enter(fa);
    }
    if (threadIdx.x == 0 && s.cont_bits && s.out_cap > 0) s.out[0] = (uint8_t)s.carry_byte;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        int cnt = nb - b0 < 1024 ? nb - b0 : 1024;
        for (int i = threadIdx.x; i < cnt; i += 256) {
            const BlockInfo bi = info[s.blk_off + b0 + i];
            const BlockRec r = blocks[s.blk_off + b0 + i];
            sh_type[i] = bi.type, sh_bits[i] = bi.bits, sh_len[i] = r.stored_len, sh_eof[i] = r.eof;
        }
        __syncthreads();
        // Send_bits is a concatenation: when no block of the batch needs byte alignment before its end (no stored block,
        // end-of-stream only on the last one) the starts are a prefix sum; otherwise thread 0 walks the blocks
        int irregular = 0;
        for (int i = threadIdx.x; i < cnt; i += 256) irregular |= (sh_type[i] == 0) || (sh_eof[i] && i != cnt - 1);
        irregular = __syncthreads_or(irregular | (int)flushing);
        if (!irregular) {
            // thread t sums 4 consecutive blocks, then a scan over the 256 partial sums
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            int64_t b4[4], mine = 0;
            for (int k = 0; k < 4; k++) {
                const int i = threadIdx.x * 4 + k;
                b4[k] = i < cnt ? sh_bits[i] : 0;
                mine += b4[k];
            }
            int64_t inc = mine;
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t x = __shfl_up(inc, o);
                if (lane >= o) inc += x;
            }
            if (lane == 63) sh_wsum[wave] = inc;
            __syncthreads();
            int64_t before = sh_pos;
            for (int k = 0; k < wave; k++) before += sh_wsum[k];
            int64_t pos = before + inc - mine;
            for (int k = 0; k < 4; k++) {
                const int i = threadIdx.x * 4 + k;
                if (i < cnt) sh_start[i] = pos;
                pos += b4[k];
            }
            __syncthreads();
            if (threadIdx.x == 255) {
                int64_t end = pos;  // thread 255 holds the end of the last block of the batch
                if (sh_eof[cnt - 1] & 1) end = (end + 7) & ~7LL;
                sh_pos = end;
            }
        } else if (threadIdx.x == 0) {
            int64_t pos = sh_pos;
            for (int i = 0; i < cnt; i++) {
                if (flushing)  // Writes that began before this block was flushed: each is a new Deflate call
                    while (fa_w + 1 < s.n_wr && s.wr_blk[fa_w + 1] <= b0 + i) fa_w++, fa_enter(fa);
                sh_start[i] = pos;
                if (sh_type[i] == 0) {
                    // the reference copies a stored block through its pending buffer (64 KiB; 32 KiB at level 0) and throws
                    // when it does not fit (Deflate.cs:710-722, 757-761): report instead of emitting what it cannot produce
                    if (sh_len[i] + 5 > (level == 0 ? 32768 : 65536)) sh_bad = 1;
                    pos += 3;
                    pos = (pos + 7) & ~7LL;
                    pos += 32 + 8LL * sh_len[i];
                } else {
                    pos += sh_bits[i];
                }
                if (sh_eof[i] & 1) pos = (pos + 7) & ~7LL;
                if (flushing) {
                    fa.bits = pos;
                    fa.last_eob_len = sh_type[i] == 0 ? 8 : sh_type[i] == 1 ? 7 : trees[s.blk_off + b0 + i].ltree[kEndBlock].dl;
                    const int f = sh_eof[i] >> 1;
                    if (f) fa_end_of_write(fa, f, fa_put);
                    else fa_after_block(fa);
                    pos = fa.bits;
                }
            }
            sh_pos = pos;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += 256) info[s.blk_off + b0 + i].bit_start = sh_start[i] - bit_base;
        __syncthreads();
    }
    // Adler-32 of the whole input: tree combine of the 64 KiB pieces (adler_combine is associative)
    uint32_t acc = 1;   // adler of the empty string
    uint64_t acc_len = 0;
    {
        // thread t folds pieces [t*per, (t+1)*per) left to right
        const int per = (s.n_adler + 255) / 256;
        for (int k = 0; k < per; k++) {
            int i = threadIdx.x * per + k;
            if (i >= s.n_adler) break;
            int64_t len = (int64_t)s.n - (int64_t)i * kAdlerPiece;
            if (len > kAdlerPiece) len = kAdlerPiece;
            acc = adler_combine(acc, adler_pieces[s.adler_off + i], (uint64_t)len);
            acc_len += (uint64_t)len;
        }
        ad_v[threadIdx.x] = acc, ad_len[threadIdx.x] = acc_len;
    }
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        if ((threadIdx.x & (2 * off - 1)) == 0) {
            ad_v[threadIdx.x] = adler_combine(ad_v[threadIdx.x], ad_v[threadIdx.x + off], ad_len[threadIdx.x + off]);
            ad_len[threadIdx.x] += ad_len[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const uint32_t ad = ps ? s.adler_stream : ad_v[0];  // an incremental stream's checksum is kept by its owner
    const int64_t pos = sh_pos;
    ss.adler = ad;
    ss.end_bits = pos;
    if (ps && !s.final_run) {
        // the stream goes on: complete bytes are the run's output, the bits of the last one stay with the stream
        if (flushing) fa.bits = pos, ps->fa = fa, ps->fa_valid = 1;
        const int64_t bytes = (pos - bit_base) >> 3;
        ss.out_len = bytes;
        ss.status = sh_bad ? -2 : (bytes + 8 > s.out_cap ? -5 : 0);
        return;
    }
    int64_t total = (pos - bit_base) / 8 + 4;
    ss.out_len = total;
    if (sh_bad) {
        ss.status = -2;  // ZS_STREAM_ERROR
        return;
    }
    if (total > s.out_cap) {
        ss.status = -5;  // ZS_BUF_ERROR
        return;
    }
    ss.status = 0;
    if (!s.cont_bits) {
        unsigned hdr = zlib_header(level);
        s.out[0] = (uint8_t)(hdr >> 8);
        s.out[1] = (uint8_t)hdr;
    }
    uint8_t *t = s.out + (pos - bit_base) / 8;
    t[0] = (uint8_t)(ad >> 24), t[1] = (uint8_t)(ad >> 16), t[2] = (uint8_t)(ad >> 8), t[3] = (uint8_t)ad;
}

// ------------------------------------------------------------------ K9
struct OrPut {
    uint8_t *out;
    int64_t pos;
    __device__ void operator()(unsigned value, int nbits) {
        or_bits(out, pos, value, nbits);
        pos += nbits;
    }
};
struct LdsTree {
    const uint32_t *t;  // fc | dl << 16
    struct E {
        uint16_t fc, dl;
    };
    __device__ E operator[](int i) const {
        uint32_t v = t[i];
        return {(uint16_t)v, (uint16_t)(v >> 16)};
    }
};
// One workgroup per block.  The block's bit string is assembled in LDS and leaves as whole 32-bit words: thread 0
// puts the block header (and the dynamic trees) at the front; then, 2048 symbols at a time, every thread takes 8
// consecutive symbols, a workgroup scan of their code lengths gives each thread its bit offset, the codes are
// OR-ed into the LDS words (ds_or), and the complete words are stored coalesced.  The partial word at the end of a
// tile is carried into the next tile; only the first and the last word of a block can be shared with its
// neighbours, and only those go out with atomicOr (the output was zeroed by K0).
constexpr int kEbTile = 2048;                            // symbols per tile
constexpr int kEbWords = kEbTile * 48 / 32 + 256;        // worst case 48 bits per symbol + dynamic header (<= 141 words) + carry
struct LdsBitPut {
    uint32_t *w;   // zeroed words
    uint32_t pos;  // bit position
    __device__ void put64(uint64_t v, int nbits) {  // nbits <= 57
        if (nbits == 0) return;
        const uint32_t k = pos >> 5, sh = pos & 31;
        const uint64_t lo = v << sh;
        const uint32_t w0 = (uint32_t)lo, w1 = (uint32_t)(lo >> 32), w2 = sh ? (uint32_t)(v >> (64 - sh)) : 0;
        if (w0) atomicOr(&w[k], w0);
        if (w1) atomicOr(&w[k + 1], w1);
        if (w2) atomicOr(&w[k + 2], w2);
        pos += (uint32_t)nbits;
    }
    __device__ void operator()(unsigned value, int nbits) { put64(value, nbits); }
};
__global__ __launch_bounds__(256) void zs_emit_bits_kernel(const StreamDesc *sd, const StreamState *st, const uint2 *work,
                                                           const uint32_t *syms, const BlockRec *blocks, const TreeWork *trees,
                                                           const BlockInfo *info) {
    __shared__ uint32_t lt[kLCodes], dt[kDCodes];
    __shared__ uint32_t obuf[kEbWords];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t sh_bits;
    uint2 w = work[blockIdx.x];
    if (w.x == kNoWork || st[w.x].deferred) return;
    const StreamDesc s = sd[w.x];
    const int b = (int)w.y;
    if (b >= st[w.x].nblocks || st[w.x].status != 0) return;
    const BlockRec r = blocks[s.blk_off + b];
    const BlockInfo bi = info[s.blk_off + b];
    uint8_t *out = s.out;
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (bi.type == 0) {
        int64_t pos = bi.bit_start;
        if (tid == 0) or_bits(out, pos, (uint64_t)(r.eof & 1), 3);
        int64_t byte = (pos + 3 + 7) >> 3;
        if (tid == 0) {
            unsigned len = (unsigned)r.stored_len;
            out[byte] = (uint8_t)len, out[byte + 1] = (uint8_t)(len >> 8);
            out[byte + 2] = (uint8_t)~len, out[byte + 3] = (uint8_t)(~len >> 8);
        }
        // Copy_block (Deflate.cs:710-722): 16 bytes per lane and trip (global memory takes them at any alignment), bytes at the end
        const gcbytes src = as_global(s.in) + (r.start - s.abs_off);  // abs_off: stream position of in[0] (0 unless the run continues a stream)
        const gbytes_w dst = as_global(out) + byte + 4;
        typedef u32x4 __attribute__((aligned(1))) u32x4u;
        const int nvec = r.stored_len >> 4;
        for (int i = tid; i < nvec; i += 256)
            *(__attribute__((address_space(1))) u32x4u *)(dst + 16 * i) = *(const __attribute__((address_space(1))) u32x4u *)(src + 16 * i);
        for (int i = (nvec << 4) + tid; i < r.stored_len; i += 256) dst[i] = src[i];
        return;
    }
    const TreeWork &tw = trees[s.blk_off + b];
    if (bi.type == 2) {
        for (int i = tid; i < kLCodes; i += 256) lt[i] = (uint32_t)tw.ltree[i].fc | ((uint32_t)tw.ltree[i].dl << 16);
        if (tid < kDCodes) dt[tid] = (uint32_t)tw.dtree[tid].fc | ((uint32_t)tw.dtree[tid].dl << 16);
    } else {
        for (int i = tid; i < kLCodes; i += 256) lt[i] = static_lcode(i) | ((uint32_t)static_llen(i) << 16);
        if (tid < kDCodes) dt[tid] = bit_reverse((unsigned)tid, 5) | (5u << 16);
    }
    for (int i = tid; i < kEbWords; i += 256) obuf[i] = 0;
    __syncthreads();
    // global bit cursor, relative to the 4-byte aligned word at or below `out`
    uint32_t *const W = (uint32_t *)((uintptr_t)out & ~(uintptr_t)3);
    int64_t cur = bi.bit_start + (int64_t)(((uintptr_t)out & 3) << 3);
    if (tid == 0) {
        LdsBitPut put{obuf, (uint32_t)(cur & 31)};
        put((unsigned)(bi.type << 1) + (unsigned)(r.eof & 1), 3);
        if (bi.type == 2) emit_dyn_header(tw, put);
        sh_bits = put.pos - (uint32_t)(cur & 31);
    }
    __syncthreads();
    LdsTree L{lt}, D{dt};
    const uint32_t *sy = syms + s.sym_off + r.sym_start;
    const int total = r.nsyms + 1;  // + END_BLOCK
    bool first = true;              // the next flush starts with the block's first word
    uint32_t pending = sh_bits;     // bits already in obuf behind (cur & 31)
    // flush `nb` bits that sit in obuf from bit (cur & 31): complete words out, the partial one carried to obuf[0]
    auto flush = [&](uint32_t nb, bool last) {
        const uint32_t end = (uint32_t)(cur & 31) + nb;
        const uint32_t nfull = end >> 5;
        uint32_t *g = W + (cur >> 5);
        const uint32_t carry = obuf[nfull];
        __syncthreads();
        for (uint32_t k = tid; k < nfull; k += 256) {
            const uint32_t v = obuf[k];
            if (k == 0 && first) {
                if (v) atomicOr(g, v);
            } else {
                g[k] = v;
            }
            obuf[k] = 0;
        }
        if (tid == 0) {
            obuf[nfull] = 0;
            if (last) {
                if (carry) atomicOr(g + nfull, carry);
            } else {
                obuf[0] = carry;
            }
        }
        if (nfull) first = false;
        cur += nb;
        __syncthreads();
    };
    for (int t0 = 0; t0 < total; t0 += kEbTile) {
        // ---- this thread's 8 symbols (END_BLOCK is symbol number nsyms)
        const int i0 = t0 + tid * 8;
        uint32_t v[8];
        uint32_t mybits = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k;
            v[k] = i < r.nsyms ? sy[i] : (uint32_t)kEndBlock;
            uint64_t bits;
            if (i < total) mybits += (uint32_t)encode_symbol(L, D, (int)(v[k] >> 16), (int)(v[k] & 0xFFFF), bits);
        }
        // ---- exclusive scan over the workgroup
        uint32_t inc = mybits;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t before = 0, tile_bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = wsum[k];
            if (k < wave) before += x;
            tile_bits += x;
        }
        // ---- pack
        LdsBitPut put{obuf, (uint32_t)(cur & 31) + pending + before + inc - mybits};
        uint64_t acc = 0;
        int fill = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (i0 + k < total) {
                uint64_t bits;
                const int nb = encode_symbol(L, D, (int)(v[k] >> 16), (int)(v[k] & 0xFFFF), bits);
                acc |= bits << fill;  // nb <= 48, fill <= 15 after a flush
                fill += nb;
                if (fill >= 16) {
                    int fl = fill & ~7;
                    if (fl > 56) fl = 56;
                    put.put64(acc & ((1ull << fl) - 1), fl);
                    acc >>= fl;
                    fill -= fl;
                }
            }
        }
        if (fill) put.put64(acc, fill);
        __syncthreads();
        flush(pending + tile_bits, t0 + kEbTile >= total);
        pending = 0;
    }
}

// ------------------------------------------------------------------ KA
// Adler-32 (seed 1) of one 64 KiB piece of a stream.
__global__ __launch_bounds__(256) void zs_adler_kernel(const StreamDesc *sd, const uint2 *work, uint32_t *pieces) {
    __shared__ uint64_t ra[256], rb[256];
    uint2 w = work[blockIdx.x];
    const StreamDesc s = sd[w.x];
    int64_t beg = (int64_t)w.y * kAdlerPiece;
    int64_t len = (int64_t)s.n - beg;
    if (len > kAdlerPiece) len = kAdlerPiece;
    const gcbytes p = as_global(s.in) + beg;
    // thread t covers bytes [t*256, t*256+256): A_t = sum d, B_t = sum (L_t - j) d_j
    int64_t o = (int64_t)threadIdx.x * 256;
    int64_t lt = len - o;
    if (lt > 256) lt = 256;
    if (lt < 0) lt = 0;
    uint64_t a = 0, bsum = 0;
    if (lt == 256 && (((uintptr_t)(p + o)) & 15) == 0) {
        // 16 x 16-byte loads; within a 16-byte group byte j has weight (256 - 16 g - j)
        const gcu32x4 v = (gcu32x4)(p + o);
#pragma unroll 4
        for (int g = 0; g < 16; g++) {
            const u32x4 q = v[g];
            const uint32_t w[4] = {q[0], q[1], q[2], q[3]};
            uint32_t ga = 0, gb = 0;  // sum of the 16 bytes, sum of j * byte
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t b0 = w[k] & 0xFF, b1 = (w[k] >> 8) & 0xFF, b2 = (w[k] >> 16) & 0xFF, b3 = w[k] >> 24;
                ga += b0 + b1 + b2 + b3;
                gb += (4 * k) * b0 + (4 * k + 1) * b1 + (4 * k + 2) * b2 + (4 * k + 3) * b3;
            }
            a += ga;
            bsum += (uint64_t)(256 - 16 * g) * ga - gb;
        }
    } else {
        for (int j = 0; j < lt; j++) {
            uint64_t d = p[o + j];
            a += d;
            bsum += (uint64_t)(lt - j) * d;
        }
    }
    // contribution to the piece's s2: B_t + (len - o - lt) * A_t
    uint64_t tailw = (uint64_t)(len - o - lt);
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = lt > 0 ? bsum + tailw * a : 0;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) ra[threadIdx.x] += ra[threadIdx.x + off], rb[threadIdx.x] += rb[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint64_t s1 = (1 + ra[0]) % kAdlerBase;
        uint64_t s2 = ((uint64_t)len + rb[0]) % kAdlerBase;  // seed s1 = 1 contributes len
        pieces[s.adler_off + w.y] = (uint32_t)((s2 << 16) | s1);
    }
}

// Adler-32 of whole buffers from their 64 KiB pieces, one workgroup per buffer, and -- for inflate -- the comparison with
// the stream's trailer (Inflate.cs:300-345): `trailer[i]` points at the 4 big-endian bytes behind the last block, or is
// null.  res[i] = (adler of the buffer, 1 when it equals the trailer / no trailer given).
__global__ __launch_bounds__(256) void zs_adler_finish_kernel(const StreamDesc *sd, const uint32_t *pieces, const uint8_t *const *trailer,
                                                              uint2 *res) {
    __shared__ uint32_t ad_v[256];
    __shared__ uint64_t ad_len[256];
    const StreamDesc s = sd[blockIdx.x];
    uint32_t acc = 1;
    uint64_t acc_len = 0;
    const int per = (s.n_adler + 255) / 256;
    for (int k = 0; k < per; k++) {
        const int i = threadIdx.x * per + k;
        if (i >= s.n_adler) break;
        int64_t len = (int64_t)s.n - (int64_t)i * kAdlerPiece;
        if (len > kAdlerPiece) len = kAdlerPiece;
        acc = adler_combine(acc, pieces[s.adler_off + i], (uint64_t)len);
        acc_len += (uint64_t)len;
    }
    ad_v[threadIdx.x] = acc, ad_len[threadIdx.x] = acc_len;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        if ((threadIdx.x & (2 * off - 1)) == 0) {
            ad_v[threadIdx.x] = adler_combine(ad_v[threadIdx.x], ad_v[threadIdx.x + off], ad_len[threadIdx.x + off]);
            ad_len[threadIdx.x] += ad_len[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    uint32_t ok = 1;
    if (trailer && trailer[blockIdx.x]) {
        const uint8_t *t = trailer[blockIdx.x];
        const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        ok = want == ad_v[0];
    }
    res[blockIdx.x] = make_uint2(ad_v[0], ok);
}

// ------------------------------------------------------------------ KP: PNG scanline filters (the caller path of the sparse case)
// the reference exists because of PNG encoding (readme.md:16-19) -- ImageSharp filters every
// scanline (PNG specification 9.2: None, Sub, Up, Average, Paeth, or per row the one with the smallest sum of absolute
// values) and hands the rows to ZlibOutputStream.  One workgroup per row: the five candidate rows are never stored, each
// thread adds up |filtered byte| for its bytes under all five filters, the row's filter is the minimum (first one on
// ties, in the order None, Sub, Up, Average, Paeth), then the row is written: filter-type byte + filtered bytes.
__device__ __forceinline__ int png_paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
__device__ __forceinline__ uint8_t png_apply(int f, int x, int a, int b, int c) {
    switch (f) {
    case 1: return (uint8_t)(x - a);
    case 2: return (uint8_t)(x - b);
    case 3: return (uint8_t)(x - ((a + b) >> 1));
    case 4: return (uint8_t)(x - png_paeth(a, b, c));
    default: return (uint8_t)x;
    }
}
__global__ __launch_bounds__(256) void zs_png_filter_kernel(const uint8_t *pix, int64_t row_bytes, int64_t height, int bpp, int filter,
                                                            uint8_t *out) {
    __shared__ uint32_t sums[5][256];
    __shared__ int chosen;
    const int64_t y = blockIdx.x;
    const gcbytes cur = as_global(pix) + y * row_bytes, up = y ? cur - row_bytes : cur;
    const bool has_up = y > 0;
    int f = filter;
    if (filter == 5) {
        uint32_t acc[5] = {0, 0, 0, 0, 0};
        for (int64_t i = threadIdx.x; i < row_bytes; i += 256) {
            const int x = cur[i], a = i >= bpp ? cur[i - bpp] : 0, b = has_up ? up[i] : 0, c = (has_up && i >= bpp) ? up[i - bpp] : 0;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int v = (int8_t)png_apply(k, x, a, b, c);  // the sum is over the bytes read as signed values
                acc[k] += (uint32_t)(v < 0 ? -v : v);
            }
        }
        for (int k = 0; k < 5; k++) sums[k][threadIdx.x] = acc[k];
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off)
                for (int k = 0; k < 5; k++) sums[k][threadIdx.x] += sums[k][threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            int best = 0;
            for (int k = 1; k < 5; k++)
                if (sums[k][0] < sums[best][0]) best = k;
            chosen = best;
        }
        __syncthreads();
        f = chosen;
    }
    uint8_t *o = out + y * (row_bytes + 1);
    if (threadIdx.x == 0) o[0] = (uint8_t)f;
    for (int64_t i = threadIdx.x; i < row_bytes; i += 256) {
        const int x = cur[i], a = i >= bpp ? cur[i - bpp] : 0, b = has_up ? up[i] : 0, c = (has_up && i >= bpp) ? up[i - bpp] : 0;
        o[1 + i] = png_apply(f, x, a, b, c);
    }
}

}  // namespace zs
