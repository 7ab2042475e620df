// zs_rle.hip -- KR: CompressionStrategy.Rle (Deflate.Rle.cs:18-104) over the chip.  Included by zs_kernels.hip; zs_rle.h has
// the formulation (a position's part in the parse follows from where its run of equal bytes began) and the code shared with
// the CPU model.
//
// A wave takes a tile of 4096 positions 64 at a time, lane = position:
//   KR1  zs_rle_starts_kernel   per tile: the last position at which a run begins (-1: none)
//   KR2  zs_rle_scan_kernel     per stream: the prefix maximum of those over the tiles -> the run that reaches into each tile
//   KR3  zs_rle_pass_kernel<0>  per tile: the loop-tops below the hand-over position, counted; the first loop-top at or behind
//                               it (atomic min): where the tail engine takes over
//   KR4  zs_rle_sums_kernel     per stream: the prefix sums of the counts, the stream's state for the kernels behind
//   KR5  zs_rle_pass_kernel<1>  per tile: the same pass again, the symbols written at their places; block cuts every 16 383
// All of them are bound by HBM and launch latency: 64 MiB is read three times (0.2 GB).
constexpr int kRleTile = 4096;

struct RleTiles {
    int32_t *last;   // per tile: last run start in it, -1
    int32_t *carry;  // per tile: first position of the run its first position lies in
    int32_t *cnt;    // per tile: loop-tops below the hand-over position
    int32_t *base;   // per tile: loop-tops in the tiles before it
    int32_t *ph;     // per stream: first loop-top at or behind the hand-over position
};

__device__ __forceinline__ int rle_ntiles(const StreamDesc &s) { return (s.rle_end + kMaxMatch + 1 + kRleTile - 1) / kRleTile; }

// (a grid's y ends at 65 535: the host launches the streams in slices, si0 is the slice's first)
__global__ __launch_bounds__(256) void zs_rle_starts_kernel(const StreamDesc *sd, RleTiles rt, int si0) {
    const int si = si0 + (int)blockIdx.y;
    const StreamDesc s = sd[si];
    if (s.rle_end < 0) return;
    const int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = lane_id();
    if (tile >= rle_ntiles(s)) return;
    const gcbytes in = as_global(s.in);
    int last = -1;
    for (int it = 0; it < kRleTile / 64; it++) {
        const int p = tile * kRleTile + it * 64 + lane;
        const bool sf = p < s.n && (p == 0 || in[p] != in[p - 1]);
        const uint64_t m = __ballot(sf);
        if (m) last = tile * kRleTile + it * 64 + 63 - (int)__builtin_clzll(m);
    }
    if (lane == 0) rt.last[s.rle_tile_off + tile] = last;
    if (tile == 0 && lane == 0) rt.ph[si] = 0x7FFFFFFF;
}

__global__ __launch_bounds__(1024) void zs_rle_scan_kernel(const StreamDesc *sd, RleTiles rt) {
    const StreamDesc s = sd[blockIdx.x];
    if (s.rle_end < 0) return;
    __shared__ int sh[1024];
    const int nt = rle_ntiles(s), tid = threadIdx.x;
    int run = 0;  // the run that reaches into the next piece of 1024 tiles
    for (int t0 = 0; t0 < nt; t0 += 1024) {
        const int t = t0 + tid;
        const int v = t < nt ? rt.last[s.rle_tile_off + t] : -1;
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // inclusive prefix maximum
            const int x = tid >= o ? sh[tid - o] : -1;
            __syncthreads();
            if (x > sh[tid]) sh[tid] = x;
            __syncthreads();
        }
        const int before = tid ? sh[tid - 1] : -1;
        if (t < nt) rt.carry[s.rle_tile_off + t] = before > run ? before : run;
        const int all = sh[1023];
        __syncthreads();
        if (all > run) run = all;
    }
}

template <int EMIT>
__global__ __launch_bounds__(256) void zs_rle_pass_kernel(const StreamDesc *sd, RleTiles rt, uint32_t *syms, int32_t *blk_end, int32_t *blk_top, int si0) {
    const int si = si0 + (int)blockIdx.y;
    const StreamDesc s = sd[si];
    if (s.rle_end < 0) return;
    const int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = lane_id();
    if (tile >= rle_ntiles(s)) return;
    const gcbytes in = as_global(s.in);
    const int H = s.rle_end;
    int a_in = rt.carry[s.rle_tile_off + tile];  // where the run began that the next 64 positions start in
    int count = 0;
    const int gbase = EMIT ? rt.base[s.rle_tile_off + tile] : 0;
    uint32_t *out = syms + s.sym_off;
    for (int it = 0; it < kRleTile / 64; it++) {
        const int p0 = tile * kRleTile + it * 64, p = p0 + lane;
        if (p0 >= H + kMaxMatch + 1) break;  // (the hand-over loop-top lies below: nothing behind it is the body's)
        const bool inb = p < s.n;
        const uint8_t d0 = inb ? in[p] : 0;
        const bool sf = inb && (p == 0 || d0 != in[p - 1]);
        const uint64_t m = __ballot(sf);
        const uint64_t below = m & ((2ull << lane) - 1ull);
        const int a = below ? p0 + 63 - (int)__builtin_clzll(below) : a_in;
        if (m) a_in = p0 + 63 - (int)__builtin_clzll(m);
        // rle_role (zs_rle.h), the run's length by 8 bytes a step (p < n - 3 * 262 for every position asked: no read behind the data)
        int role = 0;
        if (inb && p < H + kMaxMatch + 1) {
            const int o = p - a;
            if (o == 0) {
                role = 1;
            } else {
                const int k = (o - 1) % kMaxMatch;
                if (k == 0) {
                    const uint64_t c8 = 0x0101010101010101ull * d0;
                    int len = 0;
                    while (len < kMaxMatch + 6) {
                        const uint64_t x = *(gcu64u)(in + p + len) ^ c8;
                        if (x) {
                            len += (int)(__builtin_ctzll(x) >> 3);
                            break;
                        }
                        len += 8;
                    }
                    len = len < kMaxMatch ? len : kMaxMatch;
                    role = len >= kMinMatch ? len : 1;
                } else if (k == 1) {
                    role = in[p + 1] != d0 ? 1 : 0;
                }
            }
        }
        const uint64_t body = __ballot(role != 0 && p < H);
        const uint64_t behind = __ballot(role != 0 && p >= H);
        if (!EMIT && behind && lane == 0) atomicMin(&rt.ph[si], p0 + (int)__builtin_ctzll(behind));
        if (EMIT && ((body >> lane) & 1ull)) {
            const int g = gbase + count + (int)__builtin_popcountll(body & lanemask_lt());
            out[g] = role == 1 ? (uint32_t)d0 : ((1u << 16) | (uint32_t)(role - kMinMatch));
            if ((g + 1) % kBlockSyms == 0) {
                blk_end[s.blk_off + g / kBlockSyms] = p + (role == 1 ? 1 : role);
                blk_top[s.blk_off + g / kBlockSyms] = p;
            }
        }
        count += (int)__builtin_popcountll(body);
    }
    if (!EMIT && lane == 0) rt.cnt[s.rle_tile_off + tile] = count;
}

__global__ __launch_bounds__(1024) void zs_rle_sums_kernel(const StreamDesc *sd, StreamState *st, RleTiles rt) {
    const StreamDesc s = sd[blockIdx.x];
    if (s.rle_end < 0) return;
    __shared__ int sh[1024];
    const int nt = rle_ntiles(s), tid = threadIdx.x;
    int total = 0;
    for (int t0 = 0; t0 < nt; t0 += 1024) {
        const int t = t0 + tid;
        const int v = t < nt ? rt.cnt[s.rle_tile_off + t] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // inclusive prefix sum
            const int x = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += x;
            __syncthreads();
        }
        if (t < nt) rt.base[s.rle_tile_off + t] = total + sh[tid] - v;
        const int all = sh[1023];
        __syncthreads();
        total += all;
    }
    if (tid == 0) {
        StreamState &ss = st[blockIdx.x];
        ss.tail_p = rt.ph[blockIdx.x];
        ss.tail_kind = kR;
        ss.tail_pend = 0;
        ss.k_done = rle_refills_fired_at(s.rle_end, s.kl);
        ss.preins = -1;
        ss.body_syms = (uint32_t)total;
    }
}
