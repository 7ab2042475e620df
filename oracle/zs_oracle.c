/*
 * zs_oracle.c -- CPU restatement of the SixLabors/ZlibStream deflate path.
 *
 * TEST INFRASTRUCTURE ONLY (see zs_oracle.h).  The product library never
 * links this file.
 *
 * Every function names the reference file:line it follows (paths relative to
 * /root/reference/src/ZlibStream unless noted).  The code is a restatement
 * written from the behaviour of those lines, with window coordinates kept
 * exactly as the reference keeps them (64 KiB window, 32 KiB slides, u16
 * head/prev with saturating slide) so that every quirk -- hash over bytes
 * str+2..str+5, the `cur != str` guard, the Fill_window pre-insert, stale
 * bytes past the end of input after a slide -- falls out of the same state
 * machine rather than being modelled separately.
 */
#include "zs_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- constants (Deflate.cs:17-78, Trees.cs:12-34) ---- */
#define MIN_MATCH 3
#define MAX_MATCH 258
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define MAX_BITS 15
#define MAX_BL_BITS 7
#define D_CODES 30
#define BL_CODES 19
#define LENGTH_CODES 29
#define LITERALS 256
#define END_BLOCK 256
#define L_CODES (LITERALS + 1 + LENGTH_CODES)
#define HEAP_SIZE (2 * L_CODES + 1)
#define REP_3_6 16
#define REPZ_3_10 17
#define REPZ_11_138 18

#define ST_INIT 42
#define ST_BUSY 113
#define ST_FINISH 666

#define BS_NEED_MORE 0
#define BS_BLOCK_DONE 1
#define BS_FINISH_STARTED 2
#define BS_FINISH_DONE 3

#define FN_STORED 0
#define FN_FAST 1
#define FN_SLOW 2

typedef struct {
    uint16_t fc; /* Freq / Code  (Trees.Static.cs:96-109: explicit-layout union) */
    uint16_t dl; /* Dad  / Len */
} ct_data;

typedef struct {
    int good, lazy, nice, chain, func;
} level_cfg;

/* Deflate.cs:80-98 */
static const level_cfg k_levels[10] = {
    {0, 0, 0, 0, FN_STORED},     {4, 4, 8, 4, FN_FAST},        {4, 5, 16, 8, FN_FAST},      {4, 6, 32, 32, FN_FAST},
    {4, 4, 16, 16, FN_SLOW},     {8, 16, 32, 32, FN_SLOW},     {8, 16, 128, 128, FN_SLOW},  {8, 32, 128, 256, FN_SLOW},
    {32, 128, 258, 1024, FN_SLOW}, {32, 258, 258, 4096, FN_SLOW},
};

/* Deflate.cs:100-112 */
static const char *const k_errmsg[10] = {
    "need dictionary", "stream end", "", "file error", "stream error", "data error", "insufficient memory",
    "buffer error", "incompatible version", "",
};

/* Trees.cs:36-62 */
static const uint8_t k_extra_lbits[LENGTH_CODES] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                                     2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint8_t k_extra_dbits[D_CODES] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4,  4,  5,  5,  6,
                                                6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t k_extra_blbits[BL_CODES] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t k_bl_order[BL_CODES] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* The reference ships DistCode / LengthCode / BaseLength / BaseDist as
 * literal tables (Trees.cs:36-129); they are the RFC 1951 code tables, so
 * they are generated here from the extra-bit counts instead. */
static uint8_t g_dist_code[512];
static uint8_t g_length_code[256];
static int g_base_length[LENGTH_CODES];
static int g_base_dist[D_CODES];
static ct_data g_static_ltree[L_CODES + 2];
static ct_data g_static_dtree[D_CODES];
static int g_tables_ready;

static uint32_t g_crc32c_tab[4][256];

static unsigned bit_reverse(unsigned code, int len) { /* Trees.cs:269-281 */
    unsigned r = 0;
    for (int i = 0; i < len; i++) {
        r = (r << 1) | (code & 1u);
        code >>= 1;
    }
    return r;
}

/* Trees.cs:1123-1151 */
static void gen_codes(ct_data *tree, int max_code, const uint16_t *bl_count) {
    uint16_t next_code[MAX_BITS + 1];
    unsigned code = 0;
    next_code[0] = 0;
    for (int bits = 1; bits <= MAX_BITS; bits++) {
        code = (code + bl_count[bits - 1]) << 1;
        next_code[bits] = (uint16_t)code;
    }
    for (int n = 0; n <= max_code; n++) {
        int len = tree[n].dl;
        if (len == 0) continue;
        tree[n].fc = (uint16_t)bit_reverse(next_code[len]++, len);
    }
}

static void init_tables(void) {
    if (g_tables_ready) return;
    /* length codes */
    int length = 0, code;
    for (code = 0; code < LENGTH_CODES - 1; code++) {
        g_base_length[code] = length;
        for (int n = 0; n < (1 << k_extra_lbits[code]); n++) g_length_code[length++] = (uint8_t)code;
    }
    g_length_code[length - 1] = (uint8_t)code; /* length 258 -> code 28 */
    g_base_length[LENGTH_CODES - 1] = 0;      /* Trees.cs:58-62: BaseLength[28] == 0 */
    /* distance codes */
    int dist = 0;
    for (code = 0; code < 16; code++) {
        g_base_dist[code] = dist;
        for (int n = 0; n < (1 << k_extra_dbits[code]); n++) g_dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < D_CODES; code++) {
        g_base_dist[code] = dist << 7;
        for (int n = 0; n < (1 << (k_extra_dbits[code] - 7)); n++) g_dist_code[256 + dist++] = (uint8_t)code;
    }
    /* static trees (Trees.Static.cs:29-92) */
    uint16_t bl_count[MAX_BITS + 1];
    memset(bl_count, 0, sizeof bl_count);
    int n = 0;
    while (n <= 143) g_static_ltree[n++].dl = 8, bl_count[8]++;
    while (n <= 255) g_static_ltree[n++].dl = 9, bl_count[9]++;
    while (n <= 279) g_static_ltree[n++].dl = 7, bl_count[7]++;
    while (n <= 287) g_static_ltree[n++].dl = 8, bl_count[8]++;
    gen_codes(g_static_ltree, L_CODES + 1, bl_count);
    for (n = 0; n < D_CODES; n++) {
        g_static_dtree[n].dl = 5;
        g_static_dtree[n].fc = (uint16_t)bit_reverse((unsigned)n, 5);
    }
    /* CRC32C (Castagnoli, reflected 0x82F63B78), slicing-by-4 */
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        g_crc32c_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = g_crc32c_tab[0][i];
        for (int t = 1; t < 4; t++) {
            c = g_crc32c_tab[0][c & 0xFF] ^ (c >> 8);
            g_crc32c_tab[t][i] = c;
        }
    }
    g_tables_ready = 1;
}

/* Deflate.Intrinsics.cs:295-307.  Sse42.Crc32(0, u32) is the x86 `crc32`
 * instruction: CRC32C, initial value 0, no final inversion. */
uint32_t zso_hash_u32(uint32_t v, int hash_variant) {
    init_tables();
    if (hash_variant == ZSO_HASH_MUL) return (v * 2654435761u) >> 16;
    return g_crc32c_tab[3][v & 0xFF] ^ g_crc32c_tab[2][(v >> 8) & 0xFF] ^ g_crc32c_tab[1][(v >> 16) & 0xFF] ^
           g_crc32c_tab[0][v >> 24];
}

/* Adler32.cs:270-326 (scalar), seed handling :61-78 */
uint32_t zso_adler32(uint32_t adler, const uint8_t *buf, size_t len) {
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    while (len > 0) {
        size_t blk = len < 5552 ? len : 5552;
        len -= blk;
        while (blk--) {
            s1 += *buf++;
            s2 += s1;
        }
        s1 %= 65521u;
        s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

/* ------------------------------------------------------------------ */

struct zso_deflate {
    /* z_stream cursor, valid during a call (ZlibStream.cs:34-94) */
    const uint8_t *next_in;
    int avail_in;
    uint8_t *next_out;
    int avail_out;
    int64_t total_in, total_out;
    uint32_t adler;
    const char *msg;

    int status, last_flush, noheader;
    int level, strategy, hash_variant;
    int w_size, w_bits, w_mask, window_size;
    int hash_size, hash_bits;
    uint32_t hash_mask;
    int lit_bufsize, d_buf, l_buf, pending_size;

    uint8_t *window;
    uint16_t *prev, *head;
    uint8_t *pending;
    int pending_n, pending_out;

    int block_start, match_length, prev_match, match_available, strstart, match_start, lookahead, prev_length;
    int max_chain, max_lazy, good_match, nice_match;

    ct_data dyn_ltree[HEAP_SIZE], dyn_dtree[2 * D_CODES + 1], bl_tree[2 * BL_CODES + 1];
    int l_max_code, d_max_code, bl_max_code;
    int heap[HEAP_SIZE], heap_len, heap_max;
    uint8_t depth[HEAP_SIZE];
    uint16_t bl_count[MAX_BITS + 1];
    int opt_len, static_len, last_lit, matches, last_eob_len;
    uint64_t bi_buf;
    int bi_valid;

    int overflow; /* the reference would have thrown (see tr_stored_block) */
    /* instrumentation only */
    int64_t abs_base;
    int64_t out_bits;
    zso_trace trace;
    int has_trace;
};

typedef struct {
    const ct_data *static_tree;
    const uint8_t *extra_bits;
    int extra_base, elems, max_length;
} static_desc;

static const static_desc k_l_desc = {g_static_ltree, k_extra_lbits, LITERALS + 1, L_CODES, MAX_BITS};
static const static_desc k_d_desc = {g_static_dtree, k_extra_dbits, 0, D_CODES, MAX_BITS};
static const static_desc k_bl_desc = {NULL, k_extra_blbits, 0, BL_CODES, MAX_BL_BITS};

static inline int d_code(int dist) { /* Trees.cs:217-226 */
    return dist < 256 ? g_dist_code[dist] : g_dist_code[256 + (dist >> 7)];
}

/* ---- pending-buffer writers (Deflate.cs:757-792) ---- */
static inline void put_byte(zso_deflate *s, unsigned c) { s->pending[s->pending_n++] = (uint8_t)c; }
static inline void put_short_lsb(zso_deflate *s, unsigned w) {
    put_byte(s, w & 0xFF);
    put_byte(s, (w >> 8) & 0xFF);
}
static inline void put_short_msb(zso_deflate *s, unsigned w) {
    put_byte(s, (w >> 8) & 0xFF);
    put_byte(s, w & 0xFF);
}
static inline void put_le(zso_deflate *s, uint64_t w, int nbytes) {
    for (int i = 0; i < nbytes; i++) put_byte(s, (unsigned)(w >> (8 * i)) & 0xFF);
}

/* Deflate.cs:799-821: 64-bit accumulator, bytes leave LSB first */
static void send_bits(zso_deflate *s, unsigned value, int length) {
    uint64_t val = value;
    int total = s->bi_valid + length;
    s->out_bits += length;
    if (total < 64) {
        s->bi_buf |= val << s->bi_valid;
        s->bi_valid = total;
    } else if (s->bi_valid == 64) {
        put_le(s, s->bi_buf, 8);
        s->bi_buf = val;
        s->bi_valid = length;
    } else {
        s->bi_buf |= val << s->bi_valid;
        put_le(s, s->bi_buf, 8);
        s->bi_buf = val >> (64 - s->bi_valid);
        s->bi_valid = total - 64;
    }
}
static inline void send_code(zso_deflate *s, int c, const ct_data *tree) { send_bits(s, tree[c].fc, tree[c].dl); }

/* Deflate.cs:640-671 */
static void bi_flush(zso_deflate *s) {
    if (s->bi_valid == 64) {
        put_le(s, s->bi_buf, 8);
        s->bi_buf = 0;
        s->bi_valid = 0;
        return;
    }
    if (s->bi_valid >= 32) {
        put_le(s, s->bi_buf, 4);
        s->bi_buf >>= 32;
        s->bi_valid -= 32;
    }
    if (s->bi_valid >= 16) {
        put_le(s, s->bi_buf, 2);
        s->bi_buf >>= 16;
        s->bi_valid -= 16;
    }
    if (s->bi_valid >= 8) {
        put_le(s, s->bi_buf, 1);
        s->bi_buf >>= 8;
        s->bi_valid -= 8;
    }
}

/* Deflate.cs:675-705.  The reference writes 8 bytes when bi_valid > 56 and
 * otherwise 4-, 2- and 1-byte pieces (thresholds > 24, > 8, > 0); in every case
 * that is the little-endian image of bi_buf cut to ceil(bi_valid / 8) bytes. */
static void bi_windup(zso_deflate *s) {
    s->out_bits += (8 - (s->bi_valid & 7)) & 7;
    put_le(s, s->bi_buf, (s->bi_valid + 7) >> 3);
    s->bi_buf = 0;
    s->bi_valid = 0;
}

/* Deflate.cs:828-854 */
static void flush_pending(zso_deflate *s) {
    bi_flush(s);
    int len = s->pending_n;
    if (len > s->avail_out) len = s->avail_out;
    if (len == 0) return;
    memcpy(s->next_out, s->pending + s->pending_out, (size_t)len);
    s->next_out += len;
    s->pending_out += len;
    s->total_out += len;
    s->avail_out -= len;
    s->pending_n -= len;
    if (s->pending_n == 0) s->pending_out = 0;
}

/* ---- trees ---- */

/* Trees.cs:782-811 */
static void init_block(zso_deflate *s) {
    for (int i = 0; i < L_CODES; i++) s->dyn_ltree[i].fc = 0;
    for (int i = 0; i < D_CODES; i++) s->dyn_dtree[i].fc = 0;
    for (int i = 0; i < BL_CODES; i++) s->bl_tree[i].fc = 0;
    s->dyn_ltree[END_BLOCK].fc = 1;
    s->opt_len = s->static_len = 0;
    s->last_lit = s->matches = 0;
}

/* Trees.cs:556-557 */
static inline int node_smaller(const ct_data *tree, int n, int m, const uint8_t *depth) {
    return tree[n].fc < tree[m].fc || (tree[n].fc == tree[m].fc && depth[n] <= depth[m]);
}

/* Trees.cs:513-544 */
static void pqdownheap(zso_deflate *s, const ct_data *tree, int k) {
    int v = s->heap[k];
    int j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && node_smaller(tree, s->heap[j + 1], s->heap[j], s->depth)) j++;
        if (node_smaller(tree, v, s->heap[j], s->depth)) break;
        s->heap[k] = s->heap[j];
        k = j;
        j <<= 1;
    }
    s->heap[k] = v;
}

/* Trees.cs:999-1109 */
static void gen_bitlen(zso_deflate *s, ct_data *tree, int max_code, const static_desc *d) {
    int h, n, m, bits, xbits, overflow = 0;
    for (bits = 0; bits <= MAX_BITS; bits++) s->bl_count[bits] = 0;
    tree[s->heap[s->heap_max]].dl = 0;
    for (h = s->heap_max + 1; h < HEAP_SIZE; h++) {
        n = s->heap[h];
        bits = tree[tree[n].dl].dl + 1;
        if (bits > d->max_length) {
            bits = d->max_length;
            overflow++;
        }
        tree[n].dl = (uint16_t)bits;
        if (n > max_code) continue;
        s->bl_count[bits]++;
        xbits = n >= d->extra_base ? d->extra_bits[n - d->extra_base] : 0;
        unsigned f = tree[n].fc;
        s->opt_len += (int)(f * (unsigned)(bits + xbits));
        if (d->static_tree) s->static_len += (int)(f * (unsigned)(d->static_tree[n].dl + xbits));
    }
    if (overflow == 0) return;
    do {
        bits = d->max_length - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--;
        s->bl_count[bits + 1] += 2;
        s->bl_count[d->max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = d->max_length; bits != 0; bits--) {
        n = s->bl_count[bits];
        while (n != 0) {
            m = s->heap[--h];
            if (m > max_code) continue;
            if (tree[m].dl != (unsigned)bits) {
                s->opt_len += bits * (int)tree[m].fc;
                s->opt_len -= (int)tree[m].dl * (int)tree[m].fc;
                tree[m].dl = (uint16_t)bits;
            }
            n--;
        }
    }
}

/* Trees.cs:404-501 */
static void build_tree(zso_deflate *s, ct_data *tree, int *max_code_out, const static_desc *d) {
    int n, m, max_code = -1, node;
    s->heap_len = 0;
    s->heap_max = HEAP_SIZE;
    for (n = 0; n < d->elems; n++) {
        if (tree[n].fc != 0) {
            s->heap[++s->heap_len] = max_code = n;
            s->depth[n] = 0;
        } else {
            tree[n].dl = 0;
        }
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = max_code < 2 ? ++max_code : 0;
        tree[node].fc = 1;
        s->depth[node] = 0;
        s->opt_len--;
        if (d->static_tree) s->static_len -= d->static_tree[node].dl;
    }
    *max_code_out = max_code;
    for (n = s->heap_len / 2; n >= 1; n--) pqdownheap(s, tree, n);
    node = d->elems;
    do {
        n = s->heap[1];
        s->heap[1] = s->heap[s->heap_len--];
        pqdownheap(s, tree, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n;
        s->heap[--s->heap_max] = m;
        tree[node].fc = (uint16_t)(tree[n].fc + tree[m].fc);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        tree[n].dl = tree[m].dl = (uint16_t)node;
        s->heap[1] = node++;
        pqdownheap(s, tree, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];
    gen_bitlen(s, tree, max_code, d);
    gen_codes(tree, max_code, s->bl_count);
}

/* Trees.cs:290-357 */
static void scan_tree(zso_deflate *s, ct_data *tree, int max_code) {
    int prevlen = -1, curlen, nextlen = tree[0].dl, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    tree[max_code + 1].dl = 0xFFFF;
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen;
        nextlen = tree[n + 1].dl;
        if (++count < max_count && curlen == nextlen) continue;
        if (count < min_count) {
            s->bl_tree[curlen].fc += (uint16_t)count;
        } else if (curlen != 0) {
            if (curlen != prevlen) s->bl_tree[curlen].fc++;
            s->bl_tree[REP_3_6].fc++;
        } else if (count <= 10) {
            s->bl_tree[REPZ_3_10].fc++;
        } else {
            s->bl_tree[REPZ_11_138].fc++;
        }
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

/* Trees.cs:879-952 */
static void send_tree(zso_deflate *s, const ct_data *tree, int max_code) {
    int prevlen = -1, curlen, nextlen = tree[0].dl, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) max_count = 138, min_count = 3;
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen;
        nextlen = tree[n + 1].dl;
        if (++count < max_count && curlen == nextlen) continue;
        if (count < min_count) {
            do send_code(s, curlen, s->bl_tree);
            while (--count != 0);
        } else if (curlen != 0) {
            if (curlen != prevlen) {
                send_code(s, curlen, s->bl_tree);
                count--;
            }
            send_code(s, REP_3_6, s->bl_tree);
            send_bits(s, (unsigned)(count - 3), 2);
        } else if (count <= 10) {
            send_code(s, REPZ_3_10, s->bl_tree);
            send_bits(s, (unsigned)(count - 3), 3);
        } else {
            send_code(s, REPZ_11_138, s->bl_tree);
            send_bits(s, (unsigned)(count - 11), 7);
        }
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) max_count = 138, min_count = 3;
        else if (curlen == nextlen) max_count = 6, min_count = 3;
        else max_count = 7, min_count = 4;
    }
}

/* Trees.cs:361-391 */
static int build_bl_tree(zso_deflate *s) {
    int max_blindex;
    scan_tree(s, s->dyn_ltree, s->l_max_code);
    scan_tree(s, s->dyn_dtree, s->d_max_code);
    build_tree(s, s->bl_tree, &s->bl_max_code, &k_bl_desc);
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; max_blindex--)
        if (s->bl_tree[k_bl_order[max_blindex]].dl != 0) break;
    s->opt_len += 3 * (max_blindex + 1) + 5 + 5 + 4;
    return max_blindex;
}

/* Trees.cs:856-870 */
static void send_all_trees(zso_deflate *s, int lcodes, int dcodes, int blcodes) {
    send_bits(s, (unsigned)(lcodes - 257), 5);
    send_bits(s, (unsigned)(dcodes - 1), 5);
    send_bits(s, (unsigned)(blcodes - 4), 4);
    for (int rank = 0; rank < blcodes; rank++) send_bits(s, s->bl_tree[k_bl_order[rank]].dl, 3);
    send_tree(s, s->dyn_ltree, lcodes - 1);
    send_tree(s, s->dyn_dtree, dcodes - 1);
}

/* Trees.cs:956-989 with Tr_emit_distance :707-732 */
static void compress_block(zso_deflate *s, const ct_data *ltree, const ct_data *dtree) {
    for (int lx = 0; lx < s->last_lit; lx++) {
        int dist = (s->pending[s->d_buf + lx * 2] << 8) | s->pending[s->d_buf + lx * 2 + 1];
        int lc = s->pending[s->l_buf + lx];
        if (dist == 0) {
            send_code(s, lc, ltree);
        } else {
            int code = g_length_code[lc];
            send_code(s, code + LITERALS + 1, ltree);
            int extra = k_extra_lbits[code];
            if (extra != 0) send_bits(s, (unsigned)(lc - g_base_length[code]), extra);
            dist--;
            code = d_code(dist);
            send_code(s, code, dtree);
            extra = k_extra_dbits[code];
            if (extra != 0) send_bits(s, (unsigned)(dist - g_base_dist[code]), extra);
        }
    }
    send_code(s, END_BLOCK, ltree);
    s->last_eob_len = ltree[END_BLOCK].dl;
}

/* Deflate.cs:710-722 (Copy_block) + Trees.cs:742-746 (Tr_stored_block) */
static void tr_stored_block(zso_deflate *s, int buf, int stored_len, int eof) {
    send_bits(s, (0u << 1) + (eof ? 1u : 0u), 3);
    bi_windup(s);
    s->last_eob_len = 8;
    if (s->pending_n + 4 + stored_len > s->pending_size) {
        /* The reference's Buffer.BlockCopy into PendingBuffer (Deflate.cs:757-761) throws here: a stored
         * block larger than the pending buffer (only reachable with level 0 + Rle on compressible data,
         * where a block can span more than 32 KiB while its start is still in the window). */
        s->overflow = 1;
        stored_len = 0;
    }
    put_short_lsb(s, (unsigned)stored_len & 0xFFFF);
    put_short_lsb(s, (unsigned)~stored_len & 0xFFFF);
    if (stored_len > 0) memcpy(s->pending + s->pending_n, s->window + buf, (size_t)stored_len);
    s->pending_n += stored_len;
    s->out_bits += 32 + 8 * (int64_t)stored_len;
}

/* Trees.cs:658-680 */
static void tr_align(zso_deflate *s) {
    send_bits(s, 1u << 1, 3);
    send_code(s, END_BLOCK, g_static_ltree);
    bi_flush(s);
    if (1 + s->last_eob_len + 10 - s->bi_valid < 9) {
        send_bits(s, 1u << 1, 3);
        send_code(s, END_BLOCK, g_static_ltree);
        bi_flush(s);
    }
    s->last_eob_len = 7;
}

/* Trees.cs:568-643 */
static void tr_flush_block(zso_deflate *s, int buf, int stored_len, int eof) {
    int opt_lenb, static_lenb, max_blindex = 0, type;
    int nsyms = s->last_lit;
    int64_t bit_start = s->out_bits;
    if (s->level > 0) {
        build_tree(s, s->dyn_ltree, &s->l_max_code, &k_l_desc);
        build_tree(s, s->dyn_dtree, &s->d_max_code, &k_d_desc);
        max_blindex = build_bl_tree(s);
        opt_lenb = (s->opt_len + 3 + 7) >> 3;
        static_lenb = (s->static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    } else {
        opt_lenb = static_lenb = stored_len + 5;
    }
    if (stored_len + 4 <= opt_lenb && buf != -1) {
        type = 0;
        tr_stored_block(s, buf, stored_len, eof);
    } else if (s->strategy == ZSO_FIXED || static_lenb == opt_lenb) {
        type = 1;
        send_bits(s, (1u << 1) + (eof ? 1u : 0u), 3);
        compress_block(s, g_static_ltree, g_static_dtree);
    } else {
        type = 2;
        send_bits(s, (2u << 1) + (eof ? 1u : 0u), 3);
        send_all_trees(s, s->l_max_code + 1, s->d_max_code + 1, max_blindex + 1);
        compress_block(s, s->dyn_ltree, s->dyn_dtree);
    }
    if (s->has_trace && s->trace.on_block)
        s->trace.on_block(s->trace.user, type, nsyms, s->abs_base + s->block_start, stored_len, eof, bit_start);
    init_block(s);
    if (eof) bi_windup(s);
}

/* Deflate.cs:951-956 */
static void flush_block_only(zso_deflate *s, int eof) {
    tr_flush_block(s, s->block_start >= 0 ? s->block_start : -1, s->strstart - s->block_start, eof);
    s->block_start = s->strstart;
    flush_pending(s);
}

/* Deflate.cs:910-948 */
static inline int tally_dist(zso_deflate *s, int dist, int len) {
    if (s->has_trace && s->trace.on_symbol)
        s->trace.on_symbol(s->trace.user, dist, len, s->abs_base + s->strstart);
    int di = s->d_buf + s->last_lit * 2;
    s->pending[di] = (uint8_t)(dist >> 8);
    s->pending[di + 1] = (uint8_t)dist;
    s->pending[s->l_buf + s->last_lit++] = (uint8_t)len;
    s->matches++;
    dist--;
    s->dyn_ltree[g_length_code[len] + LITERALS + 1].fc++;
    s->dyn_dtree[d_code(dist)].fc++;
    return s->last_lit == s->lit_bufsize - 1;
}
static inline int tally_lit(zso_deflate *s, unsigned c) {
    if (s->has_trace && s->trace.on_symbol) s->trace.on_symbol(s->trace.user, 0, (int)c, s->abs_base + s->strstart);
    int di = s->d_buf + s->last_lit * 2;
    s->pending[di] = 0;
    s->pending[di + 1] = 0;
    s->pending[s->l_buf + s->last_lit++] = (uint8_t)c;
    s->dyn_ltree[c].fc++;
    return s->last_lit == s->lit_bufsize - 1;
}

/* ---- LZ77 core ---- */

static inline uint32_t load_le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* Deflate.cs:866-877 */
static inline int insert_string(zso_deflate *s, int str) {
    uint32_t h = zso_hash_u32(load_le32(s->window + str + (MIN_MATCH - 1)), s->hash_variant) & s->hash_mask;
    unsigned cur = s->head[h];
    if ((int)cur != str) {
        s->prev[str & s->w_mask] = (uint16_t)cur;
        s->head[h] = (uint16_t)str;
    }
    return (int)cur;
}

/* Deflate.Intrinsics.cs:174-285 (all three variants: saturating subtract) */
static void slide_hash(zso_deflate *s) {
    unsigned w = (unsigned)s->w_size;
    for (int i = 0; i < s->hash_size; i++) s->head[i] = (uint16_t)(s->head[i] >= w ? s->head[i] - w : 0);
    for (int i = 0; i < s->w_size; i++) s->prev[i] = (uint16_t)(s->prev[i] >= w ? s->prev[i] - w : 0);
}

/* ZlibStream.cs:197-222 */
static int read_buffer(zso_deflate *s, uint8_t *dst, int size) {
    int len = s->avail_in;
    if (len > size) len = size;
    if (len == 0) return 0;
    s->avail_in -= len;
    if (s->noheader == 0) s->adler = zso_adler32(s->adler, s->next_in, (size_t)len);
    memcpy(dst, s->next_in, (size_t)len);
    s->next_in += len;
    s->total_in += len;
    return len;
}

/* Deflate.cs:967-1019 */
static void fill_window(zso_deflate *s) {
    do {
        int more = s->window_size - s->lookahead - s->strstart;
        if (s->strstart >= s->w_size + s->w_size - MIN_LOOKAHEAD) {
            memcpy(s->window, s->window + s->w_size, (size_t)s->w_size);
            s->match_start -= s->w_size;
            s->strstart -= s->w_size;
            s->block_start -= s->w_size;
            s->abs_base += s->w_size;
            slide_hash(s);
            more += s->w_size;
            if (s->has_trace && s->trace.on_slide) s->trace.on_slide(s->trace.user, s->abs_base + s->strstart, s->abs_base);
        }
        if (s->avail_in == 0) return;
        int n = read_buffer(s, s->window + s->strstart + s->lookahead, more);
        s->lookahead += n;
        int pre = 0;
        if (s->lookahead >= MIN_MATCH) {
            insert_string(s, s->strstart + 1);
            pre = 1;
        }
        if (s->has_trace && s->trace.on_read)
            s->trace.on_read(s->trace.user, s->abs_base + s->strstart, n, pre, s->abs_base);
    } while (s->lookahead < MIN_LOOKAHEAD && s->avail_in != 0);
}

/* Deflate.Intrinsics.cs:19-162: count of equal leading bytes, at most 256 */
static inline int compare256(const uint8_t *a, const uint8_t *b) {
    int n = 0;
    while (n < 256 && a[n] == b[n]) n++;
    return n;
}

static inline unsigned load_le16(const uint8_t *p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }

/* Deflate.cs:1022-1100 */
static int longest_match(zso_deflate *s, int cur_match) {
    int chain_length = s->max_chain;
    const uint8_t *scan = s->window + s->strstart;
    int best_len = s->prev_length;
    int limit = s->strstart > (s->w_size - MIN_LOOKAHEAD) ? s->strstart - (s->w_size - MIN_LOOKAHEAD) : 0;
    int nice = s->nice_match;
    int match_start = s->match_start;
    if (best_len == 0) best_len = 1;
    unsigned scan_start = load_le16(scan);
    unsigned scan_end = load_le16(scan + best_len - 1);
    if (s->prev_length >= s->good_match) chain_length >>= 2;
    if (nice > s->lookahead) nice = s->lookahead;
    do {
        if (cur_match >= s->strstart) break;
        const uint8_t *match = s->window + cur_match;
        if (load_le16(match + best_len - 1) != scan_end || load_le16(match) != scan_start) continue;
        int len = compare256(scan + 2, match + 2) + 2;
        if (len > best_len) {
            match_start = cur_match;
            best_len = len;
            if (len >= nice) break;
            scan_end = load_le16(scan + best_len - 1);
        }
    } while ((cur_match = s->prev[cur_match & s->w_mask]) > limit && --chain_length != 0);
    s->match_start = match_start;
    int r = best_len < s->lookahead ? best_len : s->lookahead;
    if (s->has_trace && s->trace.on_match)
        s->trace.on_match(s->trace.user, s->abs_base + s->strstart, s->prev_length, r, s->abs_base + match_start);
    return r;
}

/* Deflate.Slow.cs:18-159 */
static int deflate_slow(zso_deflate *s, int flush) {
    int hash_head = 0;
    int bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead < MIN_LOOKAHEAD && flush == ZSO_NO_FLUSH) return BS_NEED_MORE;
            if (s->lookahead == 0) break;
        }
        if (s->lookahead >= MIN_MATCH) hash_head = insert_string(s, s->strstart);
        s->prev_length = s->match_length;
        s->prev_match = s->match_start;
        s->match_length = MIN_MATCH - 1;
        if (hash_head != 0 && s->prev_length < s->max_lazy && s->strstart - hash_head <= s->w_size - MIN_LOOKAHEAD) {
            if (s->strategy != ZSO_HUFFMAN_ONLY) s->match_length = longest_match(s, hash_head);
            if (s->match_length <= 5 &&
                (s->strategy == ZSO_FILTERED || (s->match_length == MIN_MATCH && s->strstart - s->match_start > 4096)))
                s->match_length = MIN_MATCH - 1;
        }
        if (s->prev_length >= MIN_MATCH && s->match_length <= s->prev_length) {
            int max_insert = s->strstart + s->lookahead - MIN_MATCH;
            bflush = tally_dist(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
            s->lookahead -= s->prev_length - 1;
            s->prev_length -= 2;
            do {
                if (++s->strstart <= max_insert) hash_head = insert_string(s, s->strstart);
            } while (--s->prev_length != 0);
            s->match_available = 0;
            s->match_length = MIN_MATCH - 1;
            s->strstart++;
            if (bflush) {
                flush_block_only(s, 0);
                if (s->avail_out == 0) return BS_NEED_MORE;
            }
        } else if (s->match_available != 0) {
            bflush = tally_lit(s, s->window[s->strstart - 1]);
            if (bflush) flush_block_only(s, 0);
            s->strstart++;
            s->lookahead--;
            if (s->avail_out == 0) return BS_NEED_MORE;
        } else {
            s->match_available = 1;
            s->strstart++;
            s->lookahead--;
        }
    }
    if (s->match_available != 0) {
        tally_lit(s, s->window[s->strstart - 1]);
        s->match_available = 0;
    }
    flush_block_only(s, flush == ZSO_FINISH);
    if (s->avail_out == 0) return flush == ZSO_FINISH ? BS_FINISH_STARTED : BS_NEED_MORE;
    return flush == ZSO_FINISH ? BS_FINISH_DONE : BS_BLOCK_DONE;
}

/* Deflate.Fast.cs:20-128 */
static int deflate_fast(zso_deflate *s, int flush) {
    int hash_head, bflush;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead < MIN_LOOKAHEAD && flush == ZSO_NO_FLUSH) return BS_NEED_MORE;
            if (s->lookahead == 0) break;
        }
        hash_head = 0;
        if (s->lookahead >= MIN_MATCH) hash_head = insert_string(s, s->strstart);
        if (hash_head != 0 && (s->strstart - hash_head) <= s->w_size - MIN_LOOKAHEAD) {
            if (s->strategy != ZSO_HUFFMAN_ONLY) s->match_length = longest_match(s, hash_head);
        }
        if (s->match_length >= MIN_MATCH) {
            bflush = tally_dist(s, s->strstart - s->match_start, s->match_length - MIN_MATCH);
            s->lookahead -= s->match_length;
            if (s->match_length <= s->max_lazy && s->lookahead >= MIN_MATCH) {
                s->match_length--;
                do {
                    s->strstart++;
                    insert_string(s, s->strstart);
                } while (--s->match_length != 0);
                s->strstart++;
            } else {
                s->strstart += s->match_length;
                s->match_length = 0;
                /* Deflate.Fast.cs:100 recomputes a hash value and discards it */
            }
        } else {
            bflush = tally_lit(s, s->window[s->strstart]);
            s->lookahead--;
            s->strstart++;
        }
        if (bflush) {
            flush_block_only(s, 0);
            if (s->avail_out == 0) return BS_NEED_MORE;
        }
    }
    flush_block_only(s, flush == ZSO_FINISH);
    if (s->avail_out == 0) return flush == ZSO_FINISH ? BS_FINISH_STARTED : BS_NEED_MORE;
    return flush == ZSO_FINISH ? BS_FINISH_DONE : BS_BLOCK_DONE;
}

/* Deflate.Stored.cs:24-84 */
static int deflate_stored(zso_deflate *s, int flush) {
    int max_block_size = s->pending_size - 5 < s->w_size ? s->pending_size - 5 : s->w_size;
    for (;;) {
        if (s->lookahead <= 1) {
            fill_window(s);
            if (s->lookahead == 0 && flush == ZSO_NO_FLUSH) return BS_NEED_MORE;
            if (s->lookahead == 0) break;
        }
        s->strstart += s->lookahead;
        s->lookahead = 0;
        int max_start = s->block_start + max_block_size;
        if (s->strstart == 0 || s->strstart >= max_start) {
            s->lookahead = s->strstart - max_start;
            s->strstart = max_start;
            flush_block_only(s, 0);
            if (s->avail_out == 0) return BS_NEED_MORE;
        }
        if (s->strstart - s->block_start >= s->w_size - MIN_LOOKAHEAD) {
            flush_block_only(s, 0);
            if (s->avail_out == 0) return BS_NEED_MORE;
        }
    }
    flush_block_only(s, flush == ZSO_FINISH);
    if (s->avail_out == 0) return flush == ZSO_FINISH ? BS_FINISH_STARTED : BS_NEED_MORE;
    return flush == ZSO_FINISH ? BS_FINISH_DONE : BS_BLOCK_DONE;
}

/* Deflate.Rle.cs:18-104 */
static int deflate_rle(zso_deflate *s, int flush) {
    int bflush;
    for (;;) {
        if (s->lookahead <= MAX_MATCH) {
            fill_window(s);
            if (s->lookahead <= MAX_MATCH && flush == ZSO_NO_FLUSH) return BS_NEED_MORE;
        }
        if (s->lookahead == 0) break;
        s->match_length = 0;
        if (s->lookahead >= MIN_MATCH && s->strstart > 0) {
            const uint8_t *w = s->window;
            int p = s->strstart - 1;
            unsigned prev = w[p];
            if (prev == w[p + 1] && prev == w[p + 2] && prev == w[p + 3]) {
                /* the reference's unrolled scan (Deflate.Rle.cs:52-66) stops at the
                 * first byte that differs from prev or at strstart + MAX_MATCH */
                int len = 3;
                while (len < MAX_MATCH && w[s->strstart + len] == prev) len++;
                s->match_length = len;
                if (s->match_length > s->lookahead) s->match_length = s->lookahead;
            }
        }
        if (s->match_length >= MIN_MATCH) {
            bflush = tally_dist(s, 1, s->match_length - MIN_MATCH);
            s->lookahead -= s->match_length;
            s->strstart += s->match_length;
            s->match_length = 0;
        } else {
            bflush = tally_lit(s, s->window[s->strstart]);
            s->lookahead--;
            s->strstart++;
        }
        if (bflush) {
            flush_block_only(s, 0);
            if (s->avail_out == 0) return BS_NEED_MORE;
        }
    }
    flush_block_only(s, flush == ZSO_FINISH);
    if (s->avail_out == 0) return flush == ZSO_FINISH ? BS_FINISH_STARTED : BS_NEED_MORE;
    return flush == ZSO_FINISH ? BS_FINISH_DONE : BS_BLOCK_DONE;
}

/* ---- lifecycle ---- */

/* Deflate.cs:730-752 */
static void lm_init(zso_deflate *s) {
    s->window_size = 2 * s->w_size;
    memset(s->head, 0, sizeof(uint16_t) * (size_t)s->hash_size);
    s->max_lazy = k_levels[s->level].lazy;
    s->good_match = k_levels[s->level].good;
    s->nice_match = k_levels[s->level].nice;
    s->max_chain = k_levels[s->level].chain;
    s->strstart = 0;
    s->block_start = 0;
    s->lookahead = 0;
    s->match_length = s->prev_length = MIN_MATCH - 1;
    s->match_available = 0;
}

/* Deflate.cs:879-900 + Trees.cs:770-779 */
static void deflate_reset(zso_deflate *s) {
    s->total_in = s->total_out = 0;
    s->msg = NULL;
    s->pending_n = 0;
    s->pending_out = 0;
    if (s->noheader < 0) s->noheader = 0;
    s->status = s->noheader != 0 ? ST_BUSY : ST_INIT;
    s->adler = 1;
    s->last_flush = ZSO_NO_FLUSH;
    s->bi_buf = 0;
    s->bi_valid = 0;
    s->last_eob_len = 8;
    init_block(s);
    lm_init(s);
}

zso_deflate *zso_deflate_new(int level, int strategy, int window_bits, int mem_level, int hash_variant) {
    init_tables();
    int noheader = 0;
    if (level == -1) level = 6;
    if (level == 0) mem_level = 7; /* Deflate.cs:246-249 */
    if (window_bits < 0) {
        noheader = 1;
        window_bits = -window_bits;
    }
    if (mem_level < 1 || mem_level > 9) return NULL;
    if (window_bits < 9 || window_bits > 15) return NULL;
    if (level < 0 || level > 9) return NULL;
    if (strategy < ZSO_DEFAULT_STRATEGY || strategy > ZSO_FIXED) return NULL;
    zso_deflate *s = (zso_deflate *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->noheader = noheader;
    s->w_bits = window_bits;
    s->w_size = 1 << window_bits;
    s->w_mask = s->w_size - 1;
    s->hash_bits = mem_level + 7;
    s->hash_size = 1 << s->hash_bits;
    s->hash_mask = (uint32_t)s->hash_size - 1;
    s->lit_bufsize = 1 << (mem_level + 6);
    s->d_buf = s->lit_bufsize;
    s->l_buf = 3 * s->lit_bufsize;
    s->pending_size = s->lit_bufsize * 4;
    s->level = level;
    s->strategy = strategy;
    s->hash_variant = hash_variant;
    /* zero-initialised; 512 bytes of zero slack so that the 4-byte hash read
     * and Compare256 never leave the allocation */
    s->window = (uint8_t *)calloc((size_t)s->w_size * 2 + 512, 1);
    s->prev = (uint16_t *)calloc((size_t)s->w_size, sizeof(uint16_t));
    s->head = (uint16_t *)calloc((size_t)s->hash_size, sizeof(uint16_t));
    s->pending = (uint8_t *)calloc((size_t)s->pending_size + 16, 1);
    if (!s->window || !s->prev || !s->head || !s->pending) {
        zso_deflate_free(s);
        return NULL;
    }
    deflate_reset(s);
    return s;
}

void zso_deflate_free(zso_deflate *s) {
    if (!s) return;
    free(s->window);
    free(s->prev);
    free(s->head);
    free(s->pending);
    free(s);
}

void zso_deflate_set_trace(zso_deflate *s, const zso_trace *t) {
    if (t) {
        s->trace = *t;
        s->has_trace = 1;
    } else {
        s->has_trace = 0;
    }
}

const char *zso_deflate_message(const zso_deflate *s) { return s->msg; }
uint32_t zso_deflate_adler(const zso_deflate *s) { return s->adler; }
int64_t zso_deflate_total_in(const zso_deflate *s) { return s->total_in; }
int64_t zso_deflate_total_out(const zso_deflate *s) { return s->total_out; }

/* Deflate.cs:436-636 */
static int deflate_compress(zso_deflate *s, int flush) {
    if (flush > ZSO_FINISH || flush < 0) return ZSO_STREAM_ERROR;
    if (s->next_out == NULL || (s->next_in == NULL && s->avail_in != 0) ||
        (s->status == ST_FINISH && flush != ZSO_FINISH)) {
        s->msg = k_errmsg[ZSO_NEED_DICT - ZSO_STREAM_ERROR];
        return ZSO_STREAM_ERROR;
    }
    if (s->avail_out == 0) {
        s->msg = k_errmsg[ZSO_NEED_DICT - ZSO_BUF_ERROR];
        return ZSO_BUF_ERROR;
    }
    int old_flush = s->last_flush;
    s->last_flush = flush;
    if (s->status == ST_INIT) {
        /* zlib header (Deflate.cs:464-493) */
        unsigned header = (8u + ((unsigned)(s->w_bits - 8) << 4)) << 8;
        unsigned level_flags = (((unsigned)(s->level - 1)) & 0xFF) >> 1;
        if (level_flags > 3) level_flags = 3;
        header |= level_flags << 6;
        if (s->strstart != 0) header |= 0x20;
        header += 31 - (header % 31);
        s->status = ST_BUSY;
        put_short_msb(s, header);
        if (s->strstart != 0) {
            put_short_msb(s, s->adler >> 16);
            put_short_msb(s, s->adler & 0xFFFF);
        }
        s->adler = 1;
    }
    if (s->pending_n != 0) {
        flush_pending(s);
        if (s->avail_out == 0) {
            s->last_flush = -1;
            return ZSO_OK;
        }
    } else if (s->avail_in == 0 && flush <= old_flush && flush != ZSO_FINISH) {
        s->msg = k_errmsg[ZSO_NEED_DICT - ZSO_BUF_ERROR];
        return ZSO_BUF_ERROR;
    }
    if (s->status == ST_FINISH && s->avail_in != 0) {
        s->msg = k_errmsg[ZSO_NEED_DICT - ZSO_BUF_ERROR];
        return ZSO_BUF_ERROR;
    }
    if (s->avail_in != 0 || s->lookahead != 0 || (flush != ZSO_NO_FLUSH && s->status != ST_FINISH)) {
        int bstate;
        if (s->strategy == ZSO_RLE) {
            bstate = deflate_rle(s, flush);
        } else {
            switch (k_levels[s->level].func) {
            case FN_STORED: bstate = deflate_stored(s, flush); break;
            case FN_FAST: bstate = deflate_fast(s, flush); break;
            default: bstate = deflate_slow(s, flush); break;
            }
        }
        if (bstate == BS_FINISH_STARTED || bstate == BS_FINISH_DONE) s->status = ST_FINISH;
        if (bstate == BS_NEED_MORE || bstate == BS_FINISH_STARTED) {
            if (s->avail_out == 0) s->last_flush = -1;
            return ZSO_OK;
        }
        if (bstate == BS_BLOCK_DONE) {
            if (flush == ZSO_PARTIAL_FLUSH) {
                tr_align(s);
            } else {
                tr_stored_block(s, 0, 0, 0);
                if (flush == ZSO_FULL_FLUSH) memset(s->head, 0, sizeof(uint16_t) * (size_t)s->hash_size);
            }
            flush_pending(s);
            if (s->avail_out == 0) {
                s->last_flush = -1;
                return ZSO_OK;
            }
        }
    }
    if (flush != ZSO_FINISH) return ZSO_OK;
    if (s->noheader != 0) return ZSO_STREAM_END;
    put_short_msb(s, s->adler >> 16);
    put_short_msb(s, s->adler & 0xFFFF);
    flush_pending(s);
    s->noheader = -1;
    return s->pending_n != 0 ? ZSO_OK : ZSO_STREAM_END;
}

int zso_deflate_call(zso_deflate *s, const uint8_t **next_in, int *avail_in, uint8_t **next_out, int *avail_out,
                     int flush) {
    s->next_in = *next_in;
    s->avail_in = *avail_in;
    s->next_out = *next_out;
    s->avail_out = *avail_out;
    int r = deflate_compress(s, flush);
    *next_in = s->next_in;
    *avail_in = s->avail_in;
    *next_out = s->next_out;
    *avail_out = s->avail_out;
    return r;
}

size_t zso_compress_bound(size_t n) { return n + (n >> 10) * 8 + 1024; }

/* ZlibOutputStream.WriteCore / Finish (ZlibOutputStream.cs:125-168,213-256) */
size_t zso_compress_stream(const uint8_t *in, size_t n, const size_t *chunk_lens, size_t n_chunks, int level,
                           int strategy, int flush_mode, int hash_variant, uint8_t *out, size_t out_cap,
                           const zso_trace *trace) {
    return zso_compress_stream_modes(in, n, chunk_lens, n_chunks, level, strategy, flush_mode, NULL, hash_variant, out, out_cap, trace);
}

/* the same with ZlibOptions.FlushMode set anew before every Write (the property is read by every WriteCore,
 * ZlibOutputStream.cs:140): flush_modes[i] is the mode of Write i; NULL: flush_mode for all of them */
size_t zso_compress_stream_modes(const uint8_t *in, size_t n, const size_t *chunk_lens, size_t n_chunks, int level,
                                 int strategy, int flush_mode_all, const int *flush_modes, int hash_variant, uint8_t *out, size_t out_cap,
                                 const zso_trace *trace) {
    zso_deflate *s = zso_deflate_new(level, strategy, 15, 8, hash_variant);
    if (!s) return (size_t)-1;
    if (trace) zso_deflate_set_trace(s, trace);
    uint8_t chunk[512];
    size_t produced = 0, off = 0;
    size_t one = n;
    if (n_chunks == 0 || chunk_lens == NULL) {
        chunk_lens = &one;
        n_chunks = 1;
    }
    int state = ZSO_OK;
    for (size_t c = 0; c < n_chunks; c++) {
        size_t len = chunk_lens[c];
        if (len == 0) continue; /* WriteCore returns immediately on an empty span */
        /* the reference takes int lengths; split >2 GiB writes is not needed here */
        s->next_in = in + off;
        s->avail_in = (int)len;
        off += len;
        const int flush_mode = flush_modes ? flush_modes[c] : flush_mode_all;
        do {
            s->next_out = chunk;
            s->avail_out = (int)sizeof chunk;
            state = deflate_compress(s, flush_mode);
            if (state != ZSO_OK && state != ZSO_STREAM_END) goto fail;
            size_t got = sizeof chunk - (size_t)s->avail_out;
            if (produced + got > out_cap) goto fail;
            memcpy(out + produced, chunk, got);
            produced += got;
            if (state == ZSO_STREAM_END) break;
        } while (s->avail_in > 0 || s->avail_out == 0);
    }
    do {
        s->next_out = chunk;
        s->avail_out = (int)sizeof chunk;
        state = deflate_compress(s, ZSO_FINISH);
        if (state != ZSO_OK && state != ZSO_STREAM_END) goto fail;
        size_t got = sizeof chunk - (size_t)s->avail_out;
        if (produced + got > out_cap) goto fail;
        memcpy(out + produced, chunk, got);
        produced += got;
        if (state == ZSO_STREAM_END) break;
    } while (s->avail_in > 0 || s->avail_out == 0);
    if (s->overflow) goto fail;
    zso_deflate_free(s);
    return produced;
fail:
    zso_deflate_free(s);
    return (size_t)-1;
}

/* ------------------------------------------------------------------ */
/* .NET System.Random (seeded, Knuth subtractive generator) -- used only to
 * rebuild the reference tests' input buffers. */
void zso_dotnet_random_bytes(int seed, uint8_t *buf, size_t len) {
    int32_t seed_array[56];
    int32_t subtraction = seed == INT32_MIN ? INT32_MAX : (seed < 0 ? -seed : seed);
    int32_t mj = 161803398 - subtraction;
    int32_t mk = 1;
    memset(seed_array, 0, sizeof seed_array);
    seed_array[55] = mj;
    int ii = 0;
    for (int i = 1; i < 55; i++) {
        if ((ii += 21) >= 55) ii -= 55;
        seed_array[ii] = mk;
        mk = mj - mk;
        if (mk < 0) mk += INT32_MAX;
        mj = seed_array[ii];
    }
    for (int k = 1; k < 5; k++) {
        for (int i = 1; i < 56; i++) {
            int n = i + 30;
            if (n >= 55) n -= 55;
            seed_array[i] = (int32_t)((uint32_t)seed_array[i] - (uint32_t)seed_array[1 + n]);
            if (seed_array[i] < 0) seed_array[i] += INT32_MAX;
        }
    }
    int inext = 0, inextp = 21;
    for (size_t i = 0; i < len; i++) {
        if (++inext >= 56) inext = 1;
        if (++inextp >= 56) inextp = 1;
        int32_t r = (int32_t)((uint32_t)seed_array[inext] - (uint32_t)seed_array[inextp]);
        if (r == INT32_MAX) r--;
        if (r < 0) r += INT32_MAX;
        seed_array[inext] = r;
        buf[i] = (uint8_t)r;
    }
}
