/*
 * zs_inflate_oracle.c -- CPU inflate for the oracle (TEST INFRASTRUCTURE ONLY).
 *
 * RFC 1950/1951 decoding is deterministic, so any conformant decoder yields
 * the bytes the reference's Inflate.cs / InflateBlocks.cs / InfCodes.cs /
 * InfTree.cs yield.  This is a plain canonical-Huffman decoder (count/symbol
 * tables, one bit at a time -- clarity over speed) that keeps the reference's
 * error classes and message strings:
 *   Inflate.cs:134,142,166,243,339; InflateBlocks.cs:237,278,394,569;
 *   InfCodes.cs:294,349; InfTree.cs:377-427.
 */
#include "zs_oracle.h"

#include <string.h>

typedef struct {
    const uint8_t *in;
    size_t in_len, in_pos;
    uint32_t bitbuf;
    int bitcnt;
    uint8_t *out;
    size_t out_cap, out_pos;
    const char *msg;
    int err;
} inf_state;

typedef struct {
    uint16_t count[16];
    uint16_t symbol[288];
} huff;

static int need_bits(inf_state *s, int n, uint32_t *val) {
    while (s->bitcnt < n) {
        if (s->in_pos >= s->in_len) {
            s->err = ZSO_BUF_ERROR;
            return -1;
        }
        s->bitbuf |= (uint32_t)s->in[s->in_pos++] << s->bitcnt;
        s->bitcnt += 8;
    }
    *val = s->bitbuf & ((1u << n) - 1u);
    s->bitbuf >>= n;
    s->bitcnt -= n;
    return 0;
}

/* returns 0 complete, >0 incomplete (left), <0 oversubscribed */
static int build_huff(huff *h, const uint8_t *lengths, int n) {
    int offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++) h->count[lengths[i]]++;
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++)
        if (lengths[i] != 0) h->symbol[offs[lengths[i]]++] = (uint16_t)i;
    return left;
}

static int decode_sym(inf_state *s, const huff *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        uint32_t b;
        if (need_bits(s, 1, &b)) return -1;
        code |= (int)b;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2; /* ran out of codes */
}

static const uint16_t k_lbase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                     31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t k_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t k_dbase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                     193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int fail(inf_state *s, int code, const char *msg) {
    s->err = code;
    s->msg = msg;
    return -1;
}

static int inflate_codes(inf_state *s, const huff *lh, const huff *dh) {
    for (;;) {
        int sym = decode_sym(s, lh);
        if (sym == -1) return -1;
        if (sym < 0) return fail(s, ZSO_DATA_ERROR, "invalid literal/length code");
        if (sym < 256) {
            if (s->out_pos >= s->out_cap) return fail(s, ZSO_BUF_ERROR, "buffer error");
            s->out[s->out_pos++] = (uint8_t)sym;
        } else if (sym == 256) {
            return 0;
        } else {
            sym -= 257;
            if (sym >= 29) return fail(s, ZSO_DATA_ERROR, "invalid literal/length code");
            uint32_t eb;
            if (need_bits(s, k_lext[sym], &eb)) return -1;
            size_t len = k_lbase[sym] + eb;
            int ds = decode_sym(s, dh);
            if (ds == -1) return -1;
            if (ds < 0 || ds >= 30) return fail(s, ZSO_DATA_ERROR, "invalid distance code");
            if (need_bits(s, k_dext[ds], &eb)) return -1;
            size_t dist = k_dbase[ds] + eb;
            if (dist > s->out_pos || dist > 32768) return fail(s, ZSO_DATA_ERROR, "invalid distance code");
            if (s->out_pos + len > s->out_cap) return fail(s, ZSO_BUF_ERROR, "buffer error");
            for (size_t i = 0; i < len; i++, s->out_pos++) s->out[s->out_pos] = s->out[s->out_pos - dist];
        }
    }
}

int zso_inflate_oneshot(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len,
                        size_t *in_used, const char **msg) {
    inf_state st;
    memset(&st, 0, sizeof st);
    st.in = in;
    st.in_len = in_len;
    st.out = out;
    st.out_cap = out_cap;
    st.err = ZSO_OK;
    inf_state *s = &st;
    if (msg) *msg = NULL;
#define BAIL(code, m)            \
    do {                         \
        if (msg) *msg = (m);     \
        if (out_len) *out_len = st.out_pos; \
        if (in_used) *in_used = st.in_pos;  \
        return (code);           \
    } while (0)
    /* zlib header: Inflate.cs:120-170 */
    if (in_len < 2) BAIL(ZSO_BUF_ERROR, "buffer error");
    unsigned cmf = in[0], flg = in[1];
    st.in_pos = 2;
    if ((cmf & 0x0F) != 8) BAIL(ZSO_DATA_ERROR, "unknown compression method");
    if ((cmf >> 4) + 8 > 15) BAIL(ZSO_DATA_ERROR, "invalid window size");
    if (((cmf << 8) + flg) % 31 != 0) BAIL(ZSO_DATA_ERROR, "incorrect header check");
    if (flg & 0x20) BAIL(ZSO_NEED_DICT, "need dictionary");

    huff lh, dh;
    int last = 0;
    while (!last) {
        uint32_t v;
        if (need_bits(s, 1, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
        last = (int)v;
        if (need_bits(s, 2, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
        if (v == 0) {
            st.bitbuf = 0;
            st.bitcnt = 0;
            if (st.in_pos + 4 > in_len) BAIL(ZSO_BUF_ERROR, "buffer error");
            unsigned len = in[st.in_pos] | (in[st.in_pos + 1] << 8);
            unsigned nlen = in[st.in_pos + 2] | (in[st.in_pos + 3] << 8);
            st.in_pos += 4;
            if (len != (~nlen & 0xFFFF)) BAIL(ZSO_DATA_ERROR, "invalid stored block lengths");
            if (st.in_pos + len > in_len) BAIL(ZSO_BUF_ERROR, "buffer error");
            if (st.out_pos + len > out_cap) BAIL(ZSO_BUF_ERROR, "buffer error");
            memcpy(out + st.out_pos, in + st.in_pos, len);
            st.in_pos += len;
            st.out_pos += len;
        } else if (v == 1) {
            uint8_t lengths[288];
            int i = 0;
            for (; i < 144; i++) lengths[i] = 8;
            for (; i < 256; i++) lengths[i] = 9;
            for (; i < 280; i++) lengths[i] = 7;
            for (; i < 288; i++) lengths[i] = 8;
            build_huff(&lh, lengths, 288);
            for (i = 0; i < 30; i++) lengths[i] = 5;
            build_huff(&dh, lengths, 30);
            if (inflate_codes(s, &lh, &dh)) BAIL(st.err, st.msg ? st.msg : "buffer error");
        } else if (v == 2) {
            uint32_t nlen, ndist, ncode;
            if (need_bits(s, 5, &nlen) || need_bits(s, 5, &ndist) || need_bits(s, 4, &ncode))
                BAIL(ZSO_BUF_ERROR, "buffer error");
            nlen += 257;
            ndist += 1;
            ncode += 4;
            if (nlen > 286 || ndist > 30) BAIL(ZSO_DATA_ERROR, "too many length or distance symbols");
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t lengths[320];
            memset(lengths, 0, sizeof lengths);
            for (uint32_t i = 0; i < ncode; i++) {
                if (need_bits(s, 3, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
                lengths[order[i]] = (uint8_t)v;
            }
            huff blh;
            int r = build_huff(&blh, lengths, 19);
            if (r < 0) BAIL(ZSO_DATA_ERROR, "oversubscribed dynamic bit lengths tree");
            /* incomplete is accepted only for a single code of length 1 (Huft_build, InfTree.cs:364: y != 0 && g != 1) */
            if (r > 0 && !(blh.count[1] == 1 && 19 - blh.count[0] == 1)) BAIL(ZSO_DATA_ERROR, "incomplete dynamic bit lengths tree");
            uint32_t idx = 0;
            uint8_t ll[320];
            while (idx < nlen + ndist) {
                int sym = decode_sym(s, &blh);
                if (sym == -1) BAIL(ZSO_BUF_ERROR, "buffer error");
                if (sym < 0) BAIL(ZSO_DATA_ERROR, "invalid bit length repeat");
                if (sym < 16) {
                    ll[idx++] = (uint8_t)sym;
                } else {
                    unsigned prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) BAIL(ZSO_DATA_ERROR, "invalid bit length repeat");
                        prev = ll[idx - 1];
                        if (need_bits(s, 2, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
                        rep = 3 + v;
                    } else if (sym == 17) {
                        if (need_bits(s, 3, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
                        rep = 3 + v;
                    } else {
                        if (need_bits(s, 7, &v)) BAIL(ZSO_BUF_ERROR, "buffer error");
                        rep = 11 + v;
                    }
                    if (idx + rep > nlen + ndist) BAIL(ZSO_DATA_ERROR, "invalid bit length repeat");
                    while (rep--) ll[idx++] = (uint8_t)prev;
                }
            }
            r = build_huff(&lh, ll, (int)nlen);
            if (r < 0) BAIL(ZSO_DATA_ERROR, "oversubscribed literal/length tree");
            if (r > 0 && !(lh.count[1] == 1 && (int)nlen - lh.count[0] == 1)) BAIL(ZSO_DATA_ERROR, "incomplete literal/length tree");
            r = build_huff(&dh, ll + nlen, (int)ndist);
            if (r < 0) BAIL(ZSO_DATA_ERROR, "oversubscribed distance tree");
            if (r > 0) {
                if ((int)ndist - dh.count[0] == 0) {
                    if (nlen > 257) BAIL(ZSO_DATA_ERROR, "empty distance tree with lengths");
                } else if (!(dh.count[1] == 1 && (int)ndist - dh.count[0] == 1)) {
                    BAIL(ZSO_DATA_ERROR, "incomplete distance tree");
                }
            }
            if (inflate_codes(s, &lh, &dh)) BAIL(st.err, st.msg ? st.msg : "buffer error");
        } else {
            BAIL(ZSO_DATA_ERROR, "invalid block type");
        }
    }
    /* trailer: Inflate.cs:300-345 */
    if (st.in_pos + 4 > in_len) BAIL(ZSO_BUF_ERROR, "buffer error");
    uint32_t want = ((uint32_t)in[st.in_pos] << 24) | ((uint32_t)in[st.in_pos + 1] << 16) |
                    ((uint32_t)in[st.in_pos + 2] << 8) | in[st.in_pos + 3];
    st.in_pos += 4;
    if (want != zso_adler32(1, out, st.out_pos)) BAIL(ZSO_DATA_ERROR, "incorrect data check");
    BAIL(ZSO_STREAM_END, NULL);
#undef BAIL
}
