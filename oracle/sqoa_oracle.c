/*
 * sqoa_oracle.c -- TEST INFRASTRUCTURE ONLY (see sqoa_oracle.h).
 *
 * Restatement of the codec in /root/reference/seqoia.h (H:line), written from
 * the format rules of SURVEY.md Appendix A: the state carried from pixel to
 * pixel is (previous pixel, run remainder r = k mod M, last-writer-per-slot
 * table); every emitted byte is a function of that state and the pixel.
 * Scalar, single-threaded, no SIMD: it is a checker, not a product path.
 */
#include "sqoa_oracle.h"

#include <stdlib.h>
#include <string.h>

#define HDR 14
#define PIXELS_MAX 400000000u /* H:432 */

static uint32_t be32(const uint8_t *b) { /* H:448-454 */
    return (uint32_t)b[0] << 24 | (uint32_t)b[1] << 16 | (uint32_t)b[2] << 8 | b[3];
}
static void put32(uint8_t *b, uint32_t v) { /* H:441-446 */
    b[0] = v >> 24; b[1] = v >> 16; b[2] = v >> 8; b[3] = v;
}

size_t orc_encode_bound(unsigned w, unsigned h, unsigned hdr_channels) {
    return (size_t)w * h * (hdr_channels + 1) + HDR + 8 + 1;
}

/* pixel as r | g<<8 | b<<16 | a<<24, the reference's union view (H:434-437) */
static uint32_t load_px(const uint8_t *p, int col, int has_alpha, uint32_t prev) {
    uint32_t v;
    if (col == 3) v = p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16; /* H:531-535 */
    else v = (uint32_t)p[0] << 8; /* mono: only g is set, r=b=0 (H:536-538) */
    v |= has_alpha ? (uint32_t)p[col] << 24 : (prev & 0xff000000u); /* H:540-542 */
    return v;
}

static unsigned qoi_hash(uint32_t v) { /* H:414-417 */
    return ((v & 255) * 3 + ((v >> 8) & 255) * 5 + ((v >> 16) & 255) * 7 + (v >> 24) * 11) & 63;
}

/* Bytes of the literal op for a pixel that differs from its predecessor
 * (H:584-635).  `hit` is the QOI index verdict, `pos` its slot. */
static int literal_op(uint8_t *o, uint32_t px, uint32_t pv, int col, int qoi, int hit, unsigned pos) {
    int8_t vr = (int8_t)((px) - (pv));
    int8_t vg = (int8_t)((px >> 8) - (pv >> 8));
    int8_t vb = (int8_t)((px >> 16) - (pv >> 16));
    int8_t va = (int8_t)((px >> 24) - (pv >> 24));
    int8_t vg_r = (int8_t)(vr - vg), vg_b = (int8_t)(vb - vg);
    int na = va != 0, n = 0;
    uint8_t r = px, g = px >> 8, b = px >> 16, a = px >> 24;
    if (qoi) {
        if (hit) { o[0] = (uint8_t)pos; return 1; }           /* H:567-569 */
        if (na) { o[0] = 0xff; o[1] = r; o[2] = g; o[3] = b; o[4] = a; return 5; } /* H:573-580 */
        if (vr > -3 && vr < 2 && vg > -3 && vg < 2 && vb > -3 && vb < 2) {          /* H:593-600 */
            o[0] = 0x40 | (vr + 2) << 4 | (vg + 2) << 2 | (vb + 2);
            return 1;
        }
    }
    if (col == 1 && na) { o[0] = 0xff; o[1] = g; o[2] = a; return 3; } /* H:601-605 */
    if (vg_r > -9 && vg_r < 8 && vg > -33 && vg < 32 && vg_b > -9 && vg_b < 8 && va > -17 && va < 16) {
        o[n++] = 0x80 | (vg + 32);                                      /* H:612 */
        if (col == 3) {
            o[n++] = (uint8_t)((vg_r + 8) << 4 | (vg_b + 8));           /* H:614 */
            if (na) o[n++] = 0x60 | (va + 16);                          /* H:616-618 */
        }
        return n;
    }
    o[n++] = 0xfe | na;                                                 /* H:622 */
    if (col == 3) { o[n++] = r; o[n++] = g; o[n++] = b; } else o[n++] = g;
    if (na) o[n++] = a;
    return n;
}

void *orc_encode(const void *pixels, const orc_desc *d, int *out_len) {
    if (!pixels || !out_len || !d || d->width == 0 || d->height == 0 || d->channels < 1 ||
        d->channels > 6 || d->colorspace > 1 || d->height >= PIXELS_MAX / d->width)
        return NULL; /* H:465-473 */
    int qoi = d->qoi_compat != 0;
    int has_alpha = (d->channels & 1) == 0;
    int col = d->channels < 3 ? 1 : 3;
    if (col == 1 && qoi) return NULL; /* H:477-480 */
    int ch = col + has_alpha;          /* stride and header byte (H:486, H:505) */
    size_t N = (size_t)d->width * d->height;
    unsigned M = qoi ? 62 : 512;       /* H:508-514 */

    uint8_t *out = malloc(orc_encode_bound(d->width, d->height, ch));
    if (!out) return NULL;
    size_t p = 0;
    put32(out, qoi ? 0x716f6966u : 0x53716f61u); /* "qoif" / "Sqoa" H:419-424 */
    put32(out + 4, d->width);
    put32(out + 8, d->height);
    out[12] = ch; out[13] = d->colorspace;
    p = HDR;
    if (!qoi) out[p++] = 0x31; /* H:513 */

    uint32_t table[64];
    memset(table, 0, sizeof table); /* H:518 */
    const uint8_t *src = pixels;
    uint32_t pv = 0xff000000u;      /* (0,0,0,255) H:521-525 */
    unsigned r = 0;                 /* run remainder = k mod M (Appendix A.2) */
    for (size_t i = 0; i < N; i++) {
        uint32_t px = load_px(src + i * ch, col, has_alpha, pv);
        if (px == pv) {
            if (++r == M) { out[p++] = 0xfd; r = 0; } /* H:544-550 */
            continue;
        }
        if (r) { /* pending remainder: ceil(r/61) run bytes, full 61-runs first (H:554-561) */
            for (; r > 61; r -= 61) out[p++] = 0xfc;
            out[p++] = 0xc0 | (r - 1);
            r = 0;
        }
        int hit = 0; unsigned pos = 0;
        if (qoi) { /* last-writer-per-slot table (H:563-572) */
            pos = qoi_hash(px);
            hit = table[pos] == px;
            table[pos] = px;
        }
        p += literal_op(out + p, px, pv, col, qoi, hit, pos);
        pv = px; /* H:636 */
    }
    if (r) out[p++] = 0xfd; /* trailing run, any length (H:640-642) */
    memset(out + p, 0, 7); out[p + 7] = 1; p += 8; /* H:644-646 */
    *out_len = (int)p;
    return out;
}

/* ---- decode ------------------------------------------------------------- */

typedef struct { const uint8_t *b; int p, ref, refp; } cursor;

/* SQOA_NEXT (H:418): when the cursor sits on `ref` it is moved to refp+1 and
 * NOT advanced; otherwise post-increment. Returns the index to read. */
static int next_idx(cursor *c) {
    if (c->p == c->ref) { c->p = c->refp + 1; return c->p; }
    return c->p++;
}

void *orc_decode(const void *data, int size, orc_desc *d, int channels) {
    if (!data || !d || channels > 4 || channels < 0 || size < HDR + 8) return NULL; /* H:662-668 */
    const uint8_t *bytes = data;
    uint32_t magic = be32(bytes);
    d->width = be32(bytes + 4);
    d->height = be32(bytes + 8);
    d->channels = bytes[12];
    d->colorspace = bytes[13];
    d->qoi_compat = bytes[14] != 0x31; /* H:677 */
    if (d->width == 0 || d->height == 0 || d->channels < 1 || d->channels > 6 || d->colorspace > 1 ||
        !(magic == 0x716f6966u || magic == 0x53716f61u) || (magic == 0x716f6966u && !d->qoi_compat) ||
        d->height >= PIXELS_MAX / d->width)
        return NULL; /* H:679-688 */
    int col = d->channels < 3 ? 1 : 3;
    unsigned index_size = col == 1 ? 128 : 64; /* H:690-697 */
    int add_alpha = (channels & 1) == 0;
    if (channels == 0) { add_alpha = (d->channels & 1) == 0; channels = col + add_alpha; } /* H:699-702 */
    int qoi = d->qoi_compat;
    cursor c = { bytes, HDR, -1, 0 };
    if (!qoi) c.p++; /* start byte, already known to be 0x31 (H:705) */

    size_t N = (size_t)d->width * d->height;
    uint8_t *out = malloc(N * channels);
    if (!out) return NULL;
    uint32_t table[128];
    memset(table, 0, sizeof table);
    uint8_t r = 0, g = 0, b = 0, a = 255;
    int run = 0, chunks_len = size - 8;
    for (size_t i = 0; i < N; i++) {
        if (run > 0) run--;
        else if (c.p < chunks_len) {
            int b1 = bytes[next_idx(&c)];
            if (!qoi && b1 < 0x60) { /* SQOA_OP_REF H:729-738 */
                c.refp = c.p;
                c.ref = c.p - (b1 & 31);
                c.p = c.ref - 2 - (b1 >> 5);
                if (c.p < 0) { free(out); return NULL; }
                b1 = bytes[c.p++];
            }
            if (b1 >= 0xfe) { /* RGB / RGBA H:740-752 */
                if (col == 3) { r = bytes[next_idx(&c)]; g = bytes[next_idx(&c)]; b = bytes[next_idx(&c)]; }
                else g = bytes[next_idx(&c)];
                if (b1 == 0xff) a = bytes[next_idx(&c)];
            } else if (qoi && (unsigned)b1 < index_size) { /* INDEX H:753-755 */
                uint32_t v = table[b1];
                r = v; g = v >> 8; b = v >> 16; a = v >> 24;
            } else if (qoi && (b1 & 0xc0) == 0x40) { /* DIFF H:756-760 */
                r += ((b1 >> 4) & 3) - 2; g += ((b1 >> 2) & 3) - 2; b += (b1 & 3) - 2;
            } else if ((b1 & 0xc0) == 0x80) { /* LUMA H:761-769 */
                int vg = (b1 & 0x3f) - 32;
                g += vg;
                if (col == 3) {
                    int b2 = bytes[next_idx(&c)];
                    r += vg - 8 + ((b2 >> 4) & 15);
                    b += vg - 8 + (b2 & 15);
                }
            } else if (!qoi && b1 == 0xfd) run = 511; /* BIGRUN H:770-772 */
            else run = b1 & 0x3f;                     /* RUN, and every other byte H:773-775 */
            if (!qoi && col == 3 && bytes[c.p] >= 0x60 && bytes[c.p] < 0x80) { /* ALPHA suffix H:777-783 */
                int s = bytes[next_idx(&c)];
                a = a + (s & 31) - 16;
            }
            if (qoi) { /* H:785-787 */
                uint32_t v = r | (uint32_t)g << 8 | (uint32_t)b << 16 | (uint32_t)a << 24;
                table[(r * 3 + g * 5 + b * 7 + a * 11) % index_size] = v;
            }
        }
        uint8_t *o = out + i * channels; /* H:790-805 */
        if (channels >= 3 && col == 3) { o[0] = r; o[1] = g; o[2] = b; }
        else { o[0] = g; if (channels >= 3) { o[1] = g; o[2] = g; } }
        if (add_alpha) o[channels - 1] = a;
    }
    return out;
}

/* ---- synthetic inputs ----------------------------- */

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

void orc_synth(uint8_t *dst, unsigned w, unsigned h, unsigned channels, int cls_kind, uint32_t seed, uint32_t img) {
    uint32_t s0 = mix32(seed * 0x9E3779B9u + img);
    for (unsigned y = 0; y < h; y++)
        for (unsigned x = 0; x < w; x++) {
            uint32_t i = y * w + x;
            uint32_t n = mix32(s0 ^ (i * 0x85EBCA6Bu + 0x01234567u));
            uint32_t m = mix32(s0 ^ ((i >> 1) * 0x85EBCA6Bu + 0x089ABCDEu));
            uint32_t t; unsigned cls;
            if (cls_kind == ORC_CLASS_MIXED) {
                t = mix32(s0 ^ ((y >> 6) * 73856093u) ^ ((x >> 6) * 19349663u));
                cls = t & 7;
            } else if (cls_kind == ORC_CLASS_ICON) {
                t = mix32(s0 ^ ((y >> 5) * 73856093u) ^ ((x >> 5) * 19349663u));
                if ((t & 15) < 15) {
                    uint32_t k = (t & 15) < 8 ? 0 : (t >> 4) & 7;
                    t = mix32(s0 + 0x1000u + k);
                    cls = 0;
                } else cls = 4;
            } else { t = s0; cls = 3; }
            uint32_t br = (t >> 8) & 255, bg = (t >> 16) & 255, bb = t >> 24;
            uint32_t r, g, b, a = 255;
            switch (cls) {
            case 0: r = br; g = bg; b = bb; break;
            case 1: r = br + (x >> 1); g = bg + (x >> 1) + (y >> 2); b = bb + (y >> 1); break;
            case 2:
                r = br + (x >> 1) + (n & 3) - 2;
                g = bg + (x >> 1) + ((n >> 2) & 3) - 2;
                b = bb + (y >> 1) + ((n >> 4) & 3) - 2;
                break;
            case 3: {
                uint32_t dd = (n >> 8) % 25;
                r = br + (x >> 1) + dd + ((n >> 16) & 7) - 16;
                g = bg + (y >> 1) + dd - 12;
                b = bb + ((x + y) >> 2) + dd + ((n >> 20) & 7) - 16;
                break;
            }
            case 4: {
                uint32_t q = mix32(t + (m & 15));
                r = q & 255; g = (q >> 8) & 255; b = (q >> 16) & 255;
                break;
            }
            case 5: r = n & 255; g = (n >> 8) & 255; b = (n >> 16) & 255; break;
            case 6: r = n & 255; g = (n >> 8) & 255; b = (n >> 16) & 255; a = n >> 24; break;
            default: r = br; g = bg; b = bb; a = 255 - ((x & 63) << 1); break;
            }
            uint8_t *o = dst + (size_t)i * channels;
            o[0] = r; o[1] = g; o[2] = b;
            if (channels == 4) o[3] = a;
        }
}

uint64_t orc_fnv1a64(const void *data, size_t n) {
    const uint8_t *p = data;
    uint64_t hsh = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) hsh = (hsh ^ p[i]) * 0x100000001b3ull;
    return hsh;
}
