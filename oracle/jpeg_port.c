/*
 * oracle/jpeg_port.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of the reference's HOST side of the path, so that the
 * whole JPEG -> BMP pipeline can be checked on machines where /root/reference
 * does not exist (the GPU box):
 *     container scan      reference src/jpeg_scanner.cpp:6-436
 *     BitReader           reference src/headers/jpeg.h:81-122
 *     Huffman decode      reference src/jpeg_scanner.cpp:438-520,707-756
 *     metadata packing    reference src/decoder_host.cpp:125-128,156-179
 *     BMP emit            reference src/bmp_writer.cpp:19-67
 * The device stages live in oracle/dpu_stages.c.
 *
 * It works on memory buffers instead of std::ifstream; the `cur` cursor below
 * reproduces the one ifstream behaviour the scanner depends on: a get() past
 * the end yields -1 (0xFF once stored in a byte) and makes the stream "bad".
 *
 * Pinning: tests/test_oracle.py checks this file against oracle/_ref (the real
 * reference scanner + Huffman decoder + BMP writer compiled in place) on every
 * fixture, field by field and byte by byte, and against the SURVEY section 0.4
 * known-answer hash.
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* zigzag index -> natural index, INCLUDING the reference's entry 48 = 38
 * (reference src/headers/common.h:9-18; the standard value there is 58). */
static uint8_t k_zz[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    38, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

/* NOT the reference: entry 48 as ITU T.81 has it (58).  Only tests of PJD_F_STANDARD_ZIGZAG switch it on (and off
 * again); nothing checked against the reference's fixtures runs with it.  Parity of that mode is unpinned. */
void orc_set_standard_zigzag(int on) { k_zz[48] = on ? 58 : 38; }

typedef struct {
    uint8_t offsets[17];
    uint8_t symbols[162];
    uint32_t codes[162];
    uint8_t set;
} orc_huff;

typedef struct {
    int32_t valid, width, height, ncomp, hsamp, vsamp;
    int32_t mcu_w, mcu_h, mcu_w_real, mcu_h_real, restart_interval, frame_type;
    int32_t n_dpus;
    int64_t ecs_len;
    uint8_t comp_h[3], comp_v[3], comp_qt[3], comp_dc[3], comp_ac[3];
    uint8_t qt_set[4], dc_set[4], ac_set[4];
    uint32_t qt[4][64];
    uint8_t dc_offsets[4][17], dc_symbols[4][162];
    uint8_t ac_offsets[4][17], ac_symbols[4][162];
} orc_info;                       /* same layout as ref_info in ref_driver.cpp */

typedef struct {
    orc_info i;
    uint8_t used_in_frame[3], used_in_scan[3];
    int zero_based;
    int comps_in_scan, ss, se, ah, al;
    uint8_t *ecs;                 /* destuffed, RST-stripped */
    size_t ecs_cap;
    orc_huff dc[4], ac[4];
    char *log; size_t log_cap, log_len;
    char name[512];
} orc_image;

static void say(orc_image *im, const char *fmt, ...)
{
    if (!im->log || im->log_len >= im->log_cap) return;
    int n = snprintf(im->log + im->log_len, im->log_cap - im->log_len, "%s", im->name);
    if (n > 0) im->log_len += (size_t)n;
    if (im->log_len >= im->log_cap) { im->log_len = im->log_cap - 1; return; }
    va_list ap; va_start(ap, fmt);
    n = vsnprintf(im->log + im->log_len, im->log_cap - im->log_len, fmt, ap);
    va_end(ap);
    if (n > 0) im->log_len += (size_t)n;
    if (im->log_len >= im->log_cap) im->log_len = im->log_cap - 1;
}

/* ---- byte cursor with ifstream-like end-of-file behaviour ---------------- */
typedef struct { const uint8_t *p; size_t n, pos; int bad; } cur;

static int cget(cur *c)
{
    if (c->bad || c->pos >= c->n) { c->bad = 1; return -1; }
    return c->p[c->pos++];
}
static uint32_t cget16(cur *c) { int hi = cget(c); int lo = cget(c); return (uint32_t)((hi << 8) + lo); }

/* ---- segment parsers (each returns with im->i.valid possibly cleared) ---- */

/* reference src/jpeg_scanner.cpp:187-285 */
static void seg_sof(cur *c, orc_image *im)
{
    orc_info *h = &im->i;
    if (h->ncomp != 0) { say(im, ": Error - Multiple SOFs detected\n"); h->valid = 0; return; }
    uint32_t len = cget16(c);
    uint8_t prec = (uint8_t)cget(c);
    if (prec != 8) { say(im, ": Error - Invalid precision: %u\n", (unsigned)prec); h->valid = 0; return; }
    h->height = (int32_t)cget16(c);
    h->width = (int32_t)cget16(c);
    if (h->height == 0 || h->width == 0) { say(im, ": Error - Invalid dimensions\n"); h->valid = 0; return; }
    h->mcu_h = (int32_t)(((uint32_t)h->height + 7) / 8);
    h->mcu_w = (int32_t)(((uint32_t)h->width + 7) / 8);
    h->mcu_h_real = h->mcu_h; h->mcu_w_real = h->mcu_w;
    h->ncomp = (uint8_t)cget(c);
    if (h->ncomp == 4) { say(im, ": Error - CMYK color mode not supported\n"); h->valid = 0; return; }
    if (h->ncomp == 0) { say(im, ": Error - Number of color components must not be zero\n"); h->valid = 0; return; }
    for (int k = 0; k < h->ncomp; k++) {
        uint8_t id = (uint8_t)cget(c);
        if (id == 0 && k == 0) im->zero_based = 1;
        if (im->zero_based) id = (uint8_t)(id + 1);
        if (id == 4 || id == 5) { say(im, ": Error - YIQ color mode not supported\n"); h->valid = 0; return; }
        if (id == 0 || id > h->ncomp) { say(im, ": Error - Invalid component ID:%u\n", (unsigned)id); h->valid = 0; return; }
        int ci = id - 1;
        /* NB: with ncomp > 3 the reference indexes past color_components[3];
         * restrict to the range that cannot fault. */
        if (ci > 2) { say(im, ": Error - Invalid component ID:%u\n", (unsigned)id); h->valid = 0; return; }
        if (im->used_in_frame[ci]) { say(im, ": Duplicate color component ID\n"); h->valid = 0; return; }
        im->used_in_frame[ci] = 1;
        uint8_t sf = (uint8_t)cget(c);
        h->comp_h[ci] = sf >> 4; h->comp_v[ci] = sf & 15;
        if (id == 1) {
            if ((h->comp_h[ci] != 1 && h->comp_h[ci] != 2) || (h->comp_v[ci] != 1 && h->comp_v[ci] != 2)) {
                say(im, ": Error - Sampling factors not supported\n"); h->valid = 0; return;
            }
            if (h->comp_h[ci] == 2 && h->mcu_w % 2 == 1) h->mcu_w_real += 1;
            if (h->comp_v[ci] == 2 && h->mcu_h % 2 == 1) h->mcu_h_real += 1;
            h->hsamp = h->comp_h[ci]; h->vsamp = h->comp_v[ci];
        } else if (h->comp_h[ci] != 1 || h->comp_v[ci] != 1) {
            say(im, ": Error - Sampling factors not supported\n"); h->valid = 0; return;
        }
        h->comp_qt[ci] = (uint8_t)cget(c);
        if (h->comp_qt[ci] > 3) { say(im, ": Error - Invalid quantization table ID in frame components\n"); h->valid = 0; return; }
    }
    if (len - 8 - 3 * (uint32_t)h->ncomp != 0) { say(im, ": Error - SOF invalid\n"); h->valid = 0; }
}

/* reference src/jpeg_scanner.cpp:287-321 */
static void seg_dqt(cur *c, orc_image *im)
{
    orc_info *h = &im->i;
    int len = (int)cget16(c) - 2;
    while (len > 0) {
        uint8_t info = (uint8_t)cget(c);
        len -= 1;
        uint8_t id = info & 15;
        if (id > 3) { say(im, ": Error Invalid quantization table ID: %u\n", (unsigned)id); h->valid = 0; return; }
        h->qt_set[id] = 1;
        if ((info >> 4) != 0) {
            for (int k = 0; k < 64; k++) h->qt[id][k_zz[k]] = cget16(c);
            len -= 128;
        } else {
            for (int k = 0; k < 64; k++) h->qt[id][k_zz[k]] = (uint32_t)cget(c);
            len -= 64;
        }
    }
    if (len != 0) { say(im, ": Error - DQT invalid\n"); h->valid = 0; }
}

/* reference src/jpeg_scanner.cpp:140-185 */
static void seg_dht(cur *c, orc_image *im)
{
    orc_info *h = &im->i;
    int len = (int)cget16(c) - 2;
    while (len > 0) {
        uint8_t info = (uint8_t)cget(c);
        uint8_t id = info & 15;
        if (id > 3) { say(im, ": Error - Invalid Huffman table ID: %u\n", (unsigned)id); h->valid = 0; return; }
        int is_ac = (info >> 4) != 0;
        uint8_t *offs = is_ac ? h->ac_offsets[id] : h->dc_offsets[id];
        uint8_t *syms = is_ac ? h->ac_symbols[id] : h->dc_symbols[id];
        if (is_ac) h->ac_set[id] = 1; else h->dc_set[id] = 1;
        offs[0] = 0;
        uint32_t total = 0;
        for (int k = 1; k <= 16; k++) { total += (uint32_t)cget(c); offs[k] = (uint8_t)total; }
        if (total > 162) { say(im, ": : Error - Too many symbols in Huffman table\n"); h->valid = 0; return; }
        for (uint32_t k = 0; k < total; k++) syms[k] = (uint8_t)cget(c);
        len -= 17 + (int)total;
    }
    if (len != 0) { say(im, ": Error - DHT invalid\n"); h->valid = 0; }
}

/* reference src/jpeg_scanner.cpp:6-138 (baseline and progressive checks) */
static void seg_sos(cur *c, orc_image *im)
{
    orc_info *h = &im->i;
    if (h->ncomp == 0) { say(im, ": Error - SOS detected before SOF\n"); h->valid = 0; return; }
    uint32_t len = cget16(c);
    for (int k = 0; k < h->ncomp && k < 3; k++) im->used_in_scan[k] = 0;
    im->comps_in_scan = (uint8_t)cget(c);
    if (im->comps_in_scan == 0) { say(im, ": Error - Scan must include at least 1 component\n"); h->valid = 0; return; }
    for (int k = 0; k < im->comps_in_scan; k++) {
        uint8_t id = (uint8_t)cget(c);
        if (im->zero_based) id = (uint8_t)(id + 1);
        if (id == 0 || id > h->ncomp) { say(im, ": Error - Invalid color component ID: %u\n", (unsigned)id); h->valid = 0; return; }
        int ci = id - 1;
        if (!im->used_in_frame[ci]) { say(im, ": Error - Invalid color component ID: %u\n", (unsigned)id); h->valid = 0; return; }
        if (im->used_in_scan[ci]) { say(im, ": Error - Duplicate color component ID\n"); h->valid = 0; return; }
        im->used_in_scan[ci] = 1;
        uint8_t ids = (uint8_t)cget(c);
        h->comp_dc[ci] = ids >> 4; h->comp_ac[ci] = ids & 15;
        if (h->comp_dc[ci] > 3) { say(im, ": Error - Invalid Huffman DC table ID: %u\n", (unsigned)h->comp_dc[ci]); h->valid = 0; return; }
        if (h->comp_ac[ci] > 3) { say(im, ": Error - Invalid Huffman AC table ID: %u\n", (unsigned)h->comp_ac[ci]); h->valid = 0; return; }
    }
    im->ss = (uint8_t)cget(c); im->se = (uint8_t)cget(c);
    uint8_t sa = (uint8_t)cget(c);
    im->ah = sa >> 4; im->al = sa & 15;
    if (h->frame_type == 0xC0) {
        if (im->ss != 0 || im->se != 63) { say(im, ": Error - Invalid spectral selection\n"); h->valid = 0; return; }
        if (im->ah != 0 || im->al != 0) { say(im, ": Error - Invalid successive approximation\n"); h->valid = 0; return; }
    } else if (h->frame_type == 0xC2) {
        if (im->ss > im->se) { say(im, ": Error - Invalid spectral selection (start greater than end)\n"); h->valid = 0; return; }
        if (im->se > 63) { say(im, ": Error - Invalid spectral selection (end greater than 63)\n"); h->valid = 0; return; }
        if (im->ss == 0 && im->se != 0) { say(im, ": Error - Invalid spectral selection (contains DC and AC)\n"); h->valid = 0; return; }
        if (im->ss != 0 && im->comps_in_scan != 1) { say(im, ": Error - Invalid spectral selection (AC scan contains multiple components)\n"); h->valid = 0; return; }
        if (im->ah != 0 && im->al != im->ah - 1) { say(im, ": Error - Invalid succesive approximation\n"); h->valid = 0; return; }
    }
    for (int k = 0; k < h->ncomp && k < 3; k++) {
        if (!im->used_in_scan[k]) continue;
        if (!h->qt_set[h->comp_qt[k]]) { say(im, ": Error - Color component using uninitialized quantization table\n"); h->valid = 0; return; }
        if (im->ss == 0 && !h->dc_set[h->comp_dc[k]]) { say(im, ": Error - Color component using uninitialized Huffman DC table\n"); h->valid = 0; return; }
        if (im->se > 0 && !h->ac_set[h->comp_ac[k]]) { say(im, ": Error - Color component using uninitialized Huffman AC table\n"); h->valid = 0; return; }
    }
    if (len - 6 - 2 * (uint32_t)im->comps_in_scan != 0) { say(im, ": Error - SOS invalid\n"); h->valid = 0; }
}

static void seg_skip(cur *c)
{
    uint32_t len = cget16(c);
    uint32_t cnt = len - 2;             /* unsigned, as in the reference: len < 2 wraps */
    for (uint32_t k = 0; k < cnt; k++)
        if (cget(c) < 0) break;         /* the reference keeps calling a failed get(); same end state */
}

/* reference src/jpeg_scanner.cpp:345-436 */
static void scan_file(const uint8_t *file, size_t n, orc_image *im)
{
    orc_info *h = &im->i;
    cur c = { file, n, 0, 0 };
    h->valid = 1; h->hsamp = 1; h->vsamp = 1;
    for (int k = 0; k < 3; k++) { h->comp_h[k] = 1; h->comp_v[k] = 1; }

    uint8_t last = (uint8_t)cget(&c), now = (uint8_t)cget(&c);
    if (last != 0xFF || now != 0xD8) { h->valid = 0; return; }

    last = (uint8_t)cget(&c); now = (uint8_t)cget(&c);
    while (h->valid) {
        if (c.bad || last != 0xFF) {
            if (c.bad) say(im, ": Error - File ended prematurely\n");
            if (last != 0xFF) say(im, ": Error - Expected a marker\n");
            h->valid = 0;
            return;
        }
        if (now == 0xC0 || now == 0xC2) { h->frame_type = now; seg_sof(&c, im); }
        else if (now == 0xDB) seg_dqt(&c, im);
        else if (now == 0xC4) seg_dht(&c, im);
        else if (now == 0xDA) { seg_sos(&c, im); break; }
        else if (now == 0xDD) {
            uint32_t len = cget16(&c);
            h->restart_interval = (int32_t)cget16(&c);
            if (len - 4 != 0) { say(im, ": Error - DRI invalid\n"); h->valid = 0; }
        }
        else if (now >= 0xE0 && now <= 0xEF) seg_skip(&c);
        else if (now == 0xFE) seg_skip(&c);
        else if ((now >= 0xF0 && now <= 0xFD) || now == 0xDC || now == 0xDE || now == 0xDF) seg_skip(&c);
        else if (now == 0x01) { /* TEM: no payload */ }
        else if (now == 0xFF) { now = (uint8_t)cget(&c); continue; }
        else say(im, ": Error - Unknown marker: 0x%x\n", (unsigned)now);

        last = (uint8_t)cget(&c); now = (uint8_t)cget(&c);
    }
    if (!h->valid) return;

    /* entropy-coded segment: drop FF00 stuffing, drop RSTn, skip FF fill, stop at EOI */
    size_t out = 0;
    now = (uint8_t)cget(&c);
    for (;;) {
        if (c.bad) { say(im, ": Error - File ended prematurely\n"); h->valid = 0; return; }
        last = now;
        now = (uint8_t)cget(&c);
        if (last == 0xFF) {
            if (now == 0xD9) break;
            else if (now == 0x00) { im->ecs[out++] = last; now = (uint8_t)cget(&c); }
            else if (now >= 0xD0 && now <= 0xD7) now = (uint8_t)cget(&c);
            else if (now == 0xFF) continue;
            else { say(im, ": Error - Invalid marker during compressed data scan: 0x%x\n", (unsigned)now); h->valid = 0; return; }
        } else im->ecs[out++] = last;
    }
    h->ecs_len = (int64_t)out;
}

/* ---- BitReader (reference src/headers/jpeg.h:81-122) --------------------- */
typedef struct { const uint8_t *d; size_t n, byte; unsigned bit; } bitrd;

static int rd_bit(bitrd *b)
{
    if (b->byte >= b->n) return -1;
    int v = (b->d[b->byte] >> (7 - b->bit)) & 1;
    if (++b->bit == 8) { b->bit = 0; b->byte++; }
    return v;
}
static int rd_bits(bitrd *b, unsigned len)
{
    int v = 0;
    for (unsigned k = 0; k < len; k++) {
        int bit = rd_bit(b);
        if (bit == -1) return -1;
        v = (v << 1) | bit;
    }
    return v;
}
static void rd_align(bitrd *b)
{
    if (b->byte >= b->n) return;
    if (b->bit != 0) { b->bit = 0; b->byte++; }
}

/* reference src/jpeg_scanner.cpp:438-448 */
static void make_codes(orc_huff *t)
{
    uint32_t code = 0;
    for (int len = 0; len < 16; len++) {
        for (unsigned k = t->offsets[len]; k < t->offsets[len + 1]; k++) t->codes[k] = code++;
        code <<= 1;
    }
}

/* reference src/jpeg_scanner.cpp:450-465; 0xFF = failure */
static uint8_t next_symbol(bitrd *b, const orc_huff *t)
{
    uint32_t word = 0;
    for (int len = 0; len < 16; len++) {
        int bit = rd_bit(b);
        if (bit == -1) return 0xFF;
        word = (word << 1) | (uint32_t)bit;
        for (unsigned k = t->offsets[len]; k < t->offsets[len + 1]; k++)
            if (word == t->codes[k]) return t->symbols[k];
    }
    return 0xFF;
}

/* Error classes of the baseline data-unit decoder; values are shared with the
 * product's status codes (include/pjd.h). */
enum { E_OK = 0, E_DC_SYM = 1, E_DC_LEN = 2, E_DC_BITS = 3, E_AC_SYM = 4, E_AC_RUN = 5, E_AC_LEN = 6, E_AC_BITS = 7 };

/* reference src/jpeg_scanner.cpp:467-520 (frame_type == SOF0 branch) */
static int decode_unit(orc_image *im, bitrd *b, int16_t *unit, int *pred, const orc_huff *dt, const orc_huff *at)
{
    uint8_t s = next_symbol(b, dt);
    if (s == 0xFF) { say(im, ": Error - Invalid DC value (%u)\n", (unsigned)s); return E_DC_SYM; }
    if (s > 11) { say(im, ": Error - DC coefficient length greater than 11\n"); return E_DC_LEN; }
    int v = rd_bits(b, s);
    if (v == -1) { say(im, ": Error - Invalid DC value\n"); return E_DC_BITS; }
    if (s != 0 && v < (1 << (s - 1))) v -= (1 << s) - 1;
    unit[0] = (int16_t)(v + *pred);
    *pred = unit[0];

    for (unsigned k = 1; k < 64; k++) {
        uint8_t sym = next_symbol(b, at);
        if (sym == 0xFF) { say(im, ": Error - Invalid AC value\n"); return E_AC_SYM; }
        if (sym == 0x00) return E_OK;
        unsigned run = sym >> 4, len = sym & 15;
        if (k + run >= 64) { say(im, ": Error - Zero run-length exceeded block component\n"); return E_AC_RUN; }
        k += run;
        if (len > 10) { say(im, ": Error - AC coefficient length greater than 10\n"); return E_AC_LEN; }
        v = rd_bits(b, len);
        if (v == -1) { say(im, ": Error - Invalid AC value\n"); return E_AC_BITS; }
        /* len == 0 (e.g. ZRL 0xF0): the reference evaluates 1 << -1, observed
         * result "no sign extension", so a literal 0 is stored at slot k. */
        if (len != 0 && v < (1 << (len - 1))) v -= (1 << len) - 1;
        unit[k_zz[k]] = (int16_t)v;
    }
    return E_OK;
}

/* reference src/jpeg_scanner.cpp:707-756; `mcus` = n_dpus x 19200, zeroed by caller.
 * Returns the error class of the first failure (0 = none). */
static int huffman_all(orc_image *im, int16_t *mcus)
{
    orc_info *h = &im->i;
    for (int t = 0; t < 4; t++) {
        memcpy(im->dc[t].offsets, h->dc_offsets[t], 17); memcpy(im->dc[t].symbols, h->dc_symbols[t], 162);
        memcpy(im->ac[t].offsets, h->ac_offsets[t], 17); memcpy(im->ac[t].symbols, h->ac_symbols[t], 162);
        if (h->dc_set[t]) make_codes(&im->dc[t]);
        if (h->ac_set[t]) make_codes(&im->ac[t]);
    }
    bitrd b = { im->ecs, (size_t)h->ecs_len, 0, 0 };
    int pred[3] = { 0, 0, 0 };
    const uint32_t W = (uint32_t)h->mcu_w_real;
    for (uint32_t y = 0; y < (uint32_t)h->mcu_h; y += (uint32_t)h->vsamp)
        for (uint32_t x = 0; x < (uint32_t)h->mcu_w; x += (uint32_t)h->hsamp) {
            if (h->restart_interval != 0 && (y * W + x) % (uint32_t)h->restart_interval == 0) {
                pred[0] = pred[1] = pred[2] = 0;
                rd_align(&b);
            }
            for (int j = 0; j < h->ncomp; j++)
                for (uint32_t v = 0; v < h->comp_v[j]; v++)
                    for (uint32_t hh = 0; hh < h->comp_h[j]; hh++) {
                        int m = (int)((y + v) * W + (x + hh));
                        int blk = (m / (int)(W * 2)) * (int)((W + 1) / 2) + (m % (int)W) / 2;
                        int pos = ((m / (int)W) % 2) * 2 + (m % (int)W) % 2;
                        int dpu = blk / 25;
                        blk %= 25;
                        int rc = decode_unit(im, &b, mcus + (size_t)dpu * 19200 + blk * 768 + j * 256 + pos * 64,
                                             &pred[j], &im->dc[h->comp_dc[j]], &im->ac[h->comp_ac[j]]);
                        if (rc) return rc;
                    }
        }
    return E_OK;
}

/* ---- public entry points -------------------------------------------------- */

void orc_dpu_exec(const uint32_t *metadata, int16_t *mcus);

orc_image *orc_open(const uint8_t *file, int64_t n, const char *name, char *log, int64_t log_cap)
{
    orc_image *im = (orc_image *)calloc(1, sizeof *im);
    im->ecs = (uint8_t *)malloc((size_t)n + 16);
    im->ecs_cap = (size_t)n + 16;
    im->log = log; im->log_cap = (size_t)log_cap;
    snprintf(im->name, sizeof im->name, "%s", name ? name : "");
    if (log && log_cap) log[0] = 0;
    scan_file(file, (size_t)n, im);
    if (im->i.valid) {
        int pw = (im->i.mcu_w_real + 1) / 2 * 2, ph = (im->i.mcu_h_real + 1) / 2 * 2;
        im->i.n_dpus = (pw * ph + 99) / 100;
    }
    return im;
}
void orc_close(orc_image *im) { if (im) { free(im->ecs); free(im); } }
void orc_get_info(const orc_image *im, orc_info *o) { *o = im->i; }
int64_t orc_get_ecs(const orc_image *im, uint8_t *dst, int64_t cap)
{
    if (dst && cap >= im->i.ecs_len && im->i.ecs_len) memcpy(dst, im->ecs, (size_t)im->i.ecs_len);
    return im->i.ecs_len;
}

/* reference src/decoder_host.cpp:156-179 (fresh, zero-initialised vector) */
void orc_get_metadata(const orc_image *im, uint32_t *m)
{
    const orc_info *h = &im->i;
    memset(m, 0, 276 * 4);
    m[0] = (uint32_t)h->mcu_h; m[1] = (uint32_t)h->mcu_w; m[2] = (uint32_t)h->mcu_h_real; m[3] = (uint32_t)h->mcu_w_real;
    m[4] = (uint32_t)h->ncomp; m[5] = (uint32_t)h->vsamp; m[6] = (uint32_t)h->hsamp;
    for (int j = 0; j < h->ncomp; j++) {
        m[7 + j] = h->comp_qt[j];
        m[7 + h->ncomp + j] = h->comp_h[j];
        m[7 + 2 * h->ncomp + j] = h->comp_v[j];
    }
    m[17] = (uint32_t)h->height; m[18] = (uint32_t)h->width; m[19] = 100;
    for (int t = 0; t < 4; t++) {
        if (!h->qt_set[t]) break;
        for (int k = 0; k < 64; k++) m[20 + 64 * t + k] = h->qt[t][k];
    }
}

int orc_huffman(orc_image *im, int16_t *mcus) { return huffman_all(im, mcus); }

/* reference src/bmp_writer.cpp:19-67 into memory; returns the byte count
 * (pass out == NULL to query). */
int64_t orc_bmp(const uint32_t *m, const int16_t *mcus, uint8_t *out)
{
    const int w = (int)m[18], hgt = (int)m[17];
    const uint32_t pad = (uint32_t)(w % 4);
    const uint32_t size = 14 + 12 + (uint32_t)hgt * (uint32_t)w * 3 + pad * (uint32_t)hgt;
    if (!out) return (int64_t)size;
    uint8_t *p = out;
    *p++ = 'B'; *p++ = 'M';
    for (int k = 0; k < 4; k++) *p++ = (uint8_t)(size >> (8 * k));
    for (int k = 0; k < 4; k++) *p++ = 0;
    *p++ = 0x1A; *p++ = 0; *p++ = 0; *p++ = 0;
    *p++ = 12; *p++ = 0; *p++ = 0; *p++ = 0;
    *p++ = (uint8_t)w; *p++ = (uint8_t)(w >> 8);
    *p++ = (uint8_t)hgt; *p++ = (uint8_t)(hgt >> 8);
    *p++ = 1; *p++ = 0; *p++ = 24; *p++ = 0;
    const uint32_t W = m[3];
    for (int y = hgt - 1; y >= 0; y--) {
        for (int x = 0; x < w; x++) {
            uint32_t mi = (uint32_t)(y / 8) * W + (uint32_t)(x / 8);
            uint32_t blk = (mi / (W * 2)) * ((W + 1) / 2) + (mi % W) / 2;
            uint32_t pos = ((mi / W) % 2) * 2 + (mi % W) % 2;
            const int16_t *b = mcus + (size_t)(blk / 25) * 19200 + (blk % 25) * 768 + pos * 64 + (y % 8) * 8 + (x % 8);
            *p++ = (uint8_t)b[512]; *p++ = (uint8_t)b[256]; *p++ = (uint8_t)b[0];
        }
        for (uint32_t k = 0; k < pad; k++) *p++ = 0;
    }
    return (int64_t)(p - out);
}

/* Whole path, file bytes in -> BMP bytes out (malloc'd, caller frees with
 * orc_free).  Returns 0 when a BMP was produced (even after a Huffman error,
 * as the reference does), 1 when the scanner rejected the file.
 * *huff_rc receives the Huffman error class. */
int orc_decode(const uint8_t *file, int64_t n, const char *name, uint8_t **bmp, int64_t *bmp_len,
               int *huff_rc, char *log, int64_t log_cap)
{
    orc_image *im = orc_open(file, n, name, log, log_cap);
    *bmp = NULL; *bmp_len = 0; if (huff_rc) *huff_rc = 0;
    if (!im->i.valid) { say(im, ": Error - Invalid JPEG\n"); orc_close(im); return 1; }
    uint32_t meta[276];
    orc_get_metadata(im, meta);
    int16_t *mcus = (int16_t *)calloc((size_t)im->i.n_dpus * 19200, 2);
    int rc = huffman_all(im, mcus);
    if (huff_rc) *huff_rc = rc;
    for (int d = 0; d < im->i.n_dpus; d++) orc_dpu_exec(meta, mcus + (size_t)d * 19200);
    *bmp_len = orc_bmp(meta, mcus, NULL);
    *bmp = (uint8_t *)malloc((size_t)*bmp_len);
    orc_bmp(meta, mcus, *bmp);
    free(mcus);
    orc_close(im);
    return 0;
}
void orc_free(void *p) { free(p); }

/* Raster RGB8 (top-down, tight rows) of the decoded picture -- what the
 * product's C-ABI returns -- taken from the reference's planar int16 layout. */
void orc_rgb_from_mcus(const uint32_t *m, const int16_t *mcus, uint8_t *rgb)
{
    const int w = (int)m[18], hgt = (int)m[17];
    const uint32_t W = m[3];
    for (int y = 0; y < hgt; y++)
        for (int x = 0; x < w; x++) {
            uint32_t mi = (uint32_t)(y / 8) * W + (uint32_t)(x / 8);
            uint32_t blk = (mi / (W * 2)) * ((W + 1) / 2) + (mi % W) / 2;
            uint32_t pos = ((mi / W) % 2) * 2 + (mi % W) % 2;
            const int16_t *b = mcus + (size_t)(blk / 25) * 19200 + (blk % 25) * 768 + pos * 64 + (y % 8) * 8 + (x % 8);
            uint8_t *o = rgb + ((size_t)y * w + x) * 3;
            o[0] = (uint8_t)b[0]; o[1] = (uint8_t)b[256]; o[2] = (uint8_t)b[512];
        }
}
