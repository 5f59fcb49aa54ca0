/*
 * tools/jpeg_synth.c -- synthetic workload generator (bench / test tooling, not product).
 *
 * There is no dataset on the GPU box, so the benchmark's JPEGs are made here:
 *   synth_picture()  a seeded natural-looking RGB picture (fractal value noise, colour
 *                    gradients, random soft-edged shapes, a little sensor noise)
 *   synth_encode()   a plain baseline (SOF0) JPEG encoder: 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 /
 *                    grey, libjpeg-style quality scaling of the Annex K quantisation tables,
 *                    Annex K Huffman tables, optional restart interval, byte stuffing.
 * Decode parity never depends on this encoder: oracle and GPU decode the same bytes.
 *
 *   gcc -O2 -fPIC -shared -o tools/libjpegsynth.so tools/jpeg_synth.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- picture ----------------------------------------------------------------------------- */
static uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static float lattice(uint64_t seed, int x, int y)
{
    return (float)(mix64(seed ^ ((uint64_t)(uint32_t)x << 32) ^ (uint32_t)y) >> 40) / 16777216.0f;
}
static float vnoise(uint64_t seed, float x, float y)
{
    int xi = (int)floorf(x), yi = (int)floorf(y);
    float fx = x - xi, fy = y - yi;
    fx = fx * fx * (3 - 2 * fx); fy = fy * fy * (3 - 2 * fy);
    float a = lattice(seed, xi, yi), b = lattice(seed, xi + 1, yi);
    float c = lattice(seed, xi, yi + 1), d = lattice(seed, xi + 1, yi + 1);
    return (a + (b - a) * fx) + ((c + (d - c) * fx) - (a + (b - a) * fx)) * fy;
}

/* Per (channel, octave) the lattice is tabulated once; a pixel then costs 4 reads per octave. */
typedef struct { float *g; int gx, gy; float inv; } octave;

/* detail: 1.0 = the round-1 pictures (~0.32 B/px at q75-95 4:2:0); larger values add fine luminance texture and grain,
 * ~2.2 gives the density of the bundled ImageNet sample's class (~0.58 B/px) */
void synth_picture2(uint8_t *rgb, int w, int h, uint64_t seed, float detail)
{
    enum { NOCT = 6 };
    uint64_t s = mix64(seed);
    const int nshape = 6 + (int)(mix64(s + 1) % 10);
    float sx[16], sy[16], sr[16], sc[16][3];
    for (int k = 0; k < nshape; k++) {
        sx[k] = (float)(mix64(s + 10 + k) % 1000) / 1000.0f * w;
        sy[k] = (float)(mix64(s + 40 + k) % 1000) / 1000.0f * h;
        sr[k] = (0.05f + (float)(mix64(s + 70 + k) % 1000) / 4000.0f) * (w < h ? w : h);
        for (int c = 0; c < 3; c++) sc[k][c] = (float)(mix64(s + 100 + 3 * k + c) % 256);
    }
    const float base = 40.0f + (float)(mix64(s + 2) % 64);
    const float persistence = 0.55f + (float)(mix64(s + 3) % 100) / 1000.0f;
    octave oc[3][NOCT];
    for (int c = 0; c < 3; c++) {
        float scale = base;
        for (int o = 0; o < NOCT; o++, scale *= 0.5f) {
            octave *q = &oc[c][o];
            q->inv = 1.0f / scale;
            q->gx = (int)(w * q->inv) + 3; q->gy = (int)(h * q->inv) + 3;
            q->g = (float *)malloc(sizeof(float) * (size_t)q->gx * q->gy);
            for (int y = 0; y < q->gy; y++)
                for (int x = 0; x < q->gx; x++) q->g[(size_t)y * q->gx + x] = lattice(s + 1000 * c + o, x, y);
        }
    }
    float tot = 0, amp0 = 1;
    for (int o = 0; o < NOCT; o++) { tot += amp0; amp0 *= persistence; }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float v[3];
            for (int c = 0; c < 3; c++) {
                float f = 0, amp = 1;
                for (int o = 0; o < NOCT; o++) {
                    const octave *q = &oc[c][o];
                    float fx = x * q->inv, fy = y * q->inv;
                    int xi = (int)fx, yi = (int)fy;
                    fx -= xi; fy -= yi;
                    fx = fx * fx * (3 - 2 * fx); fy = fy * fy * (3 - 2 * fy);
                    const float *g = q->g + (size_t)yi * q->gx + xi;
                    float top = g[0] + (g[1] - g[0]) * fx, bot = g[q->gx] + (g[q->gx + 1] - g[q->gx]) * fx;
                    f += amp * (top + (bot - top) * fy);
                    amp *= persistence;
                }
                v[c] = 255.0f * f / tot;
            }
            /* correlate the channels like a natural picture: mostly luminance detail */
            const float l = (v[0] + v[1] + v[2]) * (1.0f / 3.0f);
            for (int c = 0; c < 3; c++) v[c] = l + 0.45f * (v[c] - l);
            for (int k = 0; k < nshape; k++) {
                float dx = x - sx[k], dy = y - sy[k];
                float d = sqrtf(dx * dx + dy * dy) - sr[k];
                float a = d < -1 ? 1.0f : (d > 1 ? 0.0f : 0.5f - 0.5f * d);
                if (a > 0) for (int c = 0; c < 3; c++) v[c] = v[c] * (1 - 0.8f * a) + sc[k][c] * 0.8f * a;
            }
            uint64_t n = mix64(s ^ ((uint64_t)y * 65537u + x));
            const float ln = ((float)(n & 0xFFFF) / 65536.0f - 0.5f) * 26.0f * detail;      /* luminance grain */
            for (int c = 0; c < 3; c++) {
                float t = (v[c] - 128.0f) * 1.6f + 128.0f + ln + ((float)((n >> (16 * (c + 1))) & 0xFFFF) / 65536.0f - 0.5f) * 4.0f;
                rgb[((size_t)y * w + x) * 3 + c] = (uint8_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
            }
        }
    for (int c = 0; c < 3; c++) for (int o = 0; o < NOCT; o++) free(oc[c][o].g);
}

void synth_picture(uint8_t *rgb, int w, int h, uint64_t seed) { synth_picture2(rgb, w, h, seed, 1.0f); }

/* ---- encoder ------------------------------------------------------------------------------ */
static const uint8_t zz[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };   /* the STANDARD order */
static const uint8_t q_luma[64] = {
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
static const uint8_t q_chroma[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };
static const uint8_t bits_dc_l[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t bits_dc_c[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t val_dc[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t bits_ac_l[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
static const uint8_t val_ac_l[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa };
static const uint8_t bits_ac_c[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t val_ac_c[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa };

typedef struct { uint16_t code[256]; uint8_t len[256]; } hcode;

static void build_codes(const uint8_t *bits, const uint8_t *vals, hcode *h)
{
    memset(h, 0, sizeof *h);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, k++) { h->code[vals[k]] = (uint16_t)code++; h->len[vals[k]] = (uint8_t)l; }
        code <<= 1;
    }
}

typedef struct { uint8_t *p; size_t cap, n; uint32_t acc; int nb; int overflow; } bitw;

static void put_byte(bitw *b, uint8_t v) { if (b->n < b->cap) b->p[b->n++] = v; else b->overflow = 1; }
static void put_bits(bitw *b, uint32_t v, int n)
{
    b->acc = (b->acc << n) | (v & ((1u << n) - 1));
    b->nb += n;
    while (b->nb >= 8) {
        uint8_t o = (uint8_t)(b->acc >> (b->nb - 8));
        put_byte(b, o);
        if (o == 0xFF) put_byte(b, 0);
        b->nb -= 8;
    }
}
static void flush_bits(bitw *b) { if (b->nb) put_bits(b, 0x7F, 8 - b->nb); b->acc = 0; b->nb = 0; }
static void put16(bitw *b, int v) { put_byte(b, (uint8_t)(v >> 8)); put_byte(b, (uint8_t)v); }

static float ctab[8][8];
static int ctab_ready = 0;
static void fdct(const float *in, float *out)
{
    if (!ctab_ready) {
        for (int u = 0; u < 8; u++)
            for (int x = 0; x < 8; x++)
                ctab[u][x] = (float)((u == 0 ? sqrt(0.125) : 0.5) * cos((2 * x + 1) * u * M_PI / 16.0));
        ctab_ready = 1;
    }
    float tmp[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int x = 0; x < 8; x++) s += in[y * 8 + x] * ctab[u][x];
            tmp[y * 8 + u] = s;
        }
    for (int u = 0; u < 8; u++)
        for (int v = 0; v < 8; v++) {
            float s = 0;
            for (int y = 0; y < 8; y++) s += tmp[y * 8 + u] * ctab[v][y];
            out[v * 8 + u] = s;
        }
}

static int bitsize(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }

/* Optimal (length-limited to 16) Huffman code lengths from symbol frequencies: ITU-T T.81 Annex K.2, as every
 * optimising encoder does it.  freq[256] is the reserved all-ones code point. */
static void gen_optimal_table(long *freq, uint8_t *bits /*16*/, uint8_t *vals /*<=256*/, int *nvals)
{
    int codesize[257], others[257];
    uint8_t nb[33];
    memset(codesize, 0, sizeof codesize);
    memset(nb, 0, sizeof nb);
    for (int i = 0; i < 257; i++) others[i] = -1;
    freq[256] = 1;
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; i++) if (codesize[i]) nb[codesize[i] > 32 ? 32 : codesize[i]]++;
    for (int i = 32; i > 16; i--)
        while (nb[i] > 0) {
            int j = i - 2;
            while (nb[j] == 0) j--;
            nb[i] -= 2; nb[i - 1]++;
            nb[j + 1] += 2; nb[j]--;
        }
    int i = 16;
    while (nb[i] == 0) i--;
    nb[i]--;                                           /* drop the reserved code point */
    for (int k = 0; k < 16; k++) bits[k] = nb[k + 1];
    int p = 0;
    for (int l = 1; l <= 32; l++)
        for (int k = 0; k <= 255; k++) if (codesize[k] == l) vals[p++] = (uint8_t)k;
    *nvals = p;
}

/* stats != NULL: count the symbols (stats[0] = DC table, stats[1] = AC table, 257 counters each) instead of writing them */
static void encode_block(bitw *b, const float *px, const uint8_t *q, int *pred, const hcode *dc, const hcode *ac, long (*stats)[257])
{
    float f[64];
    int c[64];
    fdct(px, f);
    for (int k = 0; k < 64; k++) {
        float v = f[zz[k]] / q[zz[k]];
        int iv = (int)(v < 0 ? v - 0.5f : v + 0.5f);
        if (iv > 1023) iv = 1023;
        if (iv < -1023) iv = -1023;
        c[k] = iv;
    }
    int diff = c[0] - *pred;
    *pred = c[0];
    int s = bitsize(diff);
    if (stats) stats[0][s]++;
    else {
        put_bits(b, dc->code[s], dc->len[s]);
        if (s) put_bits(b, (uint32_t)(diff < 0 ? diff - 1 : diff), s);
    }
    int run = 0;
    for (int k = 1; k < 64; k++) {
        if (c[k] == 0) { run++; continue; }
        while (run > 15) { if (stats) stats[1][0xF0]++; else put_bits(b, ac->code[0xF0], ac->len[0xF0]); run -= 16; }
        s = bitsize(c[k]);
        int sym = (run << 4) | s;
        if (stats) stats[1][sym]++;
        else {
            put_bits(b, ac->code[sym], ac->len[sym]);
            put_bits(b, (uint32_t)(c[k] < 0 ? c[k] - 1 : c[k]), s);
        }
        run = 0;
    }
    if (run) { if (stats) stats[1][0]++; else put_bits(b, ac->code[0], ac->len[0]); }
}

/* subsampling: 0 = 4:4:4, 1 = 4:2:2 (h2v1), 2 = 4:2:0, 3 = 4:4:0 (h1v2), 4 = grey.
 * Returns the number of bytes written, or -1 if `cap` was too small. */
static long scan_pass(bitw *pb, const uint8_t *rgb, int w, int h, int H, int V, int grey, int restart_interval,
                      const uint8_t *ql, const uint8_t *qc, const hcode *hdl, const hcode *hal, const hcode *hdc, const hcode *hac,
                      long (*st_l)[257], long (*st_c)[257]);

/* optimize != 0: two passes, Huffman tables fitted to this picture (4 distinct tables, as the bundled ImageNet sample has) */
long synth_encode2(const uint8_t *rgb, int w, int h, int quality, int subsampling, int restart_interval, int optimize,
                   uint8_t *out, long cap)
{
    const int grey = subsampling == 4;
    const int H = (subsampling == 1 || subsampling == 2) ? 2 : 1;
    const int V = (subsampling == 2 || subsampling == 3) ? 2 : 1;
    const int ncomp = grey ? 1 : 3;
    uint8_t ql[64], qc[64];
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int k = 0; k < 64; k++) {
        int a = (q_luma[k] * scale + 50) / 100, c = (q_chroma[k] * scale + 50) / 100;
        ql[k] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
        qc[k] = (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c));
    }
    uint8_t tb[4][16], tv[4][256];                      /* 0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma */
    memcpy(tb[0], bits_dc_l, 16); memcpy(tv[0], val_dc, 12);
    memcpy(tb[1], bits_ac_l, 16); memcpy(tv[1], val_ac_l, 162);
    memcpy(tb[2], bits_dc_c, 16); memcpy(tv[2], val_dc, 12);
    memcpy(tb[3], bits_ac_c, 16); memcpy(tv[3], val_ac_c, 162);
    if (optimize) {
        long st_l[2][257], st_c[2][257];
        memset(st_l, 0, sizeof st_l); memset(st_c, 0, sizeof st_c);
        scan_pass(NULL, rgb, w, h, H, V, grey, restart_interval, ql, qc, NULL, NULL, NULL, NULL, st_l, st_c);
        int nv;
        gen_optimal_table(st_l[0], tb[0], tv[0], &nv); gen_optimal_table(st_l[1], tb[1], tv[1], &nv);
        if (!grey) { gen_optimal_table(st_c[0], tb[2], tv[2], &nv); gen_optimal_table(st_c[1], tb[3], tv[3], &nv); }
    }
    hcode hdl, hdc, hal, hac;
    build_codes(tb[0], tv[0], &hdl); build_codes(tb[2], tv[2], &hdc);
    build_codes(tb[1], tv[1], &hal); build_codes(tb[3], tv[3], &hac);

    bitw b = { out, (size_t)cap, 0, 0, 0, 0 };
    put16(&b, 0xFFD8);
    put16(&b, 0xFFE0); put16(&b, 16);
    { const uint8_t j[14] = { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 }; for (int k = 0; k < 14; k++) put_byte(&b, j[k]); }
    put16(&b, 0xFFDB); put16(&b, grey ? 67 : 132);
    put_byte(&b, 0); for (int k = 0; k < 64; k++) put_byte(&b, ql[zz[k]]);
    if (!grey) { put_byte(&b, 1); for (int k = 0; k < 64; k++) put_byte(&b, qc[zz[k]]); }
    put16(&b, 0xFFC0); put16(&b, 8 + 3 * ncomp); put_byte(&b, 8); put16(&b, h); put16(&b, w); put_byte(&b, (uint8_t)ncomp);
    put_byte(&b, 1); put_byte(&b, (uint8_t)((H << 4) | V)); put_byte(&b, 0);
    if (!grey) { put_byte(&b, 2); put_byte(&b, 0x11); put_byte(&b, 1); put_byte(&b, 3); put_byte(&b, 0x11); put_byte(&b, 1); }
    for (int t = 0; t < (grey ? 2 : 4); t++) {
        const uint8_t *bits = tb[t], *vals = tv[t];
        int n = 0;
        for (int k = 0; k < 16; k++) n += bits[k];
        put16(&b, 0xFFC4); put16(&b, 19 + n);
        put_byte(&b, (uint8_t)(((t & 1) << 4) | (t >> 1)));
        for (int k = 0; k < 16; k++) put_byte(&b, bits[k]);
        for (int k = 0; k < n; k++) put_byte(&b, vals[k]);
    }
    if (restart_interval > 0) { put16(&b, 0xFFDD); put16(&b, 4); put16(&b, restart_interval); }
    put16(&b, 0xFFDA); put16(&b, 6 + 2 * ncomp); put_byte(&b, (uint8_t)ncomp);
    put_byte(&b, 1); put_byte(&b, 0x00);
    if (!grey) { put_byte(&b, 2); put_byte(&b, 0x11); put_byte(&b, 3); put_byte(&b, 0x11); }
    put_byte(&b, 0); put_byte(&b, 63); put_byte(&b, 0);

    scan_pass(&b, rgb, w, h, H, V, grey, restart_interval, ql, qc, &hdl, &hal, &hdc, &hac, NULL, NULL);
    flush_bits(&b);
    put16(&b, 0xFFD9);
    return b.overflow ? -1 : (long)b.n;
}

static long scan_pass(bitw *pb, const uint8_t *rgb, int w, int h, int H, int V, int grey, int restart_interval,
                      const uint8_t *ql, const uint8_t *qc, const hcode *hdl, const hcode *hal, const hcode *hdc, const hcode *hac,
                      long (*st_l)[257], long (*st_c)[257])
{
    const int mw = 8 * H, mh = 8 * V;
    const int mcux = (w + mw - 1) / mw, mcuy = (h + mh - 1) / mh;
    int pred[3] = { 0, 0, 0 }, count = 0, rst = 0;
    float Y[16 * 16], Cb[16 * 16], Cr[16 * 16], blk[64];
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (restart_interval > 0 && count == restart_interval) {
                if (pb) flush_bits(pb);
                if (pb) { put_byte(pb, 0xFF); put_byte(pb, (uint8_t)(0xD0 + (rst++ & 7))); }
                pred[0] = pred[1] = pred[2] = 0;
                count = 0;
            }
            count++;
            for (int y = 0; y < mh; y++)
                for (int x = 0; x < mw; x++) {
                    int sx = mx * mw + x, sy = my * mh + y;
                    if (sx >= w) sx = w - 1;
                    if (sy >= h) sy = h - 1;
                    const uint8_t *p = rgb + ((size_t)sy * w + sx) * 3;
                    float r = p[0], g = p[1], bl = p[2];
                    Y[y * 16 + x] = 0.299f * r + 0.587f * g + 0.114f * bl - 128.0f;
                    Cb[y * 16 + x] = -0.168736f * r - 0.331264f * g + 0.5f * bl;
                    Cr[y * 16 + x] = 0.5f * r - 0.418688f * g - 0.081312f * bl;
                }
            for (int v = 0; v < V; v++)
                for (int hh = 0; hh < H; hh++) {
                    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) blk[y * 8 + x] = Y[(v * 8 + y) * 16 + hh * 8 + x];
                    encode_block(pb, blk, ql, &pred[0], hdl, hal, st_l);
                }
            if (!grey)
                for (int c = 1; c < 3; c++) {
                    const float *src = c == 1 ? Cb : Cr;
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++) {
                            float s = 0;
                            for (int v = 0; v < V; v++) for (int hh = 0; hh < H; hh++) s += src[(y * V + v) * 16 + x * H + hh];
                            blk[y * 8 + x] = s / (H * V);
                        }
                    encode_block(pb, blk, qc, &pred[c], hdc, hac, st_c);
                }
        }
    return 0;
}

long synth_encode(const uint8_t *rgb, int w, int h, int quality, int subsampling, int restart_interval,
                  uint8_t *out, long cap)
{
    return synth_encode2(rgb, w, h, quality, subsampling, restart_interval, 0, out, cap);
}

/* convenience: picture + encode in one call (thread-safe once ctab is initialised) */
long synth_make(int w, int h, uint64_t seed, int quality, int subsampling, int restart_interval, uint8_t *out, long cap)
{
    uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3);
    if (!rgb) return -1;
    synth_picture(rgb, w, h, seed);
    long n = synth_encode(rgb, w, h, quality, subsampling, restart_interval, out, cap);
    free(rgb);
    return n;
}

long synth_make2(int w, int h, uint64_t seed, int quality, int subsampling, int restart_interval, int detail_x100, int optimize,
                 uint8_t *out, long cap)
{
    uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3);
    if (!rgb) return -1;
    synth_picture2(rgb, w, h, seed, detail_x100 / 100.0f);
    long n = synth_encode2(rgb, w, h, quality, subsampling, restart_interval, optimize, out, cap);
    free(rgb);
    return n;
}

void synth_init(void) { float a[64] = {0}, o[64]; fdct(a, o); }
