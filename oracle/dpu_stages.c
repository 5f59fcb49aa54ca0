/*
 * oracle/dpu_stages.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's UPMEM-DPU kernel, i.e. the three per-DPU
 * stages  dequantise -> 8x8 integer IDCT -> chroma upsample + YCbCr->RGB,
 * operating on exactly the reference's per-DPU payload:
 *     metadata_buffer  u32[276]   (reference src/decoder_dpu.c:57, index map
 *                                  src/decoder_host.cpp:156-178)
 *     mcus             i16[19200] (reference src/decoder_dpu.c:58; 25 "blocks"
 *                                  of 16x16 px, 768 int16 each, laid out
 *                                  [component][position][64])
 *
 * The reference file itself (src/decoder_dpu.c) cannot be compiled in this
 * image: it includes the UPMEM SDK headers <mram.h> <defs.h> <barrier.h>
 * <perfcounter.h>, which are absent, and writing stand-ins for them is not
 * allowed.  This restatement is therefore pinned by the known-answer vector of
 * SURVEY.md section 0.4 (BMP sha256 of the bundled ILSVRC2012_val_00000001.JPEG)
 * through the REAL reference scanner / Huffman decoder / BMP writer, which do
 * compile stand-alone (see oracle/Makefile, target _ref).
 *
 * Arithmetic notes: every intermediate is a 32-bit int, every store goes back
 * through int16 (wrap), every >> is an arithmetic shift -- as on the DPU.
 */
#include <stdint.h>
#include <string.h>

#define ORC_BLK16_PER_DPU_MAX 4096

typedef struct {
    int ncomp;
    int vsamp, hsamp;      /* luma sampling factors               */
    int qt_id[3];
    int total_blk16;       /* metadata[19] / 4                    */
    const uint32_t *qt;    /* 4 x 64, natural order (via the reference zigzag) */
} dpu_meta;

/* reference src/decoder_dpu.c:112-132 (load_data) */
static void unpack_meta(const uint32_t *m, dpu_meta *d)
{
    d->ncomp = (int)m[4];
    d->vsamp = (int)(uint8_t)m[5];
    d->hsamp = (int)(uint8_t)m[6];
    for (int i = 0; i < 3; i++) d->qt_id[i] = 0;
    for (int i = 0; i < d->ncomp && i < 3; i++) d->qt_id[i] = (int)(uint8_t)m[7 + i];
    d->total_blk16 = (int)m[19] / 4;
    d->qt = m + 20;
}

/* One blk16 in WRAM order: [position 0..3][component 0..2][64]
 * (reference src/decoder_dpu.c:11-17, load_block :134-144). */
typedef struct { int16_t c[4][3][64]; } wram_blk;

static void blk_load(const int16_t *mram, int idx, wram_blk *w)
{
    const int16_t *base = mram + (size_t)idx * 768;
    for (int comp = 0; comp < 3; comp++)
        for (int pos = 0; pos < 4; pos++)
            memcpy(w->c[pos][comp], base + comp * 256 + pos * 64, 128);
}

static void blk_store(int16_t *mram, int idx, const wram_blk *w)
{
    int16_t *base = mram + (size_t)idx * 768;
    for (int comp = 0; comp < 3; comp++)
        for (int pos = 0; pos < 4; pos++)
            memcpy(base + comp * 256 + pos * 64, w->c[pos][comp], 128);
}

/* reference src/decoder_dpu.c:158-177: short *= uint32, product formed in
 * 32-bit unsigned arithmetic and truncated to int16 on store. */
static void stage_dequant(wram_blk *w, const dpu_meta *d)
{
    for (int comp = 0; comp < d->ncomp; comp++) {
        const uint32_t *q = d->qt + d->qt_id[comp] * 64;
        for (int k = 0; k < 64; k++)
            for (int pos = 0; pos < 4; pos++) {
                uint32_t prod = (uint32_t)(int32_t)w->c[pos][comp][k] * q[k];
                w->c[pos][comp][k] = (int16_t)prod;
            }
    }
}

/* One 1-D pass of the reference butterfly (src/decoder_dpu.c:219-267 /
 * :271-319).  `s` = stride between the 8 samples. */
static void idct_1d(int16_t *p, int s)
{
    int g0 = (p[0 * s] * 181) >> 5;
    int g1 = (p[4 * s] * 181) >> 5;
    int g2 = (p[2 * s] * 59) >> 3;
    int g3 = (p[6 * s] * 49) >> 4;
    int g4 = (p[5 * s] * 71) >> 4;
    int g5 = (p[1 * s] * 251) >> 5;
    int g6 = (p[7 * s] * 25) >> 4;
    int g7 = (p[3 * s] * 213) >> 5;

    int f4 = g4 - g7, f5 = g5 + g6, f6 = g5 - g6, f7 = g4 + g7;
    int e2 = g2 - g3, e3 = g2 + g3, e5 = f5 - f7, e7 = f5 + f7, e8 = f4 + f6;

    int d2 = (e2 * 181) >> 7;
    int d4 = (f4 * 277) >> 8;
    int d5 = (e5 * 181) >> 7;
    int d6 = (f6 * 669) >> 8;
    int d8 = (e8 * 49) >> 6;

    int c0 = g0 + g1, c1 = g0 - g1, c2 = d2 - e3, c4 = d4 + d8;
    int c5 = d5 + e7, c6 = d6 - d8, c8 = c5 - c6;
    int b0 = c0 + e3, b1 = c1 + c2, b2 = c1 - c2, b3 = c0 - e3;
    int b4 = c4 - c8, b6 = c6 - e7;

    p[0 * s] = (int16_t)((b0 + e7) >> 4);
    p[1 * s] = (int16_t)((b1 + b6) >> 4);
    p[2 * s] = (int16_t)((b2 + c8) >> 4);
    p[3 * s] = (int16_t)((b3 + b4) >> 4);
    p[4 * s] = (int16_t)((b3 - b4) >> 4);
    p[5 * s] = (int16_t)((b2 - c8) >> 4);
    p[6 * s] = (int16_t)((b1 - b6) >> 4);
    p[7 * s] = (int16_t)((b0 - e7) >> 4);
}

/* reference src/decoder_dpu.c:179-207 + :210-321: all 12 data units of every
 * blk16, rows first, then columns, int16 storage between the passes. */
static void stage_idct(wram_blk *w)
{
    for (int pos = 0; pos < 4; pos++)
        for (int comp = 0; comp < 3; comp++) {
            int16_t *u = w->c[pos][comp];
            for (int r = 0; r < 8; r++) idct_1d(u + r * 8, 1);
            for (int col = 0; col < 8; col++) idct_1d(u + col, 8);
        }
}

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* reference src/decoder_dpu.c:361-390; in place, reverse raster order. */
static void colour_unit(wram_blk *w, const dpu_meta *d, int cidx, int yidx, int v, int h)
{
    for (int y = 7; y >= 0; y--)
        for (int x = 7; x >= 0; x--) {
            int p = y * 8 + x;
            int q = ((y / d->vsamp) + 4 * v) * 8 + (x / d->hsamp) + 4 * h;
            int Y = w->c[yidx][0][p];
            int Cb = w->c[cidx][1][q];
            int Cr = w->c[cidx][2][q];
            /* each product is a 32-bit int multiply (wraps), shifted on its own */
            int r = Y + ((int32_t)(5880414u * (uint32_t)Cr) >> 22) + 128;
            int g = Y - ((int32_t)(1442840u * (uint32_t)Cb) >> 22)
                      - ((int32_t)(2994733u * (uint32_t)Cr) >> 22) + 128;
            int b = Y + ((int32_t)(7432306u * (uint32_t)Cb) >> 22) + 128;
            w->c[yidx][0][p] = (int16_t)clamp255(r);
            w->c[yidx][1][p] = (int16_t)clamp255(g);
            w->c[yidx][2][p] = (int16_t)clamp255(b);
        }
}

/* reference src/decoder_dpu.c:323-359: the four (cbcr, y, v, h) tuples per
 * luma sampling mode; the chroma-carrying position is converted last. */
static void stage_colour(wram_blk *w, const dpu_meta *d)
{
    if (d->vsamp == 1 && d->hsamp == 1) {
        for (int i = 0; i < 4; i++) colour_unit(w, d, i, i, 0, 0);
    }
    if (d->vsamp == 2 && d->hsamp == 1) {
        colour_unit(w, d, 0, 2, 1, 0); colour_unit(w, d, 0, 0, 0, 0);
        colour_unit(w, d, 1, 3, 1, 0); colour_unit(w, d, 1, 1, 0, 0);
    }
    if (d->vsamp == 1 && d->hsamp == 2) {
        colour_unit(w, d, 0, 1, 0, 1); colour_unit(w, d, 0, 0, 0, 0);
        colour_unit(w, d, 2, 3, 0, 1); colour_unit(w, d, 2, 2, 0, 0);
    }
    if (d->vsamp == 2 && d->hsamp == 2) {
        colour_unit(w, d, 0, 3, 1, 1); colour_unit(w, d, 0, 2, 1, 0);
        colour_unit(w, d, 0, 1, 0, 1); colour_unit(w, d, 0, 0, 0, 0);
    }
}

/* One DPU "exec": reference src/decoder_dpu.c:82-110.  The reference runs the
 * three stages as three sweeps over MRAM; per blk16 the result is identical to
 * running them back to back, because no stage reads another blk16. */
void orc_dpu_exec(const uint32_t *metadata, int16_t *mcus)
{
    dpu_meta d;
    wram_blk w;
    unpack_meta(metadata, &d);
    for (int b = 0; b < d.total_blk16; b++) {
        blk_load(mcus, b, &w);
        stage_dequant(&w, &d);
        blk_store(mcus, b, &w);
    }
    for (int b = 0; b < d.total_blk16; b++) {
        blk_load(mcus, b, &w);
        stage_idct(&w);
        blk_store(mcus, b, &w);
    }
    for (int b = 0; b < d.total_blk16; b++) {
        blk_load(mcus, b, &w);
        stage_colour(&w, &d);
        blk_store(mcus, b, &w);
    }
}

/* Stage-level entry points for parity tests (same payload, one stage only). */
void orc_dpu_stage(const uint32_t *metadata, int16_t *mcus, int stage)
{
    dpu_meta d;
    wram_blk w;
    unpack_meta(metadata, &d);
    for (int b = 0; b < d.total_blk16; b++) {
        blk_load(mcus, b, &w);
        if (stage == 0) stage_dequant(&w, &d);
        else if (stage == 1) stage_idct(&w);
        else stage_colour(&w, &d);
        blk_store(mcus, b, &w);
    }
}
