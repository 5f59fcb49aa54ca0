/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C-ABI driver around the REAL reference translation units that compile
 * stand-alone in this image:
 *     /root/reference/src/jpeg_scanner.cpp   (read_JPEG, decode_Huffman_data)
 *     /root/reference/src/bmp_writer.cpp     (write_BMP)
 * They are compiled where they lie (oracle/Makefile, target _ref); nothing of
 * the reference is copied into this repository.  The reference's device stage
 * (src/decoder_dpu.c) and its orchestrator (src/decoder_host.cpp) need the
 * UPMEM SDK headers and are NOT buildable here; the driver therefore runs the
 * C restatement oracle/dpu_stages.c between the reference's Huffman decoder and
 * the reference's BMP writer, and repeats -- in its own words -- the few lines
 * of host glue that size the per-DPU buffers and pack metadata
 * (reference src/decoder_host.cpp:125-128,156-179).
 *
 * Output library: oracle/_ref/libpjdref.so, CLI: oracle/_ref/ref_decode.
 */
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <cstring>
#include <string>
#include <vector>

#include "headers/jpeg.h"   /* reference header, found through -I/root/reference/src */
#include "headers/bmp.h"

extern "C" void orc_dpu_exec(const uint32_t *metadata, int16_t *mcus);

namespace {

const int kSlotsPerDpu = MAX_MCU_PER_DPU;             /* 100 mcu8 = 25 blk16 */
const int kShortsPerDpu = MAX_MCU_PER_DPU * 3 * 64;   /* 19200               */

int dpus_needed(const Header *h)
{
    int pw = (int)(h->mcu_width_real + 1) / 2 * 2;
    int ph = (int)(h->mcu_height_real + 1) / 2 * 2;
    return (pw * ph + kSlotsPerDpu - 1) / kSlotsPerDpu;
}

void pack_metadata(const Header *h, uint32_t *m)
{
    std::memset(m, 0, 276 * sizeof(uint32_t));
    m[0] = h->mcu_height;       m[1] = h->mcu_width;
    m[2] = h->mcu_height_real;  m[3] = h->mcu_width_real;
    m[4] = h->num_components;
    m[5] = h->v_sampling_factor; m[6] = h->h_sampling_factor;
    const unsigned n = h->num_components;
    for (unsigned j = 0; j < n; j++) {
        m[7 + j] = h->color_components[j].QT_ID;
        m[7 + n + j] = h->color_components[j].h_sampling_factor;
        m[7 + 2 * n + j] = h->color_components[j].v_sampling_factor;
    }
    m[17] = h->height; m[18] = h->width; m[19] = MAX_MCU_PER_DPU;
    for (unsigned t = 0; t < 4; t++) {
        if (!h->quantization_tables[t].set) break;   /* stops at first unset table */
        for (unsigned k = 0; k < 64; k++) m[20 + 64 * t + k] = h->quantization_tables[t].table[k];
    }
}

}  // namespace

extern "C" {

struct ref_info {
    int32_t valid, width, height, ncomp, hsamp, vsamp;
    int32_t mcu_w, mcu_h, mcu_w_real, mcu_h_real, restart_interval, frame_type;
    int32_t n_dpus;
    int64_t ecs_len;
    uint8_t comp_h[3], comp_v[3], comp_qt[3], comp_dc[3], comp_ac[3];
    uint8_t qt_set[4], dc_set[4], ac_set[4];
    uint32_t qt[4][64];
    uint8_t dc_offsets[4][17], dc_symbols[4][162];
    uint8_t ac_offsets[4][17], ac_symbols[4][162];
};

/* read_JPEG (reference src/jpeg_scanner.cpp:345). Returns an opaque handle or
 * NULL (file could not be opened). */
void *ref_open(const char *path) { return read_JPEG(std::string(path)); }

void ref_close(void *hp) { delete static_cast<Header *>(hp); }

void ref_get_info(void *hp, ref_info *o)
{
    const Header *h = static_cast<Header *>(hp);
    std::memset(o, 0, sizeof *o);
    o->valid = h->valid; o->width = h->width; o->height = h->height;
    o->ncomp = h->num_components; o->hsamp = h->h_sampling_factor; o->vsamp = h->v_sampling_factor;
    o->mcu_w = h->mcu_width; o->mcu_h = h->mcu_height;
    o->mcu_w_real = h->mcu_width_real; o->mcu_h_real = h->mcu_height_real;
    o->restart_interval = h->restart_interval; o->frame_type = h->frame_type;
    o->ecs_len = (int64_t)h->huffman_data.size();
    o->n_dpus = h->valid ? dpus_needed(h) : 0;
    for (int i = 0; i < 3; i++) {
        o->comp_h[i] = h->color_components[i].h_sampling_factor;
        o->comp_v[i] = h->color_components[i].v_sampling_factor;
        o->comp_qt[i] = h->color_components[i].QT_ID;
        o->comp_dc[i] = h->color_components[i].DHT_ID;
        o->comp_ac[i] = h->color_components[i].AHT_ID;
    }
    for (int t = 0; t < 4; t++) {
        o->qt_set[t] = h->quantization_tables[t].set;
        o->dc_set[t] = h->huffman_DC_tables[t].set;
        o->ac_set[t] = h->huffman_AC_tables[t].set;
        std::memcpy(o->qt[t], h->quantization_tables[t].table, sizeof o->qt[t]);
        std::memcpy(o->dc_offsets[t], h->huffman_DC_tables[t].offsets, 17);
        std::memcpy(o->dc_symbols[t], h->huffman_DC_tables[t].symbols, 162);
        std::memcpy(o->ac_offsets[t], h->huffman_AC_tables[t].offsets, 17);
        std::memcpy(o->ac_symbols[t], h->huffman_AC_tables[t].symbols, 162);
    }
}

/* Destuffed, RST-stripped entropy-coded bytes exactly as the reference holds
 * them (Header::huffman_data, reference src/headers/jpeg.h:168). */
int64_t ref_get_ecs(void *hp, uint8_t *dst, int64_t cap)
{
    const Header *h = static_cast<Header *>(hp);
    int64_t n = (int64_t)h->huffman_data.size();
    if (dst && cap >= n && n) std::memcpy(dst, h->huffman_data.data(), (size_t)n);
    return n;
}

void ref_get_metadata(void *hp, uint32_t *m276) { pack_metadata(static_cast<Header *>(hp), m276); }

/* decode_Huffman_data (reference src/jpeg_scanner.cpp:707) into a zeroed
 * n_dpus x 19200 int16 buffer; returns the reference's bool (which the
 * reference host ignores, src/decoder_host.cpp:181). */
int ref_huffman(void *hp, int16_t *mcus, int n_dpus)
{
    Header *h = static_cast<Header *>(hp);
    std::vector<std::vector<short>> buf((size_t)n_dpus, std::vector<short>(kShortsPerDpu));
    bool ok = decode_Huffman_data(h, buf, 0);
    for (int d = 0; d < n_dpus; d++) std::memcpy(mcus + (size_t)d * kShortsPerDpu, buf[d].data(), kShortsPerDpu * 2);
    return ok ? 1 : 0;
}

/* write_BMP (reference src/bmp_writer.cpp:19). */
void ref_write_bmp(const uint32_t *m276, const int16_t *mcus, int n_dpus, const char *out_path)
{
    std::vector<uint32_t> meta(m276, m276 + 276);
    std::vector<std::vector<short>> buf((size_t)n_dpus, std::vector<short>(kShortsPerDpu));
    for (int d = 0; d < n_dpus; d++) std::memcpy(buf[d].data(), mcus + (size_t)d * kShortsPerDpu, kShortsPerDpu * 2);
    write_BMP(meta, buf, 0, std::string(out_path));
}

/* Whole pipeline for one file: reference scanner -> reference Huffman ->
 * restated DPU stages -> reference BMP writer.  Mirrors the control flow of
 * reference src/decoder_host.cpp:118-183,326-331 for a single image.
 * Returns 0 = BMP written, 1 = rejected ("Invalid JPEG"), 2 = cannot open. */
int ref_decode_file(const char *in_path, const char *out_path)
{
    Header *h = read_JPEG(std::string(in_path));
    if (h == nullptr || !h->valid) {
        std::cout << in_path << ": Error - Invalid JPEG\n";   /* reference src/decoder_host.cpp:120-123 */
        std::cout.flush();
        int rc = h ? 1 : 2;
        delete h;
        return rc;
    }
    int n = dpus_needed(h);
    std::vector<uint32_t> meta(276);
    pack_metadata(h, meta.data());
    std::vector<std::vector<short>> buf((size_t)n, std::vector<short>(kShortsPerDpu));
    decode_Huffman_data(h, buf, 0);      /* result ignored, as in the reference */
    for (int d = 0; d < n; d++) orc_dpu_exec(meta.data(), buf[d].data());
    write_BMP(meta, buf, 0, std::string(out_path));
    std::cout.flush();
    delete h;
    return 0;
}

}  /* extern "C" */

#ifdef REF_DRIVER_MAIN
int main(int argc, char **argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s in.jpg out.bmp\n", argv[0]); return 64; }
    int rc = ref_decode_file(argv[1], argv[2]);
    std::cout.flush();
    return rc;
}
#endif
