// pjd_kernels.h -- launchers of the gfx950 kernels (implemented in pjd_k_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pjd_internal.h"

// All device pointers of one batch, as the kernels see them.
struct PjdDevBatch {
    const PjdDevImage *images;
    const PjdDevTset *tsets;
    const PjdDevHuffRaw *raw_tables;     // n_tsets * PJD_MAX_TABLES
    uint8_t *luts;                       // decode tables, one blob per table set (PjdDevTset::lut_off16)
    const uint16_t *qtab;                // n_images * 3 * 64
    const PjdDevSegment *segs;
    const PjdDevSub *lanes;
    const PjdDevHuffWave *hwaves;
    const PjdDevHuffWg *hwgs;
    const PjdDevIdctWg *iwgs;
    const PjdDevScan *pscans;            // scans of the progressive frames of the batch (PjdDevImage::pscan_base)
    const uint32_t *group_images;        // picture groups (pjd_internal.h): image indices, group after group
    const uint32_t *iwg_order;           // ... and back-end workgroup indices (into iwgs / marks), group after group
    const uint8_t *ecs;
    uint32_t *words;                     // transposed bitstream words: [wave][PJD_WORD_ROWS][64]
    int16_t *coef;                       // DENSE scratch (exact-kernel path): dense_du * 64 int16, zigzag-slot order
    uint16_t *ent;                       // lane streams: lane q of image `im` owns slots [im.ent_base + (q - im.lane_base) * im.lane_cap, + im.lane_cap)
    PjdDevLaneInfo *lane_info;           // per lane
    PjdDevLaneDc *lane_dc;               // per lane
    uint16_t *dc_blk;                    // per DC scan block: aggregate {Y, Cb, Cr, has_head}, then carry-in {Y, Cb, Cr, -}
    PjdDevMark *marks;                   // per IDCT workgroup of the parallel path
    uint8_t *out;
    int32_t *status;                     // per image
    PjdDevImState *imstate;              // per image: first entropy-coding error / earliest unresolved irregularity of this decode
    // Huffman kernel scratch (zeroed before every launch)
    uint64_t *wave_gen;                  // [PJD_GENS][n_hwave]: exit state of a wave's last lane | flag, one word per generation (pjd_internal.h)
    uint64_t *wave_desc;                 // per Huffman wave: look-back descriptor (status | poison | head | units)
    uint32_t *ticket;                    // workgroup index dispenser
    uint32_t *ready_list;                // pull back end (pjd_internal.h): [n_iwg] range index + 1 in the order pictures complete; 0: not yet
    uint32_t *ready_tail;                // entries appended so far
    uint32_t *range_done;                // [n_iwg] 1: the range's pixels are written
    uint32_t pull;                       // 1: this launch takes part in the pull protocol (Huffman waves publish, back-end workgroups take from the list)
    uint32_t *dbg;                       // PJD_DEBUG_STATS: per wave, 8 timestamps (10 ns units); else null
    unsigned long long *stats;           // [16] diagnostics: 0 re-sync rounds, 1 lane passes in them, 2 / 3 the same for the stitch
                                         //      stage, PJD_STAT_FLAG0.. waves that flagged their image, by reason, 12 / 13 cooperative walks and the lanes walked in them
    uint32_t n_images, n_tsets, n_lanes, n_hwave, n_hwg, n_iwg, n_dcblk;
    uint32_t sub_bytes;                  // Huffman subsequence size of this batch
    uint32_t word_rows;                  // PJD_WORD_ROWS(sub_bytes)
    uint32_t lane_cap;                   // largest PjdDevImage::lane_cap of the batch (diagnostics)
    uint32_t max_lut_bytes;              // largest PjdDevTset::lut_bytes in the batch (sizes the dynamic LDS of the Huffman kernel)
};

// ---- back end (pjd_k_backend.hip) ------------------------------------------------
void pjd_launch_dpu_payload(hipStream_t s, const uint32_t *metadata, int16_t *mcus, int n_dpus);
void pjd_launch_zero(hipStream_t s, void *p, size_t bytes);          // bytes: a multiple of 16
void pjd_launch_reset(hipStream_t s, const PjdDevBatch &b, const int32_t *status_init, uint64_t *opstate, size_t opstate_words, uint32_t dbg_words);   // per-decode state
void pjd_launch_copy_out(hipStream_t s, const void *src, void *dst_mapped, uint64_t bytes);   // HBM -> mapped page-locked host memory
// dense input (exact path): wgs[k].pad_ = index into dense_base[] (data unit 0 of that image's scratch)
void pjd_launch_idct_colour(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, const uint64_t *dense_base, uint32_t n_wg);
void pjd_launch_idct_colour_lanes(hipStream_t s, const PjdDevBatch &b);                                     // lane-stream input
void pjd_launch_lane_dc_scan(hipStream_t s, const PjdDevBatch &b);      // three kernels: per-image verdict (status words), local scan, carry
// one picture group (pjd_internal.h): verdict + DC predictors of its pictures in one launch, then its back-end workgroups
void pjd_launch_group_dc(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g);
void pjd_launch_group_idct(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g);
// pull back end (pjd_internal.h): the launch that runs beside the entropy decoder, and the sweep over what it left
void pjd_launch_idct_pull(hipStream_t s, const PjdDevBatch &b);
void pjd_launch_idct_sweep(hipStream_t s, const PjdDevBatch &b);
// ---- stage-level parity (pjd_k_coefdump.hip): coefficients in the reference's MCU_buffer layout; `out` is zeroed by the caller
void pjd_launch_coefdump_lanes(hipStream_t s, const PjdDevBatch &b, uint32_t image, uint32_t n_iwg, int16_t *out);
void pjd_launch_coefdump_dense(hipStream_t s, const PjdDevBatch &b, uint32_t image, const int16_t *scratch, uint32_t first_du, uint32_t n_du, int16_t *out);
// ---- entropy decode (pjd_k_huffman.hip, pjd_k_huffman_seq.hip) ---------------------
// exact kernel: image_list[k] decodes into the dense scratch from data unit dense_base[k]
void pjd_launch_huff_sequential(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n);
void pjd_launch_huff_exact_lut(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n);   // the exact decoder, table-driven
// progressive frames among image_list (the others are skipped): scan by scan into the dense scratch (pjd_k_progressive.hip)
void pjd_launch_progressive(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n);
void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b);
void pjd_launch_lane_words(hipStream_t s, const PjdDevBatch &b);     // bitstream -> per-lane big-endian words, transposed per wave
void pjd_launch_lane_words_group(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g);   // the same for the waves of one picture group
void pjd_launch_huff_lanes(hipStream_t s, const PjdDevBatch &b);     // synchronise + stitch + scan + write
void pjd_launch_huff_lanes_group(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g, uint32_t group_index);   // the same for one picture group
