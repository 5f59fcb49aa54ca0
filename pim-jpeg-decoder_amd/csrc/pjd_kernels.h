// pjd_kernels.h -- launchers of the gfx950 kernels (implemented in pjd_k_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pjd_internal.h"

// All device pointers of one batch, as the kernels see them.
struct PjdDevBatch {
    const PjdDevImage *images;
    const PjdDevHuffRaw *raw_tables;     // n_images * PJD_MAX_TABLES
    uint8_t *luts;                       // decode tables, one blob per image (PjdDevImage::lut_off16)
    const uint16_t *qtab;                // n_images * 3 * 64
    const PjdDevSegment *segs;
    const PjdDevSub *subs;
    const PjdDevHuffWg *hwgs;
    const PjdDevIdctWg *iwgs;
    const uint8_t *ecs;
    int16_t *coef;                       // DENSE scratch (exact-kernel path): dense_du * 64 int16, zigzag-slot order
    uint32_t *ent;                       // coefficient entries of the parallel path: (value << 16) | slot, AC only
    uint32_t *du_end;                    // per data unit: entry index (image-relative) just past its last entry
    uint32_t *seg_ent;                   // per restart segment: entry index (image-relative) of its first entry
    int16_t *dcv;                        // per data unit: DC difference, integrated in place by pjd_k_dc_*
    uint8_t *out;
    int32_t *status;                     // per image
    // Huffman one-pass scratch (zeroed before every launch)
    uint64_t *wg_exit;                   // [2][n_hwg]: exit state of a wave's last owned subsequence | flag; generation 0 / 1
    uint64_t *wg_desc;                   // per Huffman workgroup: look-back descriptor (status | poison | head | units | entries)
    uint32_t *ticket;                    // wave index dispenser
    uint32_t *dbg;                       // PJD_DEBUG_STATS: per wave, 8 timestamps (10 ns units); else null
    // DC prediction scratch
    uint32_t *dc_agg;                    // per DC block: {sumY, sumCb, sumCr, has_head}
    uint32_t *dc_carry;                  // per DC block: carry-in {Y, Cb, Cr, pad}
    const uint32_t *dcblk_image;         // per DC block: owning image
    unsigned long long *stats;           // [16] diagnostics: 0 re-sync rounds (stage A), 1 lane passes in them, 2 / 3 the same for the stitch stage (B)
    uint32_t n_images, n_hwg, n_iwg, n_dcblk;
    uint32_t sub_bytes;                  // Huffman subsequence size of this batch
    uint32_t max_lut_bytes;              // largest PjdDevImage::lut_bytes in the batch (sizes the dynamic LDS of the Huffman kernels)
};

// ---- back end (pjd_k_backend.hip) ------------------------------------------------
void pjd_launch_dpu_payload(hipStream_t s, const uint32_t *metadata, int16_t *mcus, int n_dpus);
void pjd_launch_idct_colour(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, uint32_t n_wg);          // dense input (exact path)
void pjd_launch_idct_colour_sparse(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, uint32_t n_wg);   // entry-stream input
void pjd_launch_dc_scan(hipStream_t s, const PjdDevBatch &b);      // two kernels: local scan + carry
// ---- entropy decode (pjd_k_huffman.hip) -----------------------------------------
void pjd_launch_huff_sequential(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, uint32_t n);
void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b);
void pjd_launch_huff_onepass(hipStream_t s, const PjdDevBatch &b);   // synchronise + stitch + scan + write
