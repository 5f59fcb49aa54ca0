// pjd_internal.h -- data layout shared by the host planner and the gfx950 kernels.
//
// Everything the kernels read lives in HBM in these plain structs / arrays; the
// host planner (pjd_plan.cpp) fills them from pjd_image_desc, the API
// (pjd_api.hip) uploads them once per batch.
#pragma once
#include <stdint.h>

// ---- tunables -----------------------------------------------------------------
#define PJD_SUB_BYTES_MIN  128      // Huffman subsequence (bytes of bitstream per decode lane): chosen per batch
#define PJD_SUB_BYTES_MAX  1024     //   by the planner (multiple of 64 in this range), see pjd_plan.cpp
#define PJD_HUFF_THREADS   64       // one wave per Huffman workgroup: lanes exchange states by shuffles, no barriers
#define PJD_HUFF_OWNED     63       // subsequences owned per workgroup (lane 0 = predecessor overlap)
#ifndef PJD_NCHK
#define PJD_NCHK           8        // checkpoints per subsequence (trajectory states a re-sync bridge can merge into)
#endif
#define PJD_LUT_BITS       10       // first-level Huffman LUT width
#define PJD_L1_BYTES       (2 << PJD_LUT_BITS)   // one first-level table: 1024 x u16
#define PJD_LUT_LDS_MAX    (6 * PJD_L1_BYTES + 8192)  // decode tables of one image in LDS; larger -> exact kernel
#define PJD_MAX_TABLES     6        // distinct Huffman tables one image can reference (3 DC + 3 AC)
#define PJD_SYNC_MAX_ITERS 24       // re-sync rounds per wave before giving up (-> exact fallback)
#define PJD_DC_BLOCK       256      // MCUs per DC-prediction scan block
#define PJD_IDCT_THREADS   256
#define PJD_IDCT_MAX_DU    96       // data units staged in LDS per IDCT workgroup
#define PJD_COEF_SENTINEL  (-32768) // "slot 52 was visited with an explicit 0" (zigzag quirk)

// ---- image flags (device side) -----------------------------------------------------
#define PJD_IF_STANDARD_RESTART 1u  // restart every RI-th MCU; else the reference's (y*Wr+x)%RI rule
#define PJD_IF_SEQUENTIAL       2u  // routed to the exact one-lane kernel up front
#define PJD_IF_BMP              4u  // output is a BMP file image (else tight RGB8)

// status word per image: low 8 bits = PJD_ST_* class, bit 8 = "fast path gave up, needs exact kernel"
#define PJD_STW_NEEDS_EXACT 0x100

struct PjdDevImage {
    uint32_t width, height;
    uint32_t mcux, mcuy, n_mcu;        // MCU grid (MCU = 8*hs x 8*vs pixels)
    uint32_t ncomp, hs, vs;
    uint32_t n_luma;                   // hs*vs
    uint32_t dus_per_mcu;              // n_luma + ncomp - 1
    uint32_t restart_interval;
    uint32_t flags;
    uint32_t ref_mcu_w, ref_mcu_h, ref_mcu_w_real;   // reference Header::mcu_width / mcu_height / mcu_width_real
    uint32_t ecs_len;                  // bytes
    uint64_t ecs_off;                  // into the batch bitstream buffer (16-byte aligned)
    uint64_t du_base;                  // first data unit of this image in the per-unit arrays (dcv, and du_off at du_base + image index)
    uint64_t ent_base;                 // first entry of this image in the coefficient-entry stream
    uint64_t dense_base;               // first data unit of this image in the DENSE scratch (exact-kernel path only)
    uint32_t n_du;
    uint32_t image_index;
    uint32_t out_stride;               // bytes per output row (BMP: 3W + W%4, RGB8: 3W)
    uint64_t out_off;                  // into the batch output buffer (256-byte aligned)
    uint32_t seg_base, n_seg;          // into PjdDevSegment[]
    uint32_t sub_base, n_sub;          // into PjdDevSub[]
    uint32_t hwg_base, n_hwg;          // Huffman workgroups of this image
    uint32_t dcblk_base, n_dcblk;      // DC scan blocks of this image
    uint32_t first_mcu, last_mcu;      // MCU range this shard decodes: [first_mcu, last_mcu)
    uint8_t  tbl_slot[3][2];           // [component][0=DC,1=AC] -> table slot 0..n_tables-1
    uint8_t  n_tables;
    uint8_t  pad_[1];
    // decode tables of this image inside PjdDevBatch::luts (layout: see "decode-ready tables" below)
    uint32_t lut_off16;                // blob offset in 16-byte units
    uint32_t lut_bytes;                // multiple of 16; 0 for images routed to the exact kernel
    uint16_t l2_off[PJD_MAX_TABLES];   // second-level region of table k: first u16 index, relative to the image's blob
    uint16_t l2_p0[PJD_MAX_TABLES];    // 10-bit prefixes [p0, p1) hold codes longer than PJD_LUT_BITS
    uint16_t l2_p1[PJD_MAX_TABLES];
    uint16_t pad2_[2];
};

// raw Huffman table as shipped by the host (reference HuffmanTable, jpeg.h:129-134)
struct PjdDevHuffRaw {
    uint8_t offsets[17];
    uint8_t symbols[162];
    uint8_t is_ac;
};                                     // 180 bytes

// decode-ready tables, built on the device by pjd_k_build_tables.  One blob per image:
//   [table 0 L1][table 1 L1]...[table n-1 L1][second-level regions, 64 u16 per long prefix]
// L1 is indexed by the next PJD_LUT_BITS bits.  Entry (u16):
//   bit 15 = 0 : (length << 8) | symbol, length 1..10;  bit 14 set = no code starts with these bits
//                (then length = 16, symbol = 0: what the reference's get_next_symbol consumes before failing)
//   bit 15 = 1 : codes with this prefix are longer than 10 bits; bits 14..0 = u16 index (relative to the blob)
//                of the prefix's 64-entry second-level table, indexed by the following 6 bits; entries there
//                have the first form with length 11..16.
// Canonical codes keep all long codes in one contiguous range of prefixes [p0, p1), so the second level
// costs 128 bytes per long prefix (Annex K tables: 5 prefixes for an AC table, 0..1 for a DC table).

struct PjdDevSegment {                 // one restart segment
    uint32_t byte_start;               // relative to the image's ecs
    uint32_t byte_end;
    uint32_t first_du;                 // image-relative index of its first data unit
    uint32_t n_du;
};

struct PjdDevSub {                     // one Huffman subsequence (decode lane)
    uint32_t byte_start;               // relative to the image's ecs
    uint32_t seg;                      // global segment index | (1u<<31 if first subsequence of its segment)
};

struct PjdDevHuffWg {                  // one Huffman workgroup
    uint32_t image;
    uint32_t first_sub;                // global subsequence index of the first OWNED subsequence
    uint32_t n_sub;                    // owned, 1..PJD_HUFF_OWNED
    uint32_t pad_;
};

struct PjdDevIdctWg {                  // one IDCT/colour workgroup
    uint32_t image;
    uint32_t first_mcu;
    uint32_t n_mcu;
    uint32_t pad_;
};

// packed decoder state at a subsequence boundary: bit position (relative to the
// image's ecs), data-unit phase within the MCU, zigzag slot
static inline __host__ __device__ uint64_t pjd_pack_state(uint32_t p, uint32_t c, uint32_t z)
{
    return (uint64_t)p | ((uint64_t)c << 32) | ((uint64_t)z << 40);
}
