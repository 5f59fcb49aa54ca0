// pjd_plan.h -- host-side batch planner: pjd_image_desc[] -> device work lists.
#pragma once
#include <string>
#include <vector>

#include "../../include/pjd.h"
#include "pjd_internal.h"

struct PjdHostImage {
    const uint8_t *ecs_src;     // caller memory: first byte to upload
    uint64_t ecs_copy_len;      // bytes to upload
    uint64_t out_bytes;         // size of this picture in the chosen output format
    bool sequential;            // routed to the exact one-lane kernel up front (or a progressive frame: pjd_k_progressive)
};

struct PjdHostScan {            // bytes of one scan of a progressive frame: where they come from, where they go
    const uint8_t *src;
    uint64_t len, off;
};

struct PjdPlan {
    int out_format = 0;
    uint32_t sub_bytes = 512;              // Huffman subsequence size chosen for this batch
    uint32_t min_sym_bits = 16;            // fewest bits any Huffman symbol of the batch's tables consumes (code + value bits)
    uint32_t lane_cap = 0;                 // slots of the largest lane region of the batch (a picture's own: PjdDevImage::lane_cap)
    std::vector<PjdDevImage> images;
    std::vector<PjdHostImage> host;
    std::vector<PjdDevTset> tsets;         // table sets: images with identical Huffman tables share one
    std::vector<PjdDevHuffRaw> tables;     // n_tsets * PJD_MAX_TABLES
    std::vector<uint16_t> qtab;            // n_images * 3 * 64, natural order, low 16 bits
    std::vector<PjdDevSegment> segs;
    std::vector<PjdDevSub> subs;           // Huffman lanes
    std::vector<PjdDevHuffWave> hwaves;    // 64 consecutive lanes of one image
    std::vector<PjdDevHuffWg> hwgs;        // up to 4 consecutive waves of one table set
    std::vector<PjdDevIdctWg> iwgs;        // images decoded by the parallel path (lane-stream back end)
    std::vector<PjdDevIdctWg> iwgs_dense;  // images routed to the exact kernel up front (dense back end)
    std::vector<PjdDevScan> pscans;        // scans of progressive frames
    std::vector<PjdHostScan> host_scans;   // one per entry of pscans
    std::vector<PjdDevGroup> groups;       // picture groups (pjd_internal.h); one group: the whole batch in one chain of launches
    std::vector<uint32_t> group_images;    // image indices, group after group
    std::vector<uint32_t> iwg_order;       // back-end workgroup indices, group after group
    std::vector<uint32_t> seq_images;      // indices of `sequential` images (progressive frames included)
    std::vector<uint32_t> fast_images;     // the others
    uint64_t ecs_buf_bytes = 0;            // size of the packed bitstream buffer (incl. padding)
    uint64_t n_du = 0;                     // data units in the batch
    uint64_t n_ent = 0;                    // capacity of the lane streams in 16-bit slots: sum over pictures of lanes * PjdDevImage::lane_cap
    uint64_t n_words = 0;                  // transposed bitstream words: waves * PJD_WORD_ROWS * 64
    uint64_t dense_du = 0;                 // data units of the dense scratch (exact-kernel images + one fallback image)
    uint64_t out_buf_bytes = 0;
    uint64_t n_dcblk = 0;
    uint64_t lut_buf_bytes = 0;            // decode-table blobs of all table sets the parallel path uses
    uint32_t max_lut_bytes = 0;            // largest blob (dynamic LDS of the Huffman kernel)
    uint64_t pixels = 0, ecs_bytes = 0, out_bytes = 0;
    int plan_mode = 0;                     // PJD_PLAN_*
    std::vector<uint32_t> tset_step_bits;  // per table set: fewest bits of stream per step of the write pass, x 256 (sizes the lane regions)
};

// Returns PJD_OK or PJD_E_ARG (with a message in `err`).
// sub_bytes_override: 0 = choose from the batch size, else a multiple of 64 in [PJD_SUB_BYTES_MIN, PJD_SUB_BYTES_MAX].
// plan_mode: PJD_PLAN_LATENCY / PJD_PLAN_THROUGHPUT (include/pjd.h): how much stream a lane takes.
int pjd_make_plan(const pjd_image_desc *images, int n, int out_format, PjdPlan &plan, std::string &err,
                  uint32_t sub_bytes_override = 0, int plan_mode = 0);
