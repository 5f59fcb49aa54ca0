/*
 * pjd.h -- C ABI of the MI355X-native JPEG decode path (libpjd.so).
 *
 * This is the drop-in boundary for the UPMEM-DPU dispatch of
 * jeun-990806/pim-jpeg-decoder.  Every entry point names the reference
 * interface it replaces (paths relative to the reference tree).
 *
 *   reference                                         this library
 *   ------------------------------------------------  -------------------------------
 *   DpuSet::allocate + load   decoder_host.cpp:32,268   pjd_open / pjd_close
 *   decode_Huffman_data       jpeg_scanner.cpp:707      \
 *   copy("metadata_buffer")   decoder_host.cpp:276       |  pjd_batch_create + _upload
 *   copy("mcus")              decoder_host.cpp:277      /
 *   exec()                    decoder_host.cpp:292      pjd_batch_decode (+ _sync)
 *   copy(batch.mcus,"mcus")   decoder_host.cpp:308      pjd_batch_download
 *   4 x dpus()[0]->copy(ctr)  decoder_host.cpp:309-312  pjd_batch_decode_timed
 *   one picture over all DPUs decoder_host.cpp:125-149   pjd_split_decode (restart segments over GPUs)
 *   the per-DPU payload T0    decoder_dpu.c:57-58       pjd_exec_dpu_payload (literal)
 *
 * Plain pointers and sizes only; no C++ or torch types.  Thread model: one
 * submitting thread per pjd_ctx (the reference has one consumer thread,
 * decoder_host.cpp:213).  All functions return 0 on success or a negative
 * PJD_E_* code; nothing here falls back to a CPU implementation -- without a
 * usable gfx950 device pjd_open fails.
 */
#ifndef PJD_H
#define PJD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct in this header changes size or layout (pjd_image_desc gained qt_slot48 and
 * pjd_batch_info grew in version 2; version 3 added the coefficient download and pjd_split_*; version 4 the progressive scans of
 * pjd_image_desc and the exact-path figures of pjd_batch_info; version 5 pjd_batch_info::n_steps).
 * A caller built against another version must not pass its structs: check pjd_version() == PJD_VERSION after loading.     */
#define PJD_VERSION 5

/* ---- error codes (library level) ---------------------------------------- */
#define PJD_OK              0
#define PJD_E_NODEVICE     -1   /* no HIP device / not gfx950                   */
#define PJD_E_HIP          -2   /* a HIP call failed (see pjd_last_error)       */
#define PJD_E_ARG          -3   /* bad argument / descriptor outside envelope   */
#define PJD_E_NOMEM        -4
#define PJD_E_STATE        -5   /* call order violated (e.g. decode before upload) */

/* ---- per-image status: the reference's Huffman error classes ------------- *
 * (jpeg_scanner.cpp:470-514).  As in the reference (decoder_host.cpp:181 drops
 * the bool), an image with a non-zero status still has its partial picture
 * written: everything decoded before the error, grey (128) after it.          */
#define PJD_ST_OK        0
#define PJD_ST_DC_SYM    1   /* "Error - Invalid DC value (255)"                        */
#define PJD_ST_DC_LEN    2   /* "Error - DC coefficient length greater than 11"         */
#define PJD_ST_DC_BITS   3   /* "Error - Invalid DC value"                              */
#define PJD_ST_AC_SYM    4   /* "Error - Invalid AC value"  (symbol)                    */
#define PJD_ST_AC_RUN    5   /* "Error - Zero run-length exceeded block component"      */
#define PJD_ST_AC_LEN    6   /* "Error - AC coefficient length greater than 10"         */
#define PJD_ST_AC_BITS   7   /* "Error - Invalid AC value"  (value bits)                */

/* ---- output formats ------------------------------------------------------- */
#define PJD_OUT_RGB8   0   /* top-down, tightly packed R,G,B bytes: 3*W*H bytes              */
#define PJD_OUT_BMP    1   /* the complete file image bmp_writer.cpp:19-67 would emit:
                              26-byte BITMAPCOREHEADER file header, bottom-up B,G,R rows,
                              (W % 4) zero bytes after each row                               */

/* ---- descriptor flags ----------------------------------------------------- */
#define PJD_F_STANDARD_RESTART  1u  /* restart at every `restart_interval`-th MCU (ITU T.81).
                                       Default (flag clear) reproduces the reference's rule
                                       (jpeg_scanner.cpp:723), which differs -- and garbles --
                                       when luma sampling is not 1x1.                         */
#define PJD_F_FORCE_SEQUENTIAL  2u  /* decode with the one-lane exact kernel (debug/diagnosis) */
#define PJD_F_STANDARD_ZIGZAG   4u  /* zigzag slot 48 -> natural position 58 (ITU T.81) instead of
                                       the reference's 38 (common.h:16), for coefficients and for
                                       the quantiser (qt_slot48).  NOT reference-comparable: the
                                       reference has no such mode, parity for it is unpinned.   */

#define PJD_F_PROGRESSIVE       8u  /* a progressive (SOF2) frame: `scans` / `n_scans` describe its scans, `ecs` is unused.  The
                                       reference cannot decode such files -- its scanner stops at the first marker between scans
                                       (jpeg_scanner.cpp:425-430) and its progressive branches (:521-704) handle one scan only --
                                       so this mode is NOT reference-comparable, parity for it is unpinned; it exists for
                                       SURVEY 8(f) N4 and is opt-in (pjd_scan_*_ex with PJD_SCAN_PROGRESSIVE).                */

/* Huffman table as the reference's scanner holds it (jpeg.h:129-134):
 * offsets[k] = number of codes of length <= k (offsets[0] = 0).               */
typedef struct pjd_huff_table {
    uint8_t offsets[17];
    uint8_t symbols[162];
    uint8_t set;
} pjd_huff_table;

/* One scan of a progressive frame (ITU T.81 G.1; the fields the reference's progressive branches read from `Header`:
 * start_of_selection, end_of_selection, successive_approximation_high / _low, jpeg.h:160-163) with the Huffman tables in
 * force when the scan starts (tables may be redefined between scans) and its own entropy-coded bytes.                     */
typedef struct pjd_scan_desc {
    uint8_t  n_comp;                   /* components in this scan, 1..3 (AC scans: 1)                          */
    uint8_t  comp[3];                  /* their indices 0..2 (frame order)                                     */
    uint8_t  ss, se, ah, al;           /* spectral selection, successive approximation                         */
    uint32_t restart_interval;         /* DRI in force for this scan, in MCUs OF THE SCAN; 0 = none            */
    pjd_huff_table table[3];           /* per scan component: its DC table (ss == 0) or its AC table (ss > 0)   */
    const uint8_t *ecs;                /* destuffed, RSTn removed                                              */
    uint64_t ecs_len;
} pjd_scan_desc;

/* One parsed baseline JPEG -- exactly the fields of the reference `Header`
 * (jpeg.h:146-179) that its hot path consumes, plus the restart-segment
 * offsets the reference's scanner throws away.                                */
typedef struct pjd_image_desc {
    uint32_t width, height;            /* pixels                                              */
    uint8_t  num_components;           /* 1..3                                                */
    uint8_t  h_samp, v_samp;           /* luma sampling factors, each 1 or 2                  */
    uint8_t  comp_h[3], comp_v[3];     /* per component (chroma must be 1x1)                  */
    uint8_t  comp_qt[3], comp_dc[3], comp_ac[3];   /* table selectors, each 0..3              */
    uint8_t  qt_set[4];
    uint32_t qt[4][64];                /* NATURAL order as filled through the reference's
                                          zigzag_map (common.h:9-18, entry 48 = 38)           */
    pjd_huff_table dc[4], ac[4];
    uint32_t restart_interval;         /* DRI value, 0 = none                                 */
    const uint8_t *ecs;                /* entropy-coded bytes, destuffed (FF00 -> FF) and with
                                          RSTn removed == Header::huffman_data (jpeg.h:168)   */
    uint64_t ecs_len;
    const uint64_t *seg_offsets;       /* byte offset in `ecs` of each restart segment,
                                          seg_offsets[0] == 0; NULL => one segment            */
    uint32_t n_segments;
    uint32_t flags;                    /* PJD_F_*                                             */
    /* Sharding of ONE image over several devices (restart segments are
     * independent): decode only segments [shard_first_seg, +shard_n_segs);
     * shard_n_segs == 0 means "all".  The output buffer is always full-size;
     * only the MCUs of the selected segments are written.                      */
    uint32_t shard_first_seg, shard_n_segs;
    uint32_t qt_slot48[4];             /* DQT entry 48 of each table: the reference's map sends it
                                          to natural 38, where entry 52 overwrites it, so `qt`
                                          does not hold it.  Read only with PJD_F_STANDARD_ZIGZAG. */
    const pjd_scan_desc *scans;        /* PJD_F_PROGRESSIVE only: the frame's scans in file order    */
    uint32_t n_scans;
    uint32_t reserved_;
} pjd_image_desc;

typedef struct pjd_ctx pjd_ctx;
typedef struct pjd_batch pjd_batch;

/* Kernel-level timing of one decode, HIP events on the context's stream.     */
#define PJD_MAX_KERNELS 16
typedef struct pjd_timings {
    int32_t n;
    float   ms[PJD_MAX_KERNELS];
    char    name[PJD_MAX_KERNELS][32];
    float   total_ms;                  /* first-start .. last-stop                            */
} pjd_timings;

typedef struct pjd_batch_info {
    int32_t  n_images;
    uint64_t pixels;                   /* sum of width*height                                 */
    uint64_t ecs_bytes;                /* sum of ecs_len                                      */
    uint64_t out_bytes;                /* sum of output sizes                                 */
    uint64_t coef_bytes;               /* lane streams + transposed bitstream words + dense scratch in HBM */
    uint64_t n_data_units;
    uint64_t n_subsequences;           /* Huffman decode lanes                                */
    uint64_t device_bytes;             /* everything this batch holds in HBM                  */
    int32_t  n_sequential;             /* images routed to the exact one-lane kernel up front */
    int32_t  n_fallback;               /* images re-decoded by it after the last decode       */
    uint64_t n_huff_workgroups;        /* Huffman workgroups (up to 2 waves of 64 lanes, one table set) */
    /* diagnostics of the last decode: self-synchronisation effort                           */
    uint64_t sync_rounds;              /* re-sync rounds summed over workgroups               */
    uint64_t sync_lane_passes;         /* lanes that re-decoded their subsequence, summed     */
    uint64_t fix_rounds, fix_lane_passes;   /* the same for the boundary-stitch stage          */
    uint32_t sub_bytes;                /* bytes of bitstream per Huffman lane chosen for this batch */
    uint32_t n_table_sets;             /* distinct Huffman table sets (images with identical tables share one) */
    uint64_t n_huff_waves;
    uint64_t n_entries;                /* 16-bit coefficient entries (= Huffman symbols) emitted by the last decode */
    float    exact_fallback_ms;        /* GPU time (HIP events) the exact one-lane kernel + its back end took to re-decode the
                                          pictures the parallel decoder could not resolve in the last decode; 0 if none.  The exact
                                          kernel is a single dependent chain per picture: about 10 MPix/s per picture */
    uint32_t n_entropy_errors;         /* pictures of the last decode with an entropy-coding error (status != 0) that the parallel
                                          decoder settled itself                                                               */
    uint64_t flag_waves[8];            /* waves that reported something unresolved in the last decode, by reason:
                                          0 invalid symbol, 1 irregular segment end / phase, 2 re-sync did not converge,
                                          3 wave boundary did not stitch, 4 wait timed out, 5 lane output overflow,
                                          6 write pass did not reproduce the synchronised state                     */
    uint32_t huff_lds_bytes;           /* LDS of one entropy-decode workgroup: the largest table set of the batch + wave areas   */
    uint32_t plan_mode;                /* PJD_PLAN_* the batch was planned with                                                    */
    uint64_t walks, walk_lanes;        /* last decode: re-sync rounds that a wave finished as a cooperative walk (few lanes left: the
                                          whole wave decodes one lane's subsequence several times faster), and the lanes walked */
    uint64_t n_steps;                  /* last decode: steps of the write pass (a step emits one entry, or the two entries of a symbol
                                          pair that one table lookup yields): n_entries / n_steps = symbols per step               */
    uint32_t lane_fill_x1024;          /* last decode: the fullest lane region, slots written x 1024 / its capacity (capacities are a
                                          bound computed from the picture's Huffman tables: never above 1024)                    */
    uint32_t reserved2_;
} pjd_batch_info;

/* ---- context --------------------------------------------------------------- */
int  pjd_version(void);
int  pjd_open(int device_ordinal, pjd_ctx **out);
void pjd_close(pjd_ctx *ctx);
/* How batches created on this context from now on are planned.  The entropy decoder cuts every bitstream into lanes; short lanes
 * finish ONE batch sooner (every pass of every chain is shorter), long lanes cost less work per byte (fewer re-synchronisation
 * passes), which is what counts when several batches are decoded at once (pjd_pipeline, a serving loop).  Measured on the default
 * workload of bench.py: latency plan 2.15 ms for a batch alone / 121 GPix/s with four in flight, throughput plan 2.50 ms / 130.
 * Pictures are identical either way.  Default: PJD_PLAN_LATENCY (PJD_PLAN_MODE=throughput in the environment changes it).      */
#define PJD_PLAN_LATENCY     0
#define PJD_PLAN_THROUGHPUT  1
int  pjd_set_plan_mode(pjd_ctx *ctx, int mode);
const char *pjd_last_error(pjd_ctx *ctx);   /* text of the last failure on this context       */
const char *pjd_status_string(int status);  /* the reference's message for a PJD_ST_* value   */
void *pjd_stream(pjd_ctx *ctx);             /* the hipStream_t all work is issued on          */

/* ---- batch life cycle ------------------------------------------------------- *
 * create : plan + allocate (host pinned staging and HBM); copies the descriptors,
 *          the caller's ecs buffers may be released after pjd_batch_upload returns.
 * upload : H2D of bitstreams, tables and work lists (asynchronous on the stream).
 * decode : enqueue the kernels; inputs and outputs stay resident in HBM.
 * download: D2H of pictures and statuses into caller memory, synchronises.
 * A batch can be decoded any number of times after one upload.                 */
int  pjd_batch_create(pjd_ctx *ctx, const pjd_image_desc *images, int n_images,
                      int out_format, pjd_batch **out);
int  pjd_batch_upload(pjd_batch *b);
int  pjd_batch_decode(pjd_batch *b);
int  pjd_batch_decode_timed(pjd_batch *b, pjd_timings *t);   /* same work, events per kernel  */
int  pjd_batch_capture(pjd_batch *b);       /* record the decode as a hipGraph; later
                                               pjd_batch_decode calls replay it               */
int  pjd_batch_sync(pjd_batch *b);
int  pjd_batch_download(pjd_batch *b, uint8_t *const *out, int32_t *status);
/* All pictures in ONE device-to-host copy: `host` receives the batch's output buffer as it lies
 * in HBM (picture i at pjd_batch_output_offset(b, i), 256-byte aligned, pjd_batch_packed_size(b)
 * bytes in total).  Give it memory from pjd_host_alloc for a full-rate PCIe transfer.  This is the
 * copy(batch.mcus, "mcus") of decoder_host.cpp:308 for a whole batch.                           */
int  pjd_batch_download_packed(pjd_batch *b, uint8_t *host, uint64_t capacity, int32_t *status);
uint64_t pjd_batch_packed_size(pjd_batch *b);
uint64_t pjd_batch_output_offset(pjd_batch *b, int image);
int  pjd_batch_get_info(pjd_batch *b, pjd_batch_info *info);
uint64_t pjd_batch_output_size(pjd_batch *b, int image);
void *pjd_batch_device_output(pjd_batch *b, int image);      /* device pointer (HBM)          */
void *pjd_batch_device_status(pjd_batch *b);                 /* int32[n_images] in HBM        */
void pjd_batch_destroy(pjd_batch *b);

/* ---- stage-level parity (debug; not on the product path) ------------------------------------ *
 * The coefficients of one image as the entropy decoder left them, in the layout of the reference's
 * MCU_buffer after decode_Huffman_data (jpeg_scanner.cpp:733-741): n_dpus x int16[19200], index
 * blk16 * 768 + component * 256 + position * 64 + natural index (through the reference's zigzag_map,
 * common.h:9-18), absolute DC values, zero where nothing was decoded.  n_dpus x 19200 =
 * pjd_coefficients_size(...) as decoder_host.cpp:125-128 sizes it.  Works after pjd_batch_decode for
 * images of either path (lane streams / exact kernel); an image the parallel decoder handed to the
 * exact kernel is decoded once more for this call.  Synchronises the stream.                          */
uint64_t pjd_coefficients_size(uint32_t width, uint32_t height, uint8_t h_samp, uint8_t v_samp);   /* in int16 */
int  pjd_batch_download_coefficients(pjd_batch *b, int image, int16_t *out, uint64_t capacity_int16);

/* One call: create + upload + decode + download + destroy.                    */
int  pjd_decode_batch(pjd_ctx *ctx, const pjd_image_desc *images, int n_images,
                      int out_format, uint8_t *const *out, int32_t *status);

/* ---- one picture over several devices (BASELINE config 5) -------------------------------------- *
 * Replaces what the reference does with every picture -- spread it over all allocated DPUs, copy the metadata
 * to each and collect the samples (decoder_host.cpp:125-149,225,262-312) -- for a picture with restart
 * intervals: device k of `devices` decodes a contiguous range of restart segments (pjd_split_plan), the
 * descriptor (tables + segment offsets, ~20 KB) is broadcast from the first device's HBM with one
 * ncclBroadcast (RCCL over xGMI; loaded on first use; plain copies if RCCL is missing), every device
 * uploads only its slice of `desc->ecs`, and the rows come back into `out` (pjd_output_size bytes).
 * One host thread per device inside the call.  A picture that cannot be split (no DRI; subsampled luma
 * under the reference's restart rule) is decoded by devices[0] alone; so is, a second time, a picture whose
 * entropy decode reports an error, so that status and partial picture are the reference's.               */
#define PJD_SPLIT_MAX_DEVICES 16
typedef struct pjd_split_stats {
    double   wall_s, broadcast_s, upload_s, exec_s, download_s;   /* the three stages: slowest rank          */
    uint64_t blob_bytes;               /* size of the broadcast descriptor                                  */
    uint64_t ecs_bytes[PJD_SPLIT_MAX_DEVICES];   /* entropy-coded bytes each rank uploaded                  */
    uint32_t n_segments, n_ranks;      /* restart segments of the picture; ranks that had work              */
    uint32_t n_exact;                  /* shards the exact one-lane kernel decoded                          */
    int32_t  rccl_used;                /* 1: the descriptor travelled by ncclBroadcast                      */
    int32_t  redone_whole;             /* 1: decoded again on one device after an entropy-coding error      */
} pjd_split_stats;
int  pjd_split_decode(const pjd_image_desc *desc, const int32_t *devices, int n_devices, int out_format,
                      uint8_t *out, uint64_t capacity, int32_t *status, pjd_split_stats *stats);
/* The range arithmetic on its own (no device needed): rank `rank` of `world` takes segments
 * [shard_first_seg, +shard_n_segs) = bytes [*byte_lo, *byte_hi) of desc->ecs = MCUs [*first_mcu, *last_mcu).
 * `shard` (optional, with `seg_scratch`: desc->n_segments words that must outlive it) becomes the descriptor
 * that rank passes to pjd_batch_create: its ecs points at the slice, its offsets are relative to the slice.
 * Returns PJD_OK, 1 if the rank has no segment (more ranks than segments), PJD_E_ARG.                      */
int  pjd_split_plan(const pjd_image_desc *desc, int world, int rank, pjd_image_desc *shard, uint64_t *seg_scratch,
                    uint64_t *byte_lo, uint64_t *byte_hi, uint32_t *first_mcu, uint32_t *last_mcu);
void pjd_split_release(void);          /* drops the cached RCCL communicators                               */
/* Diagnostic: the RCCL leg of pjd_split_decode on ONE device -- librccl is loaded, a one-rank communicator created, `bytes` bytes of
 * a pattern broadcast (root 0) from one HBM buffer into another on the library's stream, the result compared, the communicator
 * destroyed.  Exercises exactly the calls pjd_split_decode makes (ncclCommInitAll / ncclGroupStart / ncclBroadcast / ncclGroupEnd /
 * ncclCommDestroy) on boxes that have a single GPU.  0: ok; PJD_E_STATE: librccl cannot be loaded; PJD_E_HIP: a call failed.          */
int  pjd_split_rccl_selftest(int device_ordinal, uint64_t bytes);

/* ---- the literal DPU contract ---------------------------------------------- *
 * metadata: n_dpus x u32[276]   (decoder_host.cpp:156-178 index map)
 * mcus    : n_dpus x i16[19200] coefficients in, R/G/B samples out, both in the
 *           reference's blk16 layout (decoder_dpu.c:134-156,361-390).
 * Replaces copy/copy/exec/copy of decoder_host.cpp:276-308 one for one.        */
int  pjd_exec_dpu_payload(pjd_ctx *ctx, const uint32_t *metadata, int16_t *mcus, int n_dpus);

/* Host-only planning: what a batch of these images would occupy (no device needed).
 * Fills everything in `info` except device_bytes / n_fallback.                   */
int  pjd_plan_info(const pjd_image_desc *images, int n_images, int out_format, pjd_batch_info *info);
/* Debug, host-only: the bound behind the size of a picture's lane streams -- the fewest bits of bitstream per step of the entropy
 * decoder's write pass (one symbol, or the pair one table lookup yields) that ANY stream coded with this picture's Huffman tables can
 * sustain, x 256 (the minimum mean weight of a cycle of the step graph, pim-jpeg-decoder_amd/csrc/pjd_plan.cpp).  PJD_E_ARG for a
 * picture that does not take the parallel decoder.  tests/test_planner_bound.py recomputes it another way.                        */
int  pjd_plan_step_bits(const pjd_image_desc *image, uint32_t *step_bits_x256);

/* Page-locked host memory (hipHostMalloc) for pjd_batch_download_packed; NULL on failure.       */
void *pjd_host_alloc(uint64_t bytes);
void pjd_host_free(void *p);

/* Size in bytes of one picture in a given output format.                      */
uint64_t pjd_output_size(uint32_t width, uint32_t height, int out_format);

#ifdef __cplusplus
}
#endif
#endif /* PJD_H */
