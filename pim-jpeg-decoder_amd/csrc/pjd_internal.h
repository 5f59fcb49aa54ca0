// pjd_internal.h -- data layout shared by the host planner and the gfx950 kernels.
//
// Everything the kernels read lives in HBM in these plain structs / arrays; the
// host planner (pjd_plan.cpp) fills them from pjd_image_desc, the API
// (pjd_api.hip) uploads them once per batch.
#pragma once
#include <stdint.h>

// ---- tunables -----------------------------------------------------------------
#ifndef PJD_SUB_BYTES_MIN
#define PJD_SUB_BYTES_MIN  128      // Huffman subsequence (bytes of bitstream per decode lane): chosen per batch
#endif
#define PJD_SUB_BYTES_MAX  1024     //   by the planner (multiple of 64 in this range), see pjd_plan.cpp
#define PJD_HUFF_LANES     64       // subsequences per wave: lanes exchange states by shuffles, no barriers
#ifndef PJD_HUFF_WAVES
#define PJD_HUFF_WAVES     2        // waves per Huffman workgroup; they share one table set in LDS
#endif
#define PJD_HUFF_THREADS   (PJD_HUFF_LANES * PJD_HUFF_WAVES)
#ifndef PJD_NCHK
#define PJD_NCHK           4        // checkpoints per subsequence (trajectory states a re-sync bridge can merge into); a power of two
#endif
#define PJD_STAGE_ENTRIES  16       // slots a lane stages in LDS between flushes: one group (a head + 7 step words)
#define PJD_CHK_BYTES      (2 * PJD_NCHK * 64 * 4)
#define PJD_STAGE_BYTES    (PJD_STAGE_ENTRIES * 2 * 64)
// per wave: checkpoints (sync passes) / entry staging (write pass).  LDS sets the kernel's occupancy; measured in
// profiles/r02_occupancy.md: more LDS per workgroup costs throughput at once, and 2 waves per workgroup with 2 KB wave areas
// (4 checkpoints, 16 staged entries) measured best with batches in flight (+1 % on the default input, +5 % on the lighter one
// against 4 waves with 4 KB areas), equal within noise for one batch at a time.
#define PJD_WAVE_LDS       (PJD_CHK_BYTES > PJD_STAGE_BYTES ? PJD_CHK_BYTES : PJD_STAGE_BYTES)
// per wave: the phase table -- one 16-byte record per data unit of the MCU: what the NEXT unit decodes with (table offsets,
// its own record's address, the DC-sum selectors of its component); a unit's completion is one LDS read instead of ~9 VALU
#define PJD_PHASE_LDS      256      // 16 records (an MCU has at most 4 + 2 = 6 units; sampling 2x2 with three components)
#ifndef PJD_DIRECT_ECS
#define PJD_DIRECT_ECS     0        // 1: the Huffman lanes read the batch's bitstream directly (unaligned dwords, byte-swapped) instead of the
#endif                              //    transposed copy pjd_k_lane_words makes (experiment, profiles/r03_experiments.md)
#ifndef PJD_TAIL_PRIO
#define PJD_TAIL_PRIO      1        // waves in their second and later re-sync rounds raise their issue priority (s_setprio), see wave_rounds
#endif
#ifndef PJD_IDCT_PRIO
#define PJD_IDCT_PRIO      0        // issue priority of the back end's waves (0..3)
#endif
// The cooperative walker (pjd_k_huffman.hip, walk_lane): a re-sync round with at most PjdDevImage::walk_max active lanes is finished
// by the whole wave taking the lanes one after the other.  A walk has a start-up cost per lane (a window of the bitstream is fetched),
// a round costs the same whatever the number of lanes, so the threshold grows with the subsequence: 3 lanes per 512 bytes (6 at 1024),
// and at least PJD_WALK_DENSE where chains are long -- subsequences shorter than PJD_WALK_DENSE_MCUS MCUs' worth of this picture's
// stream: lanes rarely merge inside their own subsequence.  Measured in profiles/r03_experiments.md (8 then) and again in round 4 after
// the second landing pad per lane made rounds shorter: 4 / 6 / 8 / 12 / 16 lanes -> 2.56 / 2.54 / 2.62 / 2.88 / 3.11 ms for a batch alone,
// no difference with batches in flight.  PJD_WALK_MAX in the environment overrides it for every picture of a batch (0: never walk).
#define PJD_WALK_LANES(sub_bytes)  (((sub_bytes) * 3u) >> 9)
#define PJD_WALK_DENSE     6
#define PJD_WALK_DENSE_MCUS 6
#define PJD_LUT_BITS       9        // first-level Huffman LUT width
#define PJD_L2_BITS        (16 - PJD_LUT_BITS)    // a second-level table is indexed by the bits after the prefix
#define PJD_L1_BYTES       (4 << PJD_LUT_BITS)   // one first-level table: 512 x u32 (low half: the symbol; high half: the symbol PAIR, below)
#define PJD_LUT_LDS_MAX    (6 * PJD_L1_BYTES + 8192)  // decode tables of one table set in LDS; larger -> exact kernel
#define PJD_MAX_TABLES     6        // distinct Huffman tables one image can reference (3 DC + 3 AC)
#ifndef PJD_GENS
#define PJD_GENS           6        // generations of a wave's published exit state (wave_gen): 0 = speculative, 1 = after its own re-sync rounds, g + 1 = after
#endif                              // comparing its entry with generation g of its predecessor (and re-bridging if that differs); the last one is only checked.
                                    // A chain of non-merging lanes that crosses k wave boundaries needs k + 2 generations; never a chain over the whole image.
#define PJD_SYNC_MAX_ITERS 72       // re-sync rounds per wave before giving up (a non-merging chain moves one lane per round)
#define PJD_DC_BLOCK       256      // lanes per DC-prediction scan block
#ifndef PJD_IDCT_THREADS
#define PJD_IDCT_THREADS   256
#endif
#ifndef PJD_IDCT_MAX_DU
#define PJD_IDCT_MAX_DU    96       // data units staged in LDS per IDCT workgroup (<= PJD_IDCT_THREADS); with the group parser (round 3): 72 / 90 / 96 -> 0.59 / 0.53 / 0.525 ms on cfg3
#endif
#define PJD_COEF_SENTINEL  (-32768) // "slot 52 was visited with an explicit 0" (zigzag quirk)

// Bitstream words of one wave, transposed: row k holds big-endian word k of each of its 64 lanes, counted from
// the lane's own first byte.  A lane reads at most 31 bits past its subsequence, holds three words and keeps two more in flight.
#define PJD_WORD_ROWS(sub_bytes)  ((sub_bytes) / 4 + 8)
// Slots (2 bytes each) of a lane's region.  The write pass emits one 32-bit STEP word per decode step -- one symbol, or the PAIR of
// symbols one table lookup yields (below).  How many steps a lane of `sub_bytes` bytes can take follows from the PICTURE's table set:
// the planner computes the fewest bits per step any stream can sustain with it (Annex-K tables: 3.5 -- a chroma unit of DC 0, one
// +-1 coefficient and an EOB is the pair DC + coefficient, then the EOB: 7 bits in two steps; tables fitted to a dense picture: 4-5;
// without the DC pairs it was 2).
// A region is a sequence of 32-byte GROUPS: one head word + PJD_GROUP_STEPS (7) step words (below), a whole number of groups.
// The write pass still checks the bound (PJD_FLAG_OVERFLOW) before every flush.
#define PJD_GROUP          16       // slots per group = what the write pass stages between two flushes (8 dwords)
#define PJD_GROUP_STEPS    (PJD_GROUP / 2 - 1)
// step_bits_x256: fewest bits of stream per step the picture's table set can be made to sustain, x 256 (pjd_plan.cpp, min_step_bits_x256:
// the minimum mean weight mu of a cycle of the step graph, 17 nodes).  A lane's steps are ONE walk in that graph: cycles (each of mean
// >= mu) and a simple path of at most 16 edges of at least one bit each, so n steps consume at least n * mu - 16 * (mu - 1) bits: n <
// bits / mu + 16.  The bits are the lane's own (steps START inside it) plus what its last step reads past its end (two symbols: < 27 bits,
// at most 6 steps' worth at mu >= 4... 27 at mu = 1), and the pair the lane's end breaks is one more step: 16 + 27 + 1, rounded up.
#define PJD_LANE_SLACK_STEPS 48u
#define PJD_LANE_CAP(sub_bytes, step_bits_x256)   ((((8u * (sub_bytes) * 256u + (step_bits_x256) - 1) / (step_bits_x256) + PJD_LANE_SLACK_STEPS + PJD_GROUP_STEPS - 1) / PJD_GROUP_STEPS + 1u) * PJD_GROUP)

// ---- coefficient entries (lane streams) --------------------------------------------------------
// The write pass turns every decoded Huffman symbol into one 16-bit entry; a STEP word holds the one or two entries of a step:
// low half = entry A, high half = entry B (the second symbol of a pair) or PJD_ENT_NONE.
//   DC symbol            the step's low half IS the DC difference as int16 (12 significant bits); the high half is the unit's first
//                        AC symbol where the table holds that pair (short DC symbols: a flat unit, DC + EOB, is ONE step), else
//                        PJD_ENT_NONE.  Which steps hold a DC difference follows from the position: the first entry of every unit.
//   AC run/size symbol   bits 15..5 = value (11-bit two's complement; 0 for a size-0 symbol such as ZRL), bits 4..0 = run + 1
//                        (1..16): the coefficient lands on slot (next free slot + run)
//   EOB                  0x0000: "run + 1" = 0, value 0 -- completes the unit, stores nothing
//   PJD_ENT_NONE         0x001f ("run + 1" = 31): no entry (only in the high half)
// A unit is [DC][AC ...] up to and including an EOB or the entry that lands on slot 63 (next free slot = 64).  A size-0 symbol
// stores an explicit 0 (reference src/jpeg_scanner.cpp:516-517), which matters at slot 52 only (zigzag quirk).
// Steps are kept in GROUPS of 8 dwords = 32 bytes, the unit the write pass flushes: dword 0 is the group's HEAD, `[units completed
// in this lane before the group's first entry : 24][zigzag slot that entry fills from : 8]` (slot 0: the entry is a unit's DC
// difference; else the AC coefficient lands on slot + run), dwords 1..7 hold 7 steps.  With the head a back-end thread parses its
// group without looking at anything before it (pjd_k_idct_colour_lanes); the head travels in the same 32-byte store as its steps.
// Positions in a lane's region (PjdDevLaneInfo::n_ent, PjdDevMark::ent_off) count 16-bit SLOTS, heads included (always even).
// Round 4: the write pass used to emit one entry per step (one symbol); with the pair tables it takes two symbols in 60 % of
// its steps on dense streams, and a step's two entries travel in one LDS word whatever the other lanes of the wave decoded.
#define PJD_ENT_NONE       0x001fu
#define PJD_ENT_EOB        0x0000u

// ---- image flags (device side) -----------------------------------------------------
#define PJD_IF_STANDARD_RESTART 1u  // restart every RI-th MCU; else the reference's (y*Wr+x)%RI rule
#define PJD_IF_SEQUENTIAL       2u  // routed to the exact one-lane kernel up front
#define PJD_IF_BMP              4u  // output is a BMP file image (else tight RGB8)
#define PJD_IF_STANDARD_ZIGZAG  8u  // zigzag slot 48 -> natural 58, no slot-52 override (PJD_F_STANDARD_ZIGZAG)
#define PJD_IF_PROGRESSIVE      32u // a progressive frame: decoded scan by scan into the dense scratch by pjd_k_progressive (PJD_F_PROGRESSIVE)
#define PJD_IF_ENDS_STREAM      16u // the last restart segment this image (or shard) decodes is the last of its bitstream: running out of
                                    // bits there is the reference's end-of-data error, handled by the parallel decoder itself

// status word per image: low 8 bits = PJD_ST_* class, bit 8 = "fast path gave up, needs exact kernel"
#define PJD_STW_NEEDS_EXACT 0x100

// What the parallel decoder found wrong in an image, kept per image and ordered by POSITION in the stream (PjdDevBatch::imstate):
//   err_key   the first entropy-coding error of the true decode (the reference stops there, jpeg_scanner.cpp:470-514; everything
//             decoded before it is kept, the rest of the picture stays undecoded): minimum over the lanes that met one of
//             (bit position of the offending symbol << 32) | (index of its data unit << 4) | (PJD_ST_* class << 1 ... see below);
//             the write pass finds it by itself -- bad symbol, bad size, run past slot 63, end of data -- so such pictures
//             need no second decode
//   flag_pos  bit position (start of the lane, or of the wave's first lane) of the earliest thing the decoder could NOT resolve
//             (PJD_FLAG_*): only that, and only if it lies before the first error, sends the picture to the exact kernel
// layout of err_key's low word: bits 31..4 data unit, bits 3..1 PJD_ST_* class, bit 0 = the error is in the unit's DC symbol
struct PjdDevImState {
    unsigned long long err_key;        // ~0: none
    uint32_t flag_pos;                 // ~0: none
    uint32_t waves_done;               // Huffman waves of the picture that have finished their write pass (the "pull" back end, below)
};

// The PULL back end (round 4; the form a decode takes on an otherwise idle device).  With one launch after the other the back end
// starts when the LAST wave of the entropy decoder has finished although most pictures were complete long before.  Here the last
// wave of a PICTURE to finish (waves_done reaches the picture's wave count) settles the picture itself -- verdict, DC predictors at
// its lane starts (what pjd_k_image_verdict / pjd_k_lane_dc_* do) -- and appends the picture's back-end ranges to ready_list; a
// back-end launch that runs BESIDE the entropy decoder (second stream) takes range k of the list in workgroup k, waiting for the
// entry to appear (bounded: a workgroup that gives up leaves its range to the sweep -- an ordinary back-end launch that follows
// and skips the ranges marked done).  All words live in the per-decode operation state (zeroed by pjd_k_reset).
#define PJD_PULL_SPIN_LIMIT (1u << 12)     // x s_sleep(64): a few milliseconds

// why the parallel decoder flagged an image (PjdDevBatch::stats[PJD_STAT_FLAG0 + reason], counted per wave)
enum {
    PJD_FLAG_SYMBOL = 0,     // invalid code, size or run outside the baseline limits
    PJD_FLAG_SEGMENT,        // a restart segment ends early / late / off a byte boundary, phase or count mismatch
    PJD_FLAG_NOSYNC,         // re-sync rounds did not converge within PJD_SYNC_MAX_ITERS
    PJD_FLAG_STITCH,         // the entry a wave used is not what its predecessor finally produced
    PJD_FLAG_TIMEOUT,        // a bounded wait on another wave expired, or that wave was poisoned
    PJD_FLAG_OVERFLOW,       // a lane needed more than PJD_LANE_CAP entries
    PJD_FLAG_VERIFY,         // the write pass did not reproduce the synchronised exit state / unit count
    PJD_FLAG_REASONS
};
#define PJD_STAT_FLAG0 4
#define PJD_STAT_ENTRIES 11  // PjdDevBatch::stats[]: entries (= Huffman symbols) the lanes emitted in this decode
#define PJD_STAT_STEPS   14  // step words they took for it (a step holds one symbol or a pair)
#define PJD_STAT_FILL    15  // fullest lane region of the decode: slots written x 1024 / PJD_LANE_CAP of its picture (atomic max)
#define PJD_STAT_WALKS   12  // cooperative walks (a wave taking over its few remaining active lanes), 13: lanes walked in them

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
    uint64_t dense_base;               // first data unit of this image in the DENSE scratch (exact-kernel path only)
    uint32_t n_du;
    uint32_t image_index;
    uint32_t out_stride;               // bytes per output row (BMP: 3W + W%4, RGB8: 3W)
    uint64_t out_off;                  // into the batch output buffer (256-byte aligned)
    uint32_t seg_base, n_seg;          // into PjdDevSegment[]
    uint32_t lane_base, n_lane;        // into PjdDevSub[] (global lane index)
    uint32_t hwave_base, n_hwave;      // Huffman waves of this image (global wave index)
    uint32_t iwg_base, n_iwg;          // IDCT workgroups (= coefficient ranges) of this image
    uint32_t idct_mcus;                // MCUs per IDCT workgroup
    uint32_t first_mcu, last_mcu;      // MCU range this shard decodes: [first_mcu, last_mcu)
    uint32_t tset;                     // table set (PjdDevTset) of this image
    uint32_t sub_bytes;                // subsequence size of THIS image (<= the batch's, which sizes word rows and lane regions)
    uint8_t  tbl_slot[3][2];           // [component][0=DC,1=AC] -> table slot 0..n_tables-1 of the set
    uint8_t  walk_max;                 // re-sync rounds with at most this many active lanes are walked cooperatively (0: never)
    uint8_t  pad_;
    uint32_t pscan_base, n_pscan;      // progressive frames: their scans in PjdDevBatch::pscans
    uint32_t lane_cap;                 // slots of one lane region of THIS image: PJD_LANE_CAP(sub_bytes, min symbol bits of its table set)
    uint32_t pad2_;
    uint64_t ent_base;                 // first slot of lane `lane_base` in PjdDevBatch::ent; lane q of the image owns [ent_base + (q - lane_base) * lane_cap, + lane_cap)
};

// raw Huffman table as shipped by the host (reference HuffmanTable, jpeg.h:129-134)
struct PjdDevHuffRaw {
    uint8_t offsets[17];
    uint8_t symbols[162];
    uint8_t is_ac;
};                                     // 180 bytes

// One scan of a progressive frame (pjd_scan_desc): what it refines, with which tables, and where its bytes lie
struct PjdDevScan {
    uint64_t ecs_off;                  // into the batch bitstream buffer
    uint32_t ecs_len;
    uint32_t restart_interval;         // in MCUs of THIS scan
    uint8_t  n_comp, comp[3];
    uint8_t  ss, se, ah, al;
    PjdDevHuffRaw table[3];            // per scan component: DC table (ss == 0) or AC table
};

// One table set = the deduplicated Huffman tables of an image; images with identical sets share one (a batch of
// files written with the Annex-K tables has a single set), and so can the waves of one Huffman workgroup.
// Decode-ready form, built on the device by pjd_k_build_tables, one blob per set:
//   [table 0 L1][table 1 L1]...[table n-1 L1][second-level regions, 128 u16 per long prefix]
// L1 is indexed by the next PJD_LUT_BITS (9) bits and holds 32-bit entries; the LOW half describes the symbol that starts there --
// every field the per-symbol loops need comes out with one AND or one bit-field extract, and the zigzag bookkeeping is ONE
// subtraction (see PJD_LUT_ADV):
//   bits  4..0  bits consumed by the symbol (code length + value bits), 1..27; 0 marks a pointer entry (below)
//   bits 10..5  "advance": run + 1 for an AC run/size symbol, 1 for a DC symbol, 32 for an EOB
//   bit  11     EOB (AC tables only).  Bits 11..5 read as ONE 7-bit number are the slots the symbol uses up -- 96 for an EOB,
//               more than any unit has left -- so "63 - slot" minus that number going negative is "the unit is complete";
//               a run/size symbol that lands past slot 63 (an error in the reference, jpeg_scanner.cpp:500) leaves -16..-2 there,
//               an EOB -96..-34, a unit that ends exactly on slot 63 leaves -1: the write pass tells them apart with one minimum.
//               The low five bits of that number are the "run + 1" field of the symbol's lane-stream entry (0 for an EOB)
//   bits 15..12 value bits (size) 0..11, or an invalid symbol (the code alone is consumed):
//               14 = the reference's "symbol 0xFF": no code starts with these bits (then 16 bits are consumed, as its
//                    get_next_symbol does), or the table really holds the symbol 0xFF (jpeg_scanner.cpp:470,490)
//               15 = a DC size > 11 / an AC size > 10 (jpeg_scanner.cpp:474,506)
//   pointer entry (bits 4..0 == 0): codes with this 9-bit prefix are longer than 9 bits; bits 15..5 = u16 index (relative to the
//               blob) / 128 of the prefix's 128-entry second-level table (u16 entries of the form above with code lengths 10..16),
//               indexed by the following 7 bits.
// The HIGH half of an L1 entry describes the PAIR "this symbol and the one after it" where the 9 bits hold the first symbol whole
// (code and value bits), it is a valid run/size symbol (not an EOB) or a valid DC symbol, and the rest of the 9 bits determine the next
// code (any valid AC symbol, an EOB too; its value bits may lie outside) -- for a DC table: a code of the AC table its components
// decode with (PjdDevTset::pair_ac):
//   bits 20..16 bits consumed by both symbols (2..27+; the symbol's own: no pair here)        bits 27..21 slots both use up (advance 1 + advance 2)
//   bits 31..28 value bits (size) of the second symbol (its value is the last `size` of the bits both consume)
// Every pass takes a pair in ONE step when the first symbol neither completes the unit nor reaches the next checkpoint /
// subsequence end: dense streams average 5 bits per symbol, 60 % of the steps there are pairs (profiles/r03_experiments.md).
// Canonical codes keep all long codes in one contiguous range of prefixes [p0, p1), so the second level
// costs 256 bytes per long prefix.
struct PjdDevTset {
    uint32_t lut_off16;                // blob offset in PjdDevBatch::luts, 16-byte units
    uint32_t lut_bytes;                // multiple of 16
    uint32_t n_tables;
    uint16_t l2_off[PJD_MAX_TABLES];   // second-level region of table k: first u16 index, relative to the blob
    uint16_t l2_p0[PJD_MAX_TABLES];    // 9-bit prefixes [p0, p1) hold codes longer than PJD_LUT_BITS
    uint16_t l2_p1[PJD_MAX_TABLES];
    uint8_t  pair_ac[PJD_MAX_TABLES];  // of a DC table: the AC table (slot) every component that uses it decodes with -- its entries hold the pair
                                       // "DC symbol + the unit's first AC symbol" -- or 0xff (none: components disagree)
    uint8_t  pad_[2];
};
#define PJD_LUT_USED(e)  ((e) & 31u)
#define PJD_LUT_ADV(e)   (((e) >> 5) & 127u)      // run + 1; 96 for an EOB
#define PJD_LUT_SIZE(e)  (((e) >> 12) & 15u)
#define PJD_LUT_PAIR_USED(e)  (((e) >> 16) & 31u)  // == PJD_LUT_USED(e): no pair (the pair field then repeats the symbol's used / advance)
#define PJD_LUT_PAIR_ADV(e)   (((e) >> 21) & 127u)
#define PJD_LUT_PAIR_SIZE2(e) ((e) >> 28)
#define PJD_LUT_EOB      0x0800u
#define PJD_LUT_BADSYM   14u
#define PJD_LUT_BADLEN   15u
#define PJD_LUT_ENTRY(used, adv, eob, size)  ((used) | ((adv) << 5) | ((eob) ? PJD_LUT_EOB : 0u) | ((size) << 12))

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

struct PjdDevHuffWave {                // one Huffman wave: up to 64 consecutive lanes of one image
    uint32_t image;
    uint32_t first_lane;               // global lane index
    uint32_t n_lanes;                  // 1..64
    uint32_t pad_;
};

struct PjdDevHuffWg {                  // one Huffman workgroup: up to PJD_HUFF_WAVES consecutive waves of one table set
    uint32_t first_wave;
    uint32_t n_waves;
    uint32_t tset;
    uint32_t pad_;
};

struct PjdDevIdctWg {                  // one IDCT/colour workgroup = one coefficient range
    uint32_t image;
    uint32_t first_mcu;
    uint32_t n_mcu;
    uint32_t pad_;
};

// One picture GROUP: pictures of similar stream density, decoded by its own chain of launches (entropy decode -> DC predictors ->
// back end) beside the other groups' chains, so that the back end of the light pictures runs while the dense pictures' chains of
// re-sync rounds are still finishing (round 4: one batch alone was entropy-decode tail + the whole back end, one after the other).
#define PJD_MAX_GROUPS 8
struct PjdDevGroup {
    uint32_t hwg_first, hwg_count;     // its Huffman workgroups: a range of PjdDevBatch::hwgs (start order: densest pictures first)
    uint32_t img_first, img_count;     // its pictures: a range of PjdDevBatch::group_images
    uint32_t iwg_first, iwg_count;     // its back-end workgroups: a range of PjdDevBatch::iwg_order
    uint32_t pad_[2];
};

// what a Huffman lane leaves behind for the back end (written at the end of its write pass)
struct PjdDevLaneInfo {
    uint32_t n_ent;                    // slots used in the lane's region (group heads included; two per step word)
    uint32_t first_du;                 // bits 27..0: image-relative index of the data unit the lane's first entry belongs to (a unit may span lanes);
                                       // bit 31: the lane starts a restart segment (DC predictors are zero there)
    uint16_t dc_sum[3];                // sum of the DC differences decoded in this lane, per component (mod 2^16)
    uint16_t pad_;
};
#define PJD_LANE_SEG_FIRST 0x80000000u

// DC predictors at the start of a lane, from the scan over PjdDevLaneInfo::dc_sum (pjd_k_lane_dc_*)
struct PjdDevLaneDc {
    uint16_t dc_in[3];                 // relative to the start of the lane's scan block, or absolute if `abs`
    uint16_t abs;                      // 1: a segment head lies between the block start and this lane
};

// where an IDCT workgroup's first data unit starts in the lane streams (written by the lane that decodes
// that unit's DC symbol)
struct PjdDevMark {
    uint32_t lane;                     // global lane index
    uint32_t ent_off;                  // entry index inside the lane's region
    uint16_t acc[3];                   // DC differences summed over this lane up to (not including) that unit
    uint16_t pad_;
};

// packed decoder state at a subsequence boundary: bit position (relative to the
// image's ecs), data-unit phase within the MCU, zigzag slot
static inline __host__ __device__ uint64_t pjd_pack_state(uint32_t p, uint32_t c, uint32_t z)
{
    return (uint64_t)p | ((uint64_t)c << 32) | ((uint64_t)z << 40);
}
