// pjd_k_huffman.hip -- the PARALLEL entropy decoder for gfx950.
//
// Huffman decoding is one dependent chain per restart segment; a batch of ImageNet files has
// ~10^3 chains, a single 4K picture has one.  To fill 256 CUs the bitstream is cut into fixed
// SUBSEQUENCES (128..1024 bytes, chosen per batch), one decode lane each, and lanes find their true
// entry states by self-synchronisation: a decoder started at a wrong position falls into step
// with the true one after a while (the scheme Weissenberger & Schmidt describe for GPUs).
// Measured on 4:2:0 streams the distance to synchronisation is ~160 B on average with 5 % above
// 512 B -- bit position, zigzag slot AND the 6-unit MCU phase must all agree -- which shapes
// everything below.
//
//   pjd_k_build_tables   raw (offsets, symbols) tables -> two-level decode table per TABLE SET (images with the
//                        same Huffman tables share one): 9-bit first level (32-bit entries: the symbol, and the PAIR of
//                        symbols where both fit), one 128-entry second-level table
//                        per 9-bit prefix that holds longer codes; an entry already carries bits consumed /
//                        run / size / EOB / error (semantics of reference generate_codes / get_next_symbol /
//                        the size limits of decode_MCU_component, reference src/jpeg_scanner.cpp:438-520)
//   pjd_k_lane_words     the bitstream of every lane as big-endian 32-bit words counted from the lane's first
//                        byte, transposed per wave ([word][lane]): a lane's refill is one dword of a row its
//                        neighbours read too, so the 64 per-lane streams of a wave cost a few cache lines per
//                        load instead of 64
//   pjd_k_huff_lanes     one wave = 64 subsequences of one image; a workgroup = up to PJD_HUFF_WAVES (2) consecutive waves
//                        that share one table set in LDS (nothing else: waves never meet at a barrier after the tables
//                        are loaded).
//                        A  speculative pass over the own subsequence (state only, no output) that leaves
//                           PJD_NCHK checkpoints of the trajectory in LDS
//                        R  re-sync rounds: a lane restarts from its predecessor's exit state and stops as soon
//                           as its state equals a checkpoint; states travel by wave shuffles; across waves by
//                           published 64-bit words (PJD_GENS generations, never a chain over the image)
//                        C  data-unit counts: scan inside the wave, decoupled look-back over the image's waves
//                        W  write pass from the true entry states: EVERY symbol becomes one 16-bit entry in the
//                           lane's own region, one 32-bit STEP word per step -- a symbol, or the pair of symbols one
//                           lookup yields (layout: pjd_internal.h) -- staged through LDS and written as whole
//                           32-byte groups; the lane that decodes the first DC symbol of an IDCT workgroup's
//                           range leaves a mark (lane, entry offset, DC sums so far); per lane: entry count and
//                           DC sums for the predictor scan (pjd_k_lane_dc_*)
//
// Exactness: a lane that starts from the true state performs exactly the reference's
// decode_MCU_component (reference src/jpeg_scanner.cpp:467-520).  An entropy-coding error of the true
// decode -- invalid code, size or run outside the baseline limits, the data ending inside a symbol --
// is settled by the lane that meets it (careful_span: the reference's checks in the reference's order;
// the picture's first error becomes its status, pjd_k_image_verdict).  What the parallel decoder cannot
// resolve -- a segment that ends early or late, a boundary that did not stitch, a lane that does not
// reproduce its synchronised exit -- is reported with its position (PjdDevImState::flag_pos); only if
// that lies before the picture's first error does pjd_batch_sync re-decode the picture with the
// one-lane exact kernel, on the GPU.
#include <cstdlib>

#include "pjd_device_common.h"
#include "pjd_kernels.h"
#include "../../include/pjd.h"

static_assert(sizeof(PjdDevHuffRaw) == 180, "raw table layout");
static_assert(PJD_LUT_BITS + PJD_L2_BITS == 16 && PJD_L1_BYTES == (4 << PJD_LUT_BITS), "two-level table geometry");
static_assert(PJD_HUFF_LANES == 64, "one wave per 64 lanes");
static_assert(PJD_STAGE_ENTRIES == PJD_GROUP && PJD_GROUP == 16, "a group (a head word + 7 step words) is what one flush of the staging buffer writes");

#define LUT_BAD     PJD_LUT_ENTRY(16u, 1u, false, PJD_LUT_BADSYM)    // no code: consume 16 bits (as the reference's get_next_symbol)

// ---------------------------------------------------------------------------------------------
// One block per (table set, table slot): two-level decode table (layout: pjd_internal.h).
__device__ __forceinline__ uint32_t lut_entry(uint32_t len, uint32_t sym, bool is_ac)
{
    uint32_t run = 0, size, bad = 0;
    bool eob = false;
    if (sym == 0xFF) bad = PJD_LUT_BADSYM;                              // the reference reads a returned 0xFF as "no symbol" (jpeg_scanner.cpp:470,490)
    if (is_ac) {
        run = sym >> 4; size = sym & 15u;
        if (sym == 0) eob = true;
        else if (size > 10 && !bad) bad = PJD_LUT_BADLEN;               // jpeg_scanner.cpp:506
    } else {
        size = sym;
        if (sym > 11 && !bad) bad = PJD_LUT_BADLEN;                     // jpeg_scanner.cpp:474
    }
    return PJD_LUT_ENTRY(len + (bad ? 0u : size), eob ? 32u : run + 1, eob, bad ? bad : size);   // EOB: 32 + 64 = 96 slots (pjd_internal.h)
}

__global__ __launch_bounds__(256) void pjd_k_build_tables(PjdDevBatch B)
{
    const uint32_t ts = blockIdx.x / PJD_MAX_TABLES, slot = blockIdx.x % PJD_MAX_TABLES;
    const PjdDevTset &T = B.tsets[ts];
    if (slot >= T.n_tables || T.lut_bytes == 0) return;
    const PjdDevHuffRaw &r = B.raw_tables[(size_t)ts * PJD_MAX_TABLES + slot];
    uint16_t *blob = reinterpret_cast<uint16_t *>(B.luts + (size_t)T.lut_off16 * 16);
    uint32_t *L1 = reinterpret_cast<uint32_t *>(blob + slot * (PJD_L1_BYTES / 2));
    const uint32_t l2_off = T.l2_off[slot], p0 = T.l2_p0[slot], p1 = T.l2_p1[slot];
    const bool is_ac = r.is_ac != 0;
    // the table the SECOND symbol of a pair is decoded with: this one (AC), or the AC table of the components that use this DC table
    const uint32_t partner = is_ac ? slot : (uint32_t)T.pair_ac[slot];
    const bool has_partner = partner < T.n_tables;
    const PjdDevHuffRaw &r2 = B.raw_tables[(size_t)ts * PJD_MAX_TABLES + (has_partner ? partner : slot)];
    __shared__ uint32_t first[17], first2[17];
    __shared__ uint8_t offs[17], offs2[17];
    const uint32_t tid = threadIdx.x;
    if (tid < 2) {
        const PjdDevHuffRaw &q = tid ? r2 : r;
        uint32_t *f = tid ? first2 : first;
        uint32_t code = 0;                       // reference generate_codes (jpeg_scanner.cpp:438-448)
        f[0] = 0;
        for (int len = 1; len <= 16; len++) {
            f[len] = code;
            code = (code + (uint32_t)(q.offsets[len] - q.offsets[len - 1])) << 1;
        }
    }
    if (tid < 17) { offs[tid] = r.offsets[tid]; offs2[tid] = r2.offsets[tid]; }
    __syncthreads();
    // the symbol whose code is the leading bits of `bits` (`nbits` of them are known); 0 if none is determined by them.  Shortest match wins, as the reference's scan
    auto match_in = [&](const PjdDevHuffRaw &q, const uint32_t *f, const uint8_t *o, bool ac, uint32_t bits, uint32_t nbits, uint32_t lo, uint32_t hi) -> uint32_t {
        for (uint32_t len = lo; len <= hi && len <= nbits; len++) {
            const uint32_t c = bits >> (nbits - len);
            const uint32_t d = c - f[len], cnt = (uint32_t)o[len] - o[len - 1];
            if (c >= f[len] && d < cnt) return lut_entry(len, q.symbols[o[len - 1] + d], ac);
        }
        return 0u;
    };
    auto match = [&](uint32_t bits, uint32_t nbits, uint32_t lo, uint32_t hi) -> uint32_t { return match_in(r, first, offs, is_ac, bits, nbits, lo, hi); };
    for (uint32_t idx = tid; idx < (1u << PJD_LUT_BITS); idx += 256) {
        uint32_t e = (idx >= p0 && idx < p1) ? (((l2_off + ((idx - p0) << PJD_L2_BITS)) >> PJD_L2_BITS) << 5) : LUT_BAD;      // pointer entry: bits 4..0 == 0
        const uint32_t m = match(idx, PJD_LUT_BITS, 1, PJD_LUT_BITS);
        if (m) e = m;
        // the pair: this symbol whole inside the 9 bits, a valid run/size (or DC) symbol, and the bits after it determine the next code
        uint32_t pair = 0;
        if (has_partner && m && !(m & PJD_LUT_EOB) && PJD_LUT_SIZE(m) < PJD_LUT_BADSYM && PJD_LUT_USED(m) < PJD_LUT_BITS) {
            const uint32_t rest = PJD_LUT_BITS - PJD_LUT_USED(m);
            const uint32_t m2 = match_in(r2, first2, offs2, true, idx & ((1u << rest) - 1u), rest, 1, rest);
            if (m2 && PJD_LUT_SIZE(m2) < PJD_LUT_BADSYM && PJD_LUT_USED(m) + PJD_LUT_USED(m2) <= 31u)
                pair = (PJD_LUT_USED(m) + PJD_LUT_USED(m2)) | ((PJD_LUT_ADV(m) + PJD_LUT_ADV(m2)) << 5) | (PJD_LUT_SIZE(m2) << 12);   // size 2: bits 31..28 of the entry
        }
        // no pair: the pair field repeats the symbol's own used / advance (a "pair" that is the symbol alone), so the state-only
        // passes need not ask whether there is one; the write pass tells a real pair by used12 != used
        if (!pair) pair = e & 0xfffu;
        L1[idx] = (e & 0xffffu) | (pair << 16);
    }
    for (uint32_t j = tid; j < ((p1 - p0) << PJD_L2_BITS); j += 256) {
        const uint32_t w16 = (p0 << PJD_L2_BITS) + j;
        const uint32_t m = match(w16, 16, PJD_LUT_BITS + 1, 16);
        blob[l2_off + j] = (uint16_t)(m ? m : LUT_BAD);
    }
}

// ---------------------------------------------------------------------------------------------
// One block per Huffman wave: the wave's 64 lane streams as big-endian words, transposed.
// A lane's 128-byte source lines are re-read from L1 for 32 consecutive rows; the rows are written whole.
struct __attribute__((packed)) UnalignedU32 { uint32_t v; };
#ifndef PJD_WORDS_X4
#define PJD_WORDS_X4 1
#endif

struct __attribute__((packed)) UnalignedU32x4 { uint32_t v[4]; };

// hwg_count == 0: one block per wave of the batch; else the waves of the Huffman workgroups hwg_first .. (a picture group,
// pjd_internal.h): two blocks per workgroup.
__global__ __launch_bounds__(256) void pjd_k_lane_words(PjdDevBatch B, uint32_t hwg_first, uint32_t hwg_count)
{
    uint32_t wv = blockIdx.x;
    if (hwg_count) {
        const PjdDevHuffWg wg = B.hwgs[hwg_first + blockIdx.x / PJD_HUFF_WAVES];
        const uint32_t k = blockIdx.x % PJD_HUFF_WAVES;
        if (k >= wg.n_waves) return;
        wv = wg.first_wave + k;
    }
    const PjdDevHuffWave hw = B.hwaves[wv];
    const PjdDevImage &im = B.images[hw.image];
    const uint32_t l = threadIdx.x & 63, k0 = threadIdx.x >> 6;
    const uint32_t rows = PJD_WORD_ROWS(im.sub_bytes);          // of this image: at most B.word_rows, the stride between waves; a multiple of 4
    uint32_t *dst = B.words + (size_t)wv * B.word_rows * 64;
    const bool valid = l < hw.n_lanes;
    const uint8_t *src = B.ecs + im.ecs_off + (valid ? B.lanes[hw.first_lane + l].byte_start : 0u);
#if PJD_WORDS_X4
    // 16 bytes of the lane's stream per load (four rows), four groups of rows in flight per thread
    for (uint32_t k = 4 * k0; k < rows; k += 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (valid) {
            const UnalignedU32x4 v = *reinterpret_cast<const UnalignedU32x4 *>(src + 4 * k);
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = __builtin_bswap32(v.v[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) dst[(k + j) * 64 + l] = w[j];
    }
#else
    for (uint32_t k = k0; k < rows; k += 4) {
        uint32_t w = 0;
        if (valid) w = __builtin_bswap32(reinterpret_cast<const UnalignedU32 *>(src + 4 * k)->v);
        dst[k * 64 + l] = w;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Bit window over the lane's own word stream.  peek() = the next 32 bits; two words are held, a third is in
// flight.  `s` = bits of `hi` not yet consumed, minus 32 (0..31 as the shift of v_alignbit: with s == 0 the
// window is `lo` alone).
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) uint8_t *pjd_gptr;      // global memory, so that loads are global_load (a generic
                                                                         // pointer gives flat_load, which also counts as an LDS access)
#if PJD_DIRECT_ECS
#define PJD_WORD_STRIDE 4u          // the lanes read the batch's bitstream itself: a lane's words are consecutive (unaligned) dwords
#else
#define PJD_WORD_STRIDE 256u        // transposed rows (pjd_k_lane_words): word k of a lane is 256 bytes after word k-1
#endif
struct BitWin {
    pjd_gptr wb;          // wave-uniform: row 0 of the wave's word rows / the first byte of the picture's bitstream
    uint32_t hi, lo, nxt;
    uint32_t off;         // byte offset (from wb) of this lane's next word to fetch
    int s;
    __device__ __forceinline__ uint32_t word(uint32_t byte_off) const
    {
#if PJD_DIRECT_ECS
        return __builtin_bswap32(reinterpret_cast<const __attribute__((address_space(1))) UnalignedU32 *>(wb + byte_off)->v);
#else
        return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(wb + byte_off);
#endif
    }
    // col = byte offset of the lane's word 0 from wb: lane * 4 in the transposed rows, the lane's first byte in the bitstream
    __device__ __forceinline__ void init(pjd_gptr wave_words, uint32_t col, uint32_t p)
    {
        wb = wave_words;
        const int kk = (int)((p + 31) >> 5) - 1;          // word holding bit p-1 (or -1 at p == 0)
        const uint32_t o = (uint32_t)(kk + 1) * PJD_WORD_STRIDE + col;
        hi = kk >= 0 ? word(o - PJD_WORD_STRIDE) : 0u;
        lo = word(o);
        nxt = word(o + PJD_WORD_STRIDE);
        off = o + 2 * PJD_WORD_STRIDE;
        s = 32 * (kk + 1) - (int)p;                        // 0..31
    }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s); }
    __device__ __forceinline__ void drop(uint32_t n)      // n <= 32
    {
        s -= (int)n;
        if (s < 0) {
            s += 32;
            hi = lo; lo = nxt;
            // keep the copies above ahead of the load below: the load can then target nxt's register directly and is
            // first waited for at the next refill (~7 symbols later), not here
            asm volatile("" : "+v"(hi), "+v"(lo));
            nxt = word(off);
            off += PJD_WORD_STRIDE;
        }
    }
};

// The window of the passes that run with all 64 lanes (sync_span, write_span).  With BitWin a lane fetches its next word when it runs
// out of bits -- every ~4 steps per lane, so in EVERY step of the wave some lane does, and the wait before the fetched register is
// reused (s_waitcnt vmcnt(0): the counter is the wave's, in issue order) is a wait for the load of the step before: a step lasted as
// long as a load from the memory-side cache (~545 cycles; ~900 from HBM) instead of its ~300 cycles of table lookups and selects
// (profiles/r04_experiments.md #25: 54 % of the kernel's wave-cycles were such waits).  Here a lane holds 96 bits (w0..w2), two more
// words are in flight (n0, n1), and the window is SERVICED once every TWO steps, by all lanes together: shift by the 0..2 words the two
// steps used up, fetch the two words behind the new window -- the same ones again if the lane did not move (a load into a register
// that already holds the value).  What a service waits for was requested two steps earlier.
//   r = bits of w0 not yet consumed (0..31 after a service; down to -62 before the next: a step takes at most 31 bits)
struct BitWin2 {
    pjd_gptr wb;
    uint32_t w0, w1, w2, n0, n1;
    uint32_t off;         // byte offset (from wb) of n0's word
    int r;
    __device__ __forceinline__ uint32_t word(uint32_t byte_off) const
    {
        return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(wb + byte_off);
    }
    __device__ __forceinline__ void init(pjd_gptr wave_words, uint32_t col, uint32_t p)
    {
        wb = wave_words;
        const int kk = (int)((p + 31) >> 5) - 1;          // word holding bit p-1 (or -1 at p == 0)
        const uint32_t o = (uint32_t)(kk + 1) * PJD_WORD_STRIDE + col;
        w0 = kk >= 0 ? word(o - PJD_WORD_STRIDE) : 0u;
        w1 = word(o);
        w2 = word(o + PJD_WORD_STRIDE);
        n0 = word(o + 2 * PJD_WORD_STRIDE);
        n1 = word(o + 3 * PJD_WORD_STRIDE);
        off = o + 2 * PJD_WORD_STRIDE;
        r = 32 * (kk + 1) - (int)p;                        // 0..31
    }
    // the next 32 bits: right after a service (r >= 0) / anywhere between two services
    __device__ __forceinline__ uint32_t peek0() const { return __builtin_amdgcn_alignbit(w0, w1, (uint32_t)r); }
    __device__ __forceinline__ uint32_t peek1() const
    {
        const bool in1 = r < 0;
        return __builtin_amdgcn_alignbit(in1 ? w1 : w0, in1 ? w2 : w1, (uint32_t)r & 31u);
    }
    __device__ __forceinline__ void drop(uint32_t n) { r -= (int)n; }
    __device__ __forceinline__ void service()
    {
        // words used up: 0, 1 or 2 -- as two shifts by one word (a three-way choice per register is what the compiler turns into an
        // array in scratch memory indexed by the count)
        const bool j1 = r < 0, j2 = r < -32;
        uint32_t t0 = j1 ? w1 : w0, t1 = j1 ? w2 : w1, t2 = j1 ? n0 : w2, t3 = j1 ? n1 : n0;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        w0 = j2 ? t1 : t0; w1 = j2 ? t2 : t1; w2 = j2 ? t3 : t2;
        off += (j1 ? PJD_WORD_STRIDE : 0u) + (j2 ? PJD_WORD_STRIDE : 0u);
        r &= 31;
        // the copies above stay ahead of the loads: the loads then target n0 / n1 directly and are first waited for at the next service
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
        n0 = word(off);
        n1 = word(off + PJD_WORD_STRIDE);
    }
};

// LDS is addressed by ABSOLUTE byte address (the tables' offsets carry the base of the kernel's dynamic LDS): indexing a pointer
// derived from the extern array makes the compiler add that base -- a link-time constant it cannot fold -- on every access.
__device__ __forceinline__ uint32_t lds_abs(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t *)p; }
__device__ __forceinline__ uint32_t lds_u16(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(a); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(a); }
typedef uint32_t pjd_v2u32 __attribute__((ext_vector_type(2)));
typedef uint32_t pjd_v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_u32x2(uint32_t a)
{
    const pjd_v2u32 v = *reinterpret_cast<const __attribute__((address_space(3))) pjd_v2u32 *>(a);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 lds_u32x4(uint32_t a)
{
    const pjd_v4u32 v = *reinterpret_cast<const __attribute__((address_space(3))) pjd_v4u32 *>(a);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Data-unit phase inside the MCU.  Unit k of an MCU (luma units first) is kept as r = units of the MCU still to come after it
// (dus-1 .. 0): the component is 0 while r >= nc (nc = chroma components), else nc - r.  What a unit decodes with -- the absolute
// LDS byte addresses of its DC and AC tables (below 64 KB), packed DC | AC << 16 -- and which DC sum its difference goes to is looked up in the
// wave's PHASE TABLE in LDS: record r (16 bytes) describes the unit that FOLLOWS unit r, so completing a unit is one read:
//   .x  table offsets of the next unit        .y  LDS byte address of the next unit's own record
//   .z  mask of (dv | dv << 16) for the packed sums {Y | Cb << 16}         .w  the same for {Cr}
struct PhaseCtx {
    uint32_t tY, tC1, tC2;     // wave-uniform table addresses (DC | AC << 16) of the three components
    uint32_t nc, dus1;         // chroma components; data units per MCU - 1
    uint32_t xbase;            // LDS byte address of this wave's phase table
    uint32_t lbase;            // LDS byte address of the table blob (second-level tables are indexed from it)
    __device__ __forceinline__ uint32_t tabs(uint32_t r) const { return r >= nc ? tY : (r == 0 ? tC2 : tC1); }
    __device__ __forceinline__ uint32_t comp(uint32_t r) const { return r >= nc ? 0u : nc - r; }
    __device__ __forceinline__ uint32_t next(uint32_t r) const { return r == 0 ? dus1 : r - 1; }
    // lanes 0..dus1 of the wave write the table (before the wave's first pass; the wave is converged)
    __device__ __forceinline__ void build(uint32_t lane) const
    {
        if (lane <= dus1) {
            const uint32_t rn = next(lane), cn = comp(rn);
            uint4 rec;
            rec.x = tabs(rn);
            rec.y = xbase + rn * 16;
            rec.z = cn == 0 ? 0x0000ffffu : (cn == 1 ? 0xffff0000u : 0u);
            rec.w = cn == 2 ? 0x0000ffffu : 0u;
            pjd_v4u32 v; v.x = rec.x; v.y = rec.y; v.z = rec.z; v.w = rec.w;
            *reinterpret_cast<__attribute__((address_space(3))) pjd_v4u32 *>(xbase + lane * 16) = v;
        }
    }
    // the record that describes unit r itself = the one stored at its predecessor in the cycle
    __device__ __forceinline__ uint32_t self(uint32_t r) const { return xbase + (r == dus1 ? 0u : r + 1) * 16; }
};

struct ChkCtx {            // checkpoint bookkeeping of one lane (LDS, strided by lane)
    uint32_t *state;       // [PJD_NCHK][64] at this lane's column
    uint32_t *rem;         // [PJD_NCHK][64]
    uint32_t chk_bits;     // checkpoint spacing
    uint32_t walk_max;     // wave-uniform: rounds with at most this many active lanes are walked (walk_lane)
};

enum { SPAN_END = 0, SPAN_MERGED = 1, SPAN_MERGED_B = 2 };

// One more landing pad per lane, in registers ("B"): the state an OLDER trajectory of the lane had at the first checkpoint, the units
// that follow it, and that trajectory's exit state.  Truth moves through a run of non-merging lanes one lane per round, and every hop
// is a full pass of a lane whose new entry leads to a trajectory it has not decoded before; the entry states a lane sees differ mostly
// in the MCU phase, there are only a few of those, and a pass from a phase seen the round before last used to be decoded again in
// full because only the newest trajectory's checkpoints are kept (LDS).  With B such a pass ends at the first checkpoint (90 % of all
// merges happen there).  Model (tools/sync_sim.c, sim_rounds_b): -12 % round time per wave, -16 % on the slowest wave of dense pictures.
struct BCache {
    uint32_t st;           // packed state at checkpoint 1 (as sync_span's `st`); 0xffffffff: empty
    uint32_t rem;          // data units from there to the end of the subsequence
    uint32_t p, cz;        // exit state of that trajectory (as WaveState)
};

// After a sync pass: checkpoints (re)written in it hold "units so far"; make them "units still to come".
__device__ __forceinline__ void chk_finish(const ChkCtx &K, uint32_t j, uint32_t ndu)
{
    for (uint32_t i = 1; i < j; i++) K.rem[i * 64] = ndu - K.rem[i * 64];
}

// The L1 entry for the next bits (32 bits: symbol | pair << 16), or the second-level entry (16 bits: no pair) for a code longer than 9 bits
__device__ __forceinline__ uint32_t lut_lookup_pair(uint32_t lbase, uint32_t tab, uint32_t pk)
{
    uint32_t e = lds_u32(tab + 4 * __builtin_amdgcn_ubfe(pk, 32 - PJD_LUT_BITS, PJD_LUT_BITS));
    // code longer than 9 bits (a pointer entry consumes no bits): one more read, in the 128-entry table of this 9-bit prefix
    if (__builtin_expect(PJD_LUT_USED(e) == 0, 0)) {
        e = lds_u16(lbase + 2 * ((((e >> 5) & 0x7ffu) << PJD_L2_BITS) + ((pk >> 16) & ((1u << PJD_L2_BITS) - 1u))));
        e |= e << 16;                                  // no pair: as the first-level entries say it (the size lands in size2: unused)
    }
    return e;
}
// the symbol alone (the low half)
__device__ __forceinline__ uint32_t lut_lookup(uint32_t lbase, uint32_t tab, uint32_t pk) { return lut_lookup_pair(lbase, tab, pk) & 0xffffu; }

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ndu += vcc ? 1 : 0 in one instruction (the compiler emits a select and an add)
__device__ __forceinline__ uint32_t add_flag(uint32_t v, bool f)
{
    uint32_t d;
    asm("v_addc_co_u32_e64 %0, vcc, %1, 0, %2" : "=v"(d) : "v"(v), "s"(__builtin_amdgcn_ballot_w64(f)) : "vcc");
    return d;
}

// STATE-ONLY pass: decodes symbols that START before end_bit.  State (p, c, z): bit position relative to the
// lane's first byte, data-unit phase within the MCU, zigzag slot (0 = DC expected).  BRIDGE: stop as soon as
// the state equals the checkpoint recorded by an earlier pass (then ndu already includes the units still to
// come); otherwise (re)write the checkpoints passed.  Returns the number of checkpoints passed + 1 in `jout`.
// Inside the loop the slot is kept as zb = 63 - z (63: DC expected) and the phase as the LDS address of the unit's
// phase record: one symbol is a table lookup, "zb -= advance", and three selects when the unit is complete.
// BRIDGE also compares the state at checkpoint 1 with Bst (the lane's BCache) and returns SPAN_MERGED_B on a match (ndu then does
// NOT include B's units yet); oldA_st / oldA_rem: what the newest trajectory held at checkpoint 1 if this pass crossed it without
// merging into it (0xffffffff otherwise): the caller's next B.
template <bool BRIDGE>
__device__ __forceinline__ int sync_span(const PhaseCtx &P, pjd_gptr wave_words, uint32_t col,
                                         uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit,
                                         uint32_t &ndu, const ChkCtx &K, uint32_t &jout, uint32_t Bst, uint32_t &oldA_st, uint32_t &oldA_rem)
{
    uint32_t j = 1;
    jout = 1;
    oldA_st = 0xffffffffu; oldA_rem = 0;
    if (p >= end_bit) return SPAN_END;
    BitWin2 w;
    w.init(wave_words, col, p);
    uint32_t ra, x;
    {
        const uint2 cur = lds_u32x2(P.self(P.dus1 - c));
        x = cur.x; ra = cur.y;
    }
    int zb = 63 - (int)z;
    uint32_t next_chk = K.chk_bits;
    uint32_t lim = next_chk < end_bit ? next_chk : end_bit;     // one compare per symbol covers "subsequence end" and "next checkpoint"
    int res = SPAN_END;
    // one step from the window `pk`; true: the pass ends here
    auto step = [&](uint32_t pk) -> bool {
        const uint2 nx = lds_u32x2(ra);                                     // what follows this unit: issued beside the table lookup
        const uint32_t tab = (zb == 63) ? (x & 0xffffu) : (x >> 16);
        const uint32_t e = lut_lookup_pair(P.lbase, tab, pk);
        // two symbols in this step where the table holds the pair, the first one leaves the unit open and the second one still
        // starts before the next checkpoint / the subsequence's end -- else the first alone
        const uint32_t u1 = PJD_LUT_USED(e), u12 = PJD_LUT_PAIR_USED(e);
        const int z1 = zb - (int)PJD_LUT_ADV(e);
        const bool pair = z1 >= 0 && p + u1 < lim;                          // (an entry without a pair repeats the symbol in its pair field)
        const uint32_t used = pair ? u12 : u1;
        w.drop(used);
        p += used;
        // state update (reference src/jpeg_scanner.cpp:469-518): a DC symbol advances one slot and never carries the EOB bit; a
        // run past slot 63 ends the unit here (the write pass reports it)
        zb = pair ? zb - (int)PJD_LUT_PAIR_ADV(e) : z1;
        const bool done = zb < 0;                                           // EOB, or the unit's last slot was filled
        zb = done ? 63 : zb;
        ra = done ? nx.y : ra;
        x = done ? nx.x : x;
        ndu = add_flag(ndu, done);
        if (p >= lim) {
            if (p >= end_bit) return true;
            const uint32_t st = (p << 14) | ((ra - P.xbase) << 6) | (uint32_t)zb;      // p < 2^14, record offset < 256, zb < 64
            if (BRIDGE) {
                const uint32_t a = K.state[j * 64];
                if (a == st) { ndu += K.rem[j * 64]; res = SPAN_MERGED; return true; }
                if (j == 1) {
                    oldA_st = a; oldA_rem = K.rem[64];
                    if (st == Bst) { res = SPAN_MERGED_B; return true; }
                }
            }
            K.state[j * 64] = st;
            K.rem[j * 64] = ndu;                                            // turned into "still to come" after the pass
            j++;
            next_chk += K.chk_bits;
            lim = next_chk < end_bit ? next_chk : end_bit;
        }
        return false;
    };
    // on entry p < lim: p <= 26 bits (a symbol that started before the lane's first byte) and p < end_bit
    for (;;) {
        if (step(w.peek0())) break;
        if (step(w.peek1())) break;
        w.service();
    }
    c = P.dus1 - ((ra - P.xbase) >> 4);
    z = 63u - (uint32_t)zb;
    jout = j;
    return res;
}

// ---------------------------------------------------------------------------------------------
// The COOPERATIVE WALKER: one lane's subsequence decoded by the whole wave (state only, same result as sync_span<true>).
//
// A re-sync round costs a full pass of the wave whether 64 lanes or 2 are active, because one symbol of ONE lane is a chain of
// ~45 dependent instructions and two LDS round trips (~450 cycles alone on a SIMD).  When few lanes are left, the wave instead
// takes them one after the other: lane k looks up the symbol that would START at bit position p + k, for the tables of the
// current data unit and of the unit after it (64 positions x 4 tables: a handful of LDS reads per step), and a scalar loop hops
// through the results -- v_readlane + ~8 SALU per symbol, ~12 symbols per 64-bit window.  The chain of ONE lane advances
// several times faster, and because lanes are taken in ascending order a changed exit state is carried straight into the next
// lane: a chain of non-merging lanes that needed one round per lane is one walk.
// The bitstream comes from the picture's byte stream itself (contiguous across the lanes of a restart segment, so the window
// buffer survives lane boundaries): 64 big-endian dwords per lane register `cw` (+ the following 64 in `cn`, loaded a window
// ahead), gathered per bit position with ds_bpermute.
// ---------------------------------------------------------------------------------------------
// Which rounds are walked: PjdDevImage::walk_max (pjd_internal.h; chosen by the planner).
struct WalkBuf {
    pjd_gptr ecs;          // first byte of the picture's bitstream
    uint32_t clamp;        // last byte offset a dword is read at (inside the zero padding after the stream)
    uint32_t cb;           // byte offset of word 0 of `cw` (a multiple of 4)
    uint32_t cw, cn;       // per lane: word l of the window, word 64 + l
    uint32_t have;
    __device__ __forceinline__ uint32_t ld(uint32_t off) const
    {
        off = off < clamp ? off : clamp;
        return __builtin_bswap32(reinterpret_cast<const __attribute__((address_space(1))) UnalignedU32 *>(ecs + off)->v);
    }
    // make the window hold bits [pa, pa + 96) of the picture within its first 62 words; when it has to move, word 0 becomes the word of
    // bit `keep` <= pa (the oldest position the caller may still come back to: the start of the lane it is in)
    __device__ __forceinline__ void reach(uint32_t pa, uint32_t l, uint32_t keep)
    {
        cb = rfl(cb); have = rfl(have);                                        // wave-uniform by construction
        if (have && pa >= 8u * cb && pa - 8u * cb <= 1920u) return;
        const uint32_t rel = keep - 8u * cb;
        if (have && keep >= 8u * cb && rel < 2048u) {                          // slide by s <= 63 words inside (cw, cn)
            const uint32_t s = rel >> 5, idx = l + s;
            const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)cw);
            const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx & 63u) << 2), (int)cn);
            cw = idx < 64u ? a : b;
            cb += 4u * s;
        } else {
            cb = (keep >> 5) * 4u;
            cw = ld(cb + 4u * l);
        }
        cn = ld(cb + 256u + 4u * l);
        have = 1;
    }
};

// All arguments but `l` are wave-uniform.  st_col / rem_col: the checkpoint columns of the lane that is walked.
// Bst / oldA_st / oldA_rem: as sync_span (the walked lane's BCache state; all wave-uniform).
__device__ __forceinline__ int walk_lane(const PhaseCtx &P, WalkBuf &wb, uint32_t l, uint32_t base_bit,
                                         uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit, uint32_t &ndu,
                                         uint32_t *st_col, uint32_t *rem_col, uint32_t chk_bits, uint32_t &jout,
                                         uint32_t Bst, uint32_t &oldA_st, uint32_t &oldA_rem)
{
    uint32_t j = 1;
    jout = 1;
    oldA_st = 0xffffffffu; oldA_rem = 0;
    if (p >= end_bit) return SPAN_END;
    // scalar copies (intrinsic results, not loads from the context struct: selects between them stay selects)
    const uint32_t tY = rfl(P.tY), tC1 = rfl(P.tC1), tC2 = rfl(P.tC2), nc = rfl(P.nc), dus1 = rfl(P.dus1), lbase = rfl(P.lbase);
#define WALK_TABS(r_) ((r_) >= nc ? tY : ((r_) == 0 ? tC2 : tC1))
#define WALK_NEXT(r_) ((r_) == 0 ? dus1 : (r_) - 1u)
    uint32_t r = dus1 - c;                                       // units of the MCU still to come after the current one
    int zb = 63 - (int)z;
    uint32_t next_chk = chk_bits;
    uint32_t lim = next_chk < end_bit ? next_chk : end_bit;
    int res = SPAN_END;
    for (;;) {                                                   // one step = the symbols that start in one window of 64 bit positions
        wb.reach(base_bit + p, l, base_bit + p);
        // Symbols must start inside the window and before `lim`: k < klim.  The chase keeps (zb << 8) + k + 128 - klim in ONE scalar:
        // a symbol adds (bits used) - (slots advanced << 8); bit 31 then says "unit complete" (zb < 0), bit 7 "k >= klim", and
        // the low six bits select the lane that holds position k -- so lane l looks at position (l + klim) mod 64.
        const uint32_t klim = lim - p < 64u ? lim - p : 64u;
        const uint32_t bp = base_bit + p - 8u * rfl(wb.cb) + ((l + klim) & 63u), wi = bp >> 5, sh = bp & 31u;
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wi << 2), (int)wb.cw);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((wi + 1u) << 2), (int)wb.cw);
        const uint32_t pk = (uint32_t)((((((uint64_t)hi) << 32) | lo) << sh) >> 32);       // the 32 bits from that position on
        const uint32_t rn = WALK_NEXT(r), xa = WALK_TABS(r), xb = WALK_TABS(rn);
        // the symbol at every position under the DC and AC tables of the current unit and of the unit after it: four reads in flight
        const uint32_t i4 = 4u * __builtin_amdgcn_ubfe(pk, 32 - PJD_LUT_BITS, PJD_LUT_BITS);
        // (the DC table of the current unit only if the walk stands at its DC symbol; of the entries the low half: the symbol)
        const uint32_t e1 = lds_u32((xa >> 16) + i4), e2 = lds_u32((xb & 0xffffu) + i4), e3 = lds_u32((xb >> 16) + i4);
        uint32_t e0 = e1;
        if (zb == 63) e0 = lds_u32((xa & 0xffffu) + i4);
        // A position whose code is longer than 9 bits (a pointer entry) gets the mark WALK_PTR instead of its delta: the chase stops
        // there and the ONE second-level entry it needs is fetched then (long codes are rare on the path; they are not among 256 lookups).
#define WALK_PTR 0x40000000u
#define WALK_DELTA(e_) (PJD_LUT_USED(e_) ? PJD_LUT_USED(e_) - (PJD_LUT_ADV(e_) << 8) : WALK_PTR)
        uint32_t Ddc = WALK_DELTA(e0), Dac = WALK_DELTA(e1), Edc = e0, Eac = e1;
        const uint32_t DdcB = WALK_DELTA(e2), DacB = WALK_DELTA(e3);
#undef WALK_DELTA
        // st carries the mark (bit 30) and no other flag: replace it by the delta of the symbol's second-level entry
#define WALK_FIX(E_)                                                                                                        \
        do {                                                                                                                \
            const uint32_t ep_ = (uint32_t)__builtin_amdgcn_readlane((int)(E_), (int)(st & 63u));                           \
            const uint32_t pl_ = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)(st & 63u));                             \
            const uint32_t e2_ = rfl(lds_u16(lbase + 2u * ((((ep_ >> 5) & 0x7ffu) << PJD_L2_BITS) + ((pl_ >> 16) & ((1u << PJD_L2_BITS) - 1u)))));   \
            st = st - WALK_PTR + (PJD_LUT_USED(e2_) - (PJD_LUT_ADV(e2_) << 8));                                             \
        } while (0)
        const uint32_t bias = 128u - klim;
        uint32_t st = ((uint32_t)zb << 8) + bias;
        bool second = false;
        for (;;) {
            if ((st >> 8) == 63u) {                              // the unit's DC symbol
                st += (uint32_t)__builtin_amdgcn_readlane((int)Ddc, (int)(st & 63u));
                if ((int)st >= 0 && (st & WALK_PTR)) WALK_FIX(Edc);
                if (st & 0x80000080u) {
                    if ((int)st >= 0) break;                     // k >= klim
                    goto unit_done;                              // (no DC table advances past the unit; kept general)
                }
            }
            for (;;) {
                do st += (uint32_t)__builtin_amdgcn_readlane((int)Dac, (int)(st & 63u)); while (!(st & 0xC0000080u));
                if ((int)st < 0 || !(st & WALK_PTR)) break;
                WALK_FIX(Eac);
                if (st & 0x80000080u) break;
            }
            if ((int)st >= 0) break;                             // k >= klim inside the unit
        unit_done:
            ndu++; r = WALK_NEXT(r);
            st = (63u << 8) | (st & 0xffu);                      // the unit is complete: DC expected
            if ((st & 0x80u) || second) break;
            second = true; Ddc = DdcB; Dac = DacB; Edc = e2; Eac = e3;       // the unit after it: its tables were looked up too
        }
#undef WALK_FIX
#undef WALK_PTR
        zb = (int)(st >> 8);
        const uint32_t k = (st & 0xffu) - bias;
        p += k;
        if (p >= lim) {
            if (p >= end_bit) break;
            const uint32_t st = (p << 14) | (r << 10) | (uint32_t)zb;                        // as sync_span: record offset r * 16 << 6
            const uint32_t a = rfl(st_col[j * 64]);
            if (a == st) { ndu += rfl(rem_col[j * 64]); res = SPAN_MERGED; break; }
            if (j == 1) {
                oldA_st = a; oldA_rem = rfl(rem_col[64]);
                if (st == Bst) { res = SPAN_MERGED_B; break; }
            }
            if (l == 0) { st_col[j * 64] = st; rem_col[j * 64] = ndu; }
            j++;
            next_chk += chk_bits;
            lim = next_chk < end_bit ? next_chk : end_bit;
        }
    }
    c = dus1 - r;
    z = 63u - (uint32_t)zb;
    jout = j;
    return res;
#undef WALK_TABS
#undef WALK_NEXT
}

// ---------------------------------------------------------------------------------------------
// WRITE pass of one lane.
// ---------------------------------------------------------------------------------------------
struct OutCtx {
    uint16_t *region;      // the lane's entry region (HBM)
    uint32_t *stage;       // LDS: [8 rows][64 lanes] dwords at this lane's column: row 0 the group's head, rows 1..7 its step words
    uint32_t cap;          // slots the region holds
    uint32_t n;            // slots used so far (pjd_internal.h: groups of a head word + 7 step words); even
    uint32_t npair;        // steps that held a pair (symbols = steps + pairs)
    uint32_t dcA, dcB;     // DC differences summed so far, each mod 2^16: Y (low) Cb (high) | Cr (low)
    uint32_t mark_D;       // the next data unit that opens an IDCT workgroup's range
    uint32_t ru;           // data units per IDCT workgroup
    PjdDevMark *marks;     // of this image
    uint32_t mark_next;    // index of the next mark this lane would write
    uint32_t lane_q;
    uint32_t overflow;
    uint32_t D_in;         // the lane's first data unit
};

// 8 dwords of the lane's staging column -> 32 bytes of its region
__device__ __forceinline__ void stage_flush(const OutCtx &O, uint32_t first_slot)
{
    uint4 *dst = reinterpret_cast<uint4 *>(O.region + first_slot);
#pragma unroll
    for (int k = 0; k < PJD_STAGE_ENTRIES / 8; k++) {
        uint4 v;
        v.x = O.stage[(4 * k + 0) * 64]; v.y = O.stage[(4 * k + 1) * 64];
        v.z = O.stage[(4 * k + 2) * 64]; v.w = O.stage[(4 * k + 3) * 64];
        dst[k] = v;
    }
}

struct WState {            // decoder state of the write pass, in registers
    uint32_t p;
    int zb;                // 63 - zigzag slot
    uint32_t ra, x;        // phase record address, table offsets of the current unit
    uint32_t mA, mB;       // DC-sum selectors of the current unit's component
    uint32_t emax;         // max over the table entries seen: >= 0xe000 <=> an invalid symbol (size field 14 or 15)
    uint32_t umin;         // min over the steps of (63 - slot after the step) + 16, unsigned: <= 14 <=> a run past slot 63
};

typedef unsigned short pjd_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (pjd_u16x2)(__builtin_bit_cast(pjd_u16x2, a) + __builtin_bit_cast(pjd_u16x2, b)));
}

// `size` value bits sign-extended the JPEG way (reference src/jpeg_scanner.cpp:478-484,510-516)
__device__ __forceinline__ int jpeg_extend(uint32_t bits, uint32_t size)
{
    const uint32_t m1 = 1u << size;
    return (int)bits + ((((int)bits - (int)(m1 >> 1)) >> 31) & (int)(1u - m1));
}
// the same for the `size` bits that END `used` bits into the window pk (used <= 32): a leading 0 bit means bits - (2^size - 1)
__device__ __forceinline__ int jpeg_value(uint32_t pk, uint32_t used, uint32_t size)
{
    const uint32_t off = 32u - used;
    const uint32_t b = __builtin_amdgcn_ubfe(pk, off, size);
    const uint32_t m = (1u << size) - 1u;
    const uint32_t top = (uint32_t)__builtin_amdgcn_sbfe(pk, off + size - 1u, 1u);      // all ones if the leading bit is set (size 0: m == 0 anyway)
    return (int)(b - (m & ~top));
}

// One STEP of the write pass: one symbol, or the PAIR the table entry holds (AC symbols, the first leaves the unit open and the
// second still starts before `end_bit`).  Returns the step word (entry A | entry B << 16, pjd_internal.h); updates the state.
template <bool FIRST>
__device__ __forceinline__ uint32_t write_step(uint32_t lbase, BitWin2 &w, WState &S, uint32_t &D, OutCtx &O, uint32_t end_bit)
{
    const bool is_dc = (S.zb == 63);
    if (__builtin_expect(is_dc && D == O.mark_D, 0)) {                      // this unit opens an IDCT workgroup's range
        PjdDevMark m;
        m.lane = O.lane_q; m.ent_off = O.n;
        m.acc[0] = (uint16_t)O.dcA; m.acc[1] = (uint16_t)(O.dcA >> 16); m.acc[2] = (uint16_t)O.dcB; m.pad_ = 0;
        O.marks[O.mark_next++] = m;
        O.mark_D += O.ru;
    }
    const uint32_t pk = FIRST ? w.peek0() : w.peek1();                      // FIRST: the step right after a service of the window
    const uint4 nx = lds_u32x4(S.ra);                                       // the unit after this one
    const uint32_t tab = is_dc ? (S.x & 0xffffu) : (S.x >> 16);
    const uint32_t e = lut_lookup_pair(lbase, tab, pk);
    const uint32_t u1 = PJD_LUT_USED(e), size1 = PJD_LUT_SIZE(e), adv1 = PJD_LUT_ADV(e);
    const uint32_t u12 = PJD_LUT_PAIR_USED(e), adv12 = PJD_LUT_PAIR_ADV(e), size2 = PJD_LUT_PAIR_SIZE2(e);
    const bool pair = u12 != u1 && S.zb >= (int)adv1 && S.p + u1 < end_bit;
    O.npair += pair ? 1u : 0u;
    // values: `size` bits after the code(s); an invalid entry (size field 14 / 15) yields garbage here and sends the lane to the
    // careful pass through S.emax
    const int val1 = jpeg_value(pk, u1, size1);
    const int val2 = jpeg_value(pk, u12, size2);                            // no pair: zero bits
    const uint32_t used = pair ? u12 : u1, adv = pair ? adv12 : adv1;
    w.drop(used);
    S.p += used;
    { const uint32_t el = e & 0xffffu; S.emax = S.emax > el ? S.emax : el; }
    S.zb -= (int)adv;
    const bool done = S.zb < 0;
    { const uint32_t u = (uint32_t)(S.zb + 16); S.umin = S.umin < u ? S.umin : u; }      // run past slot 63 (jpeg_scanner.cpp:500): -16..-2 here
    // entries (layout: pjd_internal.h): value << 5 | run + 1 (the low five bits of the advance: 0 for an EOB); a DC difference as it is
    const uint32_t entA = is_dc ? (uint32_t)val1 : (((uint32_t)val1 << 5) | (adv1 & 31u));
    const uint32_t entB = pair ? (((uint32_t)val2 << 5) | ((adv12 - adv1) & 31u)) : PJD_ENT_NONE;
    // DC sums of this lane, per component (the predictors come from a scan over lanes)
    const uint32_t dv = is_dc ? ((uint32_t)val1 & 0xffffu) : 0u;
    const uint32_t dvv = dv | (dv << 16);
    O.dcA = pk_add16(O.dcA, dvv & S.mA);
    O.dcB = pk_add16(O.dcB, dvv & S.mB);
    S.zb = done ? 63 : S.zb;
    S.ra = done ? nx.y : S.ra;
    S.x = done ? nx.x : S.x;
    S.mA = done ? nx.z : S.mA;
    S.mB = done ? nx.w : S.mB;
    D = add_flag(D, done);
    return __builtin_amdgcn_perm(entB, entA, 0x05040100u);                  // low halves of both: A | B << 16
}

// Decodes from (p, c, z) until end_bit or until the segment's last data unit is complete (D == D_end).
// Every active lane takes exactly one step per iteration and a step is one word of the staging buffer whether it holds one
// symbol or two, so the fill is the same in all lanes still decoding: the seven steps of a group are unrolled (their rows of the
// staging buffer are constants) and the group is flushed by the whole wave at once.
__device__ __forceinline__ void write_span(const PhaseCtx &P, pjd_gptr wave_words, uint32_t col,
                                           uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit,
                                           uint32_t &err, uint32_t &D, uint32_t D_end, OutCtx &O)
{
    BitWin2 w;
    w.init(wave_words, col, p);
    WState S;
    S.p = p; S.zb = 63 - (int)z; S.emax = 0; S.umin = 0xffffffffu;
    {
        const uint4 cur = lds_u32x4(P.self(P.dus1 - c));
        S.x = cur.x; S.ra = cur.y; S.mA = cur.z; S.mB = cur.w;
    }
    O.stage[0] = z;                                                          // head of group 0: no unit completed yet, the first entry fills from slot z
    O.n = 2;
    bool running = S.p < end_bit && D < D_end;
    while (running) {                                                        // one group per iteration
#pragma unroll
        for (int k = 1; k < PJD_STAGE_ENTRIES / 2; k++) {
            if (running) {
                O.stage[k * 64] = (k & 1) ? write_step<true>(P.lbase, w, S, D, O, end_bit) : write_step<false>(P.lbase, w, S, D, O, end_bit);
                O.n += 2;
                running = S.p < end_bit && D < D_end;
            }
            if (!(k & 1) || k == PJD_STAGE_ENTRIES / 2 - 1) w.service();     // the window: after every second step, and before the next group
        }
        if ((O.n & (PJD_STAGE_ENTRIES - 1)) == 0) {                          // the group is full
            stage_flush(O, O.n - PJD_STAGE_ENTRIES);
            if (running) {
                if (O.n + PJD_STAGE_ENTRIES > O.cap) { O.overflow = 1; running = false; }
                else {
                    O.stage[0] = ((D - O.D_in) << 8) | (63u - (uint32_t)S.zb);      // head of the next group: where its first entry stands
                    O.n += 2;
                }
            }
        }
    }
    if ((O.n & (PJD_STAGE_ENTRIES - 1)) > 2) stage_flush(O, O.n & ~(uint32_t)(PJD_STAGE_ENTRIES - 1));
    else if ((O.n & (PJD_STAGE_ENTRIES - 1)) == 2) O.n -= 2;                 // a head without a step behind it is not part of the stream
    err = (S.emax >= (PJD_LUT_BADSYM << 12) || S.umin <= 14u) ? 1u : 0u;
    p = S.p;
    c = P.dus1 - ((S.ra - P.xbase) >> 4);
    z = 63u - (uint32_t)S.zb;
}

// ---------------------------------------------------------------------------------------------
// The CAREFUL write pass of one lane: run again, symbol by symbol with every check of the reference in the reference's order,
// for a lane whose fast pass saw something invalid (bad symbol or size, run past slot 63) or that reaches the end of the
// bitstream before the picture is complete.  It stops AT the offending symbol exactly as decode_MCU_component does
// (reference src/jpeg_scanner.cpp:469-518): nothing of that symbol is stored, everything before it is; an error inside a unit's
// AC part leaves the unit with what it has (closed here by an end-of-block entry), an error in the DC symbol leaves the unit
// untouched.  Rare by construction: step words go straight to HBM, two symbols to a word wherever the first leaves its unit open.
//   eof_rel: bits from the lane's first byte to the end of the stream, or ~0 if the stream does not end in this lane's segment:
//            running out of bits is get_next_symbol's 0xFF / read_bits' -1 (reference src/headers/jpeg.h:91-113)
struct Careful {
    // in: where the lane's marks start (the fast pass wrote them against ITS layout -- pairs share a step word there -- so they are written again)
    PjdDevMark *marks;
    uint32_t mark_next, mark_D, ru, lane_q;
    // out
    uint32_t n;            // slots written (heads included; two per step word)
    uint32_t ovf;          // a step word did not fit the lane's region (never, by the planner's bound: the picture goes to the exact kernel)
    uint32_t cls;          // PJD_ST_* of the error, 0: none found
    uint32_t p_err, D_err, in_dc;
    uint32_t dcA, dcB;
};

__device__ __forceinline__ void careful_span(const PhaseCtx &P, pjd_gptr wave_words, uint32_t col, uint32_t p, uint32_t c, uint32_t z,
                                          uint32_t end_bit, uint32_t eof_rel, uint32_t D, uint32_t D_end, uint16_t *region, uint32_t cap, Careful &R)
{
    const uint32_t D_in = D;
    uint32_t *region32 = reinterpret_cast<uint32_t *>(region);
    BitWin w;
    w.init(wave_words, col, p);
    uint4 cur = lds_u32x4(P.self(P.dus1 - c));      // .x tables, .y own record, .z / .w DC-sum selectors of the current unit
    int zb = 63 - (int)z;
    R.n = 0; R.ovf = 0; R.cls = 0; R.p_err = 0; R.D_err = 0; R.in_dc = 0; R.dcA = 0; R.dcB = 0;
    const bool at_end = eof_rel != 0xffffffffu;
    // A symbol that leaves its unit open waits one turn in its step word: the symbol after it (an AC symbol of the same unit) joins it
    // as entry B.  Never more steps than the fast pass takes for the same symbols (its pairs are a subset of these), which is what the
    // lane's region is sized for.
    bool open = false;                                                           // the last step word written has its B half free
    uint32_t open_at = 0, open_ent = 0;
    while (D < D_end && (p < end_bit || at_end)) {
        const bool is_dc = zb == 63;
        if (!open) {
            if ((R.n & (PJD_GROUP - 1)) == 0 && R.n + 2 <= cap) {                // a group begins: its head (pjd_internal.h)
                region32[R.n >> 1] = ((D - D_in) << 8) | (63u - (uint32_t)zb);
                R.n += 2;
            }
            if (is_dc && D == R.mark_D) {                                        // this unit opens an IDCT workgroup's range (as write_step)
                PjdDevMark m;
                m.lane = R.lane_q; m.ent_off = R.n;
                m.acc[0] = (uint16_t)R.dcA; m.acc[1] = (uint16_t)(R.dcA >> 16); m.acc[2] = (uint16_t)R.dcB; m.pad_ = 0;
                R.marks[R.mark_next++] = m;
                R.mark_D += R.ru;
            }
        }
        const uint32_t pk = w.peek();
        const uint4 nx = lds_u32x4(cur.y);
        const uint32_t e = lut_lookup(P.lbase, is_dc ? (cur.x & 0xffffu) : (cur.x >> 16), pk);
        const uint32_t used = PJD_LUT_USED(e), size = PJD_LUT_SIZE(e), adv = PJD_LUT_ADV(e);
        const uint32_t vbits = size >= PJD_LUT_BADSYM ? 0u : size, codelen = used - vbits;
        const uint32_t left = !at_end ? 64u : (eof_rel > p ? eof_rel - p : 0u);
        uint32_t bad = 0;
        if (size == PJD_LUT_BADSYM || codelen > left) bad = is_dc ? PJD_ST_DC_SYM : PJD_ST_AC_SYM;           // :470 / :490
        else if (is_dc) {
            if (size == PJD_LUT_BADLEN) bad = PJD_ST_DC_LEN;                                                    // :474
            else if (vbits > left - codelen) bad = PJD_ST_DC_BITS;                                              // :480
        } else if (!(e & PJD_LUT_EOB)) {
            if (zb - (int)adv <= -2) bad = PJD_ST_AC_RUN;                                                       // :500
            else if (size == PJD_LUT_BADLEN) bad = PJD_ST_AC_LEN;                                               // :506
            else if (vbits > left - codelen) bad = PJD_ST_AC_BITS;                                              // :512
        }
        if (bad) {
            R.cls = bad; R.p_err = p; R.D_err = D; R.in_dc = is_dc ? 1u : 0u;
            // the unit keeps what it has: an EOB closes it (its group head, if it opens a group, was written above)
            if (!is_dc) {
                if (open) region32[open_at] = open_ent | (PJD_ENT_EOB << 16);
                else if (R.n + 2 <= cap) { region32[R.n >> 1] = PJD_ENT_EOB | (PJD_ENT_NONE << 16); R.n += 2; }
                else R.ovf = 1;
            }
            break;
        }
        const int val = jpeg_extend(__builtin_amdgcn_ubfe(pk, 32u - used, size), size);
        w.drop(used);
        p += used;
        zb -= (int)adv;
        const bool done = zb < 0;
        const uint32_t ent = is_dc ? ((uint32_t)val & 0xffffu) : ((((uint32_t)val << 5) | (adv & 31u)) & 0xffffu);
        if (open) { region32[open_at] = open_ent | (ent << 16); open = false; }
        else if (R.n + 2 <= cap) {
            region32[R.n >> 1] = ent | (PJD_ENT_NONE << 16);
            open = !done; open_at = R.n >> 1; open_ent = ent;
            R.n += 2;
        } else R.ovf = 1;
        const uint32_t dv = is_dc ? ((uint32_t)val & 0xffffu) : 0u, dvv = dv | (dv << 16);
        R.dcA = pk_add16(R.dcA, dvv & cur.z);
        R.dcB = pk_add16(R.dcB, dvv & cur.w);
        if (done) { zb = 63; cur = nx; D++; }
    }
    if ((R.n & (PJD_GROUP - 1)) == 2) R.n -= 2;                                  // a head without a step behind it
}

// ---------------------------------------------------------------------------------------------
// Per-wave set-up.
// ---------------------------------------------------------------------------------------------
struct LaneGeom {
    bool valid, seg_first, seg_last;
    uint32_t q;                    // global lane index
    uint32_t seg;                  // global segment index
    uint32_t end_bit;              // end of the subsequence, bits from the lane's first byte
    uint32_t seg_end_bit;          // end of the restart segment, same origin
    uint32_t base_bit;             // the lane's first byte, in bits relative to the image's ecs
    pjd_gptr words;                // wave-uniform: row 0 of the wave's transposed word rows / the picture's bitstream
    uint32_t col;                  // byte offset of the lane's word 0 from `words`
};

extern __shared__ __attribute__((aligned(16))) uint8_t pjd_huff_lds[];   // [tables][wave areas PJD_HUFF_WAVES x PJD_WAVE_LDS][phase tables PJD_HUFF_WAVES x PJD_PHASE_LDS][ticket]

// Lanes have different origins, so states are exchanged as bit positions relative to the image's ecs.
struct WaveState { uint32_t p_img, cz, cnt; };

// Re-sync rounds inside one wave.  `changed`: this lane's exit state is new to its successor.  Lane 0 takes its
// predecessor's exit from (ext_p, ext_cz) in the first round if `ext_new`.
__device__ __forceinline__ bool wave_rounds(const PhaseCtx &P, const LaneGeom &g, const ChkCtx &K, WalkBuf &wb, WaveState &S, BCache &Bc, uint32_t changed,
                                            uint32_t ext_p, uint32_t ext_cz, bool ext_new, uint32_t &err_acc,
                                            unsigned long long *stats, int stat_base, uint32_t *rdbg)
{
    const uint32_t l = threadIdx.x & 63;
    for (int iter = 0; iter < PJD_SYNC_MAX_ITERS; iter++) {
#if PJD_TAIL_PRIO
        // A wave that needs more than one round is a straggler: everything it still has to do (more rounds with few active lanes,
        // the write pass) ends its picture's chain and, with batches in flight, the batch.  Let it issue ahead of the bulk work
        // (first passes of other workgroups and other batches) that shares its SIMD.
        if (iter == 1) __builtin_amdgcn_s_setprio(2);
        if (iter == 4) __builtin_amdgcn_s_setprio(3);
#endif
        uint32_t pp = __shfl_up(S.p_img, 1), pcz = __shfl_up(S.cz, 1), pch = __shfl_up(changed, 1);
        if (l == 0) { pp = ext_p; pcz = ext_cz; pch = (ext_new && iter == 0) ? 1u : 0u; }
        const bool act = g.valid && !g.seg_first && pch != 0;
        const uint64_t act_mask = __ballot(act);
        if (!act_mask) return true;
        const uint64_t tr0 = rdbg ? __builtin_amdgcn_s_memrealtime() : 0;
        if ((uint32_t)__popcll(act_mask) <= K.walk_max) {
            // few lanes left: the wave walks them (and whatever their new exit states set in motion) one after the other
            uint64_t mask = act_mask;
            uint32_t walked = 0;
            const uint32_t chk_bits = rfl(K.chk_bits);
            const uint64_t can = __ballot(g.valid && !g.seg_first);
            while (mask) {
                uint32_t a = (uint32_t)__builtin_ctzll(mask);
                mask &= mask - 1;
                uint32_t ep = rfl(ext_p), ecz = rfl(ext_cz);
                if (a) { ep = (uint32_t)__builtin_amdgcn_readlane((int)S.p_img, (int)(a - 1)); ecz = (uint32_t)__builtin_amdgcn_readlane((int)S.cz, (int)(a - 1)); }
                for (;;) {
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)g.base_bit, (int)a);
                    const uint32_t endb = (uint32_t)__builtin_amdgcn_readlane((int)g.end_bit, (int)a);
                    uint32_t p = ep - base, c = ecz >> 8, z = ecz & 255u, ndu = 0, j;
                    uint32_t *st_col = K.state - l + a, *rem_col = K.rem - l + a;
                    const uint32_t Bst_a = (uint32_t)__builtin_amdgcn_readlane((int)Bc.st, (int)a);
                    uint32_t oa_st, oa_rem;
                    const int res = walk_lane(P, wb, l, base, p, c, z, endb, ndu, st_col, rem_col, chk_bits, j, Bst_a, oa_st, oa_rem);
                    walked++;
                    const uint32_t op = (uint32_t)__builtin_amdgcn_readlane((int)S.p_img, (int)a), ocz = (uint32_t)__builtin_amdgcn_readlane((int)S.cz, (int)a);
                    uint32_t np = p + base, ncz = (c << 8) | z;
                    if (res == SPAN_MERGED_B) {
                        // the walk met the lane's cached older trajectory at checkpoint 1: that one is the lane's trajectory now, and the
                        // one it replaces becomes the cache
                        const uint32_t Brem_a = (uint32_t)__builtin_amdgcn_readlane((int)Bc.rem, (int)a);
                        np = (uint32_t)__builtin_amdgcn_readlane((int)Bc.p, (int)a); ncz = (uint32_t)__builtin_amdgcn_readlane((int)Bc.cz, (int)a);
                        ndu += Brem_a;
                        if (l == 0) { st_col[64] = Bst_a; rem_col[64] = Brem_a; st_col[128] = 0xffffffffu; st_col[192] = 0xffffffffu; }
                    } else if (l == 0) for (uint32_t i = 1; i < j; i++) rem_col[i * 64] = ndu - rem_col[i * 64];      // chk_finish
                    if (l == a) {
                        S.cnt = ndu;
                        if (oa_st != 0xffffffffu || res == SPAN_MERGED_B) { Bc.st = oa_st; Bc.rem = oa_rem; Bc.p = op; Bc.cz = ocz; }
                    }
                    if (res == SPAN_MERGED || (np == op && ncz == ocz)) break;
                    if (l == a) { S.p_img = np; S.cz = ncz; }
                    // the successor has a new entry state: go on into it
                    if (a == 63) break;
                    if (!((can >> (a + 1)) & 1ull)) break;
                    a++; mask &= ~(1ull << a);
                    ep = np; ecz = ncz;
                }
            }
            if (stats && l == 0) { atomicAdd(stats + stat_base, 1ull); atomicAdd(stats + stat_base + 1, (unsigned long long)walked); atomicAdd(stats + PJD_STAT_WALKS, 1ull); atomicAdd(stats + PJD_STAT_WALKS + 1, (unsigned long long)walked); }
            if (rdbg && l == 0 && iter < 24) rdbg[iter] = ((0x80u | (walked & 0x7fu)) << 24) | ((uint32_t)(__builtin_amdgcn_s_memrealtime() - tr0) & 0xffffffu);   // walked lanes
            return true;
        }
        if (stats && l == 0) { atomicAdd(stats + stat_base, 1ull); atomicAdd(stats + stat_base + 1, (unsigned long long)__popcll(act_mask)); }
        changed = 0;
        if (act) {
            uint32_t p = pp - g.base_bit, c = pcz >> 8, z = pcz & 255, ndu = 0, j, oa_st, oa_rem;
            const int res = sync_span<true>(P, g.words, g.col, p, c, z, g.end_bit, ndu, K, j, Bc.st, oa_st, oa_rem);
            const uint32_t op = S.p_img, ocz = S.cz;
            if (res == SPAN_MERGED_B) {
                // met the cached older trajectory at checkpoint 1: it is the lane's trajectory now (its later checkpoints are not known)
                ndu += Bc.rem;
                K.state[64] = Bc.st; K.rem[64] = Bc.rem; K.state[128] = 0xffffffffu; K.state[192] = 0xffffffffu;
                if (Bc.p != S.p_img || Bc.cz != S.cz) { S.p_img = Bc.p; S.cz = Bc.cz; changed = 1; }
            } else {
                chk_finish(K, j, ndu);           // also after a merge: ndu then includes the units still to come
                if (res != SPAN_MERGED) {
                    const uint32_t np = p + g.base_bit, ncz = (c << 8) | z;
                    if (np != S.p_img || ncz != S.cz) { S.p_img = np; S.cz = ncz; changed = 1; }
                }
            }
            S.cnt = ndu;
            // the trajectory this pass left at checkpoint 1 (with the exit it had) is the cache now
            if (oa_st != 0xffffffffu || res == SPAN_MERGED_B) { Bc.st = oa_st; Bc.rem = oa_rem; Bc.p = op; Bc.cz = ocz; }
        }
        if (rdbg && l == 0 && iter < 24) rdbg[iter] = ((uint32_t)__popcll(act_mask) << 24) | ((uint32_t)(__builtin_amdgcn_s_memrealtime() - tr0) & 0xffffffu);
    }
    return !__any(changed != 0);
}

// ---------------------------------------------------------------------------------------------
// Segmented combine used for data-unit counts: element = (value, head flag); a head resets.  Sums SATURATE at `cap`
// (= the image's unit count + 1, below 2^28): speculative lanes also count "units" in whatever bytes follow the picture's last
// unit, and a long such tail must read as "past the end" everywhere, never wrap back into the picture's range (a look-back
// descriptor carries 28 bits).  min(a + b, cap) is associative on non-negative values, so scan and look-back stay exact below cap.
__device__ __forceinline__ void seg_combine(uint32_t av, uint32_t af, uint32_t &bv, uint32_t &bf, uint32_t cap)   // b = a (+) b
{
    if (!bf) { const uint32_t t = bv + av; bv = t < cap ? t : cap; }
    bf |= af;
}

__device__ __forceinline__ void wave_seg_scan(uint32_t &v, uint32_t &f, uint32_t cap)      // inclusive, 64 lanes
{
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ov = __shfl_up(v, off), of = __shfl_up(f, off);
        if ((int)lane >= off) seg_combine(ov, of, v, f, cap);
    }
}

// ---------------------------------------------------------------------------------------------
// Published words.  All are self-contained 64-bit values (state or descriptor + flag bits), so relaxed
// agent-scope atomics are enough.  wave_gen / wave_desc / ticket are zeroed before every launch.
#define OP_FLAG        (1ull << 63)
#define OP_ST_AGG      (1ull << 62)            // descriptor holds this wave's aggregate
#define OP_ST_PFX      (2ull << 62)            // descriptor holds the inclusive prefix up to this wave
#define OP_ST_MASK     (3ull << 62)
#define OP_POISON      (1ull << 61)
#define OP_HEAD        (1ull << 60)
#define OP_SPIN_LIMIT  (1u << 22)

__device__ __forceinline__ uint64_t op_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void op_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint64_t op_desc(uint64_t status, uint32_t v, uint32_t f, bool poison)
{
    return status | (poison ? OP_POISON : 0ull) | (f ? OP_HEAD : 0ull) | (uint64_t)(v & 0x0fffffffu);
}

// v_readlane / v_readfirstlane return int: widen through uint32_t so that a low half >= 2^31 does not
// sign-extend over the high half
__device__ __forceinline__ uint64_t readfirstlane_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// lane 0 of the wave waits for a flagged word; the value (flag stripped) is returned to every lane
__device__ __forceinline__ uint64_t op_wait_flag(const uint64_t *p, bool &timeout)
{
    uint64_t v = 0;
    if ((threadIdx.x & 63) == 0)
        for (uint32_t it = 0; it < OP_SPIN_LIMIT; it++) {
            v = op_load(p);
            if (v & OP_FLAG) break;
            __builtin_amdgcn_s_sleep(8);
        }
    v = readfirstlane_u64(v);
    timeout = timeout || !(v & OP_FLAG);
    return v & ~OP_FLAG;
}

// ---------------------------------------------------------------------------------------------
// The pull back end (pjd_internal.h): what the LAST wave of a picture to finish does for the picture, with its 64 lanes -- the
// verdict (pjd_k_image_verdict), the DC predictors at every lane start (pjd_k_lane_dc_local / _carry: a segmented scan over the
// picture's lanes, here in chunks of 64 with absolute results) -- and the hand-over of the picture's back-end ranges.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void finish_picture(const PjdDevBatch &B, uint32_t image, uint32_t l)
{
    const PjdDevImage &im = B.images[image];
    __atomic_thread_fence(__ATOMIC_ACQUIRE);                    // what the picture's other waves wrote
    if (l == 0) {
        const int32_t st = B.status[image];
        if (!(st & PJD_STW_NEEDS_EXACT)) {
            const PjdDevImState s = B.imstate[image];
            const bool has_err = s.err_key != ~0ull;
            const uint32_t err_pos = (uint32_t)(s.err_key >> 32);
            if (s.flag_pos != 0xffffffffu && (!has_err || s.flag_pos <= err_pos)) B.status[image] = st | PJD_STW_NEEDS_EXACT;
            else if (has_err) B.status[image] = (int32_t)((s.err_key >> 1) & 7u);
        }
    }
    uint32_t cy = 0, ccb = 0, ccr = 0;                         // predictors entering the current chunk of lanes (wave-uniform)
    const uint32_t n_lane = rfl(im.n_lane), lane_base = rfl(im.lane_base);
    for (uint32_t base = 0; base < n_lane; base += 64) {
        const bool on = base + l < n_lane;
        uint32_t vy = 0, vcb = 0, vcr = 0, head = 0;
        if (on) {
            const PjdDevLaneInfo li = B.lane_info[lane_base + base + l];
            vy = li.dc_sum[0]; vcb = li.dc_sum[1]; vcr = li.dc_sum[2]; head = li.first_du >> 31;
        }
        uint32_t f = head;
        for (int off = 1; off < 64; off <<= 1) {                // inclusive segmented scan: a lane that starts a restart segment resets
            const uint32_t ay = __shfl_up(vy, off), acb = __shfl_up(vcb, off), acr = __shfl_up(vcr, off), af = __shfl_up(f, off);
            if ((int)l >= off) {
                if (!f) { vy += ay; vcb += acb; vcr += acr; }
                f |= af;
            }
        }
        // predictors entering this lane: zero at a segment head; else the inclusive result of the lane before it, plus the chunk's
        // carry-in unless a head lies between the chunk start and this lane
        const uint32_t py = __shfl_up(vy, 1), pcb = __shfl_up(vcb, 1), pcr = __shfl_up(vcr, 1), pf = __shfl_up(f, 1);
        if (on) {
            PjdDevLaneDc d;
            d.dc_in[0] = d.dc_in[1] = d.dc_in[2] = 0;
            d.abs = 1;
            if (!head) {
                uint32_t qy = cy, qcb = ccb, qcr = ccr;
                if (l > 0) { qy = (pf ? 0u : cy) + py; qcb = (pf ? 0u : ccb) + pcb; qcr = (pf ? 0u : ccr) + pcr; }
                d.dc_in[0] = (uint16_t)qy; d.dc_in[1] = (uint16_t)qcb; d.dc_in[2] = (uint16_t)qcr;
            }
            B.lane_dc[lane_base + base + l] = d;
        }
        const uint32_t ly = __shfl(vy, 63), lcb = __shfl(vcb, 63), lcr = __shfl(vcr, 63), lf = __shfl(f, 63);
        cy = (lf ? 0u : cy) + ly; ccb = (lf ? 0u : ccb) + lcb; ccr = (lf ? 0u : ccr) + lcr;
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);                    // status and predictors before the ranges appear in the list
    const uint32_t n_iwg = rfl(im.n_iwg), iwg_base = rfl(im.iwg_base);
    uint32_t base = 0;
    if (l == 0) base = atomicAdd(B.ready_tail, n_iwg);
    base = rfl(base);
    for (uint32_t k = l; k < n_iwg; k += 64) __hip_atomic_store(B.ready_list + base + k, iwg_base + k + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// The kernel.
//
// With one launch per stage every stage waits for the slowest wave of the whole batch, and that wave is
// slow by nature: somewhere in 10^5 subsequences there is a stretch of a few KB on which speculative
// decoders do not fall into step, and it can only be walked sequentially.  Here a wave depends only on waves
// of its OWN image that precede it, so everything else proceeds to the write pass while the stragglers finish:
//
//   * workgroups take their index from a ticket counter, so every wave a wave waits for has already started
//     (waits can always be satisfied; they are bounded anyway and poison the image on time-out);
//   * generation 0 of a wave = the speculative exit of its last lane, the guess its successor's first lane
//     bridges from; generation 1 = that lane's exit after the wave's own re-sync rounds; generation g + 1 = its exit after the
//     wave compared the entry it used with generation g of its predecessor and re-bridged if they differ (PJD_GENS
//     generations: a chain that crosses k wave boundaries needs k + 2) -- never a chain
//     across the image; at the very end a wave checks that the entry it used is what its predecessor finally
//     produced (else: exact kernel);
//   * data-unit counts (segmented: a restart segment's first subsequence resets to the segment's first
//     unit) are combined with a decoupled look-back over the image's waves, 64 descriptors per step;
//   * the write pass runs from registers; tables are still in LDS, the checkpoint area becomes the staging buffer.
// hwg_base / ticket: the launch covers workgroups hwg_base .. of PjdDevBatch::hwgs and draws their indices from its own ticket
// counter (one launch for the whole batch: 0 and B.ticket; one per picture group otherwise, pjd_internal.h).
#ifndef PJD_HUFF_OCC
#define PJD_HUFF_OCC 0                 // build option: waves per SIMD the register allocation must allow (0: the compiler's choice)
#endif
#if PJD_HUFF_OCC
#define PJD_HUFF_OCC_ATTR __attribute__((amdgpu_waves_per_eu(PJD_HUFF_OCC, PJD_HUFF_OCC)))
#else
#define PJD_HUFF_OCC_ATTR
#endif
__global__ __launch_bounds__(PJD_HUFF_THREADS) PJD_HUFF_OCC_ATTR void pjd_k_huff_lanes(PjdDevBatch B, uint32_t hwg_base, uint32_t *__restrict__ ticket)
{
    const uint32_t t = threadIdx.x, l = t & 63, wi = t >> 6;
    uint8_t *lds = pjd_huff_lds;
    uint8_t *areas = lds + B.max_lut_bytes;
    uint8_t *phase_tabs = areas + PJD_HUFF_WAVES * PJD_WAVE_LDS;
    uint32_t *tick = reinterpret_cast<uint32_t *>(phase_tabs + PJD_HUFF_WAVES * PJD_PHASE_LDS);
    if (t == 0) *tick = hwg_base + atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t gidx = rfl(*tick);
    const PjdDevHuffWg wg = B.hwgs[gidx];
    {   // the table set -> LDS (16 B per thread per step, 4 loads in flight per thread)
        const PjdDevTset &T = B.tsets[wg.tset];
        const uint32_t n16 = T.lut_bytes / 16;
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(B.luts) + T.lut_off16;
        uint4 *tdst = reinterpret_cast<uint4 *>(lds);
        for (uint32_t i0 = t; i0 < n16; i0 += 4 * PJD_HUFF_THREADS) {
            const uint32_t i1 = i0 + PJD_HUFF_THREADS, i2 = i1 + PJD_HUFF_THREADS, i3 = i2 + PJD_HUFF_THREADS;
            uint4 v0 = tsrc[i0], v1, v2, v3;
            if (i1 < n16) v1 = tsrc[i1];
            if (i2 < n16) v2 = tsrc[i2];
            if (i3 < n16) v3 = tsrc[i3];
            tdst[i0] = v0;
            if (i1 < n16) tdst[i1] = v1;
            if (i2 < n16) tdst[i2] = v2;
            if (i3 < n16) tdst[i3] = v3;
        }
    }
    const bool wave_on = wi < wg.n_waves;
    const uint32_t w = wg.first_wave + (wave_on ? wi : 0u);
    const PjdDevHuffWave hw = B.hwaves[w];
    const PjdDevImage &im = B.images[hw.image];
    PhaseCtx P;                                                // table offsets of this wave's image (blob at LDS offset 0)
    {
        const uint32_t nl = rfl(im.n_luma), dus = rfl(im.dus_per_mcu), nc = dus - nl;
        const uint32_t c1 = nc >= 1 ? 1u : 0u, c2 = nc >= 2 ? 2u : c1;
        const uint32_t lb = lds_abs(lds);                   // the whole layout stays below 64 KB: addresses fit the 16-bit halves
        P.lbase = lb;
        P.tY  = rfl((lb + (uint32_t)im.tbl_slot[0][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[0][1] * PJD_L1_BYTES) << 16);
        P.tC1 = rfl((lb + (uint32_t)im.tbl_slot[c1][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[c1][1] * PJD_L1_BYTES) << 16);
        P.tC2 = rfl((lb + (uint32_t)im.tbl_slot[c2][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[c2][1] * PJD_L1_BYTES) << 16);
        P.nc = nc; P.dus1 = dus - 1;
        P.xbase = lds_abs(phase_tabs) + wi * PJD_PHASE_LDS;
    }
    const uint32_t dus = P.dus1 + 1;
    P.build(l);                                                // this wave's phase table (read by this wave only)
    __syncthreads();                                           // the last barrier: from here on waves run on their own
    if (!wave_on) return;
    // beside the pull back end (pjd_internal.h) every wave of the entropy decoder issues ahead of the back end's: its chains are the
    // critical path, the back end fills what they leave
    if (B.pull) __builtin_amdgcn_s_setprio(1);

    const uint64_t ts0 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    const uint64_t tc0 = B.dbg ? __builtin_amdgcn_s_memtime() : 0;          // shader clock: with ts0..ts5 (100 MHz) it gives the clock the chip held
    uint64_t *genA = B.wave_gen, *genB = B.wave_gen + B.n_hwave;      // generation g of wave w: B.wave_gen[g * n_hwave + w]
    // ---- lane geometry
    LaneGeom g;
    g.valid = l < hw.n_lanes;
    g.q = hw.first_lane + l;
    g.seg_first = g.seg_last = false;
    g.seg = 0; g.end_bit = g.seg_end_bit = g.base_bit = 0;
#if PJD_DIRECT_ECS
    g.words = (pjd_gptr)readfirstlane_u64((uint64_t)(B.ecs + im.ecs_off));
    g.col = 0;
#else
    g.words = (pjd_gptr)readfirstlane_u64((uint64_t)(B.words + (size_t)w * B.word_rows * 64));
    g.col = l * 4;
#endif
    uint32_t seg_first_du = 0, seg_n_du = 0;
    const uint32_t sub_bytes = rfl(im.sub_bytes);        // of this wave's image
    if (g.valid) {
        const PjdDevSub sb = B.lanes[g.q];
        g.seg = sb.seg & 0x7fffffffu;
        g.seg_first = (sb.seg >> 31) != 0;
        const PjdDevSegment sg = B.segs[g.seg];
        const uint32_t end_byte = sb.byte_start + sub_bytes < sg.byte_end ? sb.byte_start + sub_bytes : sg.byte_end;
        g.seg_last = end_byte == sg.byte_end;
        g.base_bit = sb.byte_start * 8;
#if PJD_DIRECT_ECS
        g.col = sb.byte_start;
#endif
        g.end_bit = (end_byte - sb.byte_start) * 8;
        g.seg_end_bit = (sg.byte_end - sb.byte_start) * 8;
        seg_first_du = sg.first_du; seg_n_du = sg.n_du;
    }
    uint32_t *area = reinterpret_cast<uint32_t *>(areas + wi * PJD_WAVE_LDS);
    ChkCtx K;
    K.state = area + l;
    K.rem = area + PJD_NCHK * 64 + l;
    K.chk_bits = sub_bytes * 8 / PJD_NCHK;
    K.walk_max = rfl(im.walk_max);
    for (int j = 0; j < PJD_NCHK; j++) { K.state[j * 64] = 0xffffffffu; K.rem[j * 64] = 0; }
    const bool first_is_head = __shfl((uint32_t)g.seg_first, 0) != 0;     // lane 0 starts a restart segment: no predecessor wave
    const uint32_t last_lane = hw.n_lanes - 1;
    bool dead = false;                                   // wave-uniform: a wait timed out or a predecessor is poisoned
    uint32_t flag = 0;                                   // per lane: bit per PJD_FLAG_* reason

    // ---- A: speculative pass
    WaveState S;
    {
        uint32_t p = 0, c = 0, z = 0, ndu = 0, j = 1;
        if (g.valid) {
            uint32_t oa, orem;
            sync_span<false>(P, g.words, g.col, p, c, z, g.end_bit, ndu, K, j, 0xffffffffu, oa, orem);
            chk_finish(K, j, ndu);
        }
        S.p_img = p + g.base_bit; S.cz = (c << 8) | z; S.cnt = ndu;
    }
    if (l == last_lane) op_store(genA + w, pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255) | OP_FLAG);
    const uint64_t ts1 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- R: re-sync rounds; lane 0 bridges from the speculative exit of the previous wave's last lane
    uint64_t entry_used = 0;
    uint32_t err_acc = 0;
    bool ok = true;
    WalkBuf wb;
    wb.ecs = (pjd_gptr)readfirstlane_u64((uint64_t)(B.ecs + im.ecs_off));
    wb.clamp = rfl(im.ecs_len) + 40u;                      // >= 48 zero bytes follow every stream (pjd_plan.cpp)
    wb.cb = 0; wb.cw = 0; wb.cn = 0; wb.have = 0;
    if (!first_is_head) entry_used = op_wait_flag(genA + w - 1, dead);
    BCache Bc;
    Bc.st = 0xffffffffu; Bc.rem = 0; Bc.p = 0; Bc.cz = 0;
    ok = wave_rounds(P, g, K, wb, S, Bc, g.valid ? 1u : 0u, (uint32_t)entry_used,
                     (((uint32_t)(entry_used >> 32) & 255) << 8) | ((uint32_t)(entry_used >> 40) & 255), !first_is_head && !dead,
                     err_acc, B.stats, 0, B.dbg ? B.dbg + (size_t)w * 32 + 8 : nullptr);
    if (l == last_lane) op_store(genB + w, pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255) | OP_FLAG);
    const uint64_t ts2 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- stitch: compare the entry this wave used with its predecessor's next generation; redo the bridges from it if it differs
    //      (a walk of lane 0's chain, usually a few hundred bytes); publish the own exit as the next generation.  Every generation
    //      resolves one more wave boundary that a chain of non-merging lanes crosses.
    for (uint32_t gen = 1; gen + 1 < PJD_GENS; gen++) {
        if (!first_is_head && !dead) {
            const uint64_t truth = op_wait_flag(B.wave_gen + (size_t)gen * B.n_hwave + w - 1, dead);
            if (!dead && truth != entry_used) {
                entry_used = truth;
                ok = wave_rounds(P, g, K, wb, S, Bc, 0u, (uint32_t)truth,
                                 (((uint32_t)(truth >> 32) & 255) << 8) | ((uint32_t)(truth >> 40) & 255), true,
                                 err_acc, B.stats, 2, nullptr) && ok;
            }
        }
        if (l == last_lane) op_store(B.wave_gen + (size_t)(gen + 1) * B.n_hwave + w, pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255) | OP_FLAG);
    }
    if (!ok) flag |= 1u << PJD_FLAG_NOSYNC;
    const uint64_t ts3 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- C: counts.  Inside the wave: segmented scan of data units ...
    const uint32_t du_cap = rfl(im.n_du) + 1;                 // "past the last unit"; the planner keeps it below 2^28
    uint32_t cnt = 0, v = 0, f = 0;
    if (g.valid) {
        cnt = S.cnt;
        v = cnt < du_cap ? cnt : du_cap;
        if (g.seg_first) { f = 1; v = seg_first_du + v < du_cap ? seg_first_du + v : du_cap; }
    }
    wave_seg_scan(v, f, du_cap);
    const uint32_t agg_v = __shfl(v, 63), agg_f = __shfl(f, 63);
    // ... across the image's waves: decoupled look-back
    uint32_t du_in = 0;
    if (w == im.hwave_base) {
        if (l == 0) op_store(B.wave_desc + w, op_desc(OP_ST_PFX, agg_v, agg_f, dead));
    } else {
        if (l == 0) op_store(B.wave_desc + w, op_desc(OP_ST_AGG, agg_v, agg_f, dead));
        uint32_t rv = 0, rf = 0;                           // combination of the descriptors gathered so far (identity)
        bool poison = false;
        int hi = (int)w - 1;
        for (;;) {
            const int j = hi - (int)l;
            const bool inside = j >= (int)im.hwave_base;
            uint64_t d = OP_ST_PFX;                        // before the image's first wave: an empty prefix
            if (inside) {
                uint32_t it = 0;
                do { d = op_load(B.wave_desc + j); if (d & OP_ST_MASK) break; __builtin_amdgcn_s_sleep(8); } while (++it < OP_SPIN_LIMIT);
            }
            const bool ready = (d & OP_ST_MASK) != 0;
            if (__any(!ready)) { dead = true; break; }
            const uint64_t pfx_mask = __ballot((d & OP_ST_MASK) == OP_ST_PFX);   // never empty past the image start
            const int k = pfx_mask ? __builtin_ctzll(pfx_mask) : 64;              // nearest lane holding a prefix
            const bool use = (int)l <= k;
            uint32_t xv = use ? (uint32_t)d & 0x0fffffffu : 0u, xf = use ? (uint32_t)((d >> 60) & 1u) : 0u;
            poison = poison || __any(use && (d & OP_POISON));
            // lane l holds wave hi-l: larger l = earlier.  Suffix-combine so that lane 0 = X_k (+) ... (+) X_0.
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t ov = __shfl_down(xv, off), of = __shfl_down(xf, off);
                if ((int)l + off < 64) seg_combine(ov, of, xv, xf, du_cap);
            }
            const uint32_t wvv = __shfl(xv, 0), wf = __shfl(xf, 0);
            // running = window (+) running
            { uint32_t nv = rv, nf = rf; seg_combine(wvv, wf, nv, nf, du_cap); rv = nv; rf = nf; }
            if (k < 64) break;
            hi -= 64;
        }
        if (poison) dead = true;
        du_in = rv;                                        // rf set: rv is an absolute index; else a count from the image's first unit (same thing)
        uint32_t iv = agg_v, ifl = agg_f;
        seg_combine(rv, rf, iv, ifl, du_cap);
        if (l == 0) op_store(B.wave_desc + w, op_desc(OP_ST_PFX, iv, ifl, dead));
    }
    const uint64_t ts4 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- W: write pass from the true entry states
    const uint32_t prev_p = __shfl_up(S.p_img, 1), prev_cz = __shfl_up(S.cz, 1);
    const uint32_t prev_v = __shfl_up(v, 1), prev_f = __shfl_up(f, 1);      // inclusive counts of the lane before: this lane's first unit
    uint32_t npair = 0;                                  // steps of this lane's write pass that held a symbol pair
    PjdDevLaneInfo li;
    li.n_ent = 0; li.first_du = g.seg_first ? PJD_LANE_SEG_FIRST : 0u; li.dc_sum[0] = li.dc_sum[1] = li.dc_sum[2] = 0; li.pad_ = 0;
    if (g.valid && !dead) {
        // first data unit of this lane = the (saturating) counts before it; not "inclusive - own count": the lane that holds the
        // picture's last unit also counts whatever follows it, and its inclusive value may have saturated
        uint32_t D;
        if (g.seg_first) D = seg_first_du;
        else if (l == 0) D = du_in;
        else if (prev_f) D = prev_v;
        else D = du_in + prev_v < du_cap ? du_in + prev_v : du_cap;
        const uint32_t D_in = D, D_end = seg_first_du + seg_n_du;
        uint32_t p, c, z;
        if (g.seg_first) { p = 0; c = 0; z = 0; }
        else {
            if (l == 0) { p = (uint32_t)entry_used - g.base_bit; c = (uint32_t)(entry_used >> 32) & 255; z = (uint32_t)(entry_used >> 40) & 255; }
            else { p = prev_p - g.base_bit; c = prev_cz >> 8; z = prev_cz & 255; }
            if (D_in < D_end && (D_in % dus) != c) flag |= 1u << PJD_FLAG_SEGMENT;      // phase must agree with the count
        }
        if (D_in < D_end && !flag) {
            OutCtx O;
            O.region = B.ent + im.ent_base + (size_t)(g.q - im.lane_base) * im.lane_cap;
            O.stage = area + l;
            O.cap = im.lane_cap;
            O.n = 0; O.npair = 0; O.dcA = 0; O.dcB = 0; O.overflow = 0;
            O.ru = im.idct_mcus * dus;
            O.marks = B.marks + im.iwg_base;
            O.lane_q = g.q;
            O.D_in = D_in;
            li.first_du |= D_in;
            {   // the next data unit that STARTS in this lane and opens an IDCT workgroup's range
                const uint32_t first_du = im.first_mcu * dus;
                const uint32_t d_next = (z == 0) ? D_in : D_in + 1;                       // first unit starting here
                const uint32_t k = (d_next - first_du + O.ru - 1) / O.ru;                 // ranges are [first_du + k * ru, ...)
                O.mark_next = k;
                O.mark_D = first_du + k * O.ru;
            }
            uint32_t err = 0;
            const uint32_t p0 = p, c0 = c, z0 = z, mark_next0 = O.mark_next, mark_D0 = O.mark_D;
            write_span(P, g.words, g.col, p, c, z, g.end_bit, err, D, D_end, O);
            li.n_ent = O.n;
            npair = O.npair;
            li.dc_sum[0] = (uint16_t)O.dcA; li.dc_sum[1] = (uint16_t)(O.dcA >> 16); li.dc_sum[2] = (uint16_t)O.dcB;
            const bool has_next = g.seg + 1 < im.seg_base + im.n_seg;
            // the lane in which the bitstream ends: too few bits for the picture is the reference's end-of-data error, not ours
            const bool eof_lane = g.seg_last && !has_next && (im.flags & PJD_IF_ENDS_STREAM) != 0;
            bool settled = false;
            if (err || (eof_lane && (D < D_end || p > g.seg_end_bit))) {
                // an entropy-coding error of the TRUE decode (this lane started from the true state): find it exactly, keep what
                // precedes it, report it by position -- the picture's verdict takes the first one (pjd_k_image_verdict)
                Careful R;
                R.marks = O.marks; R.mark_next = mark_next0; R.mark_D = mark_D0; R.ru = O.ru; R.lane_q = g.q;
                careful_span(P, g.words, g.col, p0, c0, z0, g.end_bit, eof_lane ? g.seg_end_bit : 0xffffffffu, D_in, D_end, O.region, O.cap, R);
                // the region and the lane's marks now have the careful pass's layout (one symbol per step word), error or not
                li.n_ent = R.n;
                npair = 0;
                li.dc_sum[0] = (uint16_t)R.dcA; li.dc_sum[1] = (uint16_t)(R.dcA >> 16); li.dc_sum[2] = (uint16_t)R.dcB;
                if (R.ovf) flag |= 1u << PJD_FLAG_OVERFLOW;
                else if (R.cls) {
                    const unsigned long long key = ((unsigned long long)(g.base_bit + R.p_err) << 32) | ((unsigned long long)R.D_err << 4) | (R.cls << 1) | R.in_dc;
                    atomicMin(&B.imstate[hw.image].err_key, key);
                    settled = true;
                }
            }
            if (!settled) {
                if (err) flag |= 1u << PJD_FLAG_SYMBOL;
                if (O.overflow) flag |= 1u << PJD_FLAG_OVERFLOW;
                if (D == D_end) {
                    if (p > g.seg_end_bit) flag |= 1u << PJD_FLAG_SEGMENT;
                    if (has_next && ((p + 7) & ~7u) != g.seg_end_bit) flag |= 1u << PJD_FLAG_SEGMENT;
                } else {
                    if (S.p_img - g.base_bit != p || (S.cz >> 8) != c || (S.cz & 255) != z) flag |= 1u << PJD_FLAG_VERIFY;
                    if (D - D_in != S.cnt) flag |= 1u << PJD_FLAG_VERIFY;
                    if (g.seg_last) flag |= 1u << PJD_FLAG_SEGMENT;
                }
            }
        }
    }
    if (g.valid) B.lane_info[g.q] = li;
    // the entry this wave was synchronised with must be what its predecessor finally produced
    if (!first_is_head && !dead) {
        const uint64_t fin = op_wait_flag(B.wave_gen + (size_t)(PJD_GENS - 1) * B.n_hwave + w - 1, dead);
        if (fin != entry_used) flag |= 1u << PJD_FLAG_STITCH;
    }
    if (B.dbg && l == 0) {
        const uint64_t ts5 = __builtin_amdgcn_s_memrealtime();
        uint32_t *d = B.dbg + (size_t)w * 32;
        d[0] = (uint32_t)ts0; d[1] = (uint32_t)(ts1 - ts0); d[2] = (uint32_t)(ts2 - ts1); d[3] = (uint32_t)(ts3 - ts2);
        d[4] = (uint32_t)(ts4 - ts3); d[5] = (uint32_t)(ts5 - ts4); d[6] = hw.image; d[7] = hw.n_lanes;
        d[7] |= ((uint32_t)(((__builtin_amdgcn_s_memtime() - tc0) * 16) / ((ts5 - ts0) ? (ts5 - ts0) : 1)) & 0xffffffu) << 8;      // shader cycles per 10 ns, x16
    }
    if (dead) flag |= 1u << PJD_FLAG_TIMEOUT;
    // per wave: one counter per reason, and WHERE in the picture's stream the earliest unresolved thing lies (the start of the
    // flagged lane; conditions that concern the whole wave flag every lane, i.e. the wave's first).  Whether that sends the picture
    // to the exact kernel is decided per picture once all its waves have reported (pjd_k_image_verdict): what lies behind the
    // picture's first entropy-coding error is never decoded by the reference and does not count.
    uint32_t wflag = flag, went = npair, wsteps = g.valid ? li.n_ent / 2 - (li.n_ent + PJD_GROUP - 1) / PJD_GROUP : 0, wpos = (flag && g.valid) ? g.base_bit : 0xffffffffu;
    uint32_t wfill = g.valid ? (uint32_t)(((uint64_t)li.n_ent << 10) / im.lane_cap) : 0u;      // how full the lane's region got (the planner's bound at work)
    for (int off = 1; off < 64; off <<= 1) {
        wflag |= __shfl_xor(wflag, off); went += __shfl_xor(went, off); wsteps += __shfl_xor(wsteps, off);
        { const uint32_t o = __shfl_xor(wfill, off); wfill = o > wfill ? o : wfill; }
        const uint32_t o = __shfl_xor(wpos, off);
        wpos = o < wpos ? o : wpos;
    }
    if (l == 0) { atomicAdd(B.stats + PJD_STAT_ENTRIES, (unsigned long long)(went + wsteps)); atomicAdd(B.stats + PJD_STAT_STEPS, (unsigned long long)wsteps); atomicMax(B.stats + PJD_STAT_FILL, (unsigned long long)wfill); }
    if (wflag && l == 0) {
        atomicMin(&B.imstate[hw.image].flag_pos, wpos);
        for (int r = 0; r < PJD_FLAG_REASONS; r++)
            if (wflag & (1u << r)) atomicAdd(B.stats + PJD_STAT_FLAG0 + r, 1ull);
    }
    if (B.pull) {
        // the pull back end (pjd_internal.h): the picture's last wave to get here settles the picture and hands its ranges over
        __atomic_thread_fence(__ATOMIC_RELEASE);                // this wave's entries, lane_info, marks, error state
        uint32_t done = 0;
        if (l == 0) done = atomicAdd(&B.imstate[hw.image].waves_done, 1u) + 1u;
        if (rfl(done) == rfl(im.n_hwave)) finish_picture(B, hw.image, l);
    }
}

// ---------------------------------------------------------------------------------------------
// The EXACT decoder with the parallel decoder's tables: one lane per picture, the reference's decode_Huffman_data in its own
// order (reference src/jpeg_scanner.cpp:707-756) -- its restart rule (:723; or ITU T.81's with PJD_IF_STANDARD_RESTART), its
// end-of-data behaviour, its error classes and "stop at the first error, keep what was decoded" -- but every symbol is ONE lookup
// in the two-level table in LDS and the bitstream comes through a register window, instead of the literal bit-serial scan of
// pjd_k_huffman_seq.hip (which stays for table sets that have no decode table: over-subscribed codes, too many long codes).
// Still a single dependent chain per picture -- about eight times faster than the literal kernel, not a fast path.
// Output as there: coefficients in zigzag-SLOT order, absolute DC, PJD_COEF_SENTINEL for an explicit zero at slot 52.
// ---------------------------------------------------------------------------------------------
struct SeqWin {            // bit window over the picture's bitstream itself (unaligned big-endian dwords); p = absolute bit position
    pjd_gptr base;
    uint32_t hi, lo, nxt, off;
    int s;
    __device__ __forceinline__ uint32_t word(uint32_t byte_off) const
    {
        return __builtin_bswap32(reinterpret_cast<const __attribute__((address_space(1))) UnalignedU32 *>(base + byte_off)->v);
    }
    __device__ __forceinline__ void init(pjd_gptr b)      // at bit 0
    {
        base = b; hi = 0; lo = word(0); nxt = word(4); off = 8; s = 0;
    }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s); }
    __device__ __forceinline__ void drop(uint32_t n)      // n <= 32
    {
        s -= (int)n;
        if (s < 0) { s += 32; hi = lo; lo = nxt; nxt = word(off); off += 4; }
    }
};

__global__ __launch_bounds__(64) void pjd_k_huff_exact_lut(PjdDevBatch B, const uint32_t *__restrict__ image_list, const uint64_t *__restrict__ dense_base)
{
    const uint32_t ii = image_list[blockIdx.x];
    const PjdDevImage &im = B.images[ii];
    if (im.flags & PJD_IF_PROGRESSIVE) return;                    // pjd_k_progressive decodes it
    const PjdDevTset &T = B.tsets[im.tset];
    if (T.lut_bytes == 0) return;                                // no decode table for this set: pjd_k_huff_sequential takes the picture
    const uint32_t l = threadIdx.x;
    uint8_t *lds = pjd_huff_lds;
    {
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(B.luts) + T.lut_off16;
        uint4 *tdst = reinterpret_cast<uint4 *>(lds);
        for (uint32_t i = l; i < T.lut_bytes / 16; i += 64) tdst[i] = tsrc[i];
    }
    PhaseCtx P;
    {
        const uint32_t dus = im.dus_per_mcu, nc = dus - im.n_luma;
        const uint32_t c1 = nc >= 1 ? 1u : 0u, c2 = nc >= 2 ? 2u : c1;
        const uint32_t lb = lds_abs(lds);
        P.lbase = lb;
        P.tY  = (lb + (uint32_t)im.tbl_slot[0][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[0][1] * PJD_L1_BYTES) << 16;
        P.tC1 = (lb + (uint32_t)im.tbl_slot[c1][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[c1][1] * PJD_L1_BYTES) << 16;
        P.tC2 = (lb + (uint32_t)im.tbl_slot[c2][0] * PJD_L1_BYTES) | (lb + (uint32_t)im.tbl_slot[c2][1] * PJD_L1_BYTES) << 16;
        P.nc = nc; P.dus1 = dus - 1;
        P.xbase = 0;                                             // the phase table is not used here: tables by component below
    }
    __syncthreads();
    if (l != 0) return;

    int16_t *coef = B.coef + dense_base[blockIdx.x] * 64;        // slot 0 of the scratch = the first data unit this picture (or shard) decodes
    const uint32_t RI = im.restart_interval, Wr = im.ref_mcu_w_real;
    const bool std_rule = (im.flags & PJD_IF_STANDARD_RESTART) != 0;
    const uint32_t nbits = im.ecs_len * 8u;
    SeqWin w;
    w.init((pjd_gptr)(B.ecs + im.ecs_off));
    uint32_t p = 0, D = 0;
    int pred[3] = {0, 0, 0};
    int status = PJD_ST_OK;
    for (uint32_t m = im.first_mcu; m < im.last_mcu && !status; m++) {
        const uint32_t y = (m / im.mcux) * im.vs, x = (m % im.mcux) * im.hs;
        if (RI != 0 && (std_rule ? (m % RI == 0) : ((y * Wr + x) % RI == 0))) {
            pred[0] = pred[1] = pred[2] = 0;
            // BitReader::align() (reference src/headers/jpeg.h:115-121): a no-op once every byte is consumed
            if ((p >> 3) < im.ecs_len && (p & 7)) { const uint32_t a = 8u - (p & 7); w.drop(a); p += a; }
        }
        for (uint32_t k = 0; k < im.dus_per_mcu && !status; k++, D++) {
            const uint32_t r = P.dus1 - k, comp = P.comp(r), tabs = P.tabs(r);
            int16_t *unit = coef + (size_t)D * 64;
            // ---- DC (jpeg_scanner.cpp:469-486)
            {
                const uint32_t pk = w.peek();
                const uint32_t e = lut_lookup(P.lbase, tabs & 0xffffu, pk);
                const uint32_t used = PJD_LUT_USED(e), size = PJD_LUT_SIZE(e);
                const uint32_t vbits = size >= PJD_LUT_BADSYM ? 0u : size, codelen = used - vbits, left = nbits > p ? nbits - p : 0u;
                if (size == PJD_LUT_BADSYM || codelen > left) { status = PJD_ST_DC_SYM; break; }
                if (size == PJD_LUT_BADLEN) { status = PJD_ST_DC_LEN; break; }
                if (vbits > left - codelen) { status = PJD_ST_DC_BITS; break; }
                const uint32_t bits = __builtin_amdgcn_ubfe(pk, 32u - used, size), m1 = 1u << size;
                const int val = (int)bits + ((((int)bits - (int)(m1 >> 1)) >> 31) & (int)(1u - m1));
                w.drop(used); p += used;
                unit[0] = (int16_t)(val + pred[comp]);
                pred[comp] = unit[0];
            }
            // ---- AC (jpeg_scanner.cpp:488-518)
            for (uint32_t z = 1; z < 64; z++) {
                const uint32_t pk = w.peek();
                const uint32_t e = lut_lookup(P.lbase, tabs >> 16, pk);
                const uint32_t used = PJD_LUT_USED(e), size = PJD_LUT_SIZE(e), adv = PJD_LUT_ADV(e);
                const uint32_t vbits = size >= PJD_LUT_BADSYM ? 0u : size, codelen = used - vbits, left = nbits > p ? nbits - p : 0u;
                if (size == PJD_LUT_BADSYM || codelen > left) { status = PJD_ST_AC_SYM; break; }
                if (e & PJD_LUT_EOB) { w.drop(used); p += used; break; }
                const uint32_t run = (adv - 1u) & 15u;
                if (z + run >= 64) { status = PJD_ST_AC_RUN; break; }
                z += run;
                if (size == PJD_LUT_BADLEN) { status = PJD_ST_AC_LEN; break; }
                if (vbits > left - codelen) { status = PJD_ST_AC_BITS; break; }
                const uint32_t bits = __builtin_amdgcn_ubfe(pk, 32u - used, size), m1 = 1u << size;
                const int val = (int)bits + ((((int)bits - (int)(m1 >> 1)) >> 31) & (int)(1u - m1));
                w.drop(used); p += used;
                // size 0 stores a literal 0 (jpeg_scanner.cpp:516-517); it matters only at slot 52, whose natural position (38)
                // may already hold slot 48's value
                unit[z] = (size == 0 && z == 52) ? (int16_t)PJD_COEF_SENTINEL : (int16_t)val;
            }
        }
    }
    // keep the "decoded by the exact kernel" marker so the back end treats slot 0 as absolute
    B.status[ii] = (B.status[ii] & PJD_STW_NEEDS_EXACT) | status;
}

void pjd_launch_huff_exact_lut(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n)
{
    if (n == 0 || b.max_lut_bytes == 0) return;
    hipLaunchKernelGGL(pjd_k_huff_exact_lut, dim3(n), dim3(64), (size_t)b.max_lut_bytes + 64, s, b, image_list, dense_base);
}

// ---------------------------------------------------------------------------------------------
static size_t huff_lds_bytes(const PjdDevBatch &b)
{
    static const size_t extra = [] { const char *e = std::getenv("PJD_EXTRA_LDS"); return e ? (size_t)std::atoi(e) : (size_t)0; }();   // occupancy experiments
    return (size_t)b.max_lut_bytes + PJD_HUFF_WAVES * (PJD_WAVE_LDS + PJD_PHASE_LDS) + 16 + extra;
}

void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_tsets == 0) return;
    hipLaunchKernelGGL(pjd_k_build_tables, dim3(b.n_tsets * PJD_MAX_TABLES), dim3(256), 0, s, b);
}
void pjd_launch_lane_words(hipStream_t s, const PjdDevBatch &b)
{
#if PJD_DIRECT_ECS
    return;
#endif
    if (b.n_hwave) hipLaunchKernelGGL(pjd_k_lane_words, dim3(b.n_hwave), dim3(256), 0, s, b, 0u, 0u);
}
void pjd_launch_lane_words_group(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g)
{
#if PJD_DIRECT_ECS
    return;
#endif
    if (g.hwg_count) hipLaunchKernelGGL(pjd_k_lane_words, dim3(g.hwg_count * PJD_HUFF_WAVES), dim3(256), 0, s, b, g.hwg_first, g.hwg_count);
}
void pjd_launch_huff_lanes(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_lanes, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b, 0u, b.ticket);
}
void pjd_launch_huff_lanes_group(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g, uint32_t group_index)
{
    // ticket counters: the batch's, then one per group (the operation state is zeroed before every decode, pjd_k_reset)
    if (g.hwg_count) hipLaunchKernelGGL(pjd_k_huff_lanes, dim3(g.hwg_count), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b, g.hwg_first, b.ticket + 2 + 2 * group_index);
}
