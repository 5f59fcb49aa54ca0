// pjd_k_huffman.hip -- the PARALLEL entropy decoder for gfx950: one launch, pjd_k_huff_onepass.
//
// Huffman decoding is one dependent chain per restart segment; a batch of ImageNet files has
// ~10^3 chains, a single 4K picture has one.  To fill 256 CUs the bitstream is cut into fixed
// SUBSEQUENCES (128..1024 bytes, chosen per batch), one decode lane each, and lanes find their true
// entry states by self-synchronisation: a decoder started at a wrong position falls into step
// with the true one after a while (the scheme Weissenberger & Schmidt describe for GPUs).
// Measured on 4:2:0 streams the distance to synchronisation is ~160 B on average with 5 % above
// 512 B -- bit position, zigzag slot AND the 6-unit MCU phase must all agree -- which shapes
// everything below:
//
//   * lanes read their own subsequence straight from HBM, 16 bytes at a time with the next chunk
//     always in flight; nothing is staged in LDS, so occupancy is not LDS-bound and a workgroup is
//     ONE wave (63 owned subsequences + 1 overlap lane): states travel by wave shuffles, there are
//     no barriers;
//   * round 0 decodes every subsequence speculatively and leaves PJD_NCHK checkpoints of the
//     trajectory (state + data units still to come) in LDS;
//   * in a re-sync round a lane restarts from its predecessor's exit state and stops as soon as
//     its state equals a checkpoint ("bridge"): the usual cost is the synchronisation distance,
//     not a whole subsequence.  Only a lane that crosses its whole subsequence unmerged hands a
//     changed exit state on to its successor for the next round;
//   * waves of one image are stitched, their data-unit / entry counts scanned, and the final pass
//     written in the SAME launch: a wave waits only for earlier waves of its own image (published
//     64-bit words, decoupled look-back), never for the batch (see pjd_k_huff_onepass).
//
//   pjd_k_build_tables   raw (offsets, symbols) tables -> two-level decode table: 10-bit first level, one
//                        64-entry second-level table per 10-bit prefix that holds longer codes
//                        (semantics of reference generate_codes / get_next_symbol,
//                        reference src/jpeg_scanner.cpp:438-465)
//   pjd_k_huff_onepass   A  speculative round + re-sync rounds inside the wave
//                        B  stitch to the predecessor wave (its exit state vs the entry assumed here)
//                        C  counts: scan inside the wave, look-back across the image's waves
//                        D  final pass from the now-known entry states.  Output is COMPACT: every stored
//                           AC coefficient becomes one 4-byte entry (value << 16 | zigzag slot) in a stream
//                           whose per-lane offsets are exact prefix sums, so lanes append 16 bytes at a time
//                           and nothing has to be zero-filled; DC differences go to a dense int16 array
//                           (pjd_k_dc_* integrates it); du_end[] / seg_ent[] delimit each unit's entries
//
// Exactness: a lane that starts from the true state performs exactly the reference's
// decode_MCU_component (reference src/jpeg_scanner.cpp:467-520).  Anything irregular seen in the
// final pass -- invalid code, size or run outside the baseline limits, a segment that ends early or
// late, a boundary that did not stitch, a lane that does not reproduce its synchronised exit --
// sets PJD_STW_NEEDS_EXACT and the host re-decodes that image with the one-lane exact kernel.
#include "pjd_device_common.h"
#include "pjd_kernels.h"

static_assert(sizeof(PjdDevHuffRaw) == 180, "raw table layout");
static_assert(PJD_LUT_BITS == 10, "second level is indexed by the 6 bits after a 10-bit prefix");

#define LUT_BAD     (0x4000u | (16u << 8))       // no code: consume 16 bits (as the reference's get_next_symbol), symbol 0

// ---------------------------------------------------------------------------------------------
// One block per (image, table slot): two-level decode table (layout: pjd_internal.h).
__global__ __launch_bounds__(256) void pjd_k_build_tables(PjdDevBatch B)
{
    const uint32_t img = blockIdx.x / PJD_MAX_TABLES, slot = blockIdx.x % PJD_MAX_TABLES;
    const PjdDevImage &im = B.images[img];
    if (slot >= im.n_tables || im.lut_bytes == 0) return;
    const PjdDevHuffRaw &r = B.raw_tables[(size_t)img * PJD_MAX_TABLES + slot];
    uint16_t *blob = reinterpret_cast<uint16_t *>(B.luts + (size_t)im.lut_off16 * 16);
    uint16_t *L1 = blob + slot * (PJD_L1_BYTES / 2);
    const uint32_t l2_off = im.l2_off[slot], p0 = im.l2_p0[slot], p1 = im.l2_p1[slot];
    __shared__ uint32_t first[17];
    __shared__ uint8_t offs[17];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        uint32_t code = 0;                       // reference generate_codes (jpeg_scanner.cpp:438-448)
        first[0] = 0;
        for (int len = 1; len <= 16; len++) {
            first[len] = code;
            code = (code + (uint32_t)(r.offsets[len] - r.offsets[len - 1])) << 1;
        }
    }
    if (tid < 17) offs[tid] = r.offsets[tid];
    __syncthreads();
    for (uint32_t idx = tid; idx < (1u << PJD_LUT_BITS); idx += 256) {
        uint32_t e = (idx >= p0 && idx < p1) ? (0x8000u | (l2_off + (idx - p0) * 64)) : LUT_BAD;
        for (uint32_t len = 1; len <= PJD_LUT_BITS; len++) {     // shortest match wins, as the reference's scan
            const uint32_t c = idx >> (PJD_LUT_BITS - len);
            const uint32_t d = c - first[len], cnt = (uint32_t)offs[len] - offs[len - 1];
            if (c >= first[len] && d < cnt) { e = (len << 8) | r.symbols[offs[len - 1] + d]; break; }
        }
        L1[idx] = (uint16_t)e;
    }
    for (uint32_t j = tid; j < (p1 - p0) * 64; j += 256) {
        const uint32_t w16 = (p0 << 6) + j;
        uint32_t e = LUT_BAD;
        for (uint32_t len = PJD_LUT_BITS + 1; len <= 16; len++) {
            const uint32_t c = w16 >> (16 - len);
            const uint32_t d = c - first[len], cnt = (uint32_t)offs[len] - offs[len - 1];
            if (c >= first[len] && d < cnt) { e = (len << 8) | r.symbols[offs[len - 1] + d]; break; }
        }
        blob[l2_off + j] = (uint16_t)e;
    }
}

// ---------------------------------------------------------------------------------------------
// Bit window over the lane's own stream, fed from HBM 16 bytes at a time, one chunk ahead.
// ---------------------------------------------------------------------------------------------
struct BitWin {
    const uint4 *ptr;     // next chunk to fetch
    uint4 cur, nxt;
    uint32_t k;           // dwords left in cur
    uint64_t buf;
    int cnt;
    __device__ __forceinline__ uint32_t take()
    {
        if (k == 0) {
            cur = nxt;
            // keep the copy above ahead of the load below: the load can then target nxt's registers
            // directly and is first waited for a whole chunk (~20 symbols) later
            asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w));
            nxt = *ptr++;
            k = 4;
        }
        const uint32_t w = __builtin_bswap32(cur.x);
        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
        k--;
        return w;
    }
    // `base16`: 16-byte aligned stream origin; p: bit offset from it
    __device__ __forceinline__ void init(const uint4 *base16, uint32_t p)
    {
        ptr = base16 + (p >> 7);
        cur = ptr[0]; nxt = ptr[1]; ptr += 2;
        k = 4;
        const uint32_t skip = (p >> 5) & 3;
        if (skip >= 1) { cur.x = cur.y; cur.y = cur.z; cur.z = cur.w; k--; }
        if (skip >= 2) { cur.x = cur.y; cur.y = cur.z; k--; }
        if (skip >= 3) { cur.x = cur.y; k--; }
        const uint32_t hi = take(), lo = take();
        buf = ((uint64_t)hi << 32) | lo;
        cnt = 64 - (int)(p & 31);
        buf <<= (p & 31);
    }
    __device__ __forceinline__ uint32_t peek()
    {
        if (cnt <= 32) { buf |= (uint64_t)take() << (32 - cnt); cnt += 32; }
        return (uint32_t)(buf >> 32);
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf <<= n; cnt -= (int)n; }
};

enum { MODE_SPEC = 0, MODE_BRIDGE = 1, MODE_WRITE = 2 };

struct OutCtx {            // WRITE mode: where this lane's output goes
    uint32_t *ent;         // image's entry stream
    uint32_t *du_end;      // image's per-unit "end of entries" (image-relative entry index)
    int16_t *dcv;          // image's per-unit DC differences
    uint32_t epos;         // next entry index (image-relative)
    uint32_t epos0;        // entry index this lane started at
    uint4 acc;             // up to 4 pending entries, newest in .w
};

struct ChkCtx {            // checkpoint bookkeeping of one lane (LDS, strided by lane)
    uint32_t *state;       // [PJD_NCHK][64] at this lane's column
    uint32_t *rem;         // [PJD_NCHK][64]
    uint32_t start_bit;    // first bit of the subsequence (relative to the lane's base16)
    uint32_t chk_bits;     // checkpoint spacing
};

struct ChkCursor { uint32_t j, next_chk; };      // next checkpoint of the pass in progress

enum { SPAN_END = 0, SPAN_MERGED = 1, SPAN_YIELDED = 2 };

// After a SPEC / BRIDGE pass: checkpoints (re)written in it hold "units so far"; make them "units still to come".
__device__ __forceinline__ void chk_finish(const ChkCtx &K, uint32_t j, uint32_t ndu)
{
    for (uint32_t i = 1; i < j; i++) K.rem[i * 64] = ndu - K.rem[i * 64];
}

// Decodes symbols that START before end_bit.  State (p, c, z): bit position relative to base16,
// data-unit phase within the MCU, zigzag slot (0 = DC expected).  Returns SPAN_MERGED when a BRIDGE pass
// merged into the recorded trajectory (then ndu already includes the units still to come), SPAN_YIELDED
// when a BRIDGE pass stopped because at most `yield_lanes` lanes of the wave were still decoding (the
// caller continues those with the wave-cooperative decoder), else SPAN_END.
template <int MODE>
__device__ __forceinline__ int decode_span(const uint8_t *tabs, uint32_t tpacked, uint32_t nl, uint32_t dus,
                                           const uint4 *base16, uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit,
                                           uint32_t &ndu, uint32_t &err, const ChkCtx &K, ChkCursor &cur, uint32_t yield_lanes,
                                           OutCtx *O, uint32_t &D, uint32_t D_end)
{
    // `ndu` is a packed counter: data units completed in the low 16 bits, AC entries produced in the
    // high 16 bits (both fit for a subsequence of <= 1024 bytes)
    if (p >= end_bit) return SPAN_END;
    // wave-uniform image constants: as scalars they are waited for HERE; left in vector registers their
    // first use sits inside the loop and drags a vmcnt(0) -- i.e. a wait for the stream prefetch -- into
    // every iteration
    nl = __builtin_amdgcn_readfirstlane(nl);
    dus = __builtin_amdgcn_readfirstlane(dus);
    tpacked = __builtin_amdgcn_readfirstlane(tpacked);
    BitWin w;
    w.init(base16, p);
    uint32_t j = cur.j, next_chk = cur.next_chk;
    int res = SPAN_END;
    // per data-unit phase c: LUT slot of its DC table (bits 6c..6c+2) and of its AC table (bits 6c+3..6c+5)
    uint64_t slots = 0;
    for (uint32_t cc = 0; cc < dus; cc++) {
        const uint32_t comp = (cc >= nl ? 1u : 0u) + (cc > nl ? 1u : 0u);
        slots |= (uint64_t)(((tpacked >> (8 * comp)) & 7u) | (((tpacked >> (8 * comp + 4)) & 7u) << 3)) << (6 * cc);
    }
    // one compare per symbol covers both "subsequence end" and "next checkpoint"
    uint32_t lim = (MODE != MODE_WRITE && next_chk < end_bit) ? next_chk : end_bit;
    for (;;) {
        if (MODE == MODE_WRITE) { if (p >= end_bit || D >= D_end) break; }
        else if (p >= lim) {
            if (p >= end_bit) break;
            const uint32_t st = ((p - K.start_bit) << 12) | (c << 8) | z;      // p-start < 2^14, c < 16, z < 64
            if (MODE == MODE_BRIDGE && K.state[j * 64] == st) { ndu += K.rem[j * 64]; res = SPAN_MERGED; break; }
            K.state[j * 64] = st;
            K.rem[j * 64] = ndu;                                                // turned into "still to come" after the pass
            j++;
            next_chk += K.chk_bits;
            lim = next_chk < end_bit ? next_chk : end_bit;
        }
        if (MODE == MODE_BRIDGE && yield_lanes && (uint32_t)__popcll(__ballot(true)) <= yield_lanes) { res = SPAN_YIELDED; break; }
        const uint32_t pk = w.peek();
        const bool is_dc = (z == 0);
        const uint32_t slot = (uint32_t)(slots >> (6 * c + (is_dc ? 0u : 3u))) & 7u;
        uint32_t e = *reinterpret_cast<const uint16_t *>(tabs + slot * PJD_L1_BYTES + 2 * (pk >> (32 - PJD_LUT_BITS)));
        // code longer than 10 bits: one more read, in the 64-entry table of this 10-bit prefix
        if (__builtin_expect((e & 0x8000u) != 0, 0))
            e = *reinterpret_cast<const uint16_t *>(tabs + 2 * ((e & 0x7fffu) + ((pk >> 16) & 63u)));
        const uint32_t len = (e >> 8) & 31u, sym = e & 255u;
        err |= (e >> 14) & 1u;
        const uint32_t size = sym & 15, run = sym >> 4;
        const uint32_t used = len + size;
        w.drop(used);
        p += used;
        // state update without branches (reference src/jpeg_scanner.cpp:469-518)
        const uint32_t zr = z + run;                       // landing slot of an AC symbol
        const bool eob = (sym == 0), over = zr > 63;
        const uint32_t z_ac = (eob || over) ? 64u : zr + 1;
        const uint32_t znew = is_dc ? 1u : z_ac;
        err |= is_dc ? (sym > 11) : (!eob && (over || size > 10));
        // an AC symbol that stores something: a non-zero coefficient, or the explicit zero at slot 52
        // (the reference's zigzag_map sends slots 48 and 52 to the same natural position: DESIGN.md)
        const bool store_ac = !is_dc && !eob && !over && (size != 0 || zr == 52);
        const bool done = znew >= 64;
        if (MODE == MODE_WRITE) {
            const uint32_t bits = size ? ((pk << len) >> (32 - size)) : 0;
            int val = (int)bits;
            if (size && !(bits >> (size - 1))) val -= (int)((1u << size) - 1);
            if (is_dc) O->dcv[D] = (int16_t)val;
            if (store_ac) {
                O->acc.x = O->acc.y; O->acc.y = O->acc.z; O->acc.z = O->acc.w;
                O->acc.w = ((uint32_t)val << 16) | zr;
                O->epos++;
                if ((O->epos & 3) == 0) {
                    if (O->epos - O->epos0 >= 4) *reinterpret_cast<uint4 *>(O->ent + O->epos - 4) = O->acc;
                    else {                                   // lane started inside this group of four
                        const uint32_t n = O->epos - O->epos0;
                        O->ent[O->epos - 1] = O->acc.w;
                        if (n >= 2) O->ent[O->epos - 2] = O->acc.z;
                        if (n >= 3) O->ent[O->epos - 3] = O->acc.y;
                    }
                }
            }
            if (done) O->du_end[D] = O->epos;
        }
        z = done ? 0u : znew;
        c = done ? ((c + 1 == dus) ? 0u : c + 1) : c;
        ndu += (done ? 1u : 0u) + (store_ac ? 0x10000u : 0u);
        if (MODE == MODE_WRITE) D += done;
        else D += 1;                                       // diagnostics: symbols decoded
    }
    if (MODE == MODE_WRITE) {
        // entries of the last, incomplete group of four
        const uint32_t mine = O->epos - O->epos0, pend = (O->epos & 3) < mine ? (O->epos & 3) : mine;
        if (pend >= 1) O->ent[O->epos - 1] = O->acc.w;
        if (pend >= 2) O->ent[O->epos - 2] = O->acc.z;
        if (pend >= 3) O->ent[O->epos - 3] = O->acc.y;
    }
    cur.j = j; cur.next_chk = next_chk;
    return res;
}

// ---------------------------------------------------------------------------------------------
// Per-wave set-up.
// ---------------------------------------------------------------------------------------------
struct LaneGeom {
    bool valid, owned, seg_first, seg_last;
    uint32_t q;                    // global subsequence index
    uint32_t seg;                  // global segment index
    uint32_t start_bit, end_bit;   // relative to base16
    uint32_t seg_end_bit;
    uint32_t base_bit;             // base16, in bits relative to the image's ecs
    const uint4 *base16;           // the lane's own origin: its subsequence start rounded down to 16 bytes
};

extern __shared__ __attribute__((aligned(16))) uint8_t pjd_huff_lds[];   // [tables][chk_state 8x64 u32][chk_rem 8x64 u32][staged subsequence]

__device__ __forceinline__ void wave_setup(const PjdDevBatch &B, const PjdDevHuffWg &wg, const PjdDevImage &im,
                                           LaneGeom &g, uint32_t &tpacked, ChkCtx &K)
{
    const uint32_t t = threadIdx.x;
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    g.owned = t >= 1 && t - 1 < wg.n_sub;
    g.valid = g.owned || (t == 0 && !first_is_head);
    g.q = wg.first_sub + t - 1;
    g.seg_first = g.seg_last = false;
    g.seg = 0; g.start_bit = g.end_bit = g.seg_end_bit = g.base_bit = 0;
    g.base16 = reinterpret_cast<const uint4 *>(B.ecs + im.ecs_off);
    if (g.valid) {
        const PjdDevSub sb = B.subs[g.q];
        g.seg = sb.seg & 0x7fffffffu;
        g.seg_first = (sb.seg >> 31) != 0;
        const PjdDevSegment sg = B.segs[g.seg];
        const uint32_t end_byte = sb.byte_start + B.sub_bytes < sg.byte_end ? sb.byte_start + B.sub_bytes : sg.byte_end;
        g.seg_last = end_byte == sg.byte_end;
        const uint32_t lo16 = sb.byte_start & ~15u;
        g.base_bit = lo16 * 8;
        g.base16 = reinterpret_cast<const uint4 *>(B.ecs + im.ecs_off + lo16);
        g.start_bit = (sb.byte_start - lo16) * 8;
        g.end_bit = (end_byte - lo16) * 8;
        g.seg_end_bit = (sg.byte_end - lo16) * 8;
    }
    tpacked = 0;
    for (int cc = 0; cc < 3; cc++)
        tpacked |= ((uint32_t)im.tbl_slot[cc][0] << (8 * cc)) | ((uint32_t)im.tbl_slot[cc][1] << (8 * cc + 4));
    // tables -> LDS (16 B per lane per step, 4 loads in flight per lane)
    const uint32_t n16 = im.lut_bytes / 16;
    const uint4 *tsrc = reinterpret_cast<const uint4 *>(B.luts) + im.lut_off16;
    uint4 *tdst = reinterpret_cast<uint4 *>(pjd_huff_lds);
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
    uint32_t *chk = reinterpret_cast<uint32_t *>(pjd_huff_lds + B.max_lut_bytes);
    K.state = chk + t;
    K.rem = chk + PJD_NCHK * 64 + t;
    K.start_bit = g.start_bit;
    K.chk_bits = B.sub_bytes * 8 / PJD_NCHK;
    for (int j = 0; j < PJD_NCHK; j++) { K.state[j * 64] = 0xffffffffu; K.rem[j * 64] = 0; }
    __syncthreads();          // one wave: a wait on the LDS stores
}

// ---------------------------------------------------------------------------------------------
// Wave-cooperative BRIDGE pass: the whole wave decodes ONE subsequence.
//
// After the first re-sync round only a few lanes of a wave still have work, and what they have is the
// worst kind: stretches on which the speculative trajectory never falls into step, so a lane walks a whole
// subsequence alone at ~580 cycles per symbol -- the chains that decide how long the slowest wave lives.
// (Measured: ~400 cycles per symbol this way -- a gain, but not the factor hoped for: windows end at every
// data unit and the scalar walk is ~25 instructions per symbol.)
// Here the 64 lanes decode the AC symbol that WOULD start at each of the next 64 bit positions (one
// table look-up each, in parallel), and scalar code then follows the real chain through those candidates
// with v_readlane: ~10 scalar instructions per symbol instead of a ~70-instruction vector iteration.
// A window ends at the end of a data unit (the next unit may use other tables), after 64 bits, at the
// subsequence end or when the pass merges into a checkpoint.  Semantics are those of
// decode_span<MODE_BRIDGE>, symbol for symbol; every argument is wave-uniform.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t coop_lookup(const uint8_t *tabs, uint32_t slot, uint32_t bits)
{
    uint32_t e = *reinterpret_cast<const uint16_t *>(tabs + slot * PJD_L1_BYTES + 2 * (bits >> (32 - PJD_LUT_BITS)));
    if (e & 0x8000u) e = *reinterpret_cast<const uint16_t *>(tabs + 2 * ((e & 0x7fffu) + ((bits >> 16) & 63u)));
    return e;
}

__device__ __forceinline__ int coop_bridge(const uint8_t *tabs, uint32_t tpacked, uint32_t nl, uint32_t dus,
                                           const uint4 *base16, uint32_t sub_bytes, uint32_t &p_io, uint32_t &c_io, uint32_t &z_io,
                                           uint32_t end_bit, uint32_t &ndu_io, uint32_t *chk_state, uint32_t *chk_rem,
                                           uint32_t start_bit, uint32_t chk_bits, uint32_t j, uint32_t next_chk, uint32_t *stream)
{
    const uint32_t t = threadIdx.x;
    // the subsequence (it starts < 16 bytes after base16) + look-ahead, as big-endian words in LDS
    const uint32_t n16 = (sub_bytes + 16 + 64) / 16;
    for (uint32_t i = t; i < n16; i += PJD_HUFF_THREADS) {
        uint4 v = base16[i];
        v.x = __builtin_bswap32(v.x); v.y = __builtin_bswap32(v.y); v.z = __builtin_bswap32(v.z); v.w = __builtin_bswap32(v.w);
        reinterpret_cast<uint4 *>(stream)[i] = v;
    }
    __syncthreads();
    uint32_t p = p_io, c = c_io, z = z_io, ndu = ndu_io;
    int res = SPAN_END;
    for (;;) {
        // ---- candidates: AC symbol at every bit offset p + t; DC symbol at p
        const uint32_t comp = (c >= nl ? 1u : 0u) + (c > nl ? 1u : 0u);
        const uint32_t dc_slot = (tpacked >> (8 * comp)) & 15u, ac_slot = (tpacked >> (8 * comp + 4)) & 15u;
        const uint32_t bp = p + t;
        const uint32_t w0 = stream[bp >> 5], w1 = stream[(bp >> 5) + 1];
        const uint32_t bits = (uint32_t)((((uint64_t)w0 << 32) | w1) << (bp & 31) >> 32);
        const uint32_t e = coop_lookup(tabs, ac_slot, bits);
        const uint32_t sym = e & 255u, size = sym & 15u;
        const uint32_t pack = (((e >> 8) & 31u) + size) | ((sym >> 4) << 8) | (size ? 0x1000u : 0u) | (sym == 0 ? 0x2000u : 0u);
        uint32_t dc_used = 0;
        if (z == 0) {
            const uint32_t bits0 = __builtin_amdgcn_readlane(bits, 0);
            const uint32_t ed = coop_lookup(tabs, dc_slot, bits0);
            dc_used = __builtin_amdgcn_readfirstlane(((ed >> 8) & 31u) + (ed & 15u));
        }
        // ---- follow the chain through the window
        uint32_t pos = 0;
        bool stop = false;
        for (;;) {
            const uint32_t q = p + pos;
            if (q >= end_bit) { stop = true; break; }
            if (q >= next_chk) {
                const uint32_t st = ((q - start_bit) << 12) | (c << 8) | z;
                const uint32_t old = __builtin_amdgcn_readfirstlane(chk_state[j * 64]);
                if (old == st) { ndu += __builtin_amdgcn_readfirstlane(chk_rem[j * 64]); res = SPAN_MERGED; stop = true; break; }
                if (t == 0) { chk_state[j * 64] = st; chk_rem[j * 64] = ndu; }
                j++;
                next_chk += chk_bits;
            }
            if (z == 0) {                          // DC: only the candidate at the window start exists
                if (pos != 0) break;
                pos = dc_used;
                z = 1;
                continue;
            }
            if (pos >= 64) break;
            const uint32_t s = __builtin_amdgcn_readlane(pack, pos);
            const uint32_t zr = z + ((s >> 8) & 15u);
            const bool eob = (s & 0x2000u) != 0, over = zr > 63;
            const bool store = !eob && !over && ((s & 0x1000u) != 0 || zr == 52);
            const bool done = eob || over || zr == 63;
            ndu += (done ? 1u : 0u) + (store ? 0x10000u : 0u);
            pos += s & 63u;
            if (done) { z = 0; c = (c + 1 == dus) ? 0u : c + 1; break; }
            z = zr + 1;
        }
        p += pos;
        if (stop) break;
    }
    if (t == 0) for (uint32_t i = 1; i < j; i++) chk_rem[i * 64] = ndu - chk_rem[i * 64];
    p_io = p; c_io = c; z_io = z; ndu_io = ndu;
    return res;
}

// A re-sync round that STARTS with at most PJD_COOP_START_LANES active lanes is done cooperatively, one lane after
// the other; a round in lane-parallel mode hands over when only PJD_COOP_YIELD_LANES lanes are still decoding.
// Measured (MI355X): a cooperative pass over a whole 512-byte subsequence takes ~140 us (57..144 us per lane observed,
// depending on where the lane merges) against 190..300 us for a lane-parallel round, but cooperative passes are serial,
// so anything above ONE lane loses (1024-image batch, start/yield 1/1: 1.75 ms, 4/1: 2.08, 8/1: 2.24, 16/1: 2.68;
// single 4K picture 1.16 / 1.39 / 1.82 / 2.16, 1.30 ms without the cooperative pass; same ranking with two batches in flight).
#ifndef PJD_COOP_START_LANES
#define PJD_COOP_START_LANES 1
#endif
#ifndef PJD_COOP_YIELD_LANES
#define PJD_COOP_YIELD_LANES 1
#endif

// Lanes have different origins, so states are exchanged as bit positions relative to the image's ecs.
struct WaveState { uint32_t p_img, cz, cnt; };

__device__ __forceinline__ bool wave_rounds(const PjdDevBatch &B, const PjdDevImage &im, const LaneGeom &g, uint32_t tpacked,
                                            const ChkCtx &K, WaveState &S, uint32_t changed, unsigned long long *stats, int stat_base, uint32_t *rdbg = nullptr)
{
    const uint8_t *tabs = pjd_huff_lds;
    const uint32_t t = threadIdx.x;
    const uint32_t nl = __builtin_amdgcn_readfirstlane(im.n_luma), dus = __builtin_amdgcn_readfirstlane(im.dus_per_mcu);
    const uint32_t tp = __builtin_amdgcn_readfirstlane(tpacked);
    uint32_t *chk = reinterpret_cast<uint32_t *>(pjd_huff_lds + B.max_lut_bytes);
    uint32_t *stream = chk + 2 * PJD_NCHK * 64;
    for (int iter = 0; iter < PJD_SYNC_MAX_ITERS; iter++) {
        const uint32_t pp = __shfl_up(S.p_img, 1), pcz = __shfl_up(S.cz, 1), pch = __shfl_up(changed, 1);
        const bool act = g.owned && !g.seg_first && pch != 0;
        const uint64_t act_mask = __ballot(act);
        if (!act_mask) return true;
        const uint64_t tr0 = rdbg ? __builtin_amdgcn_s_memrealtime() : 0;
        if (stats && t == 0) atomicAdd(stats + stat_base, 1ull);
        if (stats && act) atomicAdd(stats + stat_base + 1, 1ull);
        changed = 0;
        uint32_t p = pp - g.base_bit, c = pcz >> 8, z = pcz & 255, ndu = 0, err = 0, D = 0;
        ChkCursor cur = { 1, K.start_bit + K.chk_bits };
        int res = SPAN_YIELDED;
        // many lanes: one subsequence per lane, until only a few are still at it
        if (__popcll(act_mask) > PJD_COOP_START_LANES && act)
            res = decode_span<MODE_BRIDGE>(tabs, tp, nl, dus, g.base16, p, c, z, g.end_bit, ndu, err, K, cur, PJD_COOP_YIELD_LANES, nullptr, D, 0);
        if (act && res != SPAN_YIELDED) chk_finish(K, cur.j, ndu);
        // the rest: the whole wave on one subsequence at a time
        uint64_t todo = __ballot(act && res == SPAN_YIELDED);
        while (todo) {
            const uint32_t L = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            uint32_t sp = __builtin_amdgcn_readlane(p, L), sc = __builtin_amdgcn_readlane(c, L), sz = __builtin_amdgcn_readlane(z, L);
            uint32_t sn = __builtin_amdgcn_readlane(ndu, L);
            // (the builtin returns int: without the casts a low half >= 2^31 sign-extends over the high half)
            const uint64_t bptr = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)((uint64_t)g.base16 >> 32), L) << 32) |
                                  (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(uint64_t)g.base16, L);
            const int r = coop_bridge(tabs, tp, nl, dus, reinterpret_cast<const uint4 *>(bptr), B.sub_bytes, sp, sc, sz,
                                      __builtin_amdgcn_readlane(g.end_bit, L), sn, chk + L, chk + PJD_NCHK * 64 + L,
                                      __builtin_amdgcn_readlane(K.start_bit, L), __builtin_amdgcn_readfirstlane(K.chk_bits),
                                      __builtin_amdgcn_readlane(cur.j, L), __builtin_amdgcn_readlane(cur.next_chk, L), stream);
            if (t == L) { p = sp; c = sc; z = sz; ndu = sn; res = r; }
        }
        if (rdbg && t == 0 && iter < 24) rdbg[iter] = ((uint32_t)__popcll(act_mask) << 24) | ((uint32_t)(__builtin_amdgcn_s_memrealtime() - tr0) & 0xffffffu);
        if (act) {
            S.cnt = ndu;
            if (res != SPAN_MERGED) {
                const uint32_t np = p + g.base_bit, ncz = (c << 8) | z;
                if (np != S.p_img || ncz != S.cz) { S.p_img = np; S.cz = ncz; changed = 1; }
            }
        }
    }
    return !__any(changed != 0);
}

// ---------------------------------------------------------------------------------------------
// Segmented combine used for data-unit counts: element = (value, head flag); a head resets.
__device__ __forceinline__ void seg_combine(uint32_t av, uint32_t af, uint32_t &bv, uint32_t &bf)   // b = a (+) b
{
    if (!bf) bv += av;
    bf |= af;
}

__device__ __forceinline__ void wave_seg_scan(uint32_t &v, uint32_t &f)      // inclusive, 64 lanes
{
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ov = __shfl_up(v, off), of = __shfl_up(f, off);
        if ((int)lane >= off) seg_combine(ov, of, v, f);
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel.  Synchronise, stitch, scan and write in a single launch.
//
// With one launch per stage every stage waits for the slowest wave of the whole batch, and that wave is
// slow by nature: somewhere in 10^5 subsequences there is a stretch of a few KB on which speculative
// decoders do not fall into step, and it can only be walked sequentially (measured on the 1024-image
// batch: mean wave 0.63 ms, slowest 1.38 ms in stage A).  Here a wave depends only on waves of its OWN
// image that precede it, so everything else proceeds to the write pass while the stragglers finish
// (four launches 2.31 ms -> one launch 1.78 ms on that batch):
//
//   * waves take their index from a ticket counter, so every wave a wave waits for has already started
//     (waits can always be satisfied; they are bounded anyway and poison the image on time-out);
//   * a wave publishes its speculative exit state (generation 0); its successor compares that with the
//     entry it assumed, re-bridges from the truth if they differ, and publishes generation 1 -- two hops,
//     never a chain across the image; at the very end it checks that the entry it used is what its
//     predecessor finally produced (else: exact kernel);
//   * data-unit counts (segmented: a restart segment's first subsequence resets to the segment's first
//     unit) and entry counts are combined with a decoupled look-back over the image's waves, 64
//     descriptors per step;
//   * the write pass runs from registers; tables and checkpoints are still in LDS.
// All published words are self-contained 64-bit values (state or descriptor + flag bits), so relaxed
// agent-scope atomics are enough.  wg_exit / wg_desc / ticket are zeroed before every launch.
#define OP_FLAG        (1ull << 63)
#define OP_ST_AGG      (1ull << 62)            // descriptor holds this wave's aggregate
#define OP_ST_PFX      (2ull << 62)            // descriptor holds the inclusive prefix up to this wave
#define OP_ST_MASK     (3ull << 62)
#define OP_POISON      (1ull << 61)
#define OP_HEAD        (1ull << 60)
#define OP_SPIN_LIMIT  (1u << 22)

__device__ __forceinline__ uint64_t op_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void op_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint64_t op_desc(uint64_t status, uint32_t v, uint32_t f, uint32_t e, bool poison)
{
    return status | (poison ? OP_POISON : 0ull) | (f ? OP_HEAD : 0ull) | ((uint64_t)(v & 0x0fffffffu) << 32) | e;
}

// lane 0 waits for a flagged word; the value (flag stripped) is returned to every lane
__device__ __forceinline__ uint64_t op_wait_flag(const uint64_t *p, bool &timeout)
{
    uint64_t v = 0;
    if (threadIdx.x == 0)
        for (uint32_t it = 0; it < OP_SPIN_LIMIT; it++) {
            v = op_load(p);
            if (v & OP_FLAG) break;
            __builtin_amdgcn_s_sleep(8);
        }
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    v = ((uint64_t)hi << 32) | lo;
    timeout = timeout || !(v & OP_FLAG);
    return v & ~OP_FLAG;
}

__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_onepass(PjdDevBatch B)
{
    const uint32_t t = threadIdx.x;
    uint32_t w = 0;
    if (t == 0) w = atomicAdd(B.ticket, 1u);
    w = __builtin_amdgcn_readfirstlane(w);
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    uint64_t *exit0 = B.wg_exit, *exit1 = B.wg_exit + B.n_hwg;
    LaneGeom g; uint32_t tpacked; ChkCtx K;
    const uint64_t ts0 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    wave_setup(B, wg, im, g, tpacked, K);
    const uint32_t nl = im.n_luma, dus = im.dus_per_mcu;
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    bool dead = false;                                   // wave-uniform: a wait timed out or a predecessor is poisoned
    uint32_t flag = 0;

    // ---- A: speculative round + re-sync rounds inside the wave
    WaveState S;
    {
        uint32_t p = g.start_bit, c = 0, z = 0, ndu = 0, err = 0, D = 0;
        ChkCursor cur = { 1, K.start_bit + K.chk_bits };
        if (g.valid) {
            decode_span<MODE_SPEC>(pjd_huff_lds, tpacked, nl, dus, g.base16, p, c, z, g.end_bit, ndu, err, K, cur, 0, nullptr, D, 0);
            chk_finish(K, cur.j, ndu);
        }
        S.p_img = p + g.base_bit; S.cz = (c << 8) | z; S.cnt = ndu;
    }
    const uint64_t ts1 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    const uint64_t entry0 = pjd_pack_state(__shfl(S.p_img, 0), __shfl(S.cz, 0) >> 8, __shfl(S.cz, 0) & 255);   // lane 0: the assumed entry
    bool ok = wave_rounds(B, im, g, tpacked, K, S, g.valid ? 1u : 0u, B.stats, 0, B.dbg ? B.dbg + (size_t)w * 32 + 8 : nullptr);
    if (t == wg.n_sub) op_store(exit0 + w, pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255) | OP_FLAG);
    const uint64_t ts2 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- B: stitch to the predecessor wave: redo this wave's bridges from the true entry if the guess was wrong
    uint64_t entry_used = entry0;
    if (!first_is_head) {
        const uint64_t truth = op_wait_flag(exit0 + w - 1, dead);
        if (!dead && truth != entry0) {
            if (t == 0) { S.p_img = (uint32_t)truth; S.cz = (((uint32_t)(truth >> 32) & 255) << 8) | ((uint32_t)(truth >> 40) & 255); S.cnt = 0; }
            ok = wave_rounds(B, im, g, tpacked, K, S, t == 0 ? 1u : 0u, B.stats, 2) && ok;
            entry_used = truth;
        }
    }
    if (t == wg.n_sub) op_store(exit1 + w, pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255) | OP_FLAG);
    if (!ok) flag = 1;
    const uint64_t ts3 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;

    // ---- C: counts.  Inside the wave: segmented scan of data units, plain scan of entries ...
    uint32_t cnt = 0, ecnt = 0, v = 0, f = 0, seg_first_du = 0, seg_n_du = 0;
    if (g.owned) {
        cnt = S.cnt & 0xffffu; ecnt = S.cnt >> 16;
        const PjdDevSegment sg = B.segs[g.seg];
        seg_first_du = sg.first_du; seg_n_du = sg.n_du;
        v = cnt;
        if (g.seg_first) { f = 1; v += seg_first_du; }
    }
    wave_seg_scan(v, f);
    uint32_t es = ecnt;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(es, off); if ((int)t >= off) es += o; }
    const uint32_t agg_v = __shfl(v, 63), agg_f = __shfl(f, 63), agg_e = __shfl(es, 63);
    // ... across the image's waves: decoupled look-back
    uint32_t du_in = 0, ent_in = 0;
    if (w == im.hwg_base) {
        if (t == 0) op_store(B.wg_desc + w, op_desc(OP_ST_PFX, agg_v, agg_f, agg_e, dead));
    } else {
        if (t == 0) op_store(B.wg_desc + w, op_desc(OP_ST_AGG, agg_v, agg_f, agg_e, dead));
        uint32_t rv = 0, rf = 0, re = 0;                  // combination of the descriptors gathered so far (identity)
        bool poison = false;
        int hi = (int)w - 1;
        for (;;) {
            const int j = hi - (int)t;
            const bool inside = j >= (int)im.hwg_base;
            uint64_t d = OP_ST_PFX;                        // before the image's first wave: an empty prefix
            if (inside) {
                uint32_t it = 0;
                do { d = op_load(B.wg_desc + j); if (d & OP_ST_MASK) break; __builtin_amdgcn_s_sleep(8); } while (++it < OP_SPIN_LIMIT);
            }
            const bool ready = (d & OP_ST_MASK) != 0;
            if (__any(!ready)) { dead = true; break; }
            const uint64_t pfx_mask = __ballot((d & OP_ST_MASK) == OP_ST_PFX);   // never empty past the image start
            const int k = pfx_mask ? __builtin_ctzll(pfx_mask) : 64;              // nearest lane holding a prefix
            const bool use = (int)t <= k;
            uint32_t xv = use ? (uint32_t)(d >> 32) & 0x0fffffffu : 0u, xf = use ? (uint32_t)((d >> 60) & 1u) : 0u, xe = use ? (uint32_t)d : 0u;
            poison = poison || __any(use && (d & OP_POISON));
            // lane l holds wave hi-l: larger l = earlier.  Suffix-combine so that lane 0 = X_k (+) ... (+) X_0.
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t ov = __shfl_down(xv, off), of = __shfl_down(xf, off), oe = __shfl_down(xe, off);
                if ((int)t + off < 64) { seg_combine(ov, of, xv, xf); xe += oe; }
            }
            const uint32_t wv = __shfl(xv, 0), wf = __shfl(xf, 0), we = __shfl(xe, 0);
            // running = window (+) running
            { uint32_t nv = rv, nf = rf; seg_combine(wv, wf, nv, nf); rv = nv; rf = nf; re += we; }
            if (k < 64) break;
            hi -= 64;
        }
        if (poison) dead = true;
        du_in = rv; ent_in = re;                           // rf set: rv is an absolute index; else a count from the image's first unit (same thing)
        uint32_t iv = agg_v, ifl = agg_f;
        seg_combine(rv, rf, iv, ifl);
        if (t == 0) op_store(B.wg_desc + w, op_desc(OP_ST_PFX, iv, ifl, re + agg_e, dead));
    }

    const uint64_t ts4 = B.dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    // ---- D: final pass from the true entry states
    const uint32_t prev_p = __shfl_up(S.p_img, 1), prev_cz = __shfl_up(S.cz, 1);
    if (g.owned && !dead) {
        const uint32_t D_out = f ? v : du_in + v;
        uint32_t D = D_out - cnt;
        const uint32_t D_in = D, D_end = seg_first_du + seg_n_du;
        uint32_t p, c, z;
        if (g.seg_first) { p = g.start_bit; c = 0; z = 0; }
        else {
            if (t == 1) { p = (uint32_t)entry_used - g.base_bit; c = (uint32_t)(entry_used >> 32) & 255; z = (uint32_t)(entry_used >> 40) & 255; }
            else { p = prev_p - g.base_bit; c = prev_cz >> 8; z = prev_cz & 255; }
            if (D_in < D_end && (D_in % dus) != c) flag = 1;          // phase must agree with the count
        }
        uint32_t ndu = 0, err = 0;
        OutCtx O;
        O.ent = B.ent + im.ent_base;
        O.du_end = B.du_end + im.du_base;
        O.dcv = B.dcv + im.du_base;
        O.epos = O.epos0 = ent_in + es - ecnt;
        O.acc = make_uint4(0, 0, 0, 0);
        if (g.seg_first) B.seg_ent[g.seg] = O.epos;
        if (D_in < D_end) {
            ChkCursor cur = { 1, 0 };
            decode_span<MODE_WRITE>(pjd_huff_lds, tpacked, nl, dus, g.base16, p, c, z, g.end_bit, ndu, err, K, cur, 0, &O, D, D_end);
            if (err) flag = 1;
            if (D == D_end) {
                if (p > g.seg_end_bit) flag = 1;
                const bool has_next = g.seg + 1 < im.seg_base + im.n_seg;
                if (has_next && ((p + 7) & ~7u) != g.seg_end_bit) flag = 1;
            } else {
                if (S.p_img - g.base_bit != p || (S.cz >> 8) != c || (S.cz & 255) != z) flag = 1;
                if (ndu != S.cnt) flag = 1;
                if (g.seg_last) flag = 1;
            }
        }
    }
    // the entry this wave was synchronised with must be what its predecessor finally produced
    if (!first_is_head && !dead) {
        const uint64_t fin = op_wait_flag(exit1 + w - 1, dead);
        if (fin != entry_used) flag = 1;
    }
    if (B.dbg && t == 0) {
        const uint64_t ts5 = __builtin_amdgcn_s_memrealtime();
        uint32_t *d = B.dbg + (size_t)w * 32;
        d[0] = (uint32_t)ts0; d[1] = (uint32_t)(ts1 - ts0); d[2] = (uint32_t)(ts2 - ts1); d[3] = (uint32_t)(ts3 - ts2);
        d[4] = (uint32_t)(ts4 - ts3); d[5] = (uint32_t)(ts5 - ts4); d[6] = wg.image; d[7] = wg.n_sub;
    }
    if (dead) flag = 1;
    if (flag) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
}

// ---------------------------------------------------------------------------------------------
// tables | checkpoints (state, units-to-come: PJD_NCHK x 64 each) | staged subsequence of the cooperative pass
#if PJD_COOP_START_LANES > 0
#define PJD_COOP_LDS (PJD_SUB_BYTES_MAX + 128)
#else
#define PJD_COOP_LDS 0
#endif
static size_t huff_lds_bytes(const PjdDevBatch &b) { return (size_t)b.max_lut_bytes + 2 * PJD_NCHK * 64 * sizeof(uint32_t) + PJD_COOP_LDS; }

void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_images == 0) return;
    hipLaunchKernelGGL(pjd_k_build_tables, dim3(b.n_images * PJD_MAX_TABLES), dim3(256), 0, s, b);
}
void pjd_launch_huff_onepass(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_onepass, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b);
}
