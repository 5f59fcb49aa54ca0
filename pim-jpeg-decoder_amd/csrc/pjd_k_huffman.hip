// pjd_k_huffman.hip -- the PARALLEL entropy decoder for gfx950 (v2).
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
//     changed exit state on to its successor for the next round.
//
//   pjd_k_build_tables   raw (offsets, symbols) tables -> two-level decode table: 10-bit first level, one
//                        64-entry second-level table per 10-bit prefix that holds longer codes
//                        (semantics of reference generate_codes / get_next_symbol,
//                        reference src/jpeg_scanner.cpp:438-465)
//   pjd_k_huff_sync      rounds as above; exit state and data-unit count per subsequence
//   pjd_k_huff_fix       stitches wave boundaries (overlap lane's guess vs the predecessor wave's
//                        truth; a mismatching wave is redone) and reduces per-wave unit counts
//   pjd_k_huff_carry     one wave per image: scan of those counts -> absolute data-unit index
//   pjd_k_huff_write     final pass from the now-known entry states.  Output is COMPACT: every non-zero
//                        AC coefficient becomes one 4-byte entry (value << 16 | zigzag slot) in a stream
//                        whose per-lane offsets are exact prefix sums, so lanes append 16 bytes at a time
//                        and nothing has to be zero-filled; DC differences go to a dense int16 array
//                        (pjd_k_dc_* integrates it); du_end[] / seg_ent[] delimit each unit's entries
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

// Decodes symbols that START before end_bit.  State (p, c, z): bit position relative to base16,
// data-unit phase within the MCU, zigzag slot (0 = DC expected).  Returns true when a BRIDGE pass
// merged into the recorded trajectory (then ndu already includes the units still to come).
template <int MODE>
__device__ __forceinline__ bool decode_span(const uint8_t *tabs, uint32_t tpacked, uint32_t nl, uint32_t dus,
                                            const uint4 *base16, uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit,
                                            uint32_t &ndu, uint32_t &err, const ChkCtx &K,
                                            OutCtx *O, uint32_t &D, uint32_t D_end)
{
    // `ndu` is a packed counter: data units completed in the low 16 bits, AC entries produced in the
    // high 16 bits (both fit for a subsequence of <= 1024 bytes)
    if (p >= end_bit) return false;
    // wave-uniform image constants: as scalars they are waited for HERE; left in vector registers their
    // first use sits inside the loop and drags a vmcnt(0) -- i.e. a wait for the stream prefetch -- into
    // every iteration
    nl = __builtin_amdgcn_readfirstlane(nl);
    dus = __builtin_amdgcn_readfirstlane(dus);
    tpacked = __builtin_amdgcn_readfirstlane(tpacked);
    BitWin w;
    w.init(base16, p);
    uint32_t j = 1, next_chk = K.start_bit + K.chk_bits;
    bool merged = false;
    while (p < end_bit && (MODE != MODE_WRITE || D < D_end)) {
        if (MODE != MODE_WRITE && p >= next_chk) {
            const uint32_t st = ((p - K.start_bit) << 12) | (c << 8) | z;      // p-start < 2^14, c < 16, z < 64
            if (MODE == MODE_BRIDGE && K.state[j * 64] == st) { ndu += K.rem[j * 64]; merged = true; break; }
            K.state[j * 64] = st;
            K.rem[j * 64] = ndu;                                                // turned into "still to come" after the pass
            j++;
            next_chk += K.chk_bits;
        }
        const uint32_t pk = w.peek();
        const bool is_dc = (z == 0);
        const uint32_t comp = (c >= nl ? 1u : 0u) + (c > nl ? 1u : 0u);
        const uint32_t slot = (tpacked >> (8 * comp + (is_dc ? 0u : 4u))) & 15u;
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
    if (MODE != MODE_WRITE) {
        // checkpoints (re)written in this pass hold "units so far"; make them "units still to come"
        for (uint32_t i = 1; i < j; i++) K.rem[i * 64] = ndu - K.rem[i * 64];
    }
    return merged;
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

extern __shared__ __attribute__((aligned(16))) uint8_t pjd_huff_lds[];   // [tables][chk_state 8x64 u32][chk_rem 8x64 u32]

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

// Lanes have different origins, so states are exchanged as bit positions relative to the image's ecs.
struct WaveState { uint32_t p_img, cz, cnt; };

__device__ __forceinline__ bool wave_rounds(const PjdDevImage &im, const LaneGeom &g, uint32_t tpacked,
                                            const ChkCtx &K, WaveState &S, uint32_t changed, unsigned long long *stats, int stat_base)
{
    const uint8_t *tabs = pjd_huff_lds;
    const uint32_t nl = im.n_luma, dus = im.dus_per_mcu;
    for (int iter = 0; iter < PJD_SYNC_MAX_ITERS; iter++) {
        const uint32_t pp = __shfl_up(S.p_img, 1), pcz = __shfl_up(S.cz, 1), pch = __shfl_up(changed, 1);
        const bool act = g.owned && !g.seg_first && pch != 0;
        if (!__any(act)) return true;
        if (stats && threadIdx.x == 0) atomicAdd(stats + stat_base, 1ull);
        changed = 0;
        if (act) {
            if (stats) atomicAdd(stats + stat_base + 1, 1ull);
            uint32_t p = pp - g.base_bit, c = pcz >> 8, z = pcz & 255, ndu = 0, err = 0, D = 0;
            const bool merged = decode_span<MODE_BRIDGE>(tabs, tpacked, nl, dus, g.base16, p, c, z, g.end_bit, ndu, err, K, nullptr, D, 0);
            S.cnt = ndu;
            if (!merged) {
                const uint32_t np = p + g.base_bit, ncz = (c << 8) | z;
                if (np != S.p_img || ncz != S.cz) { S.p_img = np; S.cz = ncz; changed = 1; }
            }
        }
    }
    return !__any(changed != 0);
}

__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_sync(PjdDevBatch B)
{
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    LaneGeom g; uint32_t tpacked; ChkCtx K;
    const uint64_t tc0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    wave_setup(B, wg, im, g, tpacked, K);
    const uint64_t tc1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    // round 0: every lane from the start of its own subsequence (the true state at a segment start)
    uint32_t p = g.start_bit, c = 0, z = 0, ndu = 0, err = 0, D = 0;
    if (g.valid) decode_span<MODE_SPEC>(pjd_huff_lds, tpacked, im.n_luma, im.dus_per_mcu, g.base16, p, c, z, g.end_bit, ndu, err, K, nullptr, D, 0);
    const uint64_t tc2 = __builtin_amdgcn_s_memtime(), tr2 = __builtin_amdgcn_s_memrealtime();
    WaveState S = { p + g.base_bit, (c << 8) | z, ndu };
    const uint64_t entry0 = pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255);       // lane 0: the assumed entry
    const bool ok = wave_rounds(im, g, tpacked, K, S, g.valid ? 1u : 0u, B.stats, 0);
    const uint64_t tc3 = __builtin_amdgcn_s_memtime(), tr3 = __builtin_amdgcn_s_memrealtime();
    {
        uint32_t mx = D;
        for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(mx, off); mx = o > mx ? o : mx; }
        if (t == 0 && B.stats) {
            atomicAdd(B.stats + 4, tc1 - tc0); atomicAdd(B.stats + 5, tc2 - tc1); atomicAdd(B.stats + 6, tc3 - tc2); atomicAdd(B.stats + 7, (unsigned long long)mx);
            atomicAdd(B.stats + 8, tr1 - tr0); atomicAdd(B.stats + 9, tr2 - tr1); atomicAdd(B.stats + 10, tr3 - tr2);
            atomicMax(B.stats + 11, tr3 - tr0); atomicMax(B.stats + 12, tr2 - tr1);
            // histogram of wave lifetimes in 100 us buckets is too wide for 16 slots: count waves above 0.5 / 0.8 / 1.0 ms
            if (tr3 - tr0 > 50000) atomicAdd(B.stats + 13, 1ull);
            if (tr3 - tr0 > 80000) atomicAdd(B.stats + 14, 1ull);
            if (tr3 - tr0 > 100000) atomicAdd(B.stats + 15, 1ull);
        }
    }
    if (!ok && t == 0) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
    if (g.owned) {
        B.sub_exit[g.q] = pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255);
        B.sub_cnt[g.q] = S.cnt;
        // keep the trajectory's checkpoints: a boundary re-stitch (pjd_k_huff_fix) can then bridge
        // into them instead of repeating the speculative pass
        uint4 *dst = reinterpret_cast<uint4 *>(B.sub_chk + (size_t)g.q * 2 * PJD_NCHK);
        for (int h = 0; h < 2; h++) {
            const uint32_t *src = h ? K.rem : K.state;
            dst[2 * h] = make_uint4(src[0], src[64], src[128], src[192]);
            dst[2 * h + 1] = make_uint4(src[256], src[320], src[384], src[448]);
        }
    }
    if (t == 0) {
        B.wg_entry[w] = g.valid ? entry0 : ~0ull;
    }
    if (t == wg.n_sub) B.wg_exit[w] = pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255);
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

__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_fix(PjdDevBatch B)
{
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    uint64_t *exit1 = B.wg_exit + B.n_hwg;
    bool redo = false;
    uint64_t truth = 0;
    if (!first_is_head) {
        truth = B.wg_exit[w - 1];                    // generation 0: written by pjd_k_huff_sync
        redo = truth != B.wg_entry[w];
    }
    uint32_t my_cnt = 0;
    if (redo) {                                       // wave-uniform: redo this wave from the true entry
        LaneGeom g; uint32_t tpacked; ChkCtx K;
        wave_setup(B, wg, im, g, tpacked, K);
        WaveState S = { 0, 0, 0 };
        if (g.owned) {
            const uint64_t e = B.sub_exit[g.q];
            S.p_img = (uint32_t)e; S.cz = (((uint32_t)(e >> 32) & 255) << 8) | ((uint32_t)(e >> 40) & 255);
            S.cnt = B.sub_cnt[g.q];
            const uint4 *src = reinterpret_cast<const uint4 *>(B.sub_chk + (size_t)g.q * 2 * PJD_NCHK);
            for (int h = 0; h < 2; h++) {
                uint32_t *dst = h ? K.rem : K.state;
                const uint4 a = src[2 * h], b = src[2 * h + 1];
                dst[0] = a.x; dst[64] = a.y; dst[128] = a.z; dst[192] = a.w;
                dst[256] = b.x; dst[320] = b.y; dst[384] = b.z; dst[448] = b.w;
            }
        }
        if (t == 0) { S.p_img = (uint32_t)truth; S.cz = (((uint32_t)(truth >> 32) & 255) << 8) | ((uint32_t)(truth >> 40) & 255); S.cnt = 0; }
        const bool ok = wave_rounds(im, g, tpacked, K, S, t == 0 ? 1u : 0u, B.stats, 2);
        if (!ok && t == 0) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
        if (g.owned) {
            B.sub_exit[g.q] = pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255);
            B.sub_cnt[g.q] = S.cnt;
            my_cnt = S.cnt;
        }
        if (t == 0) B.wg_entry[w] = truth;
        if (t == wg.n_sub) exit1[w] = pjd_pack_state(S.p_img, S.cz >> 8, S.cz & 255);
    } else {
        if (t == 0) exit1[w] = B.wg_exit[w];
        if (t >= 1 && t - 1 < wg.n_sub) my_cnt = B.sub_cnt[wg.first_sub + t - 1];
    }
    // per-wave aggregates: data units (absolute index after the last owned lane if a segment starts
    // inside, else the number of units completed; head flag) and coefficient entries (plain sum)
    uint32_t v = 0, f = 0, e = 0;
    if (t >= 1 && t - 1 < wg.n_sub) {
        const uint32_t sg = B.subs[wg.first_sub + t - 1].seg;
        v = my_cnt & 0xffffu;
        e = my_cnt >> 16;
        if (sg >> 31) { f = 1; v += B.segs[sg & 0x7fffffffu].first_du; }
    }
    wave_seg_scan(v, f);
    for (int off = 32; off; off >>= 1) e += __shfl_xor(e, off);
    if (t == PJD_HUFF_THREADS - 1) { B.wg_agg[2 * w] = v; B.wg_agg[2 * w + 1] = f; B.wg_eagg[w] = e; }
}

__global__ __launch_bounds__(64) void pjd_k_huff_carry(PjdDevBatch B)
{
    const PjdDevImage &im = B.images[blockIdx.x];
    const uint32_t lane = threadIdx.x, n = im.n_hwg;
    uint32_t carry = 0, ecarry = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        uint32_t v = 0, f = 0, e = 0;
        if (j < n) { v = B.wg_agg[2 * (im.hwg_base + j)]; f = B.wg_agg[2 * (im.hwg_base + j) + 1]; e = B.wg_eagg[im.hwg_base + j]; }
        wave_seg_scan(v, f);
        const uint32_t pv = __shfl_up(v, 1), pf = __shfl_up(f, 1);
        uint32_t in = carry;
        if (lane > 0) in = pf ? pv : carry + pv;
        uint32_t es = e;                                        // inclusive scan of entries
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(es, off); if ((int)lane >= off) es += o; }
        if (j < n) { B.wg_du_in[im.hwg_base + j] = in; B.wg_ent_in[im.hwg_base + j] = ecarry + es - e; }
        const uint32_t lv = __shfl(v, 63), lf = __shfl(f, 63);
        carry = lf ? lv : carry + lv;
        ecarry += __shfl(es, 63);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_write(PjdDevBatch B)
{
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    LaneGeom g; uint32_t tpacked; ChkCtx K;
    wave_setup(B, wg, im, g, tpacked, K);
    const uint32_t nl = im.n_luma, dus = im.dus_per_mcu;
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    uint32_t flag = 0;
    // the entry this wave was synchronised with must be what its predecessor finally produced
    if (t == 0 && !first_is_head && B.wg_entry[w] != B.wg_exit[B.n_hwg + w - 1]) flag = 1;

    // absolute data-unit index and entry index at the entry of every owned lane
    uint32_t cnt = 0, ecnt = 0, v = 0, f = 0, seg_first_du = 0, seg_n_du = 0;
    if (g.owned) {
        const uint32_t packed = B.sub_cnt[g.q];
        cnt = packed & 0xffffu; ecnt = packed >> 16;
        const PjdDevSegment sg = B.segs[g.seg];
        seg_first_du = sg.first_du; seg_n_du = sg.n_du;
        v = cnt;
        if (g.seg_first) { f = 1; v += seg_first_du; }
    }
    wave_seg_scan(v, f);
    uint32_t es = ecnt;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(es, off); if ((int)t >= off) es += o; }
    if (g.owned) {
        const uint32_t D_out = f ? v : B.wg_du_in[w] + v;
        uint32_t D = D_out - cnt;
        const uint32_t D_in = D, D_end = seg_first_du + seg_n_du;
        uint32_t p, c, z;
        if (g.seg_first) { p = g.start_bit; c = 0; z = 0; }
        else {
            const uint64_t e = (t == 1) ? B.wg_entry[w] : B.sub_exit[g.q - 1];
            p = (uint32_t)e - g.base_bit; c = (uint32_t)(e >> 32) & 255; z = (uint32_t)(e >> 40) & 255;
            if (D_in < D_end && (D_in % dus) != c) flag = 1;          // phase must agree with the count
        }
        uint32_t ndu = 0, err = 0;
        OutCtx O;
        O.ent = B.ent + im.ent_base;
        O.du_end = B.du_end + im.du_base;
        O.dcv = B.dcv + im.du_base;
        O.epos = O.epos0 = B.wg_ent_in[w] + es - ecnt;
        O.acc = make_uint4(0, 0, 0, 0);
        if (g.seg_first) B.seg_ent[g.seg] = O.epos;
        if (D_in < D_end) {
            decode_span<MODE_WRITE>(pjd_huff_lds, tpacked, nl, dus, g.base16, p, c, z, g.end_bit, ndu, err, K, &O, D, D_end);
            if (err) flag = 1;
            if (D == D_end) {
                // this lane completed the segment: the reference's BitReader must be able to reach
                // the next segment by align() alone, and must not have read past the data
                if (p > g.seg_end_bit) flag = 1;
                const bool has_next = g.seg + 1 < im.seg_base + im.n_seg;
                if (has_next && ((p + 7) & ~7u) != g.seg_end_bit) flag = 1;
            } else {
                // stopped at the subsequence end: must reproduce the synchronised exit state and counts
                const uint64_t e = B.sub_exit[g.q];
                if ((uint32_t)e - g.base_bit != p || ((uint32_t)(e >> 32) & 255) != c || ((uint32_t)(e >> 40) & 255) != z) flag = 1;
                if (ndu != B.sub_cnt[g.q]) flag = 1;
                if (g.seg_last) flag = 1;                              // data ended before all units were decoded
            }
        }
    }
    if (flag) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
}

// ---------------------------------------------------------------------------------------------
static size_t huff_lds_bytes(const PjdDevBatch &b) { return (size_t)b.max_lut_bytes + 2 * PJD_NCHK * 64 * sizeof(uint32_t); }

void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_images == 0) return;
    hipLaunchKernelGGL(pjd_k_build_tables, dim3(b.n_images * PJD_MAX_TABLES), dim3(256), 0, s, b);
}
void pjd_launch_huff_sync(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_sync, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b);
}
void pjd_launch_huff_fix(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_fix, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b);
}
void pjd_launch_huff_carry(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_carry, dim3(b.n_images), dim3(64), 0, s, b);
}
void pjd_launch_huff_write(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_write, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), huff_lds_bytes(b), s, b);
}
