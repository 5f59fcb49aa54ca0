// pjd_k_huffman.hip -- the PARALLEL entropy decoder for gfx950.
//
// Huffman decoding is one dependent chain per restart segment; a batch of ImageNet files has
// ~10^3 chains, a single 4K picture has one.  To fill 256 CUs the bitstream is cut into
// fixed 128-byte SUBSEQUENCES, one decode lane each, and the lanes find their true entry
// states by self-synchronisation (a decoder started at a wrong position falls into step with
// the true one after a few symbols; Weissenberger & Schmidt describe the scheme for GPUs):
//
//   pjd_k_build_tables   raw (offsets, symbols) tables -> 10-bit first-level LUT + canonical
//                        arrays; semantics of reference generate_codes / get_next_symbol
//                        (reference src/jpeg_scanner.cpp:438-465)
//   pjd_k_huff_sync      one workgroup = 255 owned subsequences of one image (+1 overlap lane).
//                        Bitstream bytes are staged ONCE into LDS with coalesced 16-byte loads,
//                        byte-swapped, 33-dword padded rows (bank-conflict free); tables in LDS.
//                        Every lane decodes its subsequence speculatively, then lanes re-decode
//                        from their predecessor's exit state until nothing changes.
//   pjd_k_huff_fix       stitches workgroup boundaries (the overlap lane's guess vs the truth)
//                        and reduces per-workgroup data-unit counts
//   pjd_k_huff_carry     one wave per image: scan of those counts -> absolute data-unit index
//   pjd_k_huff_write     final pass from the now-known entry states: coefficients are written,
//                        in zigzag-slot order, to coef[(du_base + D) * 64 + slot]; slot 0 holds
//                        the DC DIFFERENCE (pjd_k_dc_* integrates it)
//
// Exactness: a lane that starts from the true state performs exactly the reference's
// decode_MCU_component (reference src/jpeg_scanner.cpp:467-520).  Anything irregular seen in
// the final pass -- an invalid code, a size or run outside the baseline limits, a segment that
// ends early or late, a boundary that did not stitch -- sets PJD_STW_NEEDS_EXACT on the image
// and the host re-decodes it with the one-lane exact kernel (pjd_k_huffman_seq.hip).
#include "pjd_device_common.h"
#include "pjd_kernels.h"

#define LUT_BYTES        PJD_LUT_STRUCT_BYTES
#define STREAM_DWORDS    8512                 // (256*128 + 15 + 16 + 15)/4 rounded up, plus 1/32 padding
#define OFF_FIRST        (2 << PJD_LUT_BITS)  // byte offsets inside PjdDevHuffLut
#define OFF_OFFS         (OFF_FIRST + 68)
#define OFF_SYMS         (OFF_OFFS + 20)

static_assert(sizeof(PjdDevHuffLut) == LUT_BYTES, "LUT struct layout");
static_assert(sizeof(PjdDevHuffRaw) == 180, "raw table layout");

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pjd_k_build_tables(PjdDevBatch B)
{
    const uint32_t img = blockIdx.x / PJD_MAX_TABLES, slot = blockIdx.x % PJD_MAX_TABLES;
    if (slot >= B.images[img].n_tables) return;
    const PjdDevHuffRaw &r = B.raw_tables[(size_t)img * PJD_MAX_TABLES + slot];
    PjdDevHuffLut &o = B.luts[(size_t)img * PJD_MAX_TABLES + slot];
    __shared__ uint32_t first[17];
    __shared__ uint8_t offs[17];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        uint32_t code = 0;                       // reference generate_codes
        first[0] = 0;
        for (int len = 1; len <= 16; len++) {
            first[len] = code;
            code = (code + (uint32_t)(r.offsets[len] - r.offsets[len - 1])) << 1;
        }
    }
    if (tid < 17) offs[tid] = r.offsets[tid];
    __syncthreads();
    for (uint32_t idx = tid; idx < (1u << PJD_LUT_BITS); idx += 256) {
        uint16_t e = 0;
        for (uint32_t len = 1; len <= PJD_LUT_BITS; len++) {     // shortest match wins, as the reference's scan
            const uint32_t c = idx >> (PJD_LUT_BITS - len);
            const uint32_t d = c - first[len], cnt = (uint32_t)offs[len] - offs[len - 1];
            if (c >= first[len] && d < cnt) { e = (uint16_t)((len << 8) | r.symbols[offs[len - 1] + d]); break; }
        }
        o.lut[idx] = e;
    }
    if (tid < 17) { o.first[tid] = first[tid]; o.offs[tid] = offs[tid]; }
    if (tid < 164) o.symbols[tid] = tid < 162 ? r.symbols[tid] : 0;
}

// ---------------------------------------------------------------------------------------------
// Per-lane decoder over the LDS-staged stream.
// ---------------------------------------------------------------------------------------------
struct HuffLds {
    uint32_t stream[STREAM_DWORDS];                                  // big-endian dwords, row-padded
    __attribute__((aligned(4))) uint8_t tabs[PJD_MAX_TABLES * LUT_BYTES];
};

// table slot (0..5) of {DC,AC} x component, 4 bits each: bits [4*(2*comp+is_ac) +: 4].  One register,
// selected arithmetically -- a struct of six offsets gets demoted to scratch memory by the compiler.
struct LaneTables { uint32_t packed; };

__device__ __forceinline__ uint32_t stream_dword(const uint32_t *s, uint32_t d) { return s[d + (d >> 5)]; }

struct BitWin {
    uint64_t buf;
    int cnt;
    uint32_t nd;
    __device__ __forceinline__ void init(const uint32_t *s, uint32_t p)
    {
        nd = p >> 5;
        buf = ((uint64_t)stream_dword(s, nd) << 32) | stream_dword(s, nd + 1);
        nd += 2;
        cnt = 64 - (int)(p & 31);
        buf <<= (p & 31);
    }
    __device__ __forceinline__ uint32_t peek(const uint32_t *s)
    {
        if (cnt <= 32) { buf |= (uint64_t)stream_dword(s, nd++) << (32 - cnt); cnt += 32; }
        return (uint32_t)(buf >> 32);
    }
    __device__ __forceinline__ void drop(uint32_t n) { buf <<= n; cnt -= (int)n; }
};

// Decodes symbols that START before end_bit.  State (p, c, z): bit position relative to the
// staged base, data-unit phase within the MCU, zigzag slot (0 = DC expected).
template <bool WRITE>
__device__ __forceinline__ void decode_span(const HuffLds &L, const LaneTables &T, uint32_t nl, uint32_t dus,
                                            uint32_t &p, uint32_t &c, uint32_t &z, uint32_t end_bit,
                                            uint32_t &ndu, uint32_t &err,
                                            int16_t *coef_img, uint32_t &D, uint32_t D_end)
{
    if (p >= end_bit) return;
    BitWin w;
    w.init(L.stream, p);
    while (p < end_bit && (!WRITE || D < D_end)) {
        const uint32_t pk = w.peek(L.stream);
        const bool is_dc = (z == 0);
        const uint32_t comp = (c >= nl ? 1u : 0u) + (c > nl ? 1u : 0u);
        const uint32_t tb = ((T.packed >> (8 * comp + (is_dc ? 0u : 4u))) & 15u) * LUT_BYTES;
        const uint8_t *tab = L.tabs + tb;
        const uint32_t e = *reinterpret_cast<const uint16_t *>(tab + 2 * (pk >> (32 - PJD_LUT_BITS)));
        uint32_t len = e >> 8, sym = e & 255;
        if (len == 0) {                                   // code longer than the LUT (rare) or invalid
            const uint32_t *first = reinterpret_cast<const uint32_t *>(tab + OFF_FIRST);
            const uint8_t *offs = tab + OFF_OFFS;
            const uint32_t code16 = pk >> 16;
            len = 16; sym = 0;
            bool found = false;
            for (uint32_t l = PJD_LUT_BITS + 1; l <= 16; l++) {
                const uint32_t cc = code16 >> (16 - l);
                const uint32_t d = cc - first[l], n = (uint32_t)offs[l] - offs[l - 1];
                if (cc >= first[l] && d < n) { sym = tab[OFF_SYMS + offs[l - 1] + d]; len = l; found = true; break; }
            }
            if (!found) err |= 1;
        }
        const uint32_t size = sym & 15, run = sym >> 4;
        const uint32_t bits = size ? ((pk << len) >> (32 - size)) : 0;
        int val = (int)bits;
        if (size && !(bits >> (size - 1))) val -= (int)((1u << size) - 1);
        const uint32_t used = len + size;
        w.drop(used);
        p += used;
        if (is_dc) {
            err |= (sym > 11);                            // jpeg_scanner.cpp:470-477
            if (WRITE) coef_img[(size_t)D * 64] = (int16_t)val;
            z = 1;
        } else if (sym == 0) {
            z = 64;                                       // EOB
        } else {
            z += run;
            if (z > 63) { err |= 1; z = 64; }             // jpeg_scanner.cpp:500
            else {
                err |= (size > 10);                       // jpeg_scanner.cpp:506
                if (WRITE) {
                    if (size) coef_img[(size_t)D * 64 + z] = (int16_t)val;
                    else if (z == 52) coef_img[(size_t)D * 64 + z] = (int16_t)PJD_COEF_SENTINEL;
                }
                z += 1;
            }
        }
        if (z >= 64) {
            z = 0;
            c = (c + 1 == dus) ? 0 : c + 1;
            ndu++;
            if (WRITE) D++;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Common per-workgroup set-up: lane geometry, LDS staging of bitstream and tables.
// ---------------------------------------------------------------------------------------------
struct LaneGeom {
    bool valid, owned, seg_first, seg_last;
    uint32_t q;              // global subsequence index
    uint32_t seg;            // global segment index
    uint32_t start_bit, end_bit;   // relative to the staged base
    uint32_t seg_end_bit;          // relative to the staged base (may be far beyond the staged window)
    uint32_t base_bit;       // staged base, bits relative to the image's ecs
};

__device__ __forceinline__ void wg_setup(const PjdDevBatch &B, const PjdDevHuffWg &wg, const PjdDevImage &im,
                                         HuffLds &L, LaneGeom &g, LaneTables &T, bool stage)
{
    const uint32_t t = threadIdx.x;
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    g.owned = t >= 1 && t - 1 < wg.n_sub;
    g.valid = g.owned || (t == 0 && !first_is_head);
    g.q = wg.first_sub + t - 1;
    const uint32_t lane_lo = first_is_head ? 1 : 0;
    const uint32_t lo_byte = B.subs[wg.first_sub + lane_lo - 1].byte_start;
    const uint32_t lo16 = lo_byte & ~15u;
    g.base_bit = lo16 * 8;
    g.seg_first = g.seg_last = false;
    g.seg = 0; g.start_bit = g.end_bit = g.seg_end_bit = 0;
    if (g.valid) {
        const PjdDevSub sb = B.subs[g.q];
        g.seg = sb.seg & 0x7fffffffu;
        g.seg_first = (sb.seg >> 31) != 0;
        const PjdDevSegment sg = B.segs[g.seg];
        const uint32_t end_byte = sb.byte_start + PJD_SUBSEQ_BYTES < sg.byte_end ? sb.byte_start + PJD_SUBSEQ_BYTES : sg.byte_end;
        g.seg_last = end_byte == sg.byte_end;
        g.start_bit = (sb.byte_start - lo16) * 8;
        g.end_bit = (end_byte - lo16) * 8;
        g.seg_end_bit = (sg.byte_end - lo16) * 8;
    }
    T.packed = 0;
    for (int cc = 0; cc < 3; cc++)
        T.packed |= ((uint32_t)im.tbl_slot[cc][0] << (8 * cc)) | ((uint32_t)im.tbl_slot[cc][1] << (8 * cc + 4));
    if (!stage) return;
    // bitstream: [lo16, end of the last owned subsequence + 16), 16 B per lane per step
    const PjdDevSub last = B.subs[wg.first_sub + wg.n_sub - 1];
    const PjdDevSegment lseg = B.segs[last.seg & 0x7fffffffu];
    const uint32_t hi_byte = (last.byte_start + PJD_SUBSEQ_BYTES < lseg.byte_end ? last.byte_start + PJD_SUBSEQ_BYTES : lseg.byte_end) + 16;
    const uint32_t n16 = (hi_byte - lo16 + 15) / 16;
    const uint4 *src = reinterpret_cast<const uint4 *>(B.ecs + im.ecs_off + lo16);
    for (uint32_t i = t; i < n16; i += PJD_HUFF_THREADS) {
        const uint4 v = src[i];
        const uint32_t d = 4 * i, ph = d + (d >> 5);
        L.stream[ph] = __builtin_bswap32(v.x);
        L.stream[ph + 1] = __builtin_bswap32(v.y);
        L.stream[ph + 2] = __builtin_bswap32(v.z);
        L.stream[ph + 3] = __builtin_bswap32(v.w);
    }
    // zero the few dwords a refill may touch past the copied range
    if (t < 8) { const uint32_t d = 4 * n16 + t; L.stream[d + (d >> 5)] = 0; }
    // tables
    const uint32_t *tsrc = reinterpret_cast<const uint32_t *>(B.luts + (size_t)wg.image * PJD_MAX_TABLES);
    uint32_t *tdst = reinterpret_cast<uint32_t *>(L.tabs);
    const uint32_t ndw = im.n_tables * (LUT_BYTES / 4);
    for (uint32_t i = t; i < ndw; i += PJD_HUFF_THREADS) tdst[i] = tsrc[i];
    __syncthreads();
}

// Re-synchronisation rounds: lane t (owned, not at a segment start) re-decodes from E[t-1]
// whenever E[t-1] changed in the previous round.  Returns false if the cap was hit.
__device__ __forceinline__ bool resync_rounds(const HuffLds &L, const LaneTables &T, const LaneGeom &g,
                                              uint32_t nl, uint32_t dus,
                                              uint32_t *Ep, uint32_t *Ecz, uint32_t *Ecnt, uint32_t *chg /*[2][256]*/,
                                              unsigned long long *stats, int stat_base)
{
    const uint32_t t = threadIdx.x;
    int cur = 0;
    for (int iter = 0; iter < PJD_SYNC_MAX_ITERS; iter++) {
        const bool act = g.owned && !g.seg_first && chg[cur * PJD_HUFF_THREADS + t - 1] != 0;
        uint32_t p = 0, c = 0, z = 0;
        if (act) { p = Ep[t - 1]; const uint32_t cz = Ecz[t - 1]; c = cz >> 8; z = cz & 255; }
        __syncthreads();
        if (act && stats) atomicAdd(stats + stat_base + 1, 1ull);
        if (threadIdx.x == 0 && stats) atomicAdd(stats + stat_base, 1ull);
        uint32_t changed = 0;
        if (act) {
            uint32_t ndu = 0, err = 0, D = 0;
            decode_span<false>(L, T, nl, dus, p, c, z, g.end_bit, ndu, err, nullptr, D, 0);
            const uint32_t cz = (c << 8) | z;
            Ecnt[t] = ndu;
            if (p != Ep[t] || cz != Ecz[t]) { Ep[t] = p; Ecz[t] = cz; changed = 1; }
        }
        chg[(cur ^ 1) * PJD_HUFF_THREADS + t] = changed;
        if (!__syncthreads_or((int)changed)) return true;
        cur ^= 1;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_sync(PjdDevBatch B)
{
    __shared__ HuffLds L;
    __shared__ uint32_t Ep[PJD_HUFF_THREADS], Ecz[PJD_HUFF_THREADS], Ecnt[PJD_HUFF_THREADS], chg[2 * PJD_HUFF_THREADS];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    LaneGeom g; LaneTables T;
    wg_setup(B, wg, im, L, g, T, true);
    const uint32_t nl = im.n_luma, dus = im.dus_per_mcu;

    // round 0: every lane from the start of its own subsequence, state (DC of unit 0 expected)
    uint32_t p = g.start_bit, c = 0, z = 0, ndu = 0, err = 0, D = 0;
    if (g.valid) decode_span<false>(L, T, nl, dus, p, c, z, g.end_bit, ndu, err, nullptr, D, 0);
    Ep[t] = p; Ecz[t] = (c << 8) | z; Ecnt[t] = ndu;
    chg[t] = g.valid ? 1 : 0;
    __syncthreads();
    const bool ok = resync_rounds(L, T, g, nl, dus, Ep, Ecz, Ecnt, chg, B.stats, 0);
    if (!ok && t == 0) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);

    if (g.owned) {
        B.sub_exit[g.q] = pjd_pack_state(Ep[t] + g.base_bit, Ecz[t] >> 8, Ecz[t] & 255);
        B.sub_cnt[g.q] = Ecnt[t];
    }
    if (t == 0) B.wg_entry[w] = g.valid ? pjd_pack_state(Ep[0] + g.base_bit, Ecz[0] >> 8, Ecz[0] & 255) : ~0ull;
    if (t == wg.n_sub) B.wg_exit[w] = pjd_pack_state(Ep[t] + g.base_bit, Ecz[t] >> 8, Ecz[t] & 255);
}

// ---------------------------------------------------------------------------------------------
// Segmented combine used for data-unit counts: element = (value, head flag); a head resets.
__device__ __forceinline__ void seg_combine(uint32_t av, uint32_t af, uint32_t &bv, uint32_t &bf)   // b = a (+) b
{
    if (!bf) bv += av;
    bf |= af;
}

// Inclusive segmented scan over the 256 lanes of a workgroup (wave shuffles + one LDS hop).
__device__ __forceinline__ void wg_seg_scan(uint32_t &v, uint32_t &f, uint32_t *sv, uint32_t *sf)
{
    const uint32_t t = threadIdx.x, lane = t & 63, wv = t >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ov = __shfl_up(v, off), of = __shfl_up(f, off);
        if ((int)lane >= off) seg_combine(ov, of, v, f);
    }
    if (lane == 63) { sv[wv] = v; sf[wv] = f; }
    __syncthreads();
    uint32_t cv = 0, cf = 0;
    for (uint32_t k = 0; k < wv; k++) { uint32_t bv = sv[k], bf = sf[k]; seg_combine(cv, cf, bv, bf); cv = bv; cf = bf; }
    seg_combine(cv, cf, v, f);
    __syncthreads();
}

__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_fix(PjdDevBatch B)
{
    __shared__ HuffLds L;
    __shared__ uint32_t Ep[PJD_HUFF_THREADS], Ecz[PJD_HUFF_THREADS], Ecnt[PJD_HUFF_THREADS], chg[2 * PJD_HUFF_THREADS];
    __shared__ uint32_t sv[4], sf[4];
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
    if (redo) {                                       // workgroup-uniform
        LaneGeom g; LaneTables T;
        wg_setup(B, wg, im, L, g, T, true);
        uint64_t e = g.owned ? B.sub_exit[g.q] : truth;
        Ep[t] = (uint32_t)e - g.base_bit; Ecz[t] = (((uint32_t)(e >> 32) & 255) << 8) | ((uint32_t)(e >> 40) & 255);
        Ecnt[t] = g.owned ? B.sub_cnt[g.q] : 0;
        chg[t] = (t == 0) ? 1 : 0;
        __syncthreads();
        const bool ok = resync_rounds(L, T, g, im.n_luma, im.dus_per_mcu, Ep, Ecz, Ecnt, chg, B.stats, 2);
        if (!ok && t == 0) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
        if (g.owned) {
            B.sub_exit[g.q] = pjd_pack_state(Ep[t] + g.base_bit, Ecz[t] >> 8, Ecz[t] & 255);
            B.sub_cnt[g.q] = Ecnt[t];
        }
        if (t == 0) B.wg_entry[w] = truth;
        if (t == wg.n_sub) exit1[w] = pjd_pack_state(Ep[t] + g.base_bit, Ecz[t] >> 8, Ecz[t] & 255);
        __syncthreads();
    } else if (t == 0) {
        exit1[w] = B.wg_exit[w];
    }
    // per-workgroup aggregate of data-unit counts: (absolute index after the last owned lane if a
    // segment starts inside, else the number of units completed), head flag
    uint32_t v = 0, f = 0;
    if (t >= 1 && t - 1 < wg.n_sub) {
        const uint32_t q = wg.first_sub + t - 1;
        const uint32_t sg = B.subs[q].seg;
        v = redo ? Ecnt[t] : B.sub_cnt[q];
        if (sg >> 31) { f = 1; v += B.segs[sg & 0x7fffffffu].first_du; }
    }
    wg_seg_scan(v, f, sv, sf);
    if (t == PJD_HUFF_THREADS - 1) { B.wg_agg[2 * w] = v; B.wg_agg[2 * w + 1] = f; }
}

__global__ __launch_bounds__(64) void pjd_k_huff_carry(PjdDevBatch B)
{
    const PjdDevImage &im = B.images[blockIdx.x];
    const uint32_t lane = threadIdx.x, n = im.n_hwg;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        uint32_t v = 0, f = 0;
        if (j < n) { v = B.wg_agg[2 * (im.hwg_base + j)]; f = B.wg_agg[2 * (im.hwg_base + j) + 1]; }
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ov = __shfl_up(v, off), of = __shfl_up(f, off);
            if ((int)lane >= off) seg_combine(ov, of, v, f);
        }
        const uint32_t pv = __shfl_up(v, 1), pf = __shfl_up(f, 1);
        uint32_t in = carry;
        if (lane > 0) in = pf ? pv : carry + pv;
        if (j < n) B.wg_du_in[im.hwg_base + j] = in;
        const uint32_t lv = __shfl(v, 63), lf = __shfl(f, 63);
        carry = lf ? lv : carry + lv;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_HUFF_THREADS) void pjd_k_huff_write(PjdDevBatch B)
{
    __shared__ HuffLds L;
    __shared__ uint32_t sv[4], sf[4];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const PjdDevHuffWg wg = B.hwgs[w];
    const PjdDevImage &im = B.images[wg.image];
    LaneGeom g; LaneTables T;
    wg_setup(B, wg, im, L, g, T, true);
    const uint32_t nl = im.n_luma, dus = im.dus_per_mcu;
    const bool first_is_head = (B.subs[wg.first_sub].seg >> 31) != 0;
    uint32_t flag = 0;
    // the entry this workgroup was synchronised with must be what its predecessor finally produced
    if (t == 0 && !first_is_head && B.wg_entry[w] != B.wg_exit[B.n_hwg + w - 1]) flag = 1;

    // absolute data-unit index at the entry of every owned lane
    uint32_t cnt = 0, v = 0, f = 0, seg_first_du = 0, seg_n_du = 0;
    if (g.owned) {
        cnt = B.sub_cnt[g.q];
        const PjdDevSegment sg = B.segs[g.seg];
        seg_first_du = sg.first_du; seg_n_du = sg.n_du;
        v = cnt;
        if (g.seg_first) { f = 1; v += seg_first_du; }
    }
    wg_seg_scan(v, f, sv, sf);
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
        int16_t *coef_img = B.coef + im.du_base * 64;
        if (D_in < D_end || g.seg_last) {
            decode_span<true>(L, T, nl, dus, p, c, z, g.end_bit, ndu, err, coef_img, D, D_end);
            if (err) flag = 1;
            if (D == D_end && D_in < D_end) {
                // this lane completed the segment: the reference's BitReader must be able to reach
                // the next segment by align() alone, and must not have read past the data
                if (p > g.seg_end_bit) flag = 1;
                const bool has_next = g.seg + 1 < im.seg_base + im.n_seg;
                if (has_next && ((p + 7) & ~7u) != g.seg_end_bit) flag = 1;
            } else if (D < D_end) {
                // stopped at the subsequence end: must reproduce the synchronised exit state
                const uint64_t e = B.sub_exit[g.q];
                if ((uint32_t)e - g.base_bit != p || ((uint32_t)(e >> 32) & 255) != c || ((uint32_t)(e >> 40) & 255) != z) flag = 1;
                if (g.seg_last) flag = 1;                              // data ended before all units were decoded
            }
        }
    }
    if (flag) atomicOr(reinterpret_cast<unsigned int *>(B.status + wg.image), PJD_STW_NEEDS_EXACT);
}

// ---------------------------------------------------------------------------------------------
void pjd_launch_build_tables(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_images == 0) return;
    hipLaunchKernelGGL(pjd_k_build_tables, dim3(b.n_images * PJD_MAX_TABLES), dim3(256), 0, s, b);
}
void pjd_launch_huff_sync(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_sync, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), 0, s, b);
}
void pjd_launch_huff_fix(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_fix, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), 0, s, b);
}
void pjd_launch_huff_carry(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_carry, dim3(b.n_images), dim3(64), 0, s, b);
}
void pjd_launch_huff_write(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_hwg) hipLaunchKernelGGL(pjd_k_huff_write, dim3(b.n_hwg), dim3(PJD_HUFF_THREADS), 0, s, b);
}
