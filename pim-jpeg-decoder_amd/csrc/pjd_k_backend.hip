// pjd_k_backend.hip -- gfx950 kernels behind the entropy decoder:
//
//   pjd_k_dpu_payload   the reference's per-DPU contract, one workgroup per 16x16 "block"
//                       (reference src/decoder_dpu.c:82-390)
//   pjd_k_dc_local /    DC prediction as a two-level segmented scan over MCUs
//   pjd_k_dc_carry      (reference src/jpeg_scanner.cpp:485-486,723-727)
//   pjd_k_idct_colour_sparse / pjd_k_idct_colour
//                       fused de-zigzag + dequantise + 8x8 IDCT + chroma upsample + YCbCr->RGB +
//                       raster store (reference src/decoder_dpu.c:158-390 and src/bmp_writer.cpp:43-65
//                       for the BMP row order).  _sparse reads the parallel decoder's entry stream;
//                       the other reads the dense int16 scratch the exact kernel fills.
//
// HBM-bound integer work: coefficients are read once with 16-byte loads, tiles
// live in LDS (row stride 144 B so that the column pass is bank-conflict free),
// pictures are written once.
#include "pjd_device_common.h"
#include "pjd_kernels.h"

#define TILE_STRIDE 72   // int16 per data unit in LDS: 64 + 8 pad (144 B = 36 banks)

// ---------------------------------------------------------------------------------------------
// Literal DPU payload: metadata u32[276] + mcus i16[19200] per DPU.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void pjd_k_dpu_payload(const uint32_t *__restrict__ meta_all, int16_t *__restrict__ mcus_all)
{
    __shared__ __attribute__((aligned(16))) int16_t tile[12][TILE_STRIDE];   // [comp*4 + pos][64]
    const int dpu = blockIdx.x / 25, blk = blockIdx.x % 25;
    const uint32_t *m = meta_all + (size_t)dpu * 276;
    int16_t *base = mcus_all + (size_t)dpu * 19200 + blk * 768;
    const int tid = threadIdx.x;
    const int ncomp = (int)m[4];
    const int V = (int)(m[5] & 255), H = (int)(m[6] & 255);

    // dequantise (decoder_dpu.c:158-177) and row pass (:218-268): lane owns one row
    if (tid < 96) {
        const int du = tid >> 3, r = tid & 7, comp = du >> 2;
        const int4 raw = *reinterpret_cast<const int4 *>(base + du * 64 + r * 8);
        const int16_t *rv = reinterpret_cast<const int16_t *>(&raw);
        int x[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = rv[j];
        if (comp < ncomp) {
            const uint32_t *q = m + 20 + (m[7 + comp] & 255) * 64 + r * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) x[j] = pjd_dequant(x[j], q[j]);
        }
        pjd_idct8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], o);
#pragma unroll
        for (int j = 0; j < 8; j++) tile[du][r * 8 + j] = (int16_t)o[j];
    }
    __syncthreads();
    // column pass (decoder_dpu.c:270-320)
    if (tid < 96) {
        const int du = tid >> 3, c = tid & 7;
        int x[8], o[8];
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = tile[du][j * 8 + c];
        pjd_idct8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], o);
#pragma unroll
        for (int j = 0; j < 8; j++) tile[du][j * 8 + c] = (int16_t)o[j];
    }
    __syncthreads();
    // colour (decoder_dpu.c:323-390).  (cbcr_index, v, h) per position follow the four tuples
    // of each sampling mode; all reads come from LDS, all writes go to HBM, so the
    // reference's in-place ordering constraints disappear.
    for (int i = tid; i < 256; i += 128) {
        const int pos = i >> 6, p = i & 63, y = p >> 3, x = p & 7;
        int cidx, v, h;
        if (V == 1 && H == 1)      { cidx = pos;     v = 0;        h = 0; }
        else if (V == 2 && H == 1) { cidx = pos & 1; v = pos >> 1; h = 0; }
        else if (V == 1 && H == 2) { cidx = pos & 2; v = 0;        h = pos & 1; }
        else                       { cidx = 0;       v = pos >> 1; h = pos & 1; }
        const int q = ((y / V) + 4 * v) * 8 + (x / H) + 4 * h;
        int r, g, b;
        pjd_ycc_to_rgb(tile[pos][p], tile[4 + cidx][q], tile[8 + cidx][q], r, g, b);
        base[pos * 64 + p] = (int16_t)r;
        base[256 + pos * 64 + p] = (int16_t)g;
        base[512 + pos * 64 + p] = (int16_t)b;
    }
}

void pjd_launch_dpu_payload(hipStream_t s, const uint32_t *metadata, int16_t *mcus, int n_dpus)
{
    hipLaunchKernelGGL(pjd_k_dpu_payload, dim3(n_dpus * 25), dim3(128), 0, s, metadata, mcus);
}

// ---------------------------------------------------------------------------------------------
// DC prediction.  After the parallel entropy decode, dcv[] holds the DC DIFFERENCE of every data
// unit.  Level 1: each workgroup takes PJD_DC_BLOCK MCUs of one image, scans the
// per-component sums with resets at restart points, and rewrites slot 0 with the prediction
// relative to the block start.  Level 2: one wave per image scans the block aggregates.  The
// IDCT kernel adds the carry.  All sums are modulo 2^16 like the reference's `short` stores.
// ---------------------------------------------------------------------------------------------
struct DcTriple { uint32_t y, cb, cr, f; };

__global__ __launch_bounds__(PJD_DC_BLOCK) void pjd_k_dc_local(PjdDevBatch B)
{
    __shared__ uint32_t sy[PJD_DC_BLOCK], scb[PJD_DC_BLOCK], scr[PJD_DC_BLOCK], sf[PJD_DC_BLOCK];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t ii = B.dcblk_image[b];
    const PjdDevImage &im = B.images[ii];
    uint32_t *agg = B.dc_agg + (size_t)b * 4;
    const bool exact = (im.flags & PJD_IF_SEQUENTIAL) || (B.status[ii] & PJD_STW_NEEDS_EXACT);
    if (exact) {                       // slot 0 already holds absolute DC values
        if (tid < 4) agg[tid] = 0;
        return;
    }
    const uint32_t m = (b - im.dcblk_base) * PJD_DC_BLOCK + tid;
    const bool live = m >= im.first_mcu && m < im.last_mcu;
    const uint32_t RI = im.restart_interval, dus = im.dus_per_mcu, nl = im.n_luma, nc = im.ncomp;
    int16_t *du0 = B.dcv + im.du_base + (uint64_t)m * dus;
    uint32_t d[6] = {0, 0, 0, 0, 0, 0};     // fully unrolled below: stays in registers
    uint32_t vy = 0, vcb = 0, vcr = 0, head = 0;
    if (live) {
#pragma unroll
        for (uint32_t k = 0; k < 6; k++)
            if (k < dus) {
                const uint32_t dv = (uint32_t)(int32_t)du0[k];
                d[k] = dv;
                if (k < nl) vy += dv; else if (k == nl) vcb = dv; else vcr = dv;
            }
        head = (m == im.first_mcu) || (RI != 0 && m % RI == 0);
    }
    sy[tid] = vy; scb[tid] = vcb; scr[tid] = vcr; sf[tid] = head;
    __syncthreads();
    // Hillis-Steele inclusive segmented scan
    for (uint32_t off = 1; off < PJD_DC_BLOCK; off <<= 1) {
        uint32_t ay = 0, acb = 0, acr = 0, af = 0;
        const bool take = tid >= off;
        if (take) { ay = sy[tid - off]; acb = scb[tid - off]; acr = scr[tid - off]; af = sf[tid - off]; }
        const uint32_t myf = sf[tid];
        __syncthreads();
        if (take) {
            if (!myf) { sy[tid] += ay; scb[tid] += acb; scr[tid] += acr; }
            sf[tid] = myf | af;
        }
        __syncthreads();
    }
    if (live) {
        uint32_t py = 0, pcb = 0, pcr = 0;            // prediction entering this MCU
        if (!head && tid > 0) { py = sy[tid] - vy; pcb = scb[tid] - vcb; pcr = scr[tid] - vcr; }
#pragma unroll
        for (uint32_t k = 0; k < 6; k++)
            if (k < dus) {
                uint32_t val;
                if (k < nl) { py += d[k]; val = py; }
                else if (k == nl) val = pcb + d[k];
                else val = pcr + d[k];
                du0[k] = (int16_t)val;
            }
    }
    if (tid == PJD_DC_BLOCK - 1) { agg[0] = sy[tid]; agg[1] = scb[tid]; agg[2] = scr[tid]; agg[3] = sf[tid]; }
}

__device__ __forceinline__ DcTriple dc_combine(const DcTriple &a, const DcTriple &b)   // a then b
{
    DcTriple r;
    r.y = b.f ? b.y : a.y + b.y; r.cb = b.f ? b.cb : a.cb + b.cb; r.cr = b.f ? b.cr : a.cr + b.cr;
    r.f = a.f | b.f;
    return r;
}

__global__ __launch_bounds__(64) void pjd_k_dc_carry(PjdDevBatch B)
{
    const PjdDevImage &im = B.images[blockIdx.x];
    const uint32_t lane = threadIdx.x, n = im.n_dcblk;
    DcTriple carry = {0, 0, 0, 0};
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t j = base + lane;
        DcTriple v = {0, 0, 0, 0};
        if (j < n) { const uint32_t *a = B.dc_agg + (size_t)(im.dcblk_base + j) * 4; v.y = a[0]; v.cb = a[1]; v.cr = a[2]; v.f = a[3]; }
        for (int off = 1; off < 64; off <<= 1) {
            DcTriple o;
            o.y = __shfl_up(v.y, off); o.cb = __shfl_up(v.cb, off); o.cr = __shfl_up(v.cr, off); o.f = __shfl_up(v.f, off);
            if ((int)lane >= off) v = dc_combine(o, v);
        }
        DcTriple prev;                     // inclusive result of lane-1
        prev.y = __shfl_up(v.y, 1); prev.cb = __shfl_up(v.cb, 1); prev.cr = __shfl_up(v.cr, 1); prev.f = __shfl_up(v.f, 1);
        DcTriple in = (lane == 0) ? carry : dc_combine(carry, prev);
        if (j < n) { uint32_t *c = B.dc_carry + (size_t)(im.dcblk_base + j) * 4; c[0] = in.y; c[1] = in.cb; c[2] = in.cr; c[3] = 0; }
        DcTriple last;
        last.y = __shfl(v.y, 63); last.cb = __shfl(v.cb, 63); last.cr = __shfl(v.cr, 63); last.f = __shfl(v.f, 63);
        carry = dc_combine(carry, last);
        carry.f = 0;
    }
}

__device__ __forceinline__ void pjd_tile_row(int16_t (*tile)[TILE_STRIDE], uint32_t du, uint32_t r)
{
    int4 raw = *reinterpret_cast<const int4 *>(&tile[du][r * 8]);
    int16_t *rv = reinterpret_cast<int16_t *>(&raw);
    int o[8];
    pjd_idct8(rv[0], rv[1], rv[2], rv[3], rv[4], rv[5], rv[6], rv[7], o);
#pragma unroll
    for (int j = 0; j < 8; j++) rv[j] = (int16_t)o[j];
    *reinterpret_cast<int4 *>(&tile[du][r * 8]) = raw;
}

__device__ __forceinline__ void pjd_tile_col(int16_t (*tile)[TILE_STRIDE], uint32_t du, uint32_t c)
{
    int x[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = tile[du][j * 8 + c];
    pjd_idct8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], o);
#pragma unroll
    for (int j = 0; j < 8; j++) tile[du][j * 8 + c] = (int16_t)o[j];
}

// Row pass, column pass, chroma upsample + colour + raster store for the data units staged in `tile`
// (natural order, dequantised).  Shared by the sparse and the dense front ends.
// DO_IDCT = false: the caller has already run both passes on every unit (and this function's first barrier is
// the one that separates them from the colour stage).
template <bool DO_IDCT>
__device__ __forceinline__ void pjd_tile_to_pixels(int16_t (*tile)[TILE_STRIDE], uint32_t *mcu_xy, const PjdDevBatch &B,
                                                   const PjdDevImage &im, const PjdDevIdctWg &wg, uint32_t tid)
{
    if (tid < wg.n_mcu) {                                   // grid position of each MCU: the only divisions
        const uint32_t m = wg.first_mcu + tid, my = m / im.mcux;
        mcu_xy[tid] = (my << 16) | (m - my * im.mcux);
    }
    const uint32_t dus = im.dus_per_mcu, nl = im.n_luma, nc = im.ncomp, hs = im.hs, vs = im.vs;
    const uint32_t n_du = wg.n_mcu * dus;
    __syncthreads();
    if (DO_IDCT) {
        for (uint32_t i = tid; i < n_du * 8; i += PJD_IDCT_THREADS) pjd_tile_row(tile, i >> 3, i & 7);
        __syncthreads();
        for (uint32_t i = tid; i < n_du * 8; i += PJD_IDCT_THREADS) pjd_tile_col(tile, i >> 3, i & 7);
        __syncthreads();
    }

    // ---- chroma upsample (nearest neighbour, decoder_dpu.c:370), colour, raster store --------
    const uint32_t mw = 8 * hs, mh = 8 * vs;
    const bool bmp = (im.flags & PJD_IF_BMP) != 0;
    uint8_t *out = B.out + im.out_off;
    // image constants in registers: read through `im` they are re-fetched from HBM after every store
    const uint32_t width = im.width, height = im.height, stride = im.out_stride;
    if (bmp && wg.first_mcu == 0 && tid < 26) {
        // file header exactly as reference src/bmp_writer.cpp:32-41
        const uint32_t size = 26 + im.height * im.out_stride;
        uint8_t hb = 0;
        switch (tid) {
            case 0: hb = 'B'; break;  case 1: hb = 'M'; break;
            case 2: hb = size & 255; break; case 3: hb = (size >> 8) & 255; break;
            case 4: hb = (size >> 16) & 255; break; case 5: hb = (size >> 24) & 255; break;
            case 10: hb = 0x1A; break; case 14: hb = 12; break;
            case 18: hb = im.width & 255; break; case 19: hb = (im.width >> 8) & 255; break;
            case 20: hb = im.height & 255; break; case 21: hb = (im.height >> 8) & 255; break;
            case 22: hb = 1; break; case 24: hb = 24; break;
            default: hb = 0;
        }
        out[tid] = hb;
    }
    // Colour + store: a thread takes FOUR horizontally adjacent pixels (12 output bytes, written as one
    // 12-byte store -- gfx950 global stores need no alignment), thread rows of 64 items sweep `mh` picture
    // rows four at a time.  No runtime divisions: mw is 8 or 16, the MCU grid position was tabulated once.
    const uint32_t q_log = hs == 2 ? 2u : 1u;                  // log2(mw / 4): items per MCU row
    const uint32_t hs_log = hs - 1, vs_log = vs - 1;
    const uint32_t items = wg.n_mcu << q_log;
    struct __attribute__((packed)) Px12 { uint32_t a, b, c; };
    // item (MCU, 4-pixel column group) outside, picture row inside: everything that depends only on the item
    // (grid position, X, the unit indices) is computed once per item instead of once per row
    for (uint32_t it = tid & 63; it < items; it += 64) {
        const uint32_t ml = it >> q_log, px0 = (it & ((1u << q_log) - 1)) * 4;
        const uint32_t xy = mcu_xy[ml];
        const uint32_t X = __umul24(xy & 0xffffu, mw) + px0, Y0 = __umul24(xy >> 16, mh);
        if (X >= width) continue;
        const uint32_t d0 = __umul24(ml, dus);
        for (uint32_t py = tid >> 6; py < mh; py += PJD_IDCT_THREADS / 64) {
            const uint32_t cy = py >> vs_log, lrow = (py >> 3) * hs, yoff = (py & 7) * 8;
            const uint32_t Y = Y0 + py;
            if (Y >= height) continue;
            const int16_t *yp = &tile[d0 + lrow + (px0 >> 3)][yoff + (px0 & 7)];
            const uint2 yraw = *reinterpret_cast<const uint2 *>(yp);               // 4 luma samples
            const int y0 = (int16_t)(yraw.x & 0xffff), y1 = (int16_t)(yraw.x >> 16), y2 = (int16_t)(yraw.y & 0xffff), y3 = (int16_t)(yraw.y >> 16);
            // chroma terms of pjd_ycc_to_rgb, once per chroma SAMPLE (two pixels share one when hs == 2), already
            // ordered first / middle / last output byte (R,G,B -- or B,G,R for the BMP image) and with the +128
            const uint32_t q = cy * 8 + (px0 >> hs_log);
            int cf[4], cg[4], cl[4];
            const int n_chroma = hs == 2 ? 2 : 4;
            uint2 cbw = make_uint2(0, 0), crw = make_uint2(0, 0);
            if (nc > 1) { if (hs == 2) cbw.x = *reinterpret_cast<const uint32_t *>(&tile[d0 + nl][q]); else cbw = *reinterpret_cast<const uint2 *>(&tile[d0 + nl][q]); }
            if (nc > 2) { if (hs == 2) crw.x = *reinterpret_cast<const uint32_t *>(&tile[d0 + nl + 1][q]); else crw = *reinterpret_cast<const uint2 *>(&tile[d0 + nl + 1][q]); }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j < n_chroma) {
                    const uint32_t bw = j < 2 ? cbw.x : cbw.y, rw = j < 2 ? crw.x : crw.y;
                    const int cbv = (j & 1) ? (int)(int16_t)(bw >> 16) : (int)(int16_t)(bw & 0xffff);
                    const int crv = (j & 1) ? (int)(int16_t)(rw >> 16) : (int)(int16_t)(rw & 0xffff);
                    const int rC = (__mul24(5880414, crv) >> 22) + 128, bC = (__mul24(7432306, cbv) >> 22) + 128;
                    cg[j] = 128 - (__mul24(1442840, cbv) >> 22) - (__mul24(2994733, crv) >> 22);
                    cf[j] = bmp ? bC : rC;
                    cl[j] = bmp ? rC : bC;
                }
            }
            const int s1 = hs == 2 ? 0 : 1, s2 = hs == 2 ? 1 : 2, s3 = hs == 2 ? 1 : 3;   // chroma sample of pixels 1..3
            const int cf1 = s1 ? cf[1] : cf[0], cg1 = s1 ? cg[1] : cg[0], cl1 = s1 ? cl[1] : cl[0];
            const int cf2 = s2 == 2 ? cf[2] : cf[1], cg2 = s2 == 2 ? cg[2] : cg[1], cl2 = s2 == 2 ? cl[2] : cl[1];
            const int cf3 = s3 == 3 ? cf[3] : cf[1], cg3 = s3 == 3 ? cg[3] : cg[1], cl3 = s3 == 3 ? cl[3] : cl[1];
            // reference src/decoder_dpu.c:376-382: y + term + 128, clamped
            const uint32_t f0 = pjd_clamp255(y0 + cf[0]), g0 = pjd_clamp255(y0 + cg[0]), l0 = pjd_clamp255(y0 + cl[0]);
            const uint32_t f1 = pjd_clamp255(y1 + cf1), g1 = pjd_clamp255(y1 + cg1), l1 = pjd_clamp255(y1 + cl1);
            const uint32_t f2 = pjd_clamp255(y2 + cf2), g2 = pjd_clamp255(y2 + cg2), l2 = pjd_clamp255(y2 + cl2);
            const uint32_t f3 = pjd_clamp255(y3 + cf3), g3 = pjd_clamp255(y3 + cg3), l3 = pjd_clamp255(y3 + cl3);
            uint8_t *o = bmp ? out + 26 + (size_t)(height - 1 - Y) * stride + X * 3
                             : out + (size_t)Y * stride + X * 3;
            if (X + 4 <= width) {
                Px12 v;
                v.a = f0 | (g0 << 8) | (l0 << 16) | (f1 << 24);
                v.b = g1 | (l1 << 8) | (f2 << 16) | (g2 << 24);
                v.c = l2 | (f3 << 8) | (g3 << 16) | (l3 << 24);
                *reinterpret_cast<Px12 *>(o) = v;
            } else {                                                             // right picture edge
                o[0] = (uint8_t)f0; o[1] = (uint8_t)g0; o[2] = (uint8_t)l0;
                if (X + 1 < width) { o[3] = (uint8_t)f1; o[4] = (uint8_t)g1; o[5] = (uint8_t)l1; }
                if (X + 2 < width) { o[6] = (uint8_t)f2; o[7] = (uint8_t)g2; o[8] = (uint8_t)l2; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused back end.  One workgroup = up to 96 data units = a run of consecutive MCUs of one image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_IDCT_THREADS) void pjd_k_idct_colour(PjdDevBatch B, const PjdDevIdctWg *__restrict__ wgs)
{
    __shared__ __attribute__((aligned(16))) int16_t tile[PJD_IDCT_MAX_DU][TILE_STRIDE];
    __shared__ uint16_t qs[3][64];
    __shared__ uint32_t mcu_xy[PJD_IDCT_MAX_DU];

    const PjdDevIdctWg wg = wgs[blockIdx.x];
    const PjdDevImage &im = B.images[wg.image];
    const uint32_t tid = threadIdx.x;
    const uint32_t dus = im.dus_per_mcu, nl = im.n_luma;
    const uint32_t n_du = wg.n_mcu * dus;
    const uint32_t RI = im.restart_interval;

    if (tid < 192) qs[tid >> 6][tid & 63] = B.qtab[(size_t)wg.image * 192 + tid];
    __syncthreads();

    // ---- load (16 B per lane, coalesced), DC fix-up, de-zigzag, dequantise, row pass ----------
    // lane (du, r) owns zigzag slots 8r..8r+7 on load; after the scatter to natural order a
    // second sweep does the row pass.
    const int16_t *cbase = B.coef + (im.dense_base + (uint64_t)wg.first_mcu * dus) * 64;
    for (uint32_t i = tid; i < n_du * 8; i += PJD_IDCT_THREADS) {
        const uint32_t du = i >> 3, r = i & 7;
        const uint32_t ml = du / dus, k = du - ml * dus;
        const uint32_t comp = k < nl ? 0 : k - nl + 1;
        const int4 raw = *reinterpret_cast<const int4 *>(cbase + (size_t)du * 64 + r * 8);
        const int16_t *rv = reinterpret_cast<const int16_t *>(&raw);
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = rv[j];
        int16_t *t = tile[du];
        const uint16_t *q = qs[comp];
        if (r == 6) {
            // slots 48..55.  Natural position 38 is the target of slot 48 AND slot 52 (the
            // reference's zigzag_map[48] = 38): the later write wins, and an explicit zero
            // written at slot 52 (run/size symbol with size 0) is marked by the sentinel.
            const int v52 = v[4];
            const int n38 = v52 != 0 ? (v52 == PJD_COEF_SENTINEL ? 0 : v52) : v[0];
            t[38] = (int16_t)pjd_dequant(n38, q[38]);
            t[59] = (int16_t)pjd_dequant(v[1], q[59]);
            t[52] = (int16_t)pjd_dequant(v[2], q[52]);
            t[45] = (int16_t)pjd_dequant(v[3], q[45]);
            t[31] = (int16_t)pjd_dequant(v[5], q[31]);
            t[39] = (int16_t)pjd_dequant(v[6], q[39]);
            t[46] = (int16_t)pjd_dequant(v[7], q[46]);
            t[58] = 0;                    // natural 58 is never written by the reference
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t nat = c_zz[r * 8 + j];
                t[nat] = (int16_t)pjd_dequant(v[j], q[nat]);
            }
        }
    }
    __syncthreads();
    pjd_tile_to_pixels<true>(tile, mcu_xy, B, im, wg, tid);
}

// ---------------------------------------------------------------------------------------------
// Sparse front end: the parallel entropy decoder leaves, per data unit, a run of 4-byte entries
// (value << 16 | zigzag slot, AC only) delimited by du_end[] / seg_ent[], and the DC in dcv[].
// ---------------------------------------------------------------------------------------------
// The eight threads that own a data unit (one wave always holds all eight) take it from entries to
// finished samples on their own: clear, scatter (de-zigzag + dequantise), row pass, column pass.  LDS
// traffic of one wave is processed in program order, so wave-local fences are all these steps need;
// the workgroup meets once, before the colour stage, which mixes units of different waves.
#define PJD_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__global__ __launch_bounds__(PJD_IDCT_THREADS) void pjd_k_idct_colour_sparse(PjdDevBatch B, const PjdDevIdctWg *__restrict__ wgs)
{
    __shared__ __attribute__((aligned(16))) int16_t tile[PJD_IDCT_MAX_DU][TILE_STRIDE];
    __shared__ uint32_t qz[3][64];            // per component, by zigzag SLOT: quantiser of its natural position | position << 16
    __shared__ uint32_t mcu_xy[PJD_IDCT_MAX_DU];

    const PjdDevIdctWg wg = wgs[blockIdx.x];
    const PjdDevImage &im = B.images[wg.image];
    if ((im.flags & PJD_IF_SEQUENTIAL) || (B.status[wg.image] & PJD_STW_NEEDS_EXACT)) return;   // the dense path redoes it
    const uint32_t tid = threadIdx.x;
    const uint32_t dus = im.dus_per_mcu, nl = im.n_luma;
    const uint32_t n_du = wg.n_mcu * dus;
    const uint32_t RI = im.restart_interval;
    const uint32_t d0 = wg.first_mcu * dus;                     // image-relative index of the first unit

    if (tid < 192) {
        const uint32_t nat = c_zz[tid & 63];
        qz[tid >> 6][tid & 63] = (uint32_t)B.qtab[(size_t)wg.image * 192 + (tid & ~63u) + nat] | (nat << 16);
    }
    __syncthreads();

    const uint32_t *ent = B.ent + im.ent_base;
    const uint32_t *de = B.du_end + im.du_base;
    const int16_t *dcv = B.dcv + im.du_base;
    // A thread owns one eighth of up to NIT data units.  Everything it needs from HBM is requested before
    // anything is used: bounds and DC of all its units, then the first entry of each (a unit rarely has more
    // than eight entries), so a workgroup waits for two memory round trips, not six.
    constexpr int NIT = PJD_IDCT_MAX_DU * 8 / PJD_IDCT_THREADS;
    uint32_t lo[NIT], hi[NIT], comp[NIT], w0[NIT], s52[NIT];
    int dc[NIT];
    bool on[NIT];
    const uint32_t r = tid & 7;
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        const uint32_t du = (tid >> 3) + k * (PJD_IDCT_THREADS / 8);
        on[k] = du < n_du;
        lo[k] = hi[k] = 0; comp[k] = 0; dc[k] = 0;
        if (on[k]) {
            const uint32_t d = d0 + du, m = d / dus, kk = d - m * dus;
            comp[k] = kk < nl ? 0 : kk - nl + 1;
            // this unit's entries: [lo, hi) of the image's stream (the eight threads read the same two words)
            const bool seg_first = kk == 0 && (m == im.first_mcu || (RI != 0 && m % RI == 0));
            lo[k] = seg_first ? B.seg_ent[im.seg_base + (RI ? m / RI - im.first_mcu / RI : 0)] : de[d - 1];
            hi[k] = de[d];
            if (r == 0) {
                dc[k] = dcv[d];
                const uint32_t blk = m / PJD_DC_BLOCK;
                const uint32_t hm = RI ? (m / RI) * RI : 0;      // last restart point at or before m
                if (hm < blk * PJD_DC_BLOCK)                      // none inside this scan block: carry applies
                    dc[k] = (int)(int16_t)((uint32_t)dc[k] + B.dc_carry[(size_t)(im.dcblk_base + blk) * 4 + comp[k]]);
            }
            // unvisited positions are zero (the reference's buffers start zeroed): 144 bytes = 9 x 16
            int16_t *t = tile[du];
            *reinterpret_cast<uint4 *>(t + r * 8) = make_uint4(0, 0, 0, 0);
            if (r == 0) *reinterpret_cast<uint4 *>(t + 64) = make_uint4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < NIT; k++) w0[k] = (on[k] && lo[k] + r < hi[k]) ? ent[lo[k] + r] : 0xffffffffu;
    PJD_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NIT; k++) {
        s52[k] = 0;                                              // entry at slot 52, if this thread met it
        if (on[k]) {
            const uint32_t du = (tid >> 3) + k * (PJD_IDCT_THREADS / 8);
            const uint32_t *q = qz[comp[k]];
            int16_t *t = tile[du];
            if (r == 0) t[0] = (int16_t)pjd_dequant(dc[k], q[0] & 0xffffu);
            uint32_t w = w0[k];
            for (uint32_t e = lo[k] + r; e < hi[k]; ) {
                const uint32_t slot = w & 63;
                const int val = (int)(int16_t)(w >> 16);
                if (slot == 52) s52[k] = 0x80000000u | (w >> 16);   // overrides slot 48 at natural 38, even when zero
                else { const uint32_t e = q[slot]; t[e >> 16] = (int16_t)pjd_dequant(val, e & 0xffffu); }
                e += 8;
                if (e < hi[k]) w = ent[e];
            }
        }
        s52[k] |= __shfl_xor(s52[k], 1); s52[k] |= __shfl_xor(s52[k], 2); s52[k] |= __shfl_xor(s52[k], 4);   // a unit visits slot 52 at most once
    }
    PJD_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NIT; k++)
        if (on[k] && s52[k] && r == 0)
            tile[(tid >> 3) + k * (PJD_IDCT_THREADS / 8)][38] = (int16_t)pjd_dequant((int)(int16_t)(s52[k] & 0xffffu), qz[comp[k]][48] & 0xffffu);   // slot 48 -> natural 38
    PJD_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NIT; k++) if (on[k]) pjd_tile_row(tile, (tid >> 3) + k * (PJD_IDCT_THREADS / 8), r);
    PJD_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < NIT; k++) if (on[k]) pjd_tile_col(tile, (tid >> 3) + k * (PJD_IDCT_THREADS / 8), r);
    pjd_tile_to_pixels<false>(tile, mcu_xy, B, im, wg, tid);
}

void pjd_launch_idct_colour_sparse(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, uint32_t n_wg)
{
    if (n_wg == 0) return;
    hipLaunchKernelGGL(pjd_k_idct_colour_sparse, dim3(n_wg), dim3(PJD_IDCT_THREADS), 0, s, b, wgs);
}

void pjd_launch_idct_colour(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, uint32_t n_wg)
{
    if (n_wg == 0) return;
    hipLaunchKernelGGL(pjd_k_idct_colour, dim3(n_wg), dim3(PJD_IDCT_THREADS), 0, s, b, wgs);
}

void pjd_launch_dc_scan(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_dcblk == 0) return;
    hipLaunchKernelGGL(pjd_k_dc_local, dim3(b.n_dcblk), dim3(PJD_DC_BLOCK), 0, s, b);
    hipLaunchKernelGGL(pjd_k_dc_carry, dim3(b.n_images), dim3(64), 0, s, b);
}
