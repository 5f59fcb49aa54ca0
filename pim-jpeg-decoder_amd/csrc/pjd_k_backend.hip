// pjd_k_backend.hip -- gfx950 kernels behind the entropy decoder:
//
//   pjd_k_dpu_payload   the reference's per-DPU contract, one workgroup per 16x16 "block"
//                       (reference src/decoder_dpu.c:82-390)
//   pjd_k_reset / _zero / _copy_out
//                       per-decode state reset (kernels of ours, not runtime memset nodes), packed download by a kernel
//   pjd_k_lane_dc_local /  DC prediction: the entropy decoder leaves DC DIFFERENCES and per-lane sums; the predictors at every
//   pjd_k_lane_dc_carry    lane start are a two-level segmented scan over lanes (reference src/jpeg_scanner.cpp:485-486,723-727)
//   pjd_k_idct_colour_lanes / pjd_k_idct_colour
//                       fused de-zigzag + dequantise + 8x8 IDCT + chroma upsample + YCbCr->RGB + raster / BMP store
//                       (reference src/decoder_dpu.c:158-390 and src/bmp_writer.cpp:43-65 for the BMP row order).
//                       _lanes parses the parallel decoder's lane streams (one 16-bit entry per symbol, one or two per
//                       step word); the other reads the dense int16 scratch the exact kernel fills.
//
// Integer work bound by VALU instruction issue (profiles/r03_cfg3.md: valu_issue_frac 1.0 -- the entry parser is its largest
// phase), not by HBM: coefficients are read once with 16-byte loads, tiles live in LDS (row stride 144 B so that the column
// pass is bank-conflict free), pictures are written once.
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------
// Packed download by a kernel (opt-in, PJD_DOWNLOAD=kernel): pictures go from HBM straight into mapped
// page-locked host memory.  Few workgroups on purpose: stores waiting for the link hold memory-system
// queues that other kernels need (128 workgroups per copy made concurrent decode kernels 5x slower,
// profiles/r02_pcie.md).
// ---------------------------------------------------------------------------------------------
typedef unsigned int pjd_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pjd_k_copy_out(const pjd_u32x4 *__restrict__ src, pjd_u32x4 *__restrict__ dst, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

void pjd_launch_copy_out(hipStream_t s, const void *src, void *dst_mapped, uint64_t bytes)
{
    const uint64_t n16 = bytes / 16;       // the packed output buffer is a multiple of 256 bytes
    static const int wgs = [] { const char *e = std::getenv("PJD_COPY_WGS"); const int v = e ? std::atoi(e) : 0; return v > 0 && v <= 4096 ? v : 16; }();
    if (n16) hipLaunchKernelGGL(pjd_k_copy_out, dim3(wgs), dim3(256), 0, s, (const pjd_u32x4 *)src, (pjd_u32x4 *)dst_mapped, n16);
}

// ---------------------------------------------------------------------------------------------
// Per-decode reset: status words from their initial values (images routed to the exact kernel start flagged), the
// statistics, the words the Huffman waves publish to each other, the debug timeline.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pjd_k_reset(PjdDevBatch B, const int32_t *__restrict__ status_init, uint64_t *__restrict__ opstate,
                                                   uint32_t opstate_words, uint32_t dbg_words)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < B.n_images) {
        B.status[i] = status_init[i];
        PjdDevImState st; st.err_key = ~0ull; st.flag_pos = 0xffffffffu; st.waves_done = 0;
        B.imstate[i] = st;
    }
    if (i < 16) B.stats[i] = 0;
    if (i < opstate_words) opstate[i] = 0;
    for (uint32_t k = i; k < dbg_words; k += gridDim.x * 256) B.dbg[k] = 0;
}

__global__ __launch_bounds__(256) void pjd_k_zero(uint4 *__restrict__ p, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}

void pjd_launch_zero(hipStream_t s, void *p, size_t bytes)
{
    const uint64_t n16 = bytes / 16;
    if (!n16) return;
    const uint64_t blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(pjd_k_zero, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, (uint4 *)p, n16);
}

void pjd_launch_reset(hipStream_t s, const PjdDevBatch &b, const int32_t *status_init, uint64_t *opstate, size_t opstate_words, uint32_t dbg_words)
{
    size_t n = b.n_images > 16 ? b.n_images : 16;
    if (opstate_words > n) n = opstate_words;
    hipLaunchKernelGGL(pjd_k_reset, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b, status_init, opstate, (uint32_t)opstate_words, dbg_words);
}

void pjd_launch_dpu_payload(hipStream_t s, const uint32_t *metadata, int16_t *mcus, int n_dpus)
{
    hipLaunchKernelGGL(pjd_k_dpu_payload, dim3(n_dpus * 25), dim3(128), 0, s, metadata, mcus);
}

// ---------------------------------------------------------------------------------------------
// DC prediction.  The entropy decoder emits DC DIFFERENCES and, per lane, their sum per component.
// The predictors at the start of every lane are a segmented scan over lanes (a lane that starts a
// restart segment resets them, reference src/jpeg_scanner.cpp:485-486,723-727), done in two levels:
// blocks of PJD_DC_BLOCK lanes, then one workgroup over the block aggregates.  The back end adds
// the block's carry-in itself.  All sums are modulo 2^16 like the reference's `short` stores.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Per-picture verdict of the parallel entropy decode (after pjd_k_huff_lanes, before the back end): the first entropy-coding
// error of the true decode gives the status word -- the reference's error class; its picture keeps what was decoded before the
// error (reference src/decoder_host.cpp:181 ignores the failure and writes the picture) -- unless something the decoder could
// not resolve lies at or before it: then, and for any unresolved thing in a picture without an error, the exact kernel decodes it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pjd_k_image_verdict(PjdDevBatch B)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B.n_images) return;
    const int32_t st = B.status[i];
    if (st & PJD_STW_NEEDS_EXACT) return;                       // routed to the exact kernel up front
    const PjdDevImState s = B.imstate[i];
    const bool has_err = s.err_key != ~0ull;
    const uint32_t err_pos = (uint32_t)(s.err_key >> 32);
    if (s.flag_pos != 0xffffffffu && (!has_err || s.flag_pos <= err_pos)) B.status[i] = st | PJD_STW_NEEDS_EXACT;
    else if (has_err) B.status[i] = (int32_t)((s.err_key >> 1) & 7u);
}

// Data units of a range [first_du, first_du + n_du) the back end materialises: all of them, or those up to the picture's first
// entropy-coding error (the unit that holds it included, unless the error is in its DC symbol: pjd_internal.h, PjdDevImState).
__device__ __forceinline__ uint32_t pjd_units_decoded(unsigned long long err_key, uint32_t first_du, uint32_t n_du)
{
    if (err_key == ~0ull) return n_du;
    const uint32_t stop = (uint32_t)((err_key >> 4) & 0x0fffffffu) + ((err_key & 1u) ? 0u : 1u);
    return stop <= first_du ? 0u : (stop - first_du < n_du ? stop - first_du : n_du);
}

__global__ __launch_bounds__(PJD_DC_BLOCK) void pjd_k_lane_dc_local(PjdDevBatch B)
{
    __shared__ uint32_t sy[PJD_DC_BLOCK], scb[PJD_DC_BLOCK], scr[PJD_DC_BLOCK], sf[PJD_DC_BLOCK];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t q = b * PJD_DC_BLOCK + tid;
    uint32_t vy = 0, vcb = 0, vcr = 0, head = 0;
    if (q < B.n_lanes) {
        const PjdDevLaneInfo li = B.lane_info[q];
        vy = li.dc_sum[0]; vcb = li.dc_sum[1]; vcr = li.dc_sum[2]; head = li.first_du >> 31;
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
    if (q < B.n_lanes) {
        // predictors entering this lane: the inclusive result of the lane before it, zero at a segment head
        PjdDevLaneDc d;
        d.dc_in[0] = d.dc_in[1] = d.dc_in[2] = 0;
        d.abs = (uint16_t)(head | (tid > 0 ? sf[tid - 1] : 0u));
        if (!head && tid > 0) { d.dc_in[0] = (uint16_t)sy[tid - 1]; d.dc_in[1] = (uint16_t)scb[tid - 1]; d.dc_in[2] = (uint16_t)scr[tid - 1]; }
        B.lane_dc[q] = d;
    }
    if (tid == PJD_DC_BLOCK - 1) {
        uint16_t *agg = B.dc_blk + (size_t)b * 8;
        agg[0] = (uint16_t)sy[tid]; agg[1] = (uint16_t)scb[tid]; agg[2] = (uint16_t)scr[tid]; agg[3] = (uint16_t)sf[tid];
    }
}

// One workgroup: carry-in of every block = exclusive segmented scan of the block aggregates.
__global__ __launch_bounds__(256) void pjd_k_lane_dc_carry(PjdDevBatch B)
{
    __shared__ uint32_t sy[256], scb[256], scr[256], sf[256];
    const uint32_t tid = threadIdx.x, n = B.n_dcblk;
    uint32_t cy = 0, ccb = 0, ccr = 0;                       // predictors entering the current chunk of blocks
    for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t j = base + tid;
        uint32_t vy = 0, vcb = 0, vcr = 0, vf = 0;
        if (j < n) { const uint16_t *a = B.dc_blk + (size_t)j * 8; vy = a[0]; vcb = a[1]; vcr = a[2]; vf = a[3]; }
        sy[tid] = vy; scb[tid] = vcb; scr[tid] = vcr; sf[tid] = vf;
        __syncthreads();
        for (uint32_t off = 1; off < 256; off <<= 1) {
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
        if (j < n) {
            // exclusive: what the blocks before j leave behind
            uint32_t iy = cy, icb = ccb, icr = ccr;
            if (tid > 0) {
                const bool h = sf[tid - 1] != 0;
                iy = (h ? 0u : cy) + sy[tid - 1]; icb = (h ? 0u : ccb) + scb[tid - 1]; icr = (h ? 0u : ccr) + scr[tid - 1];
            }
            uint16_t *c = B.dc_blk + (size_t)j * 8 + 4;
            c[0] = (uint16_t)iy; c[1] = (uint16_t)icb; c[2] = (uint16_t)icr; c[3] = 0;
        }
        const bool h = sf[255] != 0;
        const uint32_t ny = (h ? 0u : cy) + sy[255], ncb = (h ? 0u : ccb) + scb[255], ncr = (h ? 0u : ccr) + scr[255];
        __syncthreads();
        cy = ny; ccb = ncb; ccr = ncr;
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
// Chroma upsample + colour + raster store, specialised by sampling mode and output format (the generic version is
// the tail of pjd_tile_to_pixels).  A thread takes a block of 4 x VS pixels: the VS picture rows share their chroma
// samples (nearest neighbour, reference src/decoder_dpu.c:370), so the chroma terms of the conversion
// (reference src/decoder_dpu.c:376-382) are computed once per sample.  A wave sweeps one row group across all MCUs of
// the workgroup: its stores cover whole runs of a picture row.
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed)) PjdPx12 { uint32_t a, b, c; };

template <int HS, int VS, bool BMP>
__device__ __forceinline__ void pjd_colour_store(const int16_t (*tile)[TILE_STRIDE], const uint32_t *mcu_xy, uint8_t *out,
                                                 uint32_t width, uint32_t height, uint32_t stride, uint32_t ncomp,
                                                 uint32_t n_mcu, uint32_t tid)
{
    constexpr uint32_t MW = 8 * HS, MH = 8 * VS, NL = HS * VS;
    constexpr uint32_t CG_LOG = HS == 2 ? 2 : 1;               // log2 of the 4-pixel column groups per MCU row
    constexpr int NCH = 4 / HS;                                 // chroma samples under 4 pixels
    const uint32_t dus = NL + ncomp - 1;
    const uint32_t per_row = n_mcu << CG_LOG;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    for (uint32_t rg = wv; rg < 8; rg += PJD_IDCT_THREADS / 64) {          // row group = VS picture rows = one chroma row
        for (uint32_t idx = lane; idx < per_row; idx += 64) {
            const uint32_t ml = idx >> CG_LOG, px0 = (idx & ((1u << CG_LOG) - 1)) * 4;
            const uint32_t xy = mcu_xy[ml];
            const uint32_t X = __umul24(xy & 0xffffu, MW) + px0, Y0 = __umul24(xy >> 16, MH) + rg * VS;
            if (X >= width || Y0 >= height) continue;
            const uint32_t d0 = __umul24(ml, dus);
            // chroma terms, ordered first / middle / last output byte (R,G,B -- or B,G,R for the BMP image), +128 included
            int cf[NCH], cg[NCH], cl[NCH];
            {
                const uint32_t q = rg * 8 + px0 / HS;
                uint32_t cbw[2] = {0, 0}, crw[2] = {0, 0};
                if (ncomp > 1) {
                    if (HS == 2) cbw[0] = *reinterpret_cast<const uint32_t *>(&tile[d0 + NL][q]);
                    else { const uint2 t = *reinterpret_cast<const uint2 *>(&tile[d0 + NL][q]); cbw[0] = t.x; cbw[1] = t.y; }
                }
                if (ncomp > 2) {
                    if (HS == 2) crw[0] = *reinterpret_cast<const uint32_t *>(&tile[d0 + NL + 1][q]);
                    else { const uint2 t = *reinterpret_cast<const uint2 *>(&tile[d0 + NL + 1][q]); crw[0] = t.x; crw[1] = t.y; }
                }
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    const uint32_t bw = cbw[j >> 1], rw = crw[j >> 1];
                    const int cbv = (j & 1) ? (int)bw >> 16 : (int)(int16_t)(bw & 0xffff);
                    const int crv = (j & 1) ? (int)rw >> 16 : (int)(int16_t)(rw & 0xffff);
                    const int rC = (__mul24(5880414, crv) >> 22) + 128, bC = (__mul24(7432306, cbv) >> 22) + 128;
                    cg[j] = 128 - (__mul24(1442840, cbv) >> 22) - (__mul24(2994733, crv) >> 22);
                    cf[j] = BMP ? bC : rC;
                    cl[j] = BMP ? rC : bC;
                }
            }
#pragma unroll
            for (int v = 0; v < VS; v++) {
                const uint32_t Y = Y0 + v;
                if (Y >= height) break;
                const uint32_t py = rg * VS + v;
                const int16_t *yp = &tile[d0 + (py >> 3) * HS + (px0 >> 3)][(py & 7) * 8 + (px0 & 7)];
                const uint2 yraw = *reinterpret_cast<const uint2 *>(yp);               // 4 luma samples
                const int y0 = (int16_t)(yraw.x & 0xffff), y1 = (int)yraw.x >> 16, y2 = (int16_t)(yraw.y & 0xffff), y3 = (int)yraw.y >> 16;
                constexpr int s1 = HS == 2 ? 0 : 1, s2 = HS == 2 ? 1 : 2, s3 = HS == 2 ? 1 : 3;   // chroma sample of pixels 1..3
                const uint32_t f0 = pjd_clamp255(y0 + cf[0]), g0 = pjd_clamp255(y0 + cg[0]), l0 = pjd_clamp255(y0 + cl[0]);
                const uint32_t f1 = pjd_clamp255(y1 + cf[s1]), g1 = pjd_clamp255(y1 + cg[s1]), l1 = pjd_clamp255(y1 + cl[s1]);
                const uint32_t f2 = pjd_clamp255(y2 + cf[s2]), g2 = pjd_clamp255(y2 + cg[s2]), l2 = pjd_clamp255(y2 + cl[s2]);
                const uint32_t f3 = pjd_clamp255(y3 + cf[s3]), g3 = pjd_clamp255(y3 + cg[s3]), l3 = pjd_clamp255(y3 + cl[s3]);
                uint8_t *o = BMP ? out + 26 + (size_t)(height - 1 - Y) * stride + X * 3 : out + (size_t)Y * stride + X * 3;
                if (X + 4 <= width) {
                    PjdPx12 px;
                    px.a = f0 | (g0 << 8) | (l0 << 16) | (f1 << 24);
                    px.b = g1 | (l1 << 8) | (f2 << 16) | (g2 << 24);
                    px.c = l2 | (f3 << 8) | (g3 << 16) | (l3 << 24);
                    *reinterpret_cast<PjdPx12 *>(o) = px;
                } else {                                                             // right picture edge
                    o[0] = (uint8_t)f0; o[1] = (uint8_t)g0; o[2] = (uint8_t)l0;
                    if (X + 1 < width) { o[3] = (uint8_t)f1; o[4] = (uint8_t)g1; o[5] = (uint8_t)l1; }
                    if (X + 2 < width) { o[6] = (uint8_t)f2; o[7] = (uint8_t)g2; o[8] = (uint8_t)l2; }
                }
            }
        }
    }
}

// BMP file header exactly as reference src/bmp_writer.cpp:32-41
__device__ __forceinline__ void pjd_bmp_header(uint8_t *out, uint32_t width, uint32_t height, uint32_t stride, uint32_t tid)
{
    if (tid >= 26) return;
    const uint32_t size = 26 + height * stride;
    uint8_t hb = 0;
    switch (tid) {
        case 0: hb = 'B'; break;  case 1: hb = 'M'; break;
        case 2: hb = size & 255; break; case 3: hb = (size >> 8) & 255; break;
        case 4: hb = (size >> 16) & 255; break; case 5: hb = (size >> 24) & 255; break;
        case 10: hb = 0x1A; break; case 14: hb = 12; break;
        case 18: hb = width & 255; break; case 19: hb = (width >> 8) & 255; break;
        case 20: hb = height & 255; break; case 21: hb = (height >> 8) & 255; break;
        case 22: hb = 1; break; case 24: hb = 24; break;
        default: hb = 0;
    }
    out[tid] = hb;
}

__device__ __forceinline__ void pjd_colour_dispatch(const int16_t (*tile)[TILE_STRIDE], const uint32_t *mcu_xy, const PjdDevBatch &B,
                                                    const PjdDevImage &im, const PjdDevIdctWg &wg, uint32_t tid)
{
    uint8_t *out = B.out + im.out_off;
    const uint32_t width = im.width, height = im.height, stride = im.out_stride, nc = im.ncomp, n = wg.n_mcu;
    const bool bmp = (im.flags & PJD_IF_BMP) != 0;
    if (bmp && wg.first_mcu == 0) pjd_bmp_header(out, width, height, stride, tid);
    const uint32_t mode = (im.hs - 1) | ((im.vs - 1) << 1) | (bmp ? 4u : 0u);
    switch (mode) {
        case 0: pjd_colour_store<1, 1, false>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 1: pjd_colour_store<2, 1, false>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 2: pjd_colour_store<1, 2, false>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 3: pjd_colour_store<2, 2, false>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 4: pjd_colour_store<1, 1, true>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 5: pjd_colour_store<2, 1, true>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        case 6: pjd_colour_store<1, 2, true>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
        default: pjd_colour_store<2, 2, true>(tile, mcu_xy, out, width, height, stride, nc, n, tid); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused back end.  One workgroup = up to 96 data units = a run of consecutive MCUs of one image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_IDCT_THREADS) void pjd_k_idct_colour(PjdDevBatch B, const PjdDevIdctWg *__restrict__ wgs, const uint64_t *__restrict__ dense_base)
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
    const int16_t *cbase = B.coef + (dense_base[wg.pad_] + (uint64_t)(wg.first_mcu - im.first_mcu) * dus) * 64;
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
        if (r == 6 && !(im.flags & PJD_IF_STANDARD_ZIGZAG)) {
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
                const uint32_t nat = (r == 6 && j == 0) ? 58u : c_zz[r * 8 + j];          // r == 6 here: PJD_IF_STANDARD_ZIGZAG
                t[nat] = (int16_t)pjd_dequant(v[j] == PJD_COEF_SENTINEL ? 0 : v[j], q[nat]);
            }
        }
    }
    __syncthreads();
    pjd_tile_to_pixels<true>(tile, mcu_xy, B, im, wg, tid);
}

// ---------------------------------------------------------------------------------------------
// Lane-stream front end: the parallel entropy decoder leaves one 16-bit entry per symbol in per-lane
// regions, in 32-byte groups of a head and 14 entries (layout: pjd_internal.h), the data unit every lane starts in, and,
// for the first data unit of every IDCT workgroup, a mark (lane, slot, DC sums so far).  The workgroup finds the lanes its
// range of units lies in, and one thread per group walks the group from its head: unit index and zigzag slot of every
// entry follow from the head and the entries before it in the group; it de-zigzags and dequantises into the LDS tile.
// ---------------------------------------------------------------------------------------------

// Inclusive scans over the 64 lanes of a wave with DPP moves (VALU only, no LDS round trips): shifts inside each row of
// 16 lanes, then the last lane of a row broadcast into the following rows.  Values are unsigned; 0 is the identity of both.
#define PJD_DPP_STEP(OP, v, ctrl, rmask)                                                               \
    do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, false); v = OP(v, t_); } while (0)
__device__ __forceinline__ uint32_t pjd_op_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t pjd_op_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t pjd_op_pkadd(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pjd_op_pksub(uint32_t a, uint32_t b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
#define PJD_WAVE_SCAN(OP, v)                                                                           \
    do {                                                                                                \
        PJD_DPP_STEP(OP, v, 0x111, 0xf); PJD_DPP_STEP(OP, v, 0x112, 0xf); PJD_DPP_STEP(OP, v, 0x114, 0xf);  \
        PJD_DPP_STEP(OP, v, 0x118, 0xf); PJD_DPP_STEP(OP, v, 0x142, 0xa); PJD_DPP_STEP(OP, v, 0x143, 0xc);  \
    } while (0)
// the inclusive value of the lane before (0 in lane 0)
__device__ __forceinline__ uint32_t pjd_wave_prev(uint32_t v)
{
    const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return r;
}

// tile[u][pos] = v (low 16 bits), the row offset by one full-rate multiply-add
__device__ __forceinline__ void pjd_tile_put(uint32_t tile_lds, uint32_t u, uint32_t pos, uint32_t v)
{
    const uint32_t a = pjd_mad_u24(u, TILE_STRIDE * 2u, tile_lds + 2u * pos);
    *reinterpret_cast<__attribute__((address_space(3))) int16_t *>(a) = (int16_t)v;
}

// One back-end range (PjdDevIdctWg `iwg`) by the whole workgroup; the LDS arrays are the kernel's.  Returns are workgroup-uniform.
__device__ __forceinline__ void pjd_idct_range(const PjdDevBatch &B, uint32_t iwg, int16_t (*tile)[TILE_STRIDE], uint32_t (*qz)[64], uint32_t *mcu_xy,
                                               uint8_t *comp_of, uint32_t *wagg, uint32_t *ltab)
{
    const PjdDevIdctWg wg = B.iwgs[iwg];
    const PjdDevImage &im = B.images[wg.image];
    if ((im.flags & PJD_IF_SEQUENTIAL) || (B.status[wg.image] & PJD_STW_NEEDS_EXACT)) return;   // the dense path redoes it
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t dus = im.dus_per_mcu, nl = im.n_luma;
    const uint32_t n_du = wg.n_mcu * dus;
    const uint32_t RI = im.restart_interval;

    const bool quirk = !(im.flags & PJD_IF_STANDARD_ZIGZAG);    // the reference's zigzag_map[48] = 38 (default)
    // The reference's zigzag quirk: slots 48 AND 52 land on natural position 38, and the later one -- slot 52, even an explicit zero --
    // wins.  Entries of one unit may be parsed by two threads, so slot 52 is parked in the first padding cell of the unit's tile row
    // (position 64, raw value: quantiser 1; the cell starts as PJD_COEF_SENTINEL = "no slot 52 in this unit") and moved over
    // position 38 when the rows are done.
    if (tid < 192) {
        const uint32_t nat = (!quirk && (tid & 63) == 48) ? 58u : c_zz[tid & 63];
        uint32_t v = (uint32_t)B.qtab[(size_t)wg.image * 192 + (tid & ~63u) + nat] | (nat << 16);
        if (quirk && (tid & 63) == 52) v = 1u | (64u << 16);
        qz[tid >> 6][tid & 63] = v;
    }
    // unvisited positions are zero (the reference's buffers start zeroed); a row is 9 x 16 bytes, the last of them padding
    static_assert(TILE_STRIDE == 72, "the padding cell of a tile row is element 64");
    for (uint32_t i = tid; i < n_du * (TILE_STRIDE * 2 / 16); i += PJD_IDCT_THREADS)
        reinterpret_cast<uint4 *>(&tile[0][0])[i] = make_uint4(i % 9u == 8u ? (uint32_t)(uint16_t)PJD_COEF_SENTINEL : 0u, 0, 0, 0);
    if (tid < PJD_IDCT_MAX_DU) {
        const uint32_t kk = tid % dus;                          // the range starts on an MCU boundary
        comp_of[tid] = (uint8_t)(kk < nl ? 0 : kk - nl + 1);
    }

    // units of this range that were decoded: all, unless the picture's first entropy-coding error lies in or before the range (the
    // others keep zero coefficients, as in the reference, whose buffers start zeroed and which stops at the error)
    const unsigned long long err_key = B.imstate[wg.image].err_key;
    const uint32_t n_valid = pjd_units_decoded(err_key, wg.first_mcu * dus, n_du);
    const uint32_t err_byte = (uint32_t)(err_key >> 35);       // byte of the stream the offending symbol starts in (bit positions fit 32 bits); no error: past every lane
    const PjdDevMark mark = B.marks[iwg];
    const uint32_t lane_end = im.lane_base + im.n_lane;
    uint32_t q = mark.lane, n = mark.ent_off;
    (void)n;
    if (n_valid == 0) { q = im.lane_base; n = 0; }              // nothing to parse: the mark may never have been written
    else if (q < im.lane_base || q >= lane_end) return;         // never on a verified image; keeps a stale mark harmless
    // predictors at the first unit: lane start (block-relative or absolute) + block carry + sums inside the lane
    uint32_t pred0[3];
    {
        const PjdDevLaneDc ld = B.lane_dc[q];
        const uint16_t *carry = B.dc_blk + (size_t)(q / PJD_DC_BLOCK) * 8 + 4;
#pragma unroll
        for (int c = 0; c < 3; c++) pred0[c] = (uint32_t)ld.dc_in[c] + (ld.abs ? 0u : (uint32_t)carry[c]) + mark.acc[c];
    }
    __syncthreads();
#if defined(PJD_IDCT_STOP_AFTER) && PJD_IDCT_STOP_AFTER == 0      // timing experiments only: set-up alone
    if (tile[0][0] == 12345) B.out[0] = 1;
    return;
#endif
    // ---- parse: entries -> tile, one thread per GROUP (32 bytes: a head and 14 entries, pjd_internal.h) of a lane.  The write pass
    // left in every head where the group's first entry stands (units completed in the lane before it, slot it fills from) and with
    // every lane the unit its first entry belongs to (PjdDevLaneInfo::first_du), so a thread walks its 14 entries on its own: a DC
    // entry opens a unit, an AC entry lands on slot + run, the LAST bit closes the unit -- no scans over entries, no barriers
    // between chunks (round 2: two wave scans and two barriers per 1024 entries, ~70 instructions per entry against ~25 here).
    // Lanes are taken in windows of 32 (a range of 96 units spans 3-4 lanes of a dense picture, ~20 of 128 bytes); the window's
    // table holds the groups before each lane, its first unit relative to the range and its entry count.
    {
        // ltab: [0..31] groups before lane i of the window, [32..63] first_du - U0, [64..95] entries
        const uint32_t U0 = wg.first_mcu * dus;
        const uint32_t tile_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int16_t *)&tile[0][0];
        const uint32_t g0 = mark.ent_off / PJD_GROUP;          // the range starts in this group of lane q
        // where the NEXT range starts (its mark) bounds this one; usable when every unit of this range was decoded
        uint32_t q_end = 0xffffffffu, g_end = 0;
        if (n_valid == n_du && iwg + 1 < im.iwg_base + im.n_iwg) {
            const PjdDevMark nm = B.marks[iwg + 1];
            if (nm.lane >= q && nm.lane < lane_end) { q_end = nm.lane; g_end = nm.ent_off / PJD_GROUP; }
        }
        for (uint32_t qw = q; n_valid != 0; qw += 32) {
            bool more = false;
            if (tid < 32) {
                const uint32_t ql = qw + tid;
                uint32_t ng = 0, fd = 0, ne = 0;
                if (ql < lane_end) {
                    const PjdDevLaneInfo li = B.lane_info[ql];
                    fd = (li.first_du & 0x0fffffffu) - U0;                  // "negative" for the lane the range starts in
                    ne = li.n_ent;
                    // a lane that starts BEHIND the picture's first entropy-coding error holds what the reference never decoded; when the
                    // erring unit was still open at the error (an error in its AC part), that lane's leading entries would land in it
                    const bool in = (ql == q || (int)fd < (int)n_valid) && (ql == q || B.lanes[ql].byte_start <= err_byte);
                    if (in && ql <= q_end) {
                        const uint32_t gs = ql == q ? g0 : 0u, all = (ne + PJD_GROUP - 1) / PJD_GROUP;
                        uint32_t ge = ql == q_end ? (g_end + 1 < all ? g_end + 1 : all) : all;
                        ng = ge > gs ? ge - gs : 0u;
                    }
                    more = in && ql < q_end;
                }
                uint32_t inc = ng;
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) { const uint32_t t = __shfl_up(inc, off); if ((int)tid >= off) inc += t; }
                ltab[tid] = inc - ng; ltab[32 + tid] = fd; ltab[64 + tid] = ne;
                if (tid == 31) { wagg[0] = inc; wagg[1] = more ? 1u : 0u; }   // groups in the window; the lane behind it may belong to the range too
            }
            __syncthreads();
            const uint32_t G = wagg[0];
            const bool again = wagg[1] != 0 && qw + 32 < lane_end;
            for (uint32_t w = tid; w < G; w += PJD_IDCT_THREADS) {
                uint32_t li_ = 0;                                           // last lane of the window whose groups start at or before w
#pragma unroll
                for (uint32_t step = 16; step != 0; step >>= 1) if (ltab[li_ + step] <= w) li_ += step;
                const uint32_t ql = qw + li_;
                const uint32_t g = w - ltab[li_] + (ql == q ? g0 : 0u);
                const uint32_t ne = ltab[64 + li_];
                const uint32_t cnt = ne - g * PJD_GROUP < PJD_GROUP ? ne - g * PJD_GROUP : PJD_GROUP;      // slots of this group that are in use (even)
                const uint4 *src = reinterpret_cast<const uint4 *>(B.ent + im.ent_base + (size_t)(ql - im.lane_base) * im.lane_cap + (size_t)g * PJD_GROUP);
                const uint4 r0 = src[0], r1 = src[1];                       // the group: 32 bytes, 32-byte aligned
                const uint32_t wds[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                const uint32_t head = wds[0];
                uint32_t u = ltab[32 + li_] + (head >> 8);                  // unit of the group's first entry, relative to the range ("negative" before it)
                uint32_t slot = head & 63u;                                 // 0: that entry is a DC difference; else the next free zigzag slot of the open unit
#pragma unroll
                for (int k = 1; k < PJD_GROUP / 2; k++) {                   // the group's step words: entry A, and entry B unless it is PJD_ENT_NONE
                    const uint32_t sw = wds[k];
                    const bool on = (uint32_t)(2 * k) < cnt;
                    {   // A: ONE path for both kinds of entry (the lanes of a wave stand at DC and AC entries at once): a DC difference
                        // (slot == 0) is an entry with "run + 1" = 1 that lands on position 0 and is DEQUANTISED like any other: the DC
                        // stage then sums products instead of multiplying the sum -- the same number modulo 2^16, which is all the
                        // reference keeps (src/jpeg_scanner.cpp:485-486 stores the predictor as a short, src/decoder_dpu.c:169-172 the product)
                        const bool dc = slot == 0;
                        const uint32_t f = dc ? 1u : sw & 31u;              // run + 1; 0: EOB
                        const uint32_t ns = slot + f, pos = ns - 1u;        // an EOB gives slot - 1: stores nothing (below)
                        if (on && f != 0 && pos < 64 && u < n_valid) {
                            const int val = dc ? (int)(int16_t)(sw & 0xffffu) : (int)(sw << 16) >> 21;
                            const uint32_t qe = qz[comp_of[u]][pos];        // (slot 52 under the quirk: position 64, quantiser 1)
                            // the low 16 bits of value x quantiser (reference src/decoder_dpu.c:169-172) depend on the low 16 bits of both only
                            pjd_tile_put(tile_lds, u, qe >> 16, pjd_mul_u24((uint32_t)val, qe));
                        }
                        if (on) {
                            const bool last = f == 0 || ns > 63;            // EOB, or the entry landed on slot 63 (or past it: a broken stream)
                            slot = last ? 0u : ns;
                            u += last ? 1u : 0u;
                        }
                    }
                    {   // B: the second symbol of a pair -- an AC entry of the unit A left open
                        const uint32_t f = (sw >> 16) & 31u;
                        const bool onb = on && f <= 16;                     // PJD_ENT_NONE: "run + 1" = 31
                        const uint32_t pos = slot + f - 1u;
                        if (onb && f != 0 && pos < 64 && u < n_valid) {
                            const int val = (int)sw >> 21;
                            const uint32_t qe = qz[comp_of[u]][pos];
                            pjd_tile_put(tile_lds, u, qe >> 16, pjd_mul_u24((uint32_t)val, qe));
                        }
                        if (onb) {
                            const uint32_t ns = slot + f;
                            const bool last = f == 0 || ns > 63;
                            slot = last ? 0u : ns;
                            u += last ? 1u : 0u;
                        }
                    }
                }
            }
            __syncthreads();
            if (!again) break;
        }
    }
#if defined(PJD_IDCT_STOP_AFTER) && PJD_IDCT_STOP_AFTER == 1      // timing experiments only (tools/r2_occ.sh): pictures are wrong
    if (tile[0][0] == 12345) B.out[0] = 1;
    return;
#endif
    // ---- DC prediction over the range (reference src/jpeg_scanner.cpp:485-486) by wave 0, while the other waves already do
    //      the row pass of rows 1..7 (only row 0 holds the DC coefficient) and the slot-52 rule (natural 38 lies in row 4)
    if (wv != 0) {
        for (uint32_t i = tid - 64; i < n_du * 7; i += PJD_IDCT_THREADS - 64) {
            const uint32_t u = i / 7, r = 1 + (i - u * 7);
            if (r == 4 && tile[u][64] != (int16_t)PJD_COEF_SENTINEL) tile[u][38] = (int16_t)pjd_dequant((int)tile[u][64], qz[comp_of[u]][48] & 0xffffu);   // slot 52 over natural 38 (the quantiser of slot 48 is that position's)
            pjd_tile_row(tile, u, r);
        }
    } else {
        const uint32_t d0 = wg.first_mcu * dus;
        // the parser left DEQUANTISED differences: the predictors that enter the range are scaled the same way (all modulo 2^16)
        const uint32_t q0y = qz[0][0] & 0xffffu, q0b = qz[1][0] & 0xffffu, q0r = qz[2][0] & 0xffffu;
        uint32_t cy = (pred0[0] * q0y) & 0xffffu, cc = ((pred0[1] * q0b) & 0xffffu) | ((pred0[2] * q0r) << 16);   // predictors entering the next group of 64 units
        for (uint32_t base = 0; base < n_du; base += 64) {
            const uint32_t u = base + lane;
            const bool on = u < n_valid;                        // an undecoded unit keeps DC 0: it is never predicted
            const uint32_t d = d0 + u, m = d / dus, kk = d - m * dus, comp = kk < nl ? 0 : kk - nl + 1;
            const uint32_t dv = on ? (uint32_t)(uint16_t)tile[on ? u : 0][0] : 0u;     // the unit's DC difference as the parser left it (zero if the unit has none)
            const bool head = on && kk == 0 && (m == im.first_mcu || (RI != 0 && m % RI == 0));
            // sums since the group start (inclusive), Y | Cb, Cr packed; then the same sums at the last head at or before the unit
            uint32_t vy = comp == 0 ? dv : 0u, vc = comp == 1 ? dv : (comp == 2 ? dv << 16 : 0u);
            PJD_WAVE_SCAN(pjd_op_add, vy);
            PJD_WAVE_SCAN(pjd_op_pkadd, vc);
            uint32_t hpos = head ? lane + 1 : 0u;               // 1 + lane of the last head at or before this unit
            PJD_WAVE_SCAN(pjd_op_max, hpos);
            // a head resets the predictors BEFORE its own difference is added: subtract the sums just before it
            const uint32_t hl = hpos ? hpos - 1 : 0u;           // lane of that head
            const uint32_t by = __shfl(vy, (int)hl) - __shfl(comp == 0 ? dv : 0u, (int)hl);
            const uint32_t bc = __shfl(vc, (int)hl), bc_own = __shfl(comp == 1 ? dv : (comp == 2 ? dv << 16 : 0u), (int)hl);
            uint32_t ty = vy, tc = vc;
            if (hpos) { ty -= by; tc = pjd_op_pksub(tc, pjd_op_pksub(bc, bc_own)); }
            else { ty += cy; tc = pjd_op_pkadd(tc, cc); }
            if (on) {
                const uint32_t dcv = comp == 0 ? ty : (comp == 1 ? tc : tc >> 16);
                tile[u][0] = (int16_t)dcv;
            }
            cy = __shfl(ty, 63); cc = __shfl(tc, 63);
        }
    }
    // ---- IDCT (reference src/decoder_dpu.c:210-321): rows, then columns; then colour
    if (tid < wg.n_mcu) {                                   // grid position of each MCU: the only divisions
        const uint32_t m = wg.first_mcu + tid, my = m / im.mcux;
        mcu_xy[tid] = (my << 16) | (m - my * im.mcux);
    }
    __syncthreads();
#if defined(PJD_IDCT_STOP_AFTER) && PJD_IDCT_STOP_AFTER == 2
    if (tile[0][0] == 12345) B.out[0] = 1;
    return;
#endif
    for (uint32_t u = tid; u < n_du; u += PJD_IDCT_THREADS) pjd_tile_row(tile, u, 0);
    __syncthreads();
    for (uint32_t i = tid; i < n_du * 8; i += PJD_IDCT_THREADS) pjd_tile_col(tile, i >> 3, i & 7);
    __syncthreads();
#if defined(PJD_IDCT_STOP_AFTER) && PJD_IDCT_STOP_AFTER == 3
    if (tile[0][0] == 12345) B.out[0] = 1;
    return;
#endif
    pjd_colour_dispatch(tile, mcu_xy, B, im, wg, tid);
}

// order: the launch's workgroup -> index into PjdDevBatch::iwgs / marks (null: the identity, one launch for the whole batch)
// sweep: only ranges not marked done by the pull launch (pjd_internal.h)
__global__ __launch_bounds__(PJD_IDCT_THREADS) void pjd_k_idct_colour_lanes(PjdDevBatch B, const uint32_t *__restrict__ order, int sweep)
{
    __shared__ __attribute__((aligned(16))) int16_t tile[PJD_IDCT_MAX_DU][TILE_STRIDE];
    __shared__ uint32_t qz[3][64];            // per component, by zigzag SLOT: quantiser of its natural position | position << 16
    __shared__ uint32_t mcu_xy[PJD_IDCT_MAX_DU];
    __shared__ uint8_t comp_of[PJD_IDCT_MAX_DU];
    __shared__ uint32_t wagg[2];              // group parser: groups in the lane window; whether the lane behind the window may belong to the range
    __shared__ uint32_t ltab[96];             // group parser: the window's lane table

#if PJD_IDCT_PRIO
    __builtin_amdgcn_s_setprio(PJD_IDCT_PRIO);
#endif
    const uint32_t iwg = order ? order[blockIdx.x] : blockIdx.x;
    if (sweep && __hip_atomic_load(B.range_done + iwg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    pjd_idct_range(B, iwg, tile, qz, mcu_xy, comp_of, wagg, ltab);
}

// The pull launch (pjd_internal.h; experiment switch PJD_IDLE_FORM=pull): a few workgroups per CU stay and take the ranges of ready_list
// in order -- workgroup w the entries w, w + gridDim.x, ... -- waiting for each to appear.  A kernel of its own: inlined into the
// kernel above the loop cost it 35 registers (46 -> 81) and with them its occupancy (0.50 -> 0.66 ms, 118 -> 102 GPix/s in flight).
__global__ __launch_bounds__(PJD_IDCT_THREADS) void pjd_k_idct_pull(PjdDevBatch B)
{
    __shared__ __attribute__((aligned(16))) int16_t tile[PJD_IDCT_MAX_DU][TILE_STRIDE];
    __shared__ uint32_t qz[3][64];
    __shared__ uint32_t mcu_xy[PJD_IDCT_MAX_DU];
    __shared__ uint8_t comp_of[PJD_IDCT_MAX_DU];
    __shared__ uint32_t wagg[2];
    __shared__ uint32_t ltab[96];
    // It waits only if every Huffman workgroup has started -- else the device is busy or the launches came in an unlucky order, and
    // the ranges are left to the sweep.
    if (threadIdx.x == 0) {
        uint32_t started = 0;
        for (uint32_t it = 0; it < 64 && !started; it++) {
            started = __hip_atomic_load(B.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= B.n_hwg ? 1u : 0u;
            if (!started) __builtin_amdgcn_s_sleep(64);
        }
        wagg[1] = started;
    }
    __syncthreads();
    if (wagg[1] == 0) return;
    for (uint32_t slot = blockIdx.x; slot < B.n_iwg; slot += gridDim.x) {
        __syncthreads();                                                   // the LDS arrays of the range before
        if (threadIdx.x == 0) {
            uint32_t v = 0;
            for (uint32_t it = 0; it < PJD_PULL_SPIN_LIMIT; it++) {
                v = __hip_atomic_load(B.ready_list + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) break;
                __builtin_amdgcn_s_sleep(64);
            }
            wagg[0] = v;
        }
        __syncthreads();
        const uint32_t v = wagg[0];
        __syncthreads();
        if (v == 0) return;                                                // gave up: this entry and the workgroup's later ones go to the sweep
#if defined(PJD_PULL_EXPERIMENT) && PJD_PULL_EXPERIMENT == 1             // measurement only: take the entry, do nothing (the sweep does the work)
        continue;
#endif
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                           // what the picture's waves wrote (agent scope: other CUs)
        pjd_idct_range(B, v - 1, tile, qz, mcu_xy, comp_of, wagg, ltab);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(B.range_done + (v - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by the sweep: a later launch
    }
}

void pjd_launch_idct_colour_lanes(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_iwg == 0) return;
    hipLaunchKernelGGL(pjd_k_idct_colour_lanes, dim3(b.n_iwg), dim3(PJD_IDCT_THREADS), 0, s, b, (const uint32_t *)nullptr, 0);
}

void pjd_launch_idct_pull(hipStream_t s, const PjdDevBatch &b)
{
    // two workgroups per CU: enough to keep up with the pictures as they complete, few enough to leave the entropy decoder its issue slots
    static const uint32_t workers = [] { const char *e = std::getenv("PJD_PULL_WORKERS"); const int v = e ? std::atoi(e) : 0; return (uint32_t)(v > 0 ? v : 512); }();
    const uint32_t n = b.n_iwg < workers ? b.n_iwg : workers;
    if (n) hipLaunchKernelGGL(pjd_k_idct_pull, dim3(n), dim3(PJD_IDCT_THREADS), 0, s, b);
}

void pjd_launch_idct_sweep(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_iwg) hipLaunchKernelGGL(pjd_k_idct_colour_lanes, dim3(b.n_iwg), dim3(PJD_IDCT_THREADS), 0, s, b, (const uint32_t *)nullptr, 1);
}

void pjd_launch_group_idct(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g)
{
    if (g.iwg_count) hipLaunchKernelGGL(pjd_k_idct_colour_lanes, dim3(g.iwg_count), dim3(PJD_IDCT_THREADS), 0, s, b, b.iwg_order + g.iwg_first, 0);
}

// ---------------------------------------------------------------------------------------------
// One picture group: verdict and DC predictors per PICTURE, one workgroup each (pictures of a group are small: the planner makes
// groups only if no picture has more than 8192 lanes).  The same results as pjd_k_image_verdict + pjd_k_lane_dc_local / _carry leave,
// with every lane's predictors absolute (PjdDevLaneDc::abs = 1: no block carry to add).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PJD_DC_BLOCK) void pjd_k_group_dc(PjdDevBatch B, const uint32_t *__restrict__ images)
{
    __shared__ uint32_t sy[PJD_DC_BLOCK], scb[PJD_DC_BLOCK], scr[PJD_DC_BLOCK], sf[PJD_DC_BLOCK];
    const uint32_t i = images[blockIdx.x], tid = threadIdx.x;
    const PjdDevImage &im = B.images[i];
    if (tid == 0) {                                            // pjd_k_image_verdict
        const int32_t st = B.status[i];
        if (!(st & PJD_STW_NEEDS_EXACT)) {
            const PjdDevImState s = B.imstate[i];
            const bool has_err = s.err_key != ~0ull;
            const uint32_t err_pos = (uint32_t)(s.err_key >> 32);
            if (s.flag_pos != 0xffffffffu && (!has_err || s.flag_pos <= err_pos)) B.status[i] = st | PJD_STW_NEEDS_EXACT;
            else if (has_err) B.status[i] = (int32_t)((s.err_key >> 1) & 7u);
        }
    }
    uint32_t cy = 0, ccb = 0, ccr = 0;                         // predictors entering the current chunk of lanes
    for (uint32_t base = 0; base < im.n_lane; base += PJD_DC_BLOCK) {
        const uint32_t q = im.lane_base + base + tid;
        const bool on = base + tid < im.n_lane;
        uint32_t vy = 0, vcb = 0, vcr = 0, head = 0;
        if (on) {
            const PjdDevLaneInfo li = B.lane_info[q];
            vy = li.dc_sum[0]; vcb = li.dc_sum[1]; vcr = li.dc_sum[2]; head = li.first_du >> 31;
        }
        sy[tid] = vy; scb[tid] = vcb; scr[tid] = vcr; sf[tid] = head;
        __syncthreads();
        for (uint32_t off = 1; off < PJD_DC_BLOCK; off <<= 1) {      // Hillis-Steele inclusive segmented scan
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
        if (on) {
            // predictors entering this lane: zero at a segment head; else what the lanes before it leave -- inside the chunk, plus the
            // chunk's carry-in unless a head lies between the chunk start and this lane
            PjdDevLaneDc d;
            d.dc_in[0] = d.dc_in[1] = d.dc_in[2] = 0;
            d.abs = 1;
            if (!head) {
                uint32_t py = cy, pcb = ccb, pcr = ccr;
                if (tid > 0) {
                    const bool h = sf[tid - 1] != 0;
                    py = (h ? 0u : cy) + sy[tid - 1]; pcb = (h ? 0u : ccb) + scb[tid - 1]; pcr = (h ? 0u : ccr) + scr[tid - 1];
                }
                d.dc_in[0] = (uint16_t)py; d.dc_in[1] = (uint16_t)pcb; d.dc_in[2] = (uint16_t)pcr;
            }
            B.lane_dc[q] = d;
        }
        const bool h = sf[PJD_DC_BLOCK - 1] != 0;
        const uint32_t ny = (h ? 0u : cy) + sy[PJD_DC_BLOCK - 1], ncb = (h ? 0u : ccb) + scb[PJD_DC_BLOCK - 1], ncr = (h ? 0u : ccr) + scr[PJD_DC_BLOCK - 1];
        __syncthreads();
        cy = ny; ccb = ncb; ccr = ncr;
    }
}

void pjd_launch_group_dc(hipStream_t s, const PjdDevBatch &b, const PjdDevGroup &g)
{
    if (g.img_count) hipLaunchKernelGGL(pjd_k_group_dc, dim3(g.img_count), dim3(PJD_DC_BLOCK), 0, s, b, b.group_images + g.img_first);
}

void pjd_launch_lane_dc_scan(hipStream_t s, const PjdDevBatch &b)
{
    if (b.n_dcblk == 0) return;
    hipLaunchKernelGGL(pjd_k_image_verdict, dim3((b.n_images + 255) / 256), dim3(256), 0, s, b);
    hipLaunchKernelGGL(pjd_k_lane_dc_local, dim3(b.n_dcblk), dim3(PJD_DC_BLOCK), 0, s, b);
    hipLaunchKernelGGL(pjd_k_lane_dc_carry, dim3(1), dim3(256), 0, s, b);
}

void pjd_launch_idct_colour(hipStream_t s, const PjdDevBatch &b, const PjdDevIdctWg *wgs, const uint64_t *dense_base, uint32_t n_wg)
{
    if (n_wg == 0) return;
    hipLaunchKernelGGL(pjd_k_idct_colour, dim3(n_wg), dim3(PJD_IDCT_THREADS), 0, s, b, wgs, dense_base);
}

