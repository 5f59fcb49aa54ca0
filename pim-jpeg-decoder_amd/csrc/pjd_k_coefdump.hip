// pjd_k_coefdump.hip -- stage-level parity: what the entropy decoder produced, re-laid-out as the reference's
// MCU_buffer (reference src/jpeg_scanner.cpp:733-741: int16 [blk16 * 768 + component * 256 + position * 64 + natural index],
// absolute DC values, coefficients placed through the reference's zigzag_map, everything not visited zero).
//
// Not on the product path: pjd_batch_download_coefficients (include/pjd.h) launches these so that a test can hash the GPU's
// coefficients against the hashes the reference's own decode_Huffman_data produced (tests/golden/manifest.json, coef_sha256).
// Deliberately simple -- one thread walks one IDCT workgroup's range of the lane streams entry by entry, in stream order, so
// "the later write wins" (zigzag slots 48 and 52 both land on natural 38) needs no special case -- and deliberately NOT the
// back end's parser (pjd_k_idct_colour_lanes): it reads the same marks, lane sums and entries with independent code.
#include "pjd_device_common.h"
#include "pjd_kernels.h"

namespace {

// destination of data unit D (image-relative, interleaved order) in the reference's buffer (jpeg_scanner.cpp:733-741)
__device__ __forceinline__ size_t ref_unit_base(const PjdDevImage &im, uint32_t D, uint32_t &comp)
{
    const uint32_t dus = im.dus_per_mcu, m = D / dus, k = D - m * dus;
    uint32_t v = 0, h = 0;
    if (k < im.n_luma) { comp = 0; v = k / im.hs; h = k - v * im.hs; }
    else comp = k - im.n_luma + 1;
    const uint32_t y = (m / im.mcux) * im.vs + v, x = (m % im.mcux) * im.hs + h, Wr = im.ref_mcu_w_real;
    const uint32_t m8 = y * Wr + x;
    const uint32_t blk = (m8 / (2 * Wr)) * ((Wr + 1) / 2) + (m8 % Wr) / 2;
    const uint32_t pos = ((m8 / Wr) % 2) * 2 + (m8 % Wr) % 2;
    return (size_t)blk * 768 + (size_t)comp * 256 + (size_t)pos * 64;
}

__device__ __forceinline__ uint32_t ref_natural(const PjdDevImage &im, uint32_t slot)
{
    return ((im.flags & PJD_IF_STANDARD_ZIGZAG) && slot == 48) ? 58u : c_zz[slot];
}

}  // namespace

// Lane streams -> reference layout.  One thread per IDCT workgroup (= coefficient range) of `image`.
__global__ __launch_bounds__(64) void pjd_k_coefdump_lanes(PjdDevBatch B, uint32_t image, int16_t *__restrict__ out)
{
    const PjdDevImage &im = B.images[image];
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= im.n_iwg) return;
    const PjdDevIdctWg wg = B.iwgs[im.iwg_base + k];
    const PjdDevMark mark = B.marks[im.iwg_base + k];
    const uint32_t dus = im.dus_per_mcu, RI = im.restart_interval;
    const uint32_t lane_end = im.lane_base + im.n_lane;
    uint32_t q = mark.lane, n = mark.ent_off;
    if (q < im.lane_base || q >= lane_end) return;
    int pred[3];
    {
        const PjdDevLaneDc ld = B.lane_dc[q];
        const uint16_t *carry = B.dc_blk + (size_t)(q / PJD_DC_BLOCK) * 8 + 4;
        for (int c = 0; c < 3; c++) pred[c] = (int)(int16_t)(uint16_t)(ld.dc_in[c] + (ld.abs ? 0u : (uint32_t)carry[c]) + mark.acc[c]);
    }
    uint32_t n_ent = B.lane_info[q].n_ent;
    uint32_t D = wg.first_mcu * dus;
    uint32_t D_stop = D + wg.n_mcu * dus;
    {   // nothing behind the picture's first entropy-coding error was decoded (PjdDevImState)
        const unsigned long long key = B.imstate[image].err_key;
        if (key != ~0ull) {
            const uint32_t stop = (uint32_t)((key >> 4) & 0x0fffffffu) + ((key & 1u) ? 0u : 1u);
            if (stop < D_stop) D_stop = stop;
        }
    }
    if (D >= D_stop) return;
    uint32_t slot = 0, comp = 0;
    size_t base = 0;
    bool in_unit = false;
    bool second = false;                     // the next entry to take is the high half of the step word at n
    const uint16_t *region = B.ent + im.ent_base + (size_t)(q - im.lane_base) * im.lane_cap;
    while (D < D_stop) {
        if (n >= n_ent) {
            q++; n = 0; second = false;
            if (q >= lane_end) break;
            n_ent = B.lane_info[q].n_ent;
            region = B.ent + im.ent_base + (size_t)(q - im.lane_base) * im.lane_cap;
            continue;
        }
        if ((n & (PJD_GROUP - 1)) < 2) { n = (n & ~(uint32_t)(PJD_GROUP - 1)) + 2; continue; }      // a group's head (pjd_internal.h)
        // a step word: entry A, then entry B unless it is PJD_ENT_NONE
        const uint32_t e = region[n + (second ? 1u : 0u)];
        if (second) { second = false; n += 2; if ((e & 31u) > 16u) continue; }
        else second = true;
        if (!in_unit) {                      // DC entry (layout: pjd_internal.h): the difference as int16
            const uint32_t m = D / dus, kk = D - m * dus;
            if (kk == 0 && (m == im.first_mcu || (RI != 0 && m % RI == 0))) pred[0] = pred[1] = pred[2] = 0;   // jpeg_scanner.cpp:723-727
            base = ref_unit_base(im, D, comp);
            const int diff = (int)(int16_t)e;
            pred[comp] = (int)(int16_t)(pred[comp] + diff);                                                      // :485-486
            out[base] = (int16_t)pred[comp];
            slot = 0;
            in_unit = true;
            continue;
        }
        const uint32_t f = e & 31u;          // run + 1; 0: EOB
        bool last = f == 0;
        if (f != 0) {                        // run / value, stored through the zigzag map (:517)
            slot += f;
            if (slot < 64) out[base + ref_natural(im, slot)] = (int16_t)((int)(e << 16) >> 21);
            if (slot >= 63) last = true;
        }
        if (last) { in_unit = false; D++; }
    }
}

// Dense scratch of the exact kernel (zigzag-slot order, absolute DC, PJD_COEF_SENTINEL = explicit zero at slot 52) ->
// reference layout.  One thread per data unit; `first_du` = image-relative index of the unit at scratch position 0.
__global__ __launch_bounds__(256) void pjd_k_coefdump_dense(PjdDevBatch B, uint32_t image, const int16_t *__restrict__ scratch,
                                                            uint32_t first_du, uint32_t n_du, int16_t *__restrict__ out)
{
    const PjdDevImage &im = B.images[image];
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_du) return;
    uint32_t comp;
    const size_t base = ref_unit_base(im, first_du + u, comp);
    const int16_t *src = scratch + (size_t)u * 64;
    for (uint32_t z = 0; z < 64; z++) {      // ascending slots: a later slot overwrites an earlier one on the same position
        const int v = src[z];
        if (v == 0) continue;                // unvisited (or a zero that changes nothing: the buffer starts zeroed) ...
        out[base + ref_natural(im, z)] = (int16_t)(v == PJD_COEF_SENTINEL ? 0 : v);      // ... except the explicit zero at slot 52
    }
}

void pjd_launch_coefdump_lanes(hipStream_t s, const PjdDevBatch &b, uint32_t image, uint32_t n_iwg, int16_t *out)
{
    if (n_iwg) hipLaunchKernelGGL(pjd_k_coefdump_lanes, dim3((n_iwg + 63) / 64), dim3(64), 0, s, b, image, out);
}

void pjd_launch_coefdump_dense(hipStream_t s, const PjdDevBatch &b, uint32_t image, const int16_t *scratch, uint32_t first_du,
                               uint32_t n_du, int16_t *out)
{
    if (n_du) hipLaunchKernelGGL(pjd_k_coefdump_dense, dim3((n_du + 255) / 256), dim3(256), 0, s, b, image, scratch, first_du, n_du, out);
}
