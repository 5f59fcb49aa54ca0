// pjd_k_huffman_seq.hip -- the EXACT entropy decoder: one lane per image.
//
// A literal GPU restatement of the reference's host-side Huffman stage
//     BitReader              reference src/headers/jpeg.h:81-122
//     get_next_symbol        reference src/jpeg_scanner.cpp:450-465
//     decode_MCU_component   reference src/jpeg_scanner.cpp:467-520   (baseline branch)
//     decode_Huffman_data    reference src/jpeg_scanner.cpp:707-756
// including its end-of-stream behaviour, its restart rule, its error classes and the
// "stop at the first error, keep what was decoded" outcome.  It is the path for
//   * images the parallel decoder cannot reproduce bit-exactly (restart intervals with
//     subsampled luma under the reference's restart rule, missing segment offsets),
//   * images on which the parallel decoder met anything irregular (PJD_STW_NEEDS_EXACT).
// It is slow by construction (one dependent chain per image) and is never the fast path.
//
// Output: coefficients in zigzag-SLOT order (slot k of the D-th data unit decoded at coef[(dense_base+D)*64+k]),
// absolute DC values in slot 0, and the PJD_COEF_SENTINEL mark for an explicit zero at slot 52.
#include "pjd_device_common.h"
#include "pjd_kernels.h"
#include "../../include/pjd.h"

namespace {

struct SeqReader {
    const uint8_t *d;
    uint32_t nbits;      // total bits in the stream
    uint32_t p;          // next bit
};

// up to 32 bits starting at bit p, zero-filled past the end of the stream
__device__ uint32_t seq_peek32(const SeqReader &r)
{
    const uint32_t byte = r.p >> 3, sh = r.p & 7;
    const uint32_t nbytes = (r.nbits + 7) >> 3;
    uint64_t w = 0;
    for (uint32_t k = 0; k < 5; k++) {
        const uint32_t b = byte + k;
        w = (w << 8) | (b < nbytes ? r.d[b] : 0u);
    }
    return (uint32_t)((w << sh) >> 8);
}

// reference get_next_symbol: shortest code first, 0xFF on end of data / no match in 16 bits
__device__ int seq_symbol(SeqReader &r, const PjdDevHuffRaw &t)
{
    const uint32_t win = seq_peek32(r) >> 16;
    const uint32_t left = r.nbits - r.p;
    uint32_t code0 = 0;                                  // reference generate_codes: first code of this length
    for (uint32_t len = 1; len <= 16; len++) {
        if (len > left) return -1;                       // read_bit() hit the end
        const uint32_t cnt = (uint32_t)t.offsets[len] - (uint32_t)t.offsets[len - 1];
        const uint32_t c = win >> (16 - len);
        if (c - code0 < cnt && c >= code0) { r.p += len; return t.symbols[t.offsets[len - 1] + (c - code0)]; }
        code0 = (code0 + cnt) << 1;
    }
    r.p += 16;
    return -1;
}

// reference BitReader::read_bits: -1 if the stream ends inside the field
__device__ int seq_bits(SeqReader &r, uint32_t n)
{
    if (n == 0) return 0;
    if (r.nbits - r.p < n) { r.p = r.nbits; return -1; }
    const uint32_t v = seq_peek32(r) >> (32 - n);
    r.p += n;
    return (int)v;
}

}  // namespace

__global__ __launch_bounds__(64) void pjd_k_huff_sequential(PjdDevBatch B, const uint32_t *__restrict__ image_list, const uint64_t *__restrict__ dense_base)
{
    if (threadIdx.x != 0) return;
    const uint32_t ii = image_list[blockIdx.x];
    const PjdDevImage &im = B.images[ii];
    if (im.flags & PJD_IF_PROGRESSIVE) return;                   // pjd_k_progressive decodes it
    if (B.tsets[im.tset].lut_bytes != 0) return;                // pjd_k_huff_exact_lut (pjd_k_huffman.hip) decodes it with the decode tables
    const PjdDevHuffRaw *tabs = B.raw_tables + (size_t)im.tset * PJD_MAX_TABLES;
    SeqReader r = { B.ecs + im.ecs_off, im.ecs_len * 8u, 0u };
    // slot 0 of the scratch = the first data unit this image (or this shard of it) decodes
    int16_t *coef = B.coef + dense_base[blockIdx.x] * 64;
    const uint32_t RI = im.restart_interval, Wr = im.ref_mcu_w_real;
    const bool std_rule = (im.flags & PJD_IF_STANDARD_RESTART) != 0;
    int pred[3] = {0, 0, 0};
    uint32_t D = 0;
    int status = PJD_ST_OK;

    // reference loop `for y < mcu_height step V, for x < mcu_width step H` (jpeg_scanner.cpp:721-722) by MCU
    // counter; a shard starts at a restart point, where the reference's state is (zero predictors, byte boundary)
    for (uint32_t m = im.first_mcu; m < im.last_mcu && !status; m++) {
        const uint32_t y = (m / im.mcux) * im.vs, x = (m % im.mcux) * im.hs;
        const bool restart = RI != 0 && (std_rule ? (m % RI == 0) : ((y * Wr + x) % RI == 0));
        if (restart) {
            pred[0] = pred[1] = pred[2] = 0;
            // BitReader::align(): no-op once every byte is consumed
            if ((r.p >> 3) < im.ecs_len && (r.p & 7)) r.p = (r.p + 7) & ~7u;
        }
        for (uint32_t k = 0; k < im.dus_per_mcu && !status; k++, D++) {
            const uint32_t comp = k < im.n_luma ? 0 : k - im.n_luma + 1;
            const PjdDevHuffRaw &dt = tabs[im.tbl_slot[comp][0]];
            const PjdDevHuffRaw &at = tabs[im.tbl_slot[comp][1]];
            int16_t *unit = coef + (size_t)D * 64;
            // ---- DC (jpeg_scanner.cpp:469-486)
            int s = seq_symbol(r, dt);
            if (s < 0 || s == 0xFF) { status = PJD_ST_DC_SYM; break; }
            if (s > 11) { status = PJD_ST_DC_LEN; break; }
            int v = seq_bits(r, (uint32_t)s);
            if (v == -1) { status = PJD_ST_DC_BITS; break; }
            if (s != 0 && v < (1 << (s - 1))) v -= (1 << s) - 1;
            unit[0] = (int16_t)(v + pred[comp]);
            pred[comp] = unit[0];
            // ---- AC (jpeg_scanner.cpp:488-518)
            for (uint32_t z = 1; z < 64; z++) {
                int sym = seq_symbol(r, at);
                if (sym < 0 || sym == 0xFF) { status = PJD_ST_AC_SYM; break; }
                if (sym == 0) break;
                const uint32_t run = (uint32_t)sym >> 4, len = (uint32_t)sym & 15;
                if (z + run >= 64) { status = PJD_ST_AC_RUN; break; }
                z += run;
                if (len > 10) { status = PJD_ST_AC_LEN; break; }
                v = seq_bits(r, len);
                if (v == -1) { status = PJD_ST_AC_BITS; break; }
                if (len != 0 && v < (1 << (len - 1))) v -= (1 << len) - 1;
                // len == 0 stores a literal 0 (jpeg_scanner.cpp:516-517); it matters only at
                // slot 52, whose natural position (38) may already hold slot 48's value
                unit[z] = (len == 0 && z == 52) ? (int16_t)PJD_COEF_SENTINEL : (int16_t)v;
            }
        }
    }
    // keep the "decoded by the exact kernel" marker so the back end treats slot 0 as absolute
    B.status[ii] = (B.status[ii] & PJD_STW_NEEDS_EXACT) | status;
}

void pjd_launch_huff_sequential(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n)
{
    if (n == 0) return;
    pjd_launch_huff_exact_lut(s, b, image_list, dense_base, n);         // pictures whose table set has decode tables; the rest below
    hipLaunchKernelGGL(pjd_k_huff_sequential, dim3(n), dim3(64), 0, s, b, image_list, dense_base);
}
