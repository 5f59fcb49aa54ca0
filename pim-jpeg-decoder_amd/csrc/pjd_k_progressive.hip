// pjd_k_progressive.hip -- progressive (SOF2) frames: SURVEY 8(f) N4, "standards-correct extras outside the reference envelope".
//
// The reference carries the four progressive procedures -- DC first / DC refinement / AC first / AC refinement with end-of-band
// runs -- in decode_MCU_component (reference src/jpeg_scanner.cpp:521-704) but can never use them on a real file: its scanner
// keeps one scan's parameters and rejects the first marker between scans (:425-430), and decode_Huffman_data walks every scan in
// interleaved MCU order (:721-752).  This kernel is what that code would have to become: every scan of the frame in file order
// (ITU T.81 G.1), non-interleaved scans over the component's own block grid, coefficients ACCUMULATED across scans in the dense
// scratch (zigzag-slot order, absolute DC -- what the exact kernel leaves for a baseline picture), restart intervals per scan;
// the dense back end (pjd_k_idct_colour) then does what it does for any picture.  NOT reference-comparable: parity is unpinned
// (the check is that a progressive encoding of a picture decodes to the same pixels as the baseline encoding of the same
// coefficients, tests/test_gpu_parity.py::test_progressive_*).
//
// One lane per picture: scans are sequential by definition (each refines what the previous left), a picture's scans are ten
// dependent chains.  A functional path, not a fast one (about 5 MPix/s per picture).
#include "pjd_device_common.h"
#include "pjd_kernels.h"
#include "../../include/pjd.h"

namespace {

typedef const __attribute__((address_space(1))) uint8_t *pjd_gp;
struct __attribute__((packed)) PjdUnalignedU32 { uint32_t v; };

struct ScanBits {          // MSB-first bit cursor over one scan's destuffed bytes; reads past the end give zeros (the buffer is padded)
    pjd_gp base;
    uint32_t hi, lo, nxt, off;
    int s;
    uint32_t p, nbits;
    __device__ uint32_t word(uint32_t byte_off) const
    {
        return __builtin_bswap32(reinterpret_cast<const __attribute__((address_space(1))) PjdUnalignedU32 *>(base + byte_off)->v);
    }
    __device__ void init(pjd_gp b, uint32_t n_bytes)
    {
        base = b; hi = 0; lo = word(0); nxt = word(4); off = 8; s = 0; p = 0; nbits = n_bytes * 8u;
    }
    __device__ uint32_t peek() const { return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)s); }
    __device__ void drop(uint32_t n)
    {
        p += n;
        s -= (int)n;
        if (s < 0) { s += 32; hi = lo; lo = nxt; nxt = word(off); off += 4; }
    }
    __device__ uint32_t left() const { return nbits > p ? nbits - p : 0u; }
    // `n` bits (0..16) or -1 at the end of the data (reference BitReader::read_bits, src/headers/jpeg.h:102-113)
    __device__ int bits(uint32_t n)
    {
        if (n == 0) return 0;
        if (left() < n) { p = nbits; return -1; }
        const uint32_t v = peek() >> (32 - n);
        drop(n);
        return (int)v;
    }
    __device__ void align() { if ((p >> 3) < (nbits >> 3) && (p & 7)) drop(8u - (p & 7)); }
};

// reference get_next_symbol (src/jpeg_scanner.cpp:450-465): shortest code first, -1 at the end of the data / without a match
__device__ int next_symbol(ScanBits &r, const PjdDevHuffRaw *t)
{
    const uint32_t win = r.peek() >> 16, left = r.left();
    uint32_t code0 = 0;
    for (uint32_t len = 1; len <= 16; len++) {
        if (len > left) return -1;
        const uint32_t cnt = (uint32_t)t->offsets[len] - (uint32_t)t->offsets[len - 1];
        const uint32_t c = win >> (16 - len);
        if (c >= code0 && c - code0 < cnt) { r.drop(len); return t->symbols[t->offsets[len - 1] + (c - code0)]; }
        code0 = (code0 + cnt) << 1;
    }
    r.drop(16);
    return -1;
}

__device__ __forceinline__ int extend(int v, uint32_t n) { return (n != 0 && v < (1 << (n - 1))) ? v - ((1 << n) - 1) : v; }

}  // namespace

__global__ __launch_bounds__(64) void pjd_k_progressive(PjdDevBatch B, const uint32_t *__restrict__ image_list, const uint64_t *__restrict__ dense_base)
{
    __shared__ PjdDevHuffRaw tabs[3];
    const uint32_t ii = image_list[blockIdx.x];
    const PjdDevImage &im = B.images[ii];
    if (!(im.flags & PJD_IF_PROGRESSIVE)) return;
    int16_t *coef = B.coef + dense_base[blockIdx.x] * 64;
    const uint32_t dus = im.dus_per_mcu, nl = im.n_luma;
    const uint32_t w8 = (im.width + 7) / 8, h8 = (im.height + 7) / 8;                       // luma blocks
    const uint32_t cw8 = ((im.width + im.hs - 1) / im.hs + 7) / 8, ch8 = ((im.height + im.vs - 1) / im.vs + 7) / 8;   // chroma blocks
    int status = PJD_ST_OK;
    for (uint32_t si = 0; si < im.n_pscan; si++) {
        const PjdDevScan &sc = B.pscans[im.pscan_base + si];
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < sc.n_comp * sizeof(PjdDevHuffRaw); k += 64)
            reinterpret_cast<uint8_t *>(tabs)[k] = reinterpret_cast<const uint8_t *>(sc.table)[k];
        __syncthreads();
        if (threadIdx.x != 0 || status) continue;

        ScanBits r;
        r.init((pjd_gp)(B.ecs + sc.ecs_off), sc.ecs_len);
        const uint32_t ss = sc.ss, se = sc.se, al = sc.al, RI = sc.restart_interval;
        const bool dc_scan = ss == 0, refine = sc.ah != 0;
        const bool interleaved = sc.n_comp > 1;
        // MCUs of this scan: the frame's MCU grid for an interleaved scan, the component's own block grid otherwise (T.81 A.2.3)
        uint32_t gw, gh;
        if (interleaved) { gw = im.mcux; gh = im.mcuy; }
        else if (sc.comp[0] == 0) { gw = w8; gh = h8; }
        else { gw = cw8; gh = ch8; }
        int pred[3] = {0, 0, 0};
        uint32_t eobrun = 0;
        const int p1 = 1 << al, m1 = -(1 << al);
        for (uint32_t m = 0; m < gw * gh && !status; m++) {
            if (RI != 0 && m != 0 && m % RI == 0) { pred[0] = pred[1] = pred[2] = 0; eobrun = 0; r.align(); }
            const uint32_t mx = m % gw, my = m / gw;
            for (uint32_t q = 0; q < sc.n_comp && !status; q++) {
                const uint32_t c = sc.comp[q];
                const uint32_t hc = (interleaved && c == 0) ? im.hs : 1u, vc = (interleaved && c == 0) ? im.vs : 1u;
                for (uint32_t v = 0; v < vc && !status; v++)
                    for (uint32_t h = 0; h < hc && !status; h++) {
                        // block (bx, by) of component c -> data unit D of the picture (interleaved order, as the baseline decoders count)
                        uint32_t bx, by;
                        if (interleaved) { bx = mx * hc + h; by = my * vc + v; } else { bx = mx; by = my; }
                        uint32_t mcu, k;
                        if (c == 0) { mcu = (by / im.vs) * im.mcux + (bx / im.hs); k = (by % im.vs) * im.hs + (bx % im.hs); }
                        else { mcu = by * im.mcux + bx; k = nl + c - 1; }
                        if (mcu >= im.n_mcu) continue;                      // (cannot happen: the grids cover the same picture)
                        int16_t *unit = coef + ((size_t)mcu * dus + k) * 64;
                        const PjdDevHuffRaw *t = &tabs[q];
                        if (dc_scan && !refine) {                           // DC first (jpeg_scanner.cpp:522-544)
                            const int s = next_symbol(r, t);
                            if (s < 0 || s == 0xFF) { status = PJD_ST_DC_SYM; break; }
                            if (s > 11) { status = PJD_ST_DC_LEN; break; }
                            int v2 = r.bits((uint32_t)s);
                            if (v2 == -1) { status = PJD_ST_DC_BITS; break; }
                            v2 = extend(v2, (uint32_t)s) + pred[c];
                            pred[c] = v2;
                            unit[0] = (int16_t)(v2 << al);
                        } else if (dc_scan) {                               // DC refinement (:545-552)
                            const int bit = r.bits(1);
                            if (bit == -1) { status = PJD_ST_DC_BITS; break; }
                            if (bit) unit[0] = (int16_t)(unit[0] | p1);
                        } else if (!refine) {                               // AC first (:553-607)
                            if (eobrun > 0) { eobrun--; continue; }
                            for (uint32_t z = ss; z <= se; z++) {
                                const int sym = next_symbol(r, t);
                                if (sym < 0 || sym == 0xFF) { status = PJD_ST_AC_SYM; break; }
                                const uint32_t run = (uint32_t)sym >> 4, len = (uint32_t)sym & 15;
                                if (len != 0) {
                                    if (z + run > se) { status = PJD_ST_AC_RUN; break; }
                                    z += run;
                                    if (len > 10) { status = PJD_ST_AC_LEN; break; }
                                    const int v2 = r.bits(len);
                                    if (v2 == -1) { status = PJD_ST_AC_BITS; break; }
                                    unit[z] = (int16_t)(extend(v2, len) << al);
                                } else if (run == 15) {
                                    if (z + 15 > se) { status = PJD_ST_AC_RUN; break; }
                                    z += 15;
                                } else {                                    // end of band for 2^run + extra blocks, this one included
                                    eobrun = (1u << run) - 1;
                                    const int x = r.bits(run);
                                    if (x == -1) { status = PJD_ST_AC_BITS; break; }
                                    eobrun += (uint32_t)x;
                                    break;
                                }
                            }
                        } else {                                            // AC refinement (:608-702; ITU T.81 G.1.2.3)
                            uint32_t z = ss;
                            if (eobrun == 0) {
                                for (; z <= se && !status; z++) {
                                    const int sym = next_symbol(r, t);
                                    if (sym < 0 || sym == 0xFF) { status = PJD_ST_AC_SYM; break; }
                                    int run = sym >> 4;
                                    const uint32_t len = (uint32_t)sym & 15;
                                    int newc = 0;
                                    if (len != 0) {
                                        if (len != 1) { status = PJD_ST_AC_SYM; break; }
                                        const int bit = r.bits(1);
                                        if (bit == -1) { status = PJD_ST_AC_BITS; break; }
                                        newc = bit ? p1 : m1;
                                    } else if (run != 15) {
                                        eobrun = 1u << run;
                                        const int x = r.bits((uint32_t)run);
                                        if (x == -1) { status = PJD_ST_AC_BITS; break; }
                                        eobrun += (uint32_t)x;
                                        break;                              // the rest of the band: correction bits only (below)
                                    }
                                    // skip `run` zero-history coefficients, correcting the non-zero ones passed on the way
                                    do {
                                        const int cur = unit[z];
                                        if (cur != 0) {
                                            const int bit = r.bits(1);
                                            if (bit == -1) { status = PJD_ST_AC_BITS; break; }
                                            if (bit && (cur & p1) == 0) unit[z] = (int16_t)(cur >= 0 ? cur + p1 : cur + m1);
                                        } else {
                                            if (run == 0) break;
                                            run--;
                                        }
                                        z++;
                                    } while (z <= se);
                                    if (status) break;
                                    if (newc != 0 && z <= se) unit[z] = (int16_t)newc;
                                }
                            }
                            if (!status && eobrun > 0) {
                                for (; z <= se; z++) {
                                    const int cur = unit[z];
                                    if (cur != 0) {
                                        const int bit = r.bits(1);
                                        if (bit == -1) { status = PJD_ST_AC_BITS; break; }
                                        if (bit && (cur & p1) == 0) unit[z] = (int16_t)(cur >= 0 ? cur + p1 : cur + m1);
                                    }
                                }
                                eobrun--;
                            }
                        }
                    }
            }
        }
    }
    // keep the "decoded by the exact path" marker so the back end treats slot 0 as absolute
    if (threadIdx.x == 0) B.status[ii] = (B.status[ii] & PJD_STW_NEEDS_EXACT) | status;
}

void pjd_launch_progressive(hipStream_t s, const PjdDevBatch &b, const uint32_t *image_list, const uint64_t *dense_base, uint32_t n)
{
    if (n == 0) return;
    hipLaunchKernelGGL(pjd_k_progressive, dim3(n), dim3(64), 0, s, b, image_list, dense_base);
}
