// pjd_device_common.h -- arithmetic shared by every gfx950 kernel of the path.
//
// The integer contracts restated here are the reference's device code
// (reference src/decoder_dpu.c); comments give the lines they follow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pjd_internal.h"

// zigzag slot -> natural position, with the reference's entry 48 = 38
// (reference src/headers/common.h:9-18; the standard table has 58 there).
static __constant__ uint8_t c_zz[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    38, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// One 1-D pass of the reference's integer butterfly (reference
// src/decoder_dpu.c:219-267 rows, :271-319 columns).  32-bit ints, arithmetic
// shifts, no rounding terms; the caller truncates the outputs to int16.
//
// All operands are within 24 bits when the inputs are int16 (they always are: dequantised coefficients and
// first-pass outputs are stored as int16): |g| <= 32768*251/8, sums of two or three of those stay below 2^23.
// v_mul_i32_i24 then returns exactly the low 32 bits of the reference's 32-bit product -- at full rate instead of the
// quarter-rate v_mul_lo_u32.
// sext24(a) * sext24(k), low 32 bits, in ONE full-rate instruction.  An instruction, not __mul24: for the second butterfly
// stage (sums of first-stage terms) the compiler does not prove the 24-bit range and falls back to the quarter-rate 32-bit
// multiply -- five per 1-D pass, a quarter of the pass's issue time (round 4, seen in the ISA).
__device__ __forceinline__ int pjd_mul_i24(int a, int k)
{
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "s"(k), "v"(a));
    return r;
}

__device__ __forceinline__ void pjd_idct8(int x0, int x1, int x2, int x3, int x4, int x5, int x6, int x7, int *o)
{
    const int g0 = pjd_mul_i24(x0, 181) >> 5, g1 = pjd_mul_i24(x4, 181) >> 5;
    const int g2 = pjd_mul_i24(x2, 59) >> 3,  g3 = pjd_mul_i24(x6, 49) >> 4;
    const int g4 = pjd_mul_i24(x5, 71) >> 4,  g5 = pjd_mul_i24(x1, 251) >> 5;
    const int g6 = pjd_mul_i24(x7, 25) >> 4,  g7 = pjd_mul_i24(x3, 213) >> 5;
    const int f4 = g4 - g7, f5 = g5 + g6, f6 = g5 - g6, f7 = g4 + g7;
    const int e2 = g2 - g3, e3 = g2 + g3, e5 = f5 - f7, e7 = f5 + f7, e8 = f4 + f6;
    const int d2 = pjd_mul_i24(e2, 181) >> 7, d4 = pjd_mul_i24(f4, 277) >> 8, d5 = pjd_mul_i24(e5, 181) >> 7;
    const int d6 = pjd_mul_i24(f6, 669) >> 8, d8 = pjd_mul_i24(e8, 49) >> 6;
    const int c0 = g0 + g1, c1 = g0 - g1, c2 = d2 - e3, c4 = d4 + d8;
    const int c5 = d5 + e7, c6 = d6 - d8, c8 = c5 - c6;
    const int b0 = c0 + e3, b1 = c1 + c2, b2 = c1 - c2, b3 = c0 - e3, b4 = c4 - c8, b6 = c6 - e7;
    o[0] = (b0 + e7) >> 4; o[1] = (b1 + b6) >> 4; o[2] = (b2 + c8) >> 4; o[3] = (b3 + b4) >> 4;
    o[4] = (b3 - b4) >> 4; o[5] = (b2 - c8) >> 4; o[6] = (b1 - b6) >> 4; o[7] = (b0 - e7) >> 4;
}

__device__ __forceinline__ int pjd_clamp255(int v)
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "s"(255));      // one instruction instead of max + min
    return r;
}

// Fixed-point YCbCr -> RGB (reference src/decoder_dpu.c:376-382): 1.402, 0.344,
// 0.714, 1.772 scaled by 2^22, every product shifted on its own, +128, clamp.
__device__ __forceinline__ void pjd_ycc_to_rgb(int y, int cb, int cr, int &r, int &g, int &b)
{
    // cb, cr are int16 samples and the constants are below 2^23: __mul24 == the low 32 bits of the 32-bit product
    r = pjd_clamp255(y + (__mul24(5880414, cr) >> 22) + 128);
    g = pjd_clamp255(y - (__mul24(1442840, cb) >> 22) - (__mul24(2994733, cr) >> 22) + 128);
    b = pjd_clamp255(y + (__mul24(7432306, cb) >> 22) + 128);
}

// Full-rate 24-bit multiplies as single instructions.  Written as instructions, not as __mul24 / __umul24: where only the low
// 16 bits of a product are kept the compiler proves that the operand masks do not matter, drops them, and then has to use the
// quarter-rate 32-bit multiply (v_mul_lo_u32) on the unmasked registers (seen in the entry parser of pjd_k_idct_colour_lanes).
__device__ __forceinline__ uint32_t pjd_mul_u24(uint32_t a, uint32_t b)      // (a & 0xffffff) * (b & 0xffffff), low 32 bits
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pjd_mad_u24(uint32_t a, uint32_t b, uint32_t c)      // the same + c
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Dequantise: short *= u32 with the product truncated to int16 on store
// (reference src/decoder_dpu.c:169-172); only the low 16 bits of q matter.
// coef is an int16 value, q < 2^16: within __mul24's exact range
__device__ __forceinline__ int pjd_dequant(int coef, unsigned q) { return (int)(int16_t)__mul24(coef, (int)q); }
