// ilp_probe.hip -- measurement tool (not product): what would two subsequences per thread buy the entropy decoder?
//
// pjd_k_huff_lanes is bound by in-order issue: one step of a lane is ONE dependent chain (window -> table index -> LDS lookup -> fields
// -> state selects -> new window position) of ~45 VALU + ~25 SALU instructions and two LDS round trips; a SIMD holds 5 such waves.
// This probe runs a loop of the same shape -- a 9-bit table lookup in LDS whose result decides how far the bit window moves, a second
// dependent lookup, ~30 dependent integer operations on the looked-up fields, the window serviced every two steps from transposed
// word rows in global memory -- for CHAINS = 1 or 2 independent chains per thread, at the occupancies either form would have
// (registers: 1 chain ~92 VGPRs = 5 waves per SIMD, 2 chains ~150 = 3), and prints steps per second per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/ilp_probe tools/ilp_probe.hip && bin/ilp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Chain { uint32_t w0, w1, w2, n0, n1, off; int r; int zb; uint32_t acc, x; };

template <int CHAINS>
__global__ __launch_bounds__(128) void probe(const uint32_t *__restrict__ words, uint32_t rows, uint32_t steps, uint32_t *out)
{
    extern __shared__ uint32_t lds[];                    // [4 tables x 512 entries][second-level 1024 x u16 as u32 pairs]
    for (uint32_t i = threadIdx.x; i < 4 * 512 + 512; i += blockDim.x) {
        // entries: used (2..15) | advance << 5 | pair used << 16 | pair advance << 21, pseudo-random but fixed
        const uint32_t h = (i * 2654435761u) >> 7;
        const uint32_t used = 2 + (h & 7) + ((h >> 3) & 3), adv = 1 + ((h >> 5) & 3);
        const uint32_t u12 = used + 2 + ((h >> 9) & 7), adv12 = adv + 1 + ((h >> 12) & 3);
        lds[i] = used | (adv << 5) | ((h >> 20 & 15) == 0 ? 0x0800u : 0u) | (u12 << 16) | (adv12 << 21);
    }
    __syncthreads();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    Chain c[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; k++) {
        const uint32_t base = ((wave * CHAINS + k) * rows) * 64 + lane;          // this chain's column of its wave's transposed rows
        c[k].off = base * 4;
        const uint32_t *q = words + base;
        c[k].w0 = q[0]; c[k].w1 = q[64]; c[k].w2 = q[128]; c[k].n0 = q[192]; c[k].n1 = q[256];
        c[k].off += 3 * 256;
        c[k].r = 32; c[k].zb = 63; c[k].acc = 0; c[k].x = (k * 2048u) | ((k * 2048u + 512u * 4u) << 16);
    }
    const uint8_t *wb = reinterpret_cast<const uint8_t *>(words);
    const uint32_t limit = rows * 256 - 8 * 256;
    for (uint32_t s = 0; s < steps; s += 2) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t pk[CHAINS], e[CHAINS];
#pragma unroll
            for (int k = 0; k < CHAINS; k++) {                                      // the lookups of all chains are issued before any is waited for
                const bool in1 = half && c[k].r < 0;
                pk[k] = __builtin_amdgcn_alignbit(in1 ? c[k].w1 : c[k].w0, in1 ? c[k].w2 : c[k].w1, (uint32_t)c[k].r);
                const uint32_t tab = c[k].zb == 63 ? (c[k].x & 0xffffu) : (c[k].x >> 16);
                e[k] = lds[(tab >> 2) + (pk[k] >> 23)];
            }
#pragma unroll
            for (int k = 0; k < CHAINS; k++) {
                uint32_t ee = e[k];
                if ((ee & 0x0800u) != 0) ee = lds[2048 + ((ee >> 5 & 3) << 7) + ((pk[k] >> 16) & 127)] | 2u;      // the second level, taken by some lane in most steps
                const uint32_t u1 = ee & 31u, u12 = (ee >> 16) & 31u;
                const int z1 = c[k].zb - (int)((ee >> 5) & 127u);
                const bool pair = z1 >= 0 && (c[k].acc & 7u) != 0;
                const uint32_t used = pair ? u12 : u1;
                c[k].r -= (int)used;
                int zb = pair ? c[k].zb - (int)((ee >> 21) & 127u) : z1;
                const bool done = zb < 0;
                // the value extraction, packing and sums of the write pass, as dependent integer work on what was looked up
                const uint32_t v1 = __builtin_amdgcn_ubfe(pk[k], 32u - u1, ee >> 12 & 15u), v2 = __builtin_amdgcn_ubfe(pk[k], 32u - used, ee >> 28);
                uint32_t a = c[k].acc;
                a = a * 3u + v1; a ^= a >> 7; a += v2 << 5; a = (a << 3) | (a >> 29); a += used; a ^= (uint32_t)zb; a += (a >> 11);
                a = __builtin_amdgcn_perm(a, v1, 0x05040100u); a += pair ? 0x10001u : 1u; a ^= a << 9; a += v2; a = (a >> 3) + (a << 5); a += u12;
                c[k].acc = a;
                c[k].zb = done ? 63 : zb;
                c[k].x = done ? ((c[k].x >> 16) | (c[k].x << 16)) : c[k].x;
            }
        }
#pragma unroll
        for (int k = 0; k < CHAINS; k++) {                                          // service: shift by the 0..2 words used up, fetch the two behind
            const bool j1 = c[k].r < 0, j2 = c[k].r < -32;
            uint32_t t0 = j1 ? c[k].w1 : c[k].w0, t1 = j1 ? c[k].w2 : c[k].w1, t2 = j1 ? c[k].n0 : c[k].w2, t3 = j1 ? c[k].n1 : c[k].n0;
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
            c[k].w0 = j2 ? t1 : t0; c[k].w1 = j2 ? t2 : t1; c[k].w2 = j2 ? t3 : t2;
            c[k].off += (j1 ? 256u : 0u) + (j2 ? 256u : 0u);
            if (c[k].off - (c[k].off % (rows * 256)) != 0 && (c[k].off % (rows * 256)) > limit) c[k].off -= limit - 8 * 256;       // wrap inside the wave's rows
            c[k].r &= 31;
            c[k].n0 = *reinterpret_cast<const uint32_t *>(wb + c[k].off);
            c[k].n1 = *reinterpret_cast<const uint32_t *>(wb + c[k].off + 256);
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; k++) r ^= c[k].acc + (uint32_t)c[k].r;
    if (r == 0x12345678u) out[0] = r;
}

template <int CHAINS>
static double run(const uint32_t *d_words, uint32_t rows, uint32_t waves_per_simd, uint32_t lds_bytes, uint32_t *d_out)
{
    const uint32_t steps = 4000;
    const uint32_t n_wg = 256 * 4 * waves_per_simd / 2;             // 2 waves per workgroup, as the decoder
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe<CHAINS>, dim3(n_wg), dim3(128), lds_bytes, 0, d_words, rows, 200u, d_out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(probe<CHAINS>, dim3(n_wg), dim3(128), lds_bytes, 0, d_words, rows, steps, d_out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double lane_steps = (double)n_wg * 128 * CHAINS * steps;
    printf("chains %d, %u waves per SIMD (%u chains), LDS %u B per workgroup: %.3f ms, %.2f G lane-steps/s, %.0f ns per step of a wave\n", CHAINS, waves_per_simd,
           waves_per_simd * CHAINS, lds_bytes, ms, lane_steps / ms / 1e6, ms * 1e6 / steps);
    return lane_steps / ms;
}

int main()
{
    const uint32_t rows = 256 + 8, max_waves = 256 * 4 * 6 * 2;
    std::vector<uint32_t> h((size_t)max_waves * rows * 64);
    uint32_t x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x ^ (x >> 13); }
    uint32_t *d_words, *d_out;
    CHECK(hipMalloc(&d_words, h.size() * 4)); CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemcpy(d_words, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // one chain per thread: the decoder as it is (16 KB of LDS per workgroup = 5 waves per SIMD), and fewer waves for comparison
    for (uint32_t w : {1u, 2u, 3u, 4u, 5u}) run<1>(d_words, rows, w, 16144, d_out);
    // two chains per thread: the tables once per workgroup, the per-wave areas twice (20.8 KB): at most 3 waves per SIMD by registers
    for (uint32_t w : {1u, 2u, 3u}) run<2>(d_words, rows, w, 16144 + 4608, d_out);
    return 0;
}
