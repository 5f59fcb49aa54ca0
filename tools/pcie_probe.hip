// pcie_probe.hip -- what the host link of this box can carry, measured without any of the decoder around it.
//
// The decoder's PCIe-inclusive rate (bench.py: pcie_inclusive) is bounded by the copy of the decoded pictures
// back to the host: 3 bytes per pixel go down for ~0.3-0.6 bytes per pixel that went up.  This probe times
//   * hipMemcpyAsync device -> page-locked host (what pjd_batch_download_packed issues), by size;
//   * the same with 2 and 3 copies in flight on separate streams (what 3 pipeline slots issue);
//   * host -> device;
//   * a copy KERNEL storing straight into mapped page-locked memory (no SDMA engine involved);
//   * device -> pageable host memory (what a caller without pjd_host_alloc gets);
// and prints one JSON object.  Run it once as is and once with HSA_ENABLE_SDMA=0 (copies done by blit
// kernels instead of the SDMA engines) to see which engine sets the ceiling.
//
//   hipcc --offload-arch=gfx950 -O2 -o bin/pcie_probe tools/pcie_probe.hip && bin/pcie_probe
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void copy_to_host(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(src[i], dst + i);
}

__global__ void spin_kernel(unsigned long long ticks, unsigned int *sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    unsigned int x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeef) *sink = x;
}

// `pcie_probe sim THREADS ROUNDS FLAGS`: THREADS host threads, each with its own stream and buffers, loop over
//   [memset 581 MB][H2D 110 MB][kernel busy 6 ms on every CU][D2H 581 MB], synchronising after the upload, the kernel
// and the download like a pipeline slot does.  FLAGS is a string of letters switching parts OFF: m(emset) u(pload)
// k(ernel); 'p' uploads from pageable memory in 14 small pieces as well.  Prints the aggregate D2H rate.
static int simulate(int dev, int threads, int rounds, const char *flags)
{
    const size_t out_bytes = (size_t)581 << 20, in_bytes = (size_t)110 << 20;
    const bool no_m = std::strchr(flags, 'm'), no_u = std::strchr(flags, 'u'), no_k = std::strchr(flags, 'k'), small = std::strchr(flags, 'p');
    std::vector<std::thread> th;
    std::vector<double> d2h_s((size_t)threads, 0.0), up_s((size_t)threads, 0.0);
    std::atomic<int> bad{0};
    const double t0 = now_s();
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            uint8_t *d = nullptr, *din = nullptr, *h = nullptr, *hin = nullptr;
            unsigned int *sink = nullptr;
            hipStream_t s;
            if (hipSetDevice(dev) != hipSuccess || hipMalloc(&d, out_bytes) != hipSuccess || hipMalloc(&din, in_bytes) != hipSuccess ||
                hipMalloc(&sink, 4) != hipSuccess || hipHostMalloc(&h, out_bytes, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(&hin, in_bytes, hipHostMallocDefault) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { bad++; return; }
            std::memset(h, 1, out_bytes); std::memset(hin, 1, in_bytes);
            std::vector<uint8_t> pg((size_t)2 << 20, 3);
            for (int r = 0; r < rounds; r++) {
                double a = now_s();
                if (small) for (int k = 0; k < 14; k++) (void)hipMemcpyAsync(din + k * 4096, pg.data() + (size_t)k * 100000, k == 5 ? 1700000 : 60000, hipMemcpyHostToDevice, s);
                if (!no_u) (void)hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, s);
                if (!no_m) (void)hipMemsetAsync(d, 0, out_bytes, s);
                (void)hipStreamSynchronize(s);
                up_s[t] += now_s() - a;
                if (!no_k) { spin_kernel<<<1024, 256, 0, s>>>(600000ull, sink); (void)hipStreamSynchronize(s); }
                a = now_s();
                (void)hipMemcpyAsync(h, d, out_bytes, hipMemcpyDeviceToHost, s);
                (void)hipStreamSynchronize(s);
                d2h_s[t] += now_s() - a;
            }
            (void)hipFree(d); (void)hipFree(din); (void)hipFree(sink); (void)hipHostFree(h); (void)hipHostFree(hin); (void)hipStreamDestroy(s);
        });
    for (std::thread &x : th) x.join();
    const double wall = now_s() - t0;
    double dsum = 0, usum = 0;
    for (int t = 0; t < threads; t++) { dsum += d2h_s[t]; usum += up_s[t]; }
    std::printf("{\"sim\": {\"threads\": %d, \"rounds\": %d, \"off\": \"%s\", \"ok\": %s, \"wall_ms\": %.1f, \"aggregate_d2h_GBps\": %.2f, "
                "\"mean_download_ms\": %.2f, \"mean_upload_ms\": %.2f}}\n", threads, rounds, flags, bad.load() ? "false" : "true", wall * 1e3,
                (double)out_bytes * threads * rounds / wall / 1e9, dsum / (threads * rounds) * 1e3, usum / (threads * rounds) * 1e3);
    return bad.load() ? 2 : 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "sim"))
        return simulate(0, argc > 2 ? std::atoi(argv[2]) : 3, argc > 3 ? std::atoi(argv[3]) : 6, argc > 4 ? argv[4] : "");
    const int dev = argc > 1 ? std::atoi(argv[1]) : 0;
    CHECK(hipSetDevice(dev));
    const size_t cap = (size_t)768 << 20;
    uint8_t *d = nullptr, *h = nullptr, *h2 = nullptr, *h3 = nullptr;
    CHECK(hipMalloc(&d, cap));
    CHECK(hipMemset(d, 0x5a, cap));
    CHECK(hipHostMalloc(&h, cap, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h2, cap, hipHostMallocDefault));
    CHECK(hipHostMalloc(&h3, cap, hipHostMallocDefault));
    std::memset(h, 1, cap); std::memset(h2, 1, cap); std::memset(h3, 1, cap);
    hipStream_t s[3];
    for (int k = 0; k < 3; k++) CHECK(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking));
    const char *sdma = std::getenv("HSA_ENABLE_SDMA");
    std::printf("{\"device\": %d, \"HSA_ENABLE_SDMA\": \"%s\"", dev, sdma ? sdma : "unset");

    // one copy at a time, by size
    const size_t sizes[] = {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20, (size_t)256 << 20, (size_t)640 << 20};
    std::printf(", \"d2h_pinned_GBps\": {");
    for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; k++) {
        const size_t n = sizes[k];
        const int reps = n >= ((size_t)256 << 20) ? 4 : 16;
        CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        const double t0 = now_s();
        for (int r = 0; r < reps; r++) CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        std::printf("%s\"%zuMB\": %.2f", k ? ", " : "", n >> 20, (double)n * reps / (now_s() - t0) / 1e9);
    }
    std::printf("}, \"h2d_pinned_GBps\": {");
    for (size_t k = 0; k < sizeof sizes / sizeof sizes[0]; k++) {
        const size_t n = sizes[k];
        const int reps = n >= ((size_t)256 << 20) ? 4 : 16;
        CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        const double t0 = now_s();
        for (int r = 0; r < reps; r++) CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        std::printf("%s\"%zuMB\": %.2f", k ? ", " : "", n >> 20, (double)n * reps / (now_s() - t0) / 1e9);
    }
    std::printf("}");

    // several 256 MB copies in flight on separate streams (aggregate rate)
    {
        const size_t n = (size_t)256 << 20;
        uint8_t *hs[3] = {h, h2, h3};
        std::printf(", \"d2h_concurrent_GBps\": {");
        for (int ns = 1; ns <= 3; ns++) {
            for (int k = 0; k < ns; k++) CHECK(hipMemcpyAsync(hs[k], d + (size_t)k * n, n, hipMemcpyDeviceToHost, s[k]));
            for (int k = 0; k < ns; k++) CHECK(hipStreamSynchronize(s[k]));
            const double t0 = now_s();
            for (int r = 0; r < 4; r++)
                for (int k = 0; k < ns; k++) CHECK(hipMemcpyAsync(hs[k], d + (size_t)k * n, n, hipMemcpyDeviceToHost, s[k]));
            for (int k = 0; k < ns; k++) CHECK(hipStreamSynchronize(s[k]));
            std::printf("%s\"%d\": %.2f", ns > 1 ? ", " : "", ns, (double)n * 4 * ns / (now_s() - t0) / 1e9);
        }
        std::printf("}");
        // both directions at once: 256 MB down on one stream while 64 MB go up on another
        const double t0 = now_s();
        for (int r = 0; r < 4; r++) {
            CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s[0]));
            CHECK(hipMemcpyAsync(d + 2 * n, h2, n / 4, hipMemcpyHostToDevice, s[1]));
        }
        CHECK(hipStreamSynchronize(s[0])); CHECK(hipStreamSynchronize(s[1]));
        std::printf(", \"d2h_while_h2d_GBps\": %.2f", (double)n * 4 / (now_s() - t0) / 1e9);
    }

    // a kernel storing into mapped page-locked memory
    {
        const size_t n = (size_t)256 << 20;
        u32x4 *hd = nullptr;
        CHECK(hipHostGetDevicePointer((void **)&hd, h, 0));
        std::printf(", \"kernel_store_to_host_GBps\": {");
        const int grids[] = {64, 256, 1024, 4096};
        for (size_t g = 0; g < 4; g++) {
            copy_to_host<<<grids[g], 256, 0, s[0]>>>((const u32x4 *)d, hd, n / 16);
            CHECK(hipStreamSynchronize(s[0]));
            const double t0 = now_s();
            for (int r = 0; r < 4; r++) copy_to_host<<<grids[g], 256, 0, s[0]>>>((const u32x4 *)d, hd, n / 16);
            CHECK(hipStreamSynchronize(s[0]));
            std::printf("%s\"%d_wgs\": %.2f", g ? ", " : "", grids[g], (double)n * 4 / (now_s() - t0) / 1e9);
        }
        std::printf("}");
    }

    // pageable destination
    {
        const size_t n = (size_t)256 << 20;
        std::vector<uint8_t> pg(n, 1);
        CHECK(hipMemcpy(pg.data(), d, n, hipMemcpyDeviceToHost));
        const double t0 = now_s();
        for (int r = 0; r < 2; r++) CHECK(hipMemcpy(pg.data(), d, n, hipMemcpyDeviceToHost));
        std::printf(", \"d2h_pageable_GBps\": %.2f", (double)n * 2 / (now_s() - t0) / 1e9);
    }

    // what the link says about itself
    {
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, sizeof bus, dev) == hipSuccess) std::printf(", \"pci_bus_id\": \"%s\"", bus);
    }
    std::printf("}\n");
    (void)hipFree(d); (void)hipHostFree(h); (void)hipHostFree(h2); (void)hipHostFree(h3);
    return 0;
}
