// graph_memset_probe.hip -- measurement tool, not product (make -C pim-jpeg-decoder_amd tools -> bin/graph_memset_probe).
//
// Round 2 saw, once, a 128-byte statistics buffer hold a repeating 16-byte pattern of non-zero words after a hipGraph REPLAY
// whose first node was hipMemsetAsync(stats, 0, 128): in the multi-rank rehearsal of bench.py (torch + gloo in the process,
// several ranks on one GPU, one rank holding an EMPTY batch).  The library has since reset its per-decode state with a kernel
// of its own (pjd_k_reset).  This program tries to show the runtime doing that by itself, with nothing of ours in the process:
//
//   A  a captured graph = { memset(stats, 0, 128 B), memset(small, 0, 16 B), a kernel that reads stats }, replayed after and
//      during unrelated runtime work: allocations and frees, memsets with other patterns on other streams, pageable copies
//   B  the same with ZERO-LENGTH memset / copy calls captured next to it (what an empty batch issued in round 2)
//   C  the same while a second host thread issues fills with a 16-byte-periodic pattern on its own stream all the time
//
// After every replay the 128 bytes are read back; anything non-zero is printed with the replay number.  Exit code 0 either way
// (a finding is a finding); the last line says how many replays were dirty.  Run ONCE (profiles/r03_graph_memset_probe.log).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ void dirty(unsigned long long *p, int n, unsigned long long v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void touch(const unsigned long long *stats, unsigned long long *sink) { if (threadIdx.x == 0) sink[0] += stats[threadIdx.x & 15]; }

static int run_case(const char *name, bool zero_len_nodes, bool hammer, int replays)
{
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned long long *stats, *small_buf, *sink, *other;
    CK(hipMalloc((void **)&stats, 128)); CK(hipMalloc((void **)&small_buf, 16)); CK(hipMalloc((void **)&sink, 64)); CK(hipMalloc((void **)&other, 1 << 20));
    CK(hipMemset(sink, 0, 64));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(stats, 0, 128, s));
    CK(hipMemsetAsync(small_buf, 0, 16, s));
    if (zero_len_nodes) {
        (void)hipMemsetAsync(other, 0, 0, s);                                          // what an empty batch's work lists amounted to
        (void)hipMemcpyAsync(other, other + 64, 0, hipMemcpyDeviceToDevice, s);
        (void)hipGetLastError();
    }
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, stats, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));

    std::atomic<bool> stop{false};
    std::thread th;
    if (hammer)
        th = std::thread([&] {
            hipStream_t hs; CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
            unsigned int *buf; CK(hipMalloc((void **)&buf, 1 << 16));
            unsigned k = 0;
            while (!stop.load()) {
                CK(hipMemsetD32Async((hipDeviceptr_t)buf, 0x7f001000u + (k++ & 0xff), 1 << 14, hs));     // pointer-looking words
                CK(hipMemsetAsync(buf, 0xab, 128, hs));
                if ((k & 15) == 0) CK(hipStreamSynchronize(hs));
            }
            CK(hipStreamSynchronize(hs)); CK(hipFree(buf)); CK(hipStreamDestroy(hs));
        });

    int dirty_replays = 0;
    std::vector<char> pageable(1 << 16, 7);
    unsigned long long host[16];
    for (int r = 0; r < replays; r++) {
        // unrelated runtime work between replays
        void *tmp[4];
        for (int k = 0; k < 4; k++) CK(hipMalloc(&tmp[k], (size_t)4096 << ((r + k) % 9)));
        CK(hipMemsetD32Async((hipDeviceptr_t)other, 0x7f00beefu, 4096, s2));
        CK(hipMemcpyAsync(other + 8192, pageable.data(), pageable.size(), hipMemcpyHostToDevice, s2));
        for (int k = 0; k < 4; k++) CK(hipFree(tmp[k]));
        // make the target dirty, replay, read back
        hipLaunchKernelGGL(dirty, dim3(1), dim3(64), 0, s, stats, 16, 0x00007f1234560000ull);
        hipLaunchKernelGGL(dirty, dim3(1), dim3(64), 0, s, small_buf, 2, 0x00007f1234560000ull);
        CK(hipGraphLaunch(ge, s));
        CK(hipMemcpyAsync(host, stats, 128, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        bool bad = false;
        for (int i = 0; i < 16; i++) bad = bad || host[i] != 0;
        if (bad) {
            if (dirty_replays < 5) {
                std::printf("[%s] replay %d: stats not zero:", name, r);
                for (int i = 0; i < 16; i++) std::printf(" %016llx", host[i]);
                std::printf("\n");
            }
            dirty_replays++;
        }
        if ((r & 3) == 0) CK(hipStreamSynchronize(s2));
    }
    stop.store(true);
    if (th.joinable()) th.join();
    CK(hipStreamSynchronize(s2));
    std::printf("[%s] %d replays, %d left non-zero bytes in the 128-byte target\n", name, replays, dirty_replays);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipFree(stats)); CK(hipFree(small_buf)); CK(hipFree(sink)); CK(hipFree(other));
    CK(hipStreamDestroy(s)); CK(hipStreamDestroy(s2));
    return dirty_replays;
}

int main(int argc, char **argv)
{
    const int replays = argc > 1 ? std::atoi(argv[1]) : 2000;
    int rt = 0;
    CK(hipRuntimeGetVersion(&rt));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    std::printf("graph_memset_probe: %s, HIP runtime %d, %d replays per case\n", prop.gcnArchName, rt, replays);
    int total = 0;
    total += run_case("A plain", false, false, replays);
    total += run_case("B zero-length nodes", true, false, replays);
    total += run_case("C concurrent fills", false, true, replays);
    total += run_case("D zero-length + concurrent fills", true, true, replays);
    std::printf("graph_memset_probe: %d dirty replays in total -> %s\n", total,
                total ? "the runtime's memset node did not zero its target" : "not reproduced outside the library");
    return 0;
}
