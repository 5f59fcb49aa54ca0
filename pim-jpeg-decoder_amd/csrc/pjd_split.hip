// pjd_split.hip -- ONE picture decoded by several GPUs of the node (BASELINE config 5), behind the C ABI.
//
// Replaces, for a picture too large for one device's share, what the reference does with every picture: spread its 8x8
// positions over all the DPUs it allocated (reference src/decoder_host.cpp:125-149 sizes the share, :225 takes every DPU,
// :262-312 copies metadata and coefficients to each DPU and the samples back).  Here restart segments are the unit of work:
// they are independent (DC predictors reset, bit cursor byte-aligned: reference src/jpeg_scanner.cpp:723-729), so
//   * rank r (one device, one host thread, one pjd_ctx) takes a contiguous range of segments (pjd_split_plan),
//   * the one thing every rank needs -- the descriptor: geometry, quantisation and Huffman tables, segment offsets, ~20 KB --
//     is broadcast from rank 0's HBM to every device with ONE ncclBroadcast (RCCL over xGMI; the library is loaded on first
//     use), then read back by each rank, which builds its shard descriptor from what it RECEIVED,
//   * each rank uploads only its own slice of the entropy-coded bytes, decodes with the ordinary batch path and returns only
//     the picture rows its MCUs cover; rows are assembled on the host.  No other exchange: the path has none.
// A picture without restart intervals cannot be split (one dependent chain): it is decoded by the first device alone.
// A picture with an entropy-coding error is decoded again by one device, so that status and partial picture are the
// reference's (it stops at the first error, src/decoder_host.cpp:181) whatever the split was.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pjd.h"

namespace {

double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

// ---- RCCL, loaded on first use (a decoder that never splits a picture never maps the library) ---------------------------
typedef void *rccl_comm;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(rccl_comm *, int, const int *) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
const int kNcclUint8 = 1;          // ncclDataType_t::ncclUint8 (rccl.h)

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast;
    });
    return r;
}

// communicators of the last device list used (creating them costs about a second); pjd_split_release() destroys them
std::mutex g_comm_m;
std::vector<int> g_comm_devs;
std::vector<rccl_comm> g_comms;

void drop_comms_locked()
{
    for (rccl_comm c : g_comms) if (c) rccl().CommDestroy(c);
    g_comms.clear();
    g_comm_devs.clear();
}

struct Geometry {
    uint32_t mcu_w, mcu_h, mcux, mcuy, n_mcu, stride;
    bool bmp;
    uint32_t W, H;
    // bytes [lo, hi) of picture rows [y0, y1) in the output image
    void row_bytes(uint32_t y0, uint32_t y1, uint64_t &lo, uint64_t &hi) const
    {
        if (bmp) { lo = 26 + (uint64_t)(H - y1) * stride; hi = 26 + (uint64_t)(H - y0) * stride; }
        else { lo = (uint64_t)y0 * stride; hi = (uint64_t)y1 * stride; }
    }
    uint64_t row_off(uint32_t y) const { return bmp ? 26 + (uint64_t)(H - 1 - y) * stride : (uint64_t)y * stride; }
};

Geometry geometry_of(const pjd_image_desc &d, int out_format)
{
    Geometry g;
    g.W = d.width; g.H = d.height;
    g.mcu_w = 8u * d.h_samp; g.mcu_h = 8u * d.v_samp;
    const uint32_t w8 = (d.width + 7) / 8, h8 = (d.height + 7) / 8;
    g.mcux = (w8 + d.h_samp - 1) / d.h_samp;
    g.mcuy = (h8 + d.v_samp - 1) / d.v_samp;
    g.n_mcu = g.mcux * g.mcuy;
    g.bmp = out_format == PJD_OUT_BMP;
    g.stride = g.bmp ? d.width * 3 + d.width % 4 : d.width * 3;
    return g;
}

bool splittable(const pjd_image_desc &d)
{
    if (d.restart_interval == 0 || !d.seg_offsets || d.n_segments < 2) return false;
    if (d.flags & PJD_F_FORCE_SEQUENTIAL) return false;
    // the reference's restart rule equals the MCU counter only for 1x1 luma (jpeg_scanner.cpp:723); otherwise the picture goes
    // to the exact kernel as a whole unless the caller opted into the standard rule
    if ((d.h_samp != 1 || d.v_samp != 1) && !(d.flags & PJD_F_STANDARD_RESTART)) return false;
    return true;
}

struct RankOut {
    int rc = PJD_OK;
    int32_t status = 0;
    std::string err;
    double upload_s = 0, exec_s = 0, download_s = 0;
    uint64_t ecs_bytes = 0;
    int n_fallback = 0, n_sequential = 0;
    bool has_work = false;
};

}  // namespace

extern "C" {

int pjd_split_plan(const pjd_image_desc *desc, int world, int rank, pjd_image_desc *shard, uint64_t *seg_scratch,
                   uint64_t *byte_lo, uint64_t *byte_hi, uint32_t *first_mcu, uint32_t *last_mcu)
{
    if (!desc || world <= 0 || rank < 0 || rank >= world || !desc->seg_offsets || desc->n_segments == 0 || desc->restart_interval == 0) return PJD_E_ARG;
    const uint32_t n = desc->n_segments;
    const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
    const uint32_t first = (uint32_t)rank * base + std::min((uint32_t)rank, rem), count = base + ((uint32_t)rank < rem ? 1u : 0u);
    if (count == 0) return 1;                                   // more ranks than restart segments: nothing to do here
    const uint64_t lo = desc->seg_offsets[first];
    const uint64_t hi = first + count < n ? desc->seg_offsets[first + count] : desc->ecs_len;
    if (hi < lo || hi > desc->ecs_len) return PJD_E_ARG;
    if (byte_lo) *byte_lo = lo;
    if (byte_hi) *byte_hi = hi;
    const Geometry g = geometry_of(*desc, PJD_OUT_RGB8);
    const uint64_t m0 = (uint64_t)first * desc->restart_interval, m1 = (uint64_t)(first + count) * desc->restart_interval;
    if (first_mcu) *first_mcu = (uint32_t)std::min<uint64_t>(m0, g.n_mcu);
    if (last_mcu) *last_mcu = (uint32_t)std::min<uint64_t>(m1, g.n_mcu);
    if (shard && seg_scratch) {
        // the C ABI takes segment offsets relative to the ecs pointer it is given: this rank holds only its slice, so its own
        // segments are rebased and the others collapse onto the slice's ends
        *shard = *desc;
        for (uint32_t k = 0; k < n; k++)
            seg_scratch[k] = k < first ? 0 : (k < first + count ? desc->seg_offsets[k] - lo : hi - lo);
        shard->ecs = desc->ecs ? desc->ecs + lo : nullptr;
        shard->ecs_len = hi - lo;
        shard->seg_offsets = seg_scratch;
        shard->shard_first_seg = first;
        shard->shard_n_segs = count;
    }
    return PJD_OK;
}

int pjd_split_rccl_selftest(int device_ordinal, uint64_t bytes)
{
    if (!rccl().ok) return PJD_E_STATE;          // librccl cannot be loaded
    if (bytes == 0) bytes = 20480;
    pjd_ctx *ctx = nullptr;
    int rc = pjd_open(device_ordinal, &ctx);
    if (rc != PJD_OK) return rc;
    hipStream_t s = (hipStream_t)pjd_stream(ctx);
    uint8_t *src = nullptr, *dst = nullptr;
    std::vector<uint8_t> pat(bytes), back(bytes, 0);
    for (uint64_t i = 0; i < bytes; i++) pat[i] = (uint8_t)(i * 131u + 7u);
    rccl_comm comm = nullptr;
    const int dev = device_ordinal;
    bool ok = hipSetDevice(dev) == hipSuccess && hipMalloc((void **)&src, bytes) == hipSuccess && hipMalloc((void **)&dst, bytes) == hipSuccess &&
              hipMemcpyAsync(src, pat.data(), bytes, hipMemcpyHostToDevice, s) == hipSuccess && hipMemsetAsync(dst, 0, bytes, s) == hipSuccess;
    if (ok) ok = rccl().CommInitAll(&comm, 1, &dev) == 0;
    if (ok) {
        int nr = rccl().GroupStart();
        if (nr == 0) nr = rccl().Broadcast(src, dst, bytes, kNcclUint8, 0, comm, s);
        const int ne = rccl().GroupEnd();
        ok = nr == 0 && ne == 0;
    }
    if (ok) ok = hipMemcpyAsync(back.data(), dst, bytes, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && back == pat;
    if (comm) rccl().CommDestroy(comm);
    if (src) hipFree(src);
    if (dst) hipFree(dst);
    pjd_close(ctx);
    return ok ? PJD_OK : PJD_E_HIP;
}

void pjd_split_release(void)
{
    std::lock_guard<std::mutex> l(g_comm_m);
    drop_comms_locked();
}

int pjd_split_decode(const pjd_image_desc *desc, const int32_t *devices, int n_devices, int out_format, uint8_t *out, uint64_t capacity,
                     int32_t *status, pjd_split_stats *stats_out)
{
    pjd_split_stats st;
    std::memset(&st, 0, sizeof st);
    if (stats_out) *stats_out = st;
    if (!desc || !devices || n_devices <= 0 || n_devices > PJD_SPLIT_MAX_DEVICES || !out) return PJD_E_ARG;
    if (out_format != PJD_OUT_RGB8 && out_format != PJD_OUT_BMP) return PJD_E_ARG;
    const uint64_t out_bytes = pjd_output_size(desc->width, desc->height, out_format);
    if (capacity < out_bytes) return PJD_E_ARG;
    const bool dup_ok = std::getenv("PJD_PIPE_ALLOW_DUP_DEVICES") != nullptr;     // tests on a one-GPU box: one ordinal, several ranks
    bool distinct = true;
    for (int a = 0; a < n_devices; a++) {
        if (devices[a] < 0) return PJD_E_ARG;
        for (int b = 0; b < a; b++) if (devices[a] == devices[b]) distinct = false;
    }
    if (!distinct && !dup_ok) return PJD_E_ARG;
    const double t_all = now_s();

    // ---- one device decodes the whole picture (not splittable, or one rank) ---------------------------------------------------
    auto whole = [&](int device, const pjd_image_desc &d) -> int {
        pjd_ctx *ctx = nullptr;
        int rc = pjd_open(device, &ctx);
        if (rc != PJD_OK) return rc;
        uint8_t *outs[1] = {out};
        int32_t s1 = 0;
        rc = pjd_decode_batch(ctx, &d, 1, out_format, outs, &s1);
        pjd_close(ctx);
        if (status) *status = s1;
        return rc;
    };
    int world = n_devices;
    if (!splittable(*desc)) world = 1;
    if (world > 1 && (uint32_t)world > desc->n_segments) world = (int)desc->n_segments;
    st.n_segments = desc->n_segments;
    if (world == 1) {
        const int rc = whole(devices[0], *desc);
        st.n_ranks = 1; st.wall_s = now_s() - t_all;
        if (stats_out) *stats_out = st;
        return rc;
    }

    // ---- contexts: one per rank ------------------------------------------------------------------------------------------------
    std::vector<pjd_ctx *> ctx((size_t)world, nullptr);
    auto close_all = [&] { for (pjd_ctx *c : ctx) if (c) pjd_close(c); };
    for (int r = 0; r < world; r++) {
        const int rc = pjd_open(devices[r], &ctx[(size_t)r]);
        if (rc != PJD_OK) { close_all(); return rc; }
    }

    // ---- the descriptor blob: the struct with its pointers cleared + the segment offsets -----------------------------------
    const size_t blob_bytes = sizeof(pjd_image_desc) + (size_t)desc->n_segments * sizeof(uint64_t);
    std::vector<uint8_t> blob(blob_bytes);
    {
        pjd_image_desc h = *desc;
        h.ecs = nullptr; h.seg_offsets = nullptr;
        std::memcpy(blob.data(), &h, sizeof h);
        std::memcpy(blob.data() + sizeof h, desc->seg_offsets, (size_t)desc->n_segments * sizeof(uint64_t));
    }
    st.blob_bytes = blob_bytes;
    const double t_bc = now_s();
    std::vector<uint8_t *> d_blob((size_t)world, nullptr);
    std::vector<std::vector<uint8_t>> got((size_t)world, std::vector<uint8_t>(blob_bytes));
    auto free_blobs = [&] { for (int r = 0; r < world; r++) if (d_blob[(size_t)r]) { hipSetDevice(devices[r]); hipFree(d_blob[(size_t)r]); } };
    hipError_t he = hipSuccess;
    for (int r = 0; r < world && he == hipSuccess; r++) {
        he = hipSetDevice(devices[r]);
        if (he == hipSuccess) he = hipMalloc((void **)&d_blob[(size_t)r], blob_bytes);
    }
    if (he == hipSuccess) { hipSetDevice(devices[0]); he = hipMemcpyAsync(d_blob[0], blob.data(), blob_bytes, hipMemcpyHostToDevice, (hipStream_t)pjd_stream(ctx[0])); }
    bool by_rccl = false;
    if (he == hipSuccess && distinct && rccl().ok && !std::getenv("PJD_SPLIT_NO_RCCL")) {
        // One collective at a time per process: the communicators are cached per device list, and creating them (ncclCommInitAll, about
        // a second) happens under this lock -- concurrent pjd_split_decode callers are serialised through this leg.
        std::lock_guard<std::mutex> l(g_comm_m);
        std::vector<int> devs(devices, devices + world);
        if (g_comm_devs != devs) {
            drop_comms_locked();
            g_comms.assign((size_t)world, nullptr);
            if (rccl().CommInitAll(g_comms.data(), world, devs.data()) == 0) g_comm_devs = devs;
            else { g_comms.clear(); }
        }
        if (!g_comms.empty()) {
            const bool open = rccl().GroupStart() == 0;            // a group that did not open is not closed either
            int nr = open ? 0 : -1;
            for (int r = 0; r < world && nr == 0; r++) {
                if (hipSetDevice(devices[r]) != hipSuccess) { nr = -1; break; }
                nr = rccl().Broadcast(d_blob[0], d_blob[(size_t)r], blob_bytes, kNcclUint8, 0, g_comms[(size_t)r], (hipStream_t)pjd_stream(ctx[(size_t)r]));
            }
            const int ne = open ? rccl().GroupEnd() : -1;
            by_rccl = nr == 0 && ne == 0;
            if (!by_rccl) {
                // Part of the group may have reached the streams: drain them and drop the communicators BEFORE the plain copies below
                // reuse the same streams and buffers.
                for (int r = 0; r < world; r++)
                    if (hipSetDevice(devices[r]) == hipSuccess) (void)hipStreamSynchronize((hipStream_t)pjd_stream(ctx[(size_t)r]));
                (void)hipGetLastError();
                drop_comms_locked();
                if (hipSetDevice(devices[0]) == hipSuccess)         // rank 0's copy of the blob may not have survived an aborted group
                    he = hipMemcpyAsync(d_blob[0], blob.data(), blob_bytes, hipMemcpyHostToDevice, (hipStream_t)pjd_stream(ctx[0]));
            }
        }
    }
    if (he == hipSuccess && !by_rccl) {
        // no collective available (RCCL missing, or ranks that share a device in a test): the same bytes by plain copies
        for (int r = 1; r < world && he == hipSuccess; r++) {
            he = hipSetDevice(devices[r]);
            if (he == hipSuccess) he = hipMemcpyAsync(d_blob[(size_t)r], blob.data(), blob_bytes, hipMemcpyHostToDevice, (hipStream_t)pjd_stream(ctx[(size_t)r]));
        }
    }
    for (int r = 0; r < world && he == hipSuccess; r++) {           // every rank reads back what IT received
        he = hipSetDevice(devices[r]);
        if (he == hipSuccess) he = hipMemcpyAsync(got[(size_t)r].data(), d_blob[(size_t)r], blob_bytes, hipMemcpyDeviceToHost, (hipStream_t)pjd_stream(ctx[(size_t)r]));
        if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)pjd_stream(ctx[(size_t)r]));
    }
    free_blobs();
    if (he != hipSuccess) { close_all(); return PJD_E_HIP; }
    st.rccl_used = by_rccl ? 1 : 0;
    st.broadcast_s = now_s() - t_bc;

    // ---- every rank: shard descriptor from the received blob, its slice of the bitstream, decode, its rows back -----------------
    const Geometry g = geometry_of(*desc, out_format);
    std::vector<RankOut> ro((size_t)world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++)
        th.emplace_back([&, r] {
            RankOut &o = ro[(size_t)r];
            pjd_image_desc full;
            std::memcpy(&full, got[(size_t)r].data(), sizeof full);
            std::vector<uint64_t> segs((size_t)full.n_segments), scratch((size_t)full.n_segments);
            std::memcpy(segs.data(), got[(size_t)r].data() + sizeof full, segs.size() * sizeof(uint64_t));
            full.seg_offsets = segs.data();
            full.ecs = desc->ecs;                                   // host memory of the scanning rank; only [lo, hi) of it is touched below
            pjd_image_desc shard;
            uint64_t lo = 0, hi = 0; uint32_t m0 = 0, m1 = 0;
            const int pr = pjd_split_plan(&full, world, r, &shard, scratch.data(), &lo, &hi, &m0, &m1);
            if (pr == 1) return;
            if (pr != PJD_OK) { o.rc = pr; return; }
            o.has_work = true;
            o.ecs_bytes = hi - lo;
            hipSetDevice(devices[r]);
            pjd_batch *b = nullptr;
            double t0 = now_s();
            int rc = pjd_batch_create(ctx[(size_t)r], &shard, 1, out_format, &b);
            if (rc == PJD_OK) rc = pjd_batch_upload(b);
            o.upload_s = now_s() - t0; t0 = now_s();
            if (rc == PJD_OK) rc = pjd_batch_decode(b);
            if (rc == PJD_OK) rc = pjd_batch_sync(b);
            o.exec_s = now_s() - t0; t0 = now_s();
            if (rc == PJD_OK) {
                pjd_batch_info info;
                if (pjd_batch_get_info(b, &info) == PJD_OK) { o.n_fallback = info.n_fallback; o.n_sequential = info.n_sequential; }
                int32_t s1 = 0;
                rc = pjd_batch_download(b, nullptr, &s1);           // statuses only
                o.status = s1;
            }
            if (rc == PJD_OK && m1 > m0) {
                const uint8_t *dev_out = (const uint8_t *)pjd_batch_device_output(b, 0);
                hipStream_t s = (hipStream_t)pjd_stream(ctx[(size_t)r]);
                hipError_t e = hipSuccess;
                std::vector<uint8_t> band;
                const uint32_t b_first = m0 / g.mcux, b_last = (m1 - 1) / g.mcux;
                uint32_t run_lo = 0, run_hi = 0; bool in_run = false;
                auto flush_run = [&] {                              // consecutive MCU rows this rank owns whole: straight into the caller's picture
                    if (!in_run) return;
                    uint64_t a, z;
                    g.row_bytes(run_lo * g.mcu_h, std::min(g.H, run_hi * g.mcu_h), a, z);
                    if (e == hipSuccess) e = hipMemcpyAsync(out + a, dev_out + a, z - a, hipMemcpyDeviceToHost, s);
                    in_run = false;
                };
                for (uint32_t bnd = b_first; bnd <= b_last && e == hipSuccess; bnd++) {
                    const uint32_t c0 = bnd == b_first ? m0 % g.mcux : 0, c1 = bnd == b_last ? (m1 - 1) % g.mcux + 1 : g.mcux;
                    if (c0 == 0 && c1 == g.mcux) {
                        if (!in_run) { run_lo = bnd; in_run = true; }
                        run_hi = bnd + 1;
                        continue;
                    }
                    flush_run();
                    // an MCU row shared with a neighbouring rank: fetch its rows, keep the columns of this rank's MCUs
                    const uint32_t y0 = bnd * g.mcu_h, y1 = std::min(g.H, (bnd + 1) * g.mcu_h);
                    uint64_t a, z;
                    g.row_bytes(y0, y1, a, z);
                    band.resize(z - a);
                    e = hipMemcpyAsync(band.data(), dev_out + a, z - a, hipMemcpyDeviceToHost, s);
                    if (e == hipSuccess) e = hipStreamSynchronize(s);
                    if (e != hipSuccess) break;
                    const uint64_t x0 = (uint64_t)c0 * g.mcu_w * 3;
                    const uint64_t x1 = c1 == g.mcux ? g.stride : std::min<uint64_t>((uint64_t)c1 * g.mcu_w, g.W) * 3;   // the row's last MCU also owns its padding
                    if (x1 > x0)
                        for (uint32_t y = y0; y < y1; y++) std::memcpy(out + g.row_off(y) + x0, band.data() + (g.row_off(y) - a) + x0, x1 - x0);
                }
                flush_run();
                if (e == hipSuccess && g.bmp && m0 == 0) e = hipMemcpyAsync(out, dev_out, 26, hipMemcpyDeviceToHost, s);    // the file header comes with MCU 0
                if (e == hipSuccess) e = hipStreamSynchronize(s);
                if (e != hipSuccess) { rc = PJD_E_HIP; o.err = hipGetErrorString(e); }
            }
            if (rc != PJD_OK && o.err.empty()) o.err = pjd_last_error(ctx[(size_t)r]);
            o.download_s = now_s() - t0;
            if (b) pjd_batch_destroy(b);
            o.rc = rc;
        });
    for (std::thread &t : th) t.join();
    int rc = PJD_OK;
    int32_t st_all = 0;
    for (int r = 0; r < world; r++) {
        const RankOut &o = ro[(size_t)r];
        if (o.rc != PJD_OK && rc == PJD_OK) rc = o.rc;
        if (o.status != 0 && st_all == 0) st_all = o.status;
        st.upload_s = std::max(st.upload_s, o.upload_s); st.exec_s = std::max(st.exec_s, o.exec_s); st.download_s = std::max(st.download_s, o.download_s);
        st.ecs_bytes[r] = o.ecs_bytes;
        st.n_exact += (uint32_t)(o.n_fallback + o.n_sequential);
        if (o.has_work) st.n_ranks++;
    }
    close_all();
    if (rc == PJD_E_ARG) {
        // the planner would not shard it (tables the parallel decoder does not take go to the exact kernel as a whole picture)
        st.redone_whole = 1;
        rc = whole(devices[0], *desc);
    } else if (rc == PJD_OK && st_all != 0) {
        // an entropy-coding error somewhere: the reference stops there and leaves the rest of the picture undecoded, which no
        // split reproduces -- decode the picture once more on one device
        st.redone_whole = 1;
        rc = whole(devices[0], *desc);
    } else if (status) {
        *status = 0;
    }
    st.wall_s = now_s() - t_all;
    if (stats_out) *stats_out = st;
    return rc;
}

}  // extern "C"
