// pjd_api.hip -- the C ABI of include/pjd.h on top of the gfx950 kernels.
//
// Host-side flow of one batch (what replaces the reference's consumer thread,
// reference src/decoder_host.cpp:213-350):
//   create   plan (pjd_plan.cpp) + allocate HBM and pinned staging
//   upload   one packed H2D copy of the bitstreams + the small work lists
//   decode   table build -> lane words -> parallel Huffman decode (lane streams) -> DC scan over lanes -> fused
//            IDCT/colour; images routed to the exact kernel: dense scratch -> exact kernel -> dense IDCT/colour
//   sync     read the status words; any image the parallel decoder flagged is re-decoded by the
//            exact kernel (on the GPU) and its picture regenerated
//   download one D2H copy per picture
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/pjd.h"
#include "pjd_kernels.h"
#include "pjd_plan.h"

// Buffers of destroyed batches are kept per context and handed to the next batch (a steady stream of
// batches then allocates nothing); bounded by pool_cap, freed at pjd_close.
struct PoolBlock { void *p; size_t bytes; };
struct Pool {
    std::mutex m;                          // the owner takes and gives; any context of the device may flush (out of memory)
    std::vector<PoolBlock> free_blocks;
    size_t bytes = 0;
    void *take(size_t want, size_t &got)
    {
        std::lock_guard<std::mutex> l(m);
        int best = -1;
        for (size_t k = 0; k < free_blocks.size(); k++) {
            const size_t sz = free_blocks[k].bytes;
            if (sz >= want && sz <= 2 * want + (1u << 20) && (best < 0 || sz < free_blocks[best].bytes)) best = (int)k;
        }
        if (best < 0) return nullptr;
        void *p = free_blocks[best].p;
        got = free_blocks[best].bytes;
        bytes -= got;
        free_blocks.erase(free_blocks.begin() + best);
        return p;
    }
    bool give(void *p, size_t sz, size_t cap)
    {
        std::lock_guard<std::mutex> l(m);
        if (bytes + sz > cap) return false;
        free_blocks.push_back({p, sz}); bytes += sz;
        return true;
    }
    template <class F> void flush(F release)            // hand everything cached back to the runtime
    {
        std::lock_guard<std::mutex> l(m);
        for (PoolBlock &k : free_blocks) release(k.p);
        free_blocks.clear(); bytes = 0;
    }
};

struct pjd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Pool dev_pool, pin_pool;
    size_t pool_cap = (size_t)64 << 30;   // bytes of HBM kept for reuse (PJD_POOL_GB)
    std::string err;
    bool force_sequential = false;
    uint32_t sub_bytes_override = 0;
    int plan_mode = PJD_PLAN_LATENCY;     // pjd_set_plan_mode
    // picture groups: the chains of groups 1.. run on these, forked from / joined to `stream` (created on first use)
    std::vector<hipStream_t> group_streams;
    std::vector<hipEvent_t> join_ev;
    hipEvent_t fork_ev = nullptr, tables_ev = nullptr;
};

#define HIP_TRY(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
            return PJD_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

namespace {

// Decodes issued and not yet drained, per device, over all contexts of the process.  A decode issued while the device has nothing else
// of ours to do spreads its picture groups over several streams (the back end of the light pictures then runs beside the last chains
// of the entropy decode: one batch alone finishes earlier); one issued while others run keeps to ONE stream -- several batches in
// flight fill the device by themselves, and more streams than the runtime has hardware queues serialise each other
// (measured: three groups 2.89 -> 2.59 ms for a batch alone, but 122 -> 91 GPix/s with four batches in flight).
std::atomic<int> g_active[64];

// every open context, so that a context that runs out of HBM can make the others of its device give their caches back
std::mutex g_ctx_m;
std::vector<pjd_ctx *> g_ctxs;

int pool_dev_alloc(pjd_ctx *ctx, void **out, size_t bytes, std::vector<PoolBlock> &owned)
{
    if (bytes == 0) bytes = 16;
    size_t got = 0;
    void *p = ctx->dev_pool.take(bytes, got);
    if (!p) {
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {                                                // cached memory of every context on this device goes back; retry
            (void)hipGetLastError();
            std::lock_guard<std::mutex> l(g_ctx_m);
            for (pjd_ctx *c : g_ctxs)
                if (c->device == ctx->device) c->dev_pool.flush([](void *q) { (void)hipFree(q); });
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return PJD_E_NOMEM; }
        got = bytes;
    }
    owned.push_back({p, got});
    *out = p;
    return PJD_OK;
}

int pool_pin_alloc(pjd_ctx *ctx, void **out, size_t bytes, std::vector<PoolBlock> &owned)
{
    if (bytes == 0) bytes = 16;
    size_t got = 0;
    void *p = ctx->pin_pool.take(bytes, got);
    if (!p) {
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { ctx->err = "hipHostMalloc failed"; return PJD_E_NOMEM; }
        got = bytes;
    }
    owned.push_back({p, got});
    *out = p;
    return PJD_OK;
}

}  // namespace

struct pjd_batch {
    pjd_ctx *ctx = nullptr;
    PjdPlan plan;
    PjdDevBatch dev{};
    // owned device allocations (non-const views of what `dev` points to)
    PjdDevImage *d_images = nullptr;
    PjdDevTset *d_tsets = nullptr;
    PjdDevHuffRaw *d_raw = nullptr;
    uint16_t *d_qtab = nullptr;
    PjdDevSegment *d_segs = nullptr;
    PjdDevSub *d_lanes = nullptr;
    PjdDevHuffWave *d_hwaves = nullptr;
    PjdDevHuffWg *d_hwgs = nullptr;
    PjdDevIdctWg *d_iwgs = nullptr;
    PjdDevIdctWg *d_iwgs_dense = nullptr;
    uint8_t *d_ecs = nullptr;
    uint32_t *d_seq_list = nullptr;      // images routed to the exact kernel up front ...
    uint64_t *d_seq_base = nullptr;      // ... and where each one's data units start in the dense scratch
    int32_t *d_status_init = nullptr;
    uint64_t *d_opstate = nullptr;       // wave_gen + wave_desc + ticket
    size_t opstate_bytes = 0;
    uint8_t *d_in = nullptr, *h_in = nullptr;   // the input blob (work lists + bitstreams) in HBM / page-locked memory
    size_t in_bytes = 0;
    int32_t *h_status = nullptr;         // pinned
    unsigned long long *h_stats = nullptr;   // pinned, 16 words
    std::vector<uint32_t> seq_list;      // what d_seq_list holds
    std::vector<PoolBlock> dev_blocks, pin_blocks;   // everything this batch took from the context's pools
    uint64_t device_bytes = 0;
    bool uploaded = false, decoded = false, settled = false;
    int n_fallback = 0;
    float exact_fallback_ms = 0;
    uint32_t n_entropy_errors = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph_groups = nullptr;           // the same decode with the picture groups' chains as parallel branches (pjd_internal.h)
    hipGraphExec_t graph_exec_groups = nullptr;
    bool counted = false;                        // this batch's decode is counted in g_active[device]
};

extern "C" {

int pjd_version(void) { return PJD_VERSION; }

int pjd_open(int device_ordinal, pjd_ctx **out)
{
    if (!out) return PJD_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return PJD_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return PJD_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PJD_E_NODEVICE;   // kernels are built for gfx950 only
    pjd_ctx *c = new pjd_ctx;
    c->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return PJD_E_HIP;
    }
    const char *fs = std::getenv("PJD_FORCE_SEQUENTIAL");
    c->force_sequential = fs && fs[0] == '1';
    const char *pg = std::getenv("PJD_POOL_GB");
    if (pg) c->pool_cap = (size_t)std::atoll(pg) << 30;
    const char *sb = std::getenv("PJD_SUB_BYTES");
    c->sub_bytes_override = sb ? (uint32_t)std::atoi(sb) : 0;
    if (const char *pm = std::getenv("PJD_PLAN_MODE")) c->plan_mode = (pm[0] == 't' || pm[0] == '1') ? PJD_PLAN_THROUGHPUT : PJD_PLAN_LATENCY;
    { std::lock_guard<std::mutex> l(g_ctx_m); g_ctxs.push_back(c); }
    *out = c;
    return PJD_OK;
}

int pjd_set_plan_mode(pjd_ctx *ctx, int mode)
{
    if (!ctx) return PJD_E_ARG;
    if (mode != PJD_PLAN_LATENCY && mode != PJD_PLAN_THROUGHPUT) { ctx->err = "unknown plan mode"; return PJD_E_ARG; }
    ctx->plan_mode = mode;
    return PJD_OK;
}

void pjd_close(pjd_ctx *ctx)
{
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> l(g_ctx_m);
        for (size_t k = 0; k < g_ctxs.size(); k++)
            if (g_ctxs[k] == ctx) { g_ctxs.erase(g_ctxs.begin() + k); break; }
    }
    hipSetDevice(ctx->device);
    if (ctx->stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
    for (hipStream_t st : ctx->group_streams) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (hipEvent_t ev : ctx->join_ev) hipEventDestroy(ev);
    if (ctx->fork_ev) hipEventDestroy(ctx->fork_ev);
    if (ctx->tables_ev) hipEventDestroy(ctx->tables_ev);
    ctx->dev_pool.flush([](void *q) { (void)hipFree(q); });
    ctx->pin_pool.flush([](void *q) { (void)hipHostFree(q); });
    delete ctx;
}

const char *pjd_last_error(pjd_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

const char *pjd_status_string(int status)
{
    switch (status & 0xFF) {   // reference src/jpeg_scanner.cpp:471,475,481,491,501,507,513
        case PJD_ST_OK: return "";
        case PJD_ST_DC_SYM: return "Error - Invalid DC value (255)";
        case PJD_ST_DC_LEN: return "Error - DC coefficient length greater than 11";
        case PJD_ST_DC_BITS: return "Error - Invalid DC value";
        case PJD_ST_AC_SYM: return "Error - Invalid AC value";
        case PJD_ST_AC_RUN: return "Error - Zero run-length exceeded block component";
        case PJD_ST_AC_LEN: return "Error - AC coefficient length greater than 10";
        case PJD_ST_AC_BITS: return "Error - Invalid AC value";
        default: return "Error - unknown";
    }
}

void *pjd_stream(pjd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

void pjd_batch_destroy(pjd_batch *b)
{
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->counted) { g_active[b->ctx->device].fetch_sub(1); b->counted = false; }
    if (b->graph_exec) hipGraphExecDestroy(b->graph_exec);
    if (b->graph) hipGraphDestroy(b->graph);
    if (b->graph_exec_groups) hipGraphExecDestroy(b->graph_exec_groups);
    if (b->graph_groups) hipGraphDestroy(b->graph_groups);
    pjd_ctx *ctx = b->ctx;
    for (PoolBlock &k : b->dev_blocks)
        if (!ctx->dev_pool.give(k.p, k.bytes, ctx->pool_cap)) hipFree(k.p);
    for (PoolBlock &k : b->pin_blocks)
        if (!ctx->pin_pool.give(k.p, k.bytes, ctx->pool_cap / 4)) hipHostFree(k.p);
    delete b;
}

int pjd_batch_create(pjd_ctx *ctx, const pjd_image_desc *images, int n_images, int out_format, pjd_batch **out)
{
    if (!ctx || !out) return PJD_E_ARG;
    *out = nullptr;
    pjd_batch *b = new pjd_batch;
    b->ctx = ctx;
    int rc = pjd_make_plan(images, n_images, out_format, b->plan, ctx->err, ctx->sub_bytes_override, ctx->plan_mode);
    if (rc != PJD_OK) { delete b; return rc; }
    PjdPlan &P = b->plan;
    hipSetDevice(ctx->device);

    // One input blob per batch: the work lists and tables, then the packed bitstreams.  It is assembled in page-locked
    // memory now (the caller's buffers can go away) and goes up in ONE copy: every separate copy of a pageable vector is
    // staged by the runtime and waits its turn behind other streams' transfers on the copy engine (profiles/r02_pcie.md).
    rc = PJD_OK;
    auto fail = [&](int code) { pjd_batch_destroy(b); return code; };
    b->seq_list = P.seq_images;
    std::vector<uint64_t> seq_base;
    std::vector<int32_t> st0(P.images.size(), 0);
    for (uint32_t i : b->seq_list) { st0[i] = PJD_STW_NEEDS_EXACT; seq_base.push_back(P.images[i].dense_base); }
    struct Part { const void *src; size_t bytes, off; };
    std::vector<Part> parts;
    size_t in_bytes = 0;
    auto part = [&](const auto &vec, size_t min_count) {
        using T = std::remove_cv_t<std::remove_reference_t<decltype(vec[0])>>;
        const size_t off = in_bytes;
        parts.push_back({vec.data(), vec.size() * sizeof(T), off});
        in_bytes = (off + std::max(vec.size(), std::max(min_count, (size_t)1)) * sizeof(T) + 255) & ~(size_t)255;
        return off;
    };
    const size_t o_images = part(P.images, 0), o_tsets = part(P.tsets, 0), o_raw = part(P.tables, 0), o_qtab = part(P.qtab, 0);
    const size_t o_segs = part(P.segs, 0), o_lanes = part(P.subs, 0), o_hwaves = part(P.hwaves, 0), o_hwgs = part(P.hwgs, 0);
    const size_t o_iwgs = part(P.iwgs, 0), o_iwgs_dense = part(P.iwgs_dense, 0), o_pscans = part(P.pscans, 0);
    const size_t o_gimg = part(P.group_images, 0), o_iorder = part(P.iwg_order, 0);
    const size_t o_seq_list = part(b->seq_list, (size_t)n_images), o_seq_base = part(seq_base, (size_t)n_images), o_st0 = part(st0, (size_t)n_images);
    const size_t o_ecs = in_bytes;
    in_bytes += P.ecs_buf_bytes;
    b->in_bytes = in_bytes;
    if (pool_pin_alloc(ctx, (void **)&b->h_in, in_bytes, b->pin_blocks) != PJD_OK) return fail(PJD_E_NOMEM);
    for (const Part &q : parts) if (q.bytes) std::memcpy(b->h_in + q.off, q.src, q.bytes);
    {   // streams at their offsets, zeros in between (every stream is followed by >= 48 zero bytes)
        uint8_t *h_ecs = b->h_in + o_ecs;
        uint64_t pos = 0;
        for (int i = 0; i < n_images; i++) {
            const uint64_t off = P.images[i].ecs_off, len = P.host[i].ecs_copy_len;
            if (off > pos) std::memset(h_ecs + pos, 0, off - pos);
            if (len) std::memcpy(h_ecs + off, P.host[i].ecs_src, len);
            pos = off + len;
            if (P.images[i].flags & PJD_IF_PROGRESSIVE)              // its scans follow, each at its own offset
                for (uint32_t k = 0; k < P.images[i].n_pscan; k++) {
                    const PjdHostScan &hs = P.host_scans[P.images[i].pscan_base + k];
                    if (hs.off > pos) std::memset(h_ecs + pos, 0, hs.off - pos);
                    if (hs.len) std::memcpy(h_ecs + hs.off, hs.src, hs.len);
                    pos = hs.off + hs.len;
                }
        }
        std::memset(h_ecs + pos, 0, P.ecs_buf_bytes - pos);
    }
    if (pool_pin_alloc(ctx, (void **)&b->h_status, sizeof(int32_t) * (n_images + 1), b->pin_blocks) != PJD_OK) return fail(PJD_E_NOMEM);
    if (pool_pin_alloc(ctx, (void **)&b->h_stats, 16 * sizeof(unsigned long long), b->pin_blocks) != PJD_OK) return fail(PJD_E_NOMEM);

    uint64_t &tot = b->device_bytes;
#define TRY_RC(x) do { int rc_ = (x); if (rc_ != PJD_OK) return fail(rc_); } while (0)
    auto dev_alloc = [&](pjd_ctx *c, auto *&dptr, size_t n, uint64_t &total) {
        using T = std::remove_reference_t<decltype(*dptr)>;
        if (n == 0) n = 1;
        void *p = nullptr;
        const int r = pool_dev_alloc(c, &p, n * sizeof(T), b->dev_blocks);
        dptr = (T *)p;
        total += n * sizeof(T);
        return r;
    };
    const size_t n_hwave = P.hwaves.size();
    TRY_RC(dev_alloc(ctx, b->d_in, in_bytes, tot));
    b->d_images = (PjdDevImage *)(b->d_in + o_images); b->d_tsets = (PjdDevTset *)(b->d_in + o_tsets);
    b->d_raw = (PjdDevHuffRaw *)(b->d_in + o_raw); b->d_qtab = (uint16_t *)(b->d_in + o_qtab);
    b->d_segs = (PjdDevSegment *)(b->d_in + o_segs); b->d_lanes = (PjdDevSub *)(b->d_in + o_lanes);
    b->d_hwaves = (PjdDevHuffWave *)(b->d_in + o_hwaves); b->d_hwgs = (PjdDevHuffWg *)(b->d_in + o_hwgs);
    b->d_iwgs = (PjdDevIdctWg *)(b->d_in + o_iwgs); b->d_iwgs_dense = (PjdDevIdctWg *)(b->d_in + o_iwgs_dense);
    b->dev.pscans = (const PjdDevScan *)(b->d_in + o_pscans);
    b->dev.group_images = (const uint32_t *)(b->d_in + o_gimg); b->dev.iwg_order = (const uint32_t *)(b->d_in + o_iorder);
    b->d_seq_list = (uint32_t *)(b->d_in + o_seq_list); b->d_seq_base = (uint64_t *)(b->d_in + o_seq_base);
    b->d_status_init = (int32_t *)(b->d_in + o_st0);
    b->d_ecs = b->d_in + o_ecs;
    TRY_RC(dev_alloc(ctx, b->dev.luts, (size_t)P.lut_buf_bytes, tot));
    TRY_RC(dev_alloc(ctx, b->dev.words, (size_t)P.n_words, tot));
    TRY_RC(dev_alloc(ctx, b->dev.coef, P.dense_du * 64, tot));
    TRY_RC(dev_alloc(ctx, b->dev.ent, (size_t)P.n_ent, tot));
    TRY_RC(dev_alloc(ctx, b->dev.lane_info, P.subs.size(), tot));
    TRY_RC(dev_alloc(ctx, b->dev.lane_dc, P.subs.size(), tot));
    TRY_RC(dev_alloc(ctx, b->dev.dc_blk, (size_t)P.n_dcblk * 8, tot));
    TRY_RC(dev_alloc(ctx, b->dev.marks, P.iwgs.size(), tot));
    TRY_RC(dev_alloc(ctx, b->dev.out, P.out_buf_bytes, tot));
    TRY_RC(dev_alloc(ctx, b->dev.status, (size_t)n_images, tot));
    TRY_RC(dev_alloc(ctx, b->dev.imstate, (size_t)n_images, tot));
    // wave_gen [PJD_GENS][n_hwave], wave_desc [n_hwave] and the ticket live in one allocation, zeroed before every launch
    // ... a ticket per picture group, and the pull back end's list, tail and done flags (pjd_internal.h)
    const size_t op_words = n_hwave * (PJD_GENS + 1) + 2 + PJD_MAX_GROUPS, pull_words = P.iwgs.size() + 2;      // 64-bit words: 2 x n_iwg + 2 dwords
    TRY_RC(dev_alloc(ctx, b->d_opstate, op_words + pull_words, tot));
    b->opstate_bytes = (op_words + pull_words) * sizeof(uint64_t);
    b->dev.ready_tail = reinterpret_cast<uint32_t *>(b->d_opstate + op_words);
    b->dev.ready_list = b->dev.ready_tail + 2;
    b->dev.range_done = b->dev.ready_list + P.iwgs.size();
    b->dev.pull = 0;
    b->dev.wave_gen = b->d_opstate;
    b->dev.wave_desc = b->d_opstate + n_hwave * PJD_GENS;
    b->dev.ticket = reinterpret_cast<uint32_t *>(b->d_opstate + n_hwave * (PJD_GENS + 1));
    b->dev.dbg = nullptr;
    if (std::getenv("PJD_DEBUG_STATS")) TRY_RC(dev_alloc(ctx, b->dev.dbg, n_hwave * 32, tot));
    TRY_RC(dev_alloc(ctx, b->dev.stats, 16, tot));
#undef TRY_RC
    b->dev.images = b->d_images; b->dev.tsets = b->d_tsets; b->dev.raw_tables = b->d_raw; b->dev.qtab = b->d_qtab;
    b->dev.segs = b->d_segs; b->dev.lanes = b->d_lanes; b->dev.hwaves = b->d_hwaves; b->dev.hwgs = b->d_hwgs; b->dev.iwgs = b->d_iwgs;
    b->dev.ecs = b->d_ecs;
    b->dev.n_images = (uint32_t)n_images; b->dev.n_tsets = (uint32_t)P.tsets.size(); b->dev.n_lanes = (uint32_t)P.subs.size();
    b->dev.n_hwave = (uint32_t)n_hwave; b->dev.n_hwg = (uint32_t)P.hwgs.size();
    b->dev.n_iwg = (uint32_t)P.iwgs.size(); b->dev.n_dcblk = (uint32_t)P.n_dcblk;
    b->dev.sub_bytes = P.sub_bytes;
    b->dev.word_rows = PJD_WORD_ROWS(P.sub_bytes);
    b->dev.lane_cap = P.lane_cap;
    b->dev.max_lut_bytes = P.max_lut_bytes;
    *out = b;
    return PJD_OK;
}

int pjd_batch_upload(pjd_batch *b)
{
    if (!b) return PJD_E_ARG;
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    // asynchronous: the blob is page-locked and owned by the batch
    HIP_TRY(ctx, hipMemcpyAsync(b->d_in, b->h_in, b->in_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(b->dev.out, 0, P.out_buf_bytes, s));   // BMP row padding stays zero
    b->uploaded = true;
    return PJD_OK;
}

}  // extern "C"

// ---- the decode sequence -------------------------------------------------------------------
namespace {

struct KernelTimer {
    pjd_timings *t;
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<std::string> names;
    bool debug_sync = std::getenv("PJD_DEBUG_SYNC") != nullptr;
    void mark(const char *name)
    {
        if (debug_sync) {                                  // PJD_DEBUG_SYNC=1: name the launch a fault belongs to
            const hipError_t e = hipStreamSynchronize(s);
            std::fprintf(stderr, "[pjd] %-14s %s\n", name, e == hipSuccess ? "ok" : hipGetErrorString(e));
        }
        if (!t) return;
        hipEvent_t e;
        hipEventCreate(&e);
        hipEventRecord(e, s);
        ev.push_back(e);
        names.push_back(name);
    }
    void finish()
    {
        if (!t) return;
        hipStreamSynchronize(s);
        t->n = 0;
        for (size_t k = 0; k + 1 < ev.size() && t->n < PJD_MAX_KERNELS; k++) {
            float ms = 0;
            hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
            t->ms[t->n] = ms;
            std::snprintf(t->name[t->n], sizeof t->name[0], "%s", names[k + 1].c_str());
            t->n++;
        }
        t->total_ms = 0;
        if (ev.size() >= 2) hipEventElapsedTime(&t->total_ms, ev.front(), ev.back());
        for (hipEvent_t e : ev) hipEventDestroy(e);
    }
};

// streams and events for `ng` picture groups (group 0 uses the context's own stream)
bool ctx_group_streams(pjd_ctx *ctx, size_t ng)
{
    if (!ctx->fork_ev && hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming) != hipSuccess) { ctx->fork_ev = nullptr; return false; }
    if (!ctx->tables_ev && hipEventCreateWithFlags(&ctx->tables_ev, hipEventDisableTiming) != hipSuccess) { ctx->tables_ev = nullptr; return false; }
    while (ctx->group_streams.size() + 1 < ng) {
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { hipStreamDestroy(st); return false; }
        ctx->group_streams.push_back(st);
        ctx->join_ev.push_back(ev);
    }
    return true;
}

int enqueue_decode(pjd_batch *b, pjd_timings *timings, bool use_groups)
{
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    hipStream_t s = ctx->stream;
    KernelTimer kt{timings, s, {}, {}};
    const bool parallel = !P.hwgs.empty();
    kt.mark("start");
    // One kernel of ours instead of two runtime memsets and a copy: status words from their initial values, statistics and
    // the words the Huffman waves publish to each other zeroed.  Inside a captured graph the runtime's small memset nodes are
    // not safe to replay next to other users of the runtime in the process: a 128-byte memset node replayed a 16-byte pattern
    // of stale pointers instead of zeros (seen with torch/gloo active before the capture; tools/r2 rehearsal of cfg5split).
    pjd_launch_reset(s, b->dev, b->d_status_init, parallel ? b->d_opstate : nullptr, parallel ? b->opstate_bytes / 8 : 0,
                     b->dev.dbg ? (uint32_t)(P.hwaves.size() * 32) : 0u);
    kt.mark("reset");
    // What a decode looks like on an otherwise idle device (use_groups), for a batch of many small pictures (the planner made groups):
    //   "groups" (default) three chains of launches (pictures by density) on three streams
    //   "pull"   the back end runs BESIDE the entropy decoder and takes pictures as their last wave completes them (pjd_internal.h).
    //            Bit-exact and complete (the GPU suite passes in this form), but SLOWER: the back end's waves share SIMDs with the
    //            entropy decoder's chains and stretch them -- 3.4-3.5 ms per batch against 2.6-2.7 for "groups" and 2.9 for "chain"
    //            (profiles/r04_experiments.md #16); kept as an experiment switch
    //   "chain"  as with several batches in flight: one chain of launches
    static const int idle_form = [] { const char *e = std::getenv("PJD_IDLE_FORM"); return !e ? 1 : (!std::strcmp(e, "pull") ? 2 : (!std::strcmp(e, "chain") ? 0 : 1)); }();
    const bool idle = !timings && use_groups && parallel && !P.groups.empty() && idle_form != 0;       // per-kernel timing: one chain, kernel after kernel
    const size_t ng = (idle && idle_form == 1) ? P.groups.size() : 0;
    const bool grouped = ng > 1 && ctx_group_streams(ctx, ng);
    if (idle && idle_form == 2 && ctx_group_streams(ctx, 2)) {
        // The pull back end (pjd_internal.h): entropy decode on the context's stream, the back end's pull launch on a second stream
        // beside it (inside a capture: two parallel one-node branches), then the sweep over whatever the pull launch left.
        PjdDevBatch dv = b->dev;
        dv.pull = 1;
        pjd_launch_build_tables(s, dv);
        pjd_launch_lane_words(s, dv);
        HIP_TRY(ctx, hipEventRecord(ctx->fork_ev, s));
        pjd_launch_huff_lanes(s, dv);
        hipStream_t s1 = ctx->group_streams[0];
        HIP_TRY(ctx, hipStreamWaitEvent(s1, ctx->fork_ev, 0));
        pjd_launch_idct_pull(s1, dv);
        HIP_TRY(ctx, hipEventRecord(ctx->join_ev[0], s1));
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->join_ev[0], 0));
        pjd_launch_idct_sweep(s, dv);
    } else if (grouped) {
        // Picture groups (pjd_internal.h): every group's chain bitstream words -> entropy decode -> DC predictors -> back end on a stream
        // of its own, forked from and joined to the context's stream with events (inside a stream capture these become parallel
        // branches of the graph).  Group 0 holds the densest pictures -- the longest chains of re-sync rounds -- and stays on the main
        // stream.  The chains have the same shape on purpose: with the decode tables built on the second stream beside group 0's
        // bitstream words (a branch one node longer than the other) the graph ran the two entropy decodes one after the other
        // (4.65 ms instead of 2.60 for a batch alone; profiles/r04_experiments.md).
        pjd_launch_build_tables(s, b->dev);
        HIP_TRY(ctx, hipEventRecord(ctx->fork_ev, s));
        for (size_t g = 0; g < ng; g++) {
            hipStream_t gs = g == 0 ? s : ctx->group_streams[g - 1];
            if (g) HIP_TRY(ctx, hipStreamWaitEvent(gs, ctx->fork_ev, 0));
            pjd_launch_lane_words_group(gs, b->dev, P.groups[g]);
            pjd_launch_huff_lanes_group(gs, b->dev, P.groups[g], (uint32_t)g);
            pjd_launch_group_dc(gs, b->dev, P.groups[g]);
            pjd_launch_group_idct(gs, b->dev, P.groups[g]);
            if (g) { HIP_TRY(ctx, hipEventRecord(ctx->join_ev[g - 1], gs)); }
        }
        for (size_t g = 1; g < ng; g++) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->join_ev[g - 1], 0));
    } else {
        if (parallel || !b->seq_list.empty()) { pjd_launch_build_tables(s, b->dev);  kt.mark("build_tables"); }      // the exact path uses the decode tables too
        if (parallel) {
            pjd_launch_lane_words(s, b->dev);    kt.mark("lane_words");
            pjd_launch_huff_lanes(s, b->dev);    kt.mark("huff_lanes");
            pjd_launch_lane_dc_scan(s, b->dev);  kt.mark("dc_scan");
            pjd_launch_idct_colour_lanes(s, b->dev);
            kt.mark("idct_colour");
        }
    }
    if (!b->seq_list.empty()) {
        // images routed to the exact kernel: dense int16 scratch, cleared first (unvisited slots are zero)
        pjd_launch_zero(s, b->dev.coef, P.dense_du * 64 * sizeof(int16_t));      // a kernel, not a memset node: see the reset above
        pjd_launch_huff_sequential(s, b->dev, b->d_seq_list, b->d_seq_base, (uint32_t)b->seq_list.size());
        if (!P.pscans.empty()) pjd_launch_progressive(s, b->dev, b->d_seq_list, b->d_seq_base, (uint32_t)b->seq_list.size());
        pjd_launch_idct_colour(s, b->dev, b->d_iwgs_dense, b->d_seq_base, (uint32_t)P.iwgs_dense.size());
        kt.mark("exact_path");
    }
    HIP_TRY(ctx, hipGetLastError());
    kt.finish();
    b->decoded = true;
    b->settled = false;
    return PJD_OK;
}

// After the stream drained: re-decode, with the exact kernel, every image the parallel decoder
// flagged -- all of them in ONE launch, into a dense scratch allocated for just them (a rare path: corrupt
// or otherwise irregular streams).  Runs on the GPU.  A shard is re-decoded over its own segment range.
// As in the reference (decoder_host.cpp:181 drops decode_Huffman_data's result) such an image keeps its status
// and its partial picture; the rest of the batch is unaffected.
// one stream per device for the packed downloads of all contexts (never destroyed: a process has few devices)
hipStream_t download_stream(int device)
{
    static std::mutex m;
    static hipStream_t streams[64] = {nullptr};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> l(m);
    if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) streams[device] = nullptr;
    return streams[device];
}

int settle(pjd_batch *b)
{
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    hipStream_t s = ctx->stream;
    if (b->settled || !b->decoded) return PJD_OK;
    const size_t n = P.images.size();
    HIP_TRY(ctx, hipMemcpyAsync(b->h_status, b->dev.status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (b->counted) { g_active[ctx->device].fetch_sub(1); b->counted = false; }       // this decode has left the device
    std::vector<uint32_t> fb;
    std::vector<uint64_t> fb_base;
    std::vector<PjdDevIdctWg> fb_wgs;
    std::vector<char> was_seq(n, 0);
    for (uint32_t i : b->seq_list) was_seq[i] = 1;
    uint64_t du = 0;
    for (size_t i = 0; i < n; i++)
        if ((b->h_status[i] & PJD_STW_NEEDS_EXACT) && !was_seq[i]) {
            const PjdDevImage &g = P.images[i];
            for (uint32_t k = 0; k < g.n_iwg; k++) {
                PjdDevIdctWg w = P.iwgs[g.iwg_base + k];
                w.pad_ = (uint32_t)fb.size();
                fb_wgs.push_back(w);
            }
            fb.push_back((uint32_t)i);
            fb_base.push_back(du);
            du += (uint64_t)(g.last_mcu - g.first_mcu) * g.dus_per_mcu;
        }
    b->n_fallback = (int)fb.size();
    b->exact_fallback_ms = 0;
    b->n_entropy_errors = 0;
    for (size_t i = 0; i < n; i++)
        if (!(b->h_status[i] & PJD_STW_NEEDS_EXACT) && (b->h_status[i] & 0xFF) != 0) b->n_entropy_errors++;
    if (!fb.empty()) {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
        int16_t *coef = nullptr; uint32_t *d_list = nullptr; uint64_t *d_base = nullptr; PjdDevIdctWg *d_wgs = nullptr;
        auto cleanup = [&] { hipFree(coef); hipFree(d_list); hipFree(d_base); hipFree(d_wgs); };
        if (hipMalloc((void **)&coef, du * 64 * sizeof(int16_t)) != hipSuccess || hipMalloc((void **)&d_list, fb.size() * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&d_base, fb.size() * sizeof(uint64_t)) != hipSuccess || hipMalloc((void **)&d_wgs, fb_wgs.size() * sizeof(PjdDevIdctWg)) != hipSuccess) {
            cleanup();
            ctx->err = "hipMalloc failed for the exact-kernel scratch";
            return PJD_E_NOMEM;
        }
        PjdDevBatch dv = b->dev;
        dv.coef = coef;
        pjd_launch_zero(s, coef, du * 64 * sizeof(int16_t));      // our own kernel, as everywhere on the decode path
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(d_list, fb.data(), fb.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_base, fb_base.data(), fb_base.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_wgs, fb_wgs.data(), fb_wgs.size() * sizeof(PjdDevIdctWg), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            if (ev0) (void)hipEventRecord(ev0, s);
            pjd_launch_huff_sequential(s, dv, d_list, d_base, (uint32_t)fb.size());
            pjd_launch_idct_colour(s, dv, d_wgs, d_base, (uint32_t)fb_wgs.size());
            if (ev1) (void)hipEventRecord(ev1, s);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(b->h_status, b->dev.status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);        // also: the host vectors above are pageable
        if (e == hipSuccess && ev0 && ev1) (void)hipEventElapsedTime(&b->exact_fallback_ms, ev0, ev1);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        cleanup();
        if (e != hipSuccess) { ctx->err = std::string("exact-kernel fallback: ") + hipGetErrorString(e); return PJD_E_HIP; }
    }
    b->settled = true;
    return PJD_OK;
}

}  // namespace

extern "C" {

static bool device_idle_then_count(pjd_batch *b)
{
    const int dev = b->ctx->device;
    if (dev < 0 || dev >= 64) return false;
    const int before = b->counted ? g_active[dev].load() - 1 : g_active[dev].fetch_add(1);
    b->counted = true;
    static const int force = [] { const char *e = std::getenv("PJD_GROUPS_ALWAYS"); return e ? std::atoi(e) : -1; }();   // experiments: 1 always, 0 never
    if (force >= 0) return force != 0;
    return before <= 0;
}

int pjd_batch_decode(pjd_batch *b)
{
    if (!b) return PJD_E_ARG;
    if (!b->uploaded) { b->ctx->err = "decode before upload"; return PJD_E_STATE; }
    hipSetDevice(b->ctx->device);
    const bool groups = !b->plan.groups.empty() && device_idle_then_count(b);
    if (b->graph_exec) {
        hipGraphExec_t ge = (groups && b->graph_exec_groups) ? b->graph_exec_groups : b->graph_exec;
        HIP_TRY(b->ctx, hipGraphLaunch(ge, b->ctx->stream));
        b->decoded = true; b->settled = false;
        return PJD_OK;
    }
    return enqueue_decode(b, nullptr, groups);
}

int pjd_batch_decode_timed(pjd_batch *b, pjd_timings *t)
{
    if (!b || !t) return PJD_E_ARG;
    if (!b->uploaded) { b->ctx->err = "decode before upload"; return PJD_E_STATE; }
    hipSetDevice(b->ctx->device);
    std::memset(t, 0, sizeof *t);
    return enqueue_decode(b, t, false);
}

int pjd_batch_capture(pjd_batch *b)
{
    if (!b) return PJD_E_ARG;
    if (!b->uploaded) { b->ctx->err = "capture before upload"; return PJD_E_STATE; }
    pjd_ctx *ctx = b->ctx;
    hipSetDevice(ctx->device);
    if (b->graph_exec) return PJD_OK;
    for (int variant = 0; variant < (b->plan.groups.empty() ? 1 : 2); variant++) {
        // variant 0: the whole batch in one chain of launches; variant 1: the picture groups' chains as parallel branches
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_decode(b, nullptr, variant == 1);
        hipGraph_t &gr = variant ? b->graph_groups : b->graph;
        hipError_t e = hipStreamEndCapture(ctx->stream, &gr);
        b->decoded = false;
        if (rc != PJD_OK) return rc;
        if (e != hipSuccess) { ctx->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return PJD_E_HIP; }
        HIP_TRY(ctx, hipGraphInstantiate(variant ? &b->graph_exec_groups : &b->graph_exec, gr, nullptr, nullptr, 0));
    }
    return PJD_OK;
}

int pjd_batch_sync(pjd_batch *b)
{
    if (!b) return PJD_E_ARG;
    hipSetDevice(b->ctx->device);
    if (b->decoded) return settle(b);
    HIP_TRY(b->ctx, hipStreamSynchronize(b->ctx->stream));
    return PJD_OK;
}

int pjd_batch_download(pjd_batch *b, uint8_t *const *out, int32_t *status)
{
    if (!b) return PJD_E_ARG;
    if (!b->decoded) { b->ctx->err = "download before decode"; return PJD_E_STATE; }
    hipSetDevice(b->ctx->device);
    int rc = settle(b);
    if (rc != PJD_OK) return rc;
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    if (out)
        for (size_t i = 0; i < P.images.size(); i++)
            if (out[i]) HIP_TRY(ctx, hipMemcpyAsync(out[i], b->dev.out + P.images[i].out_off, P.host[i].out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (status)
        for (size_t i = 0; i < P.images.size(); i++) status[i] = b->h_status[i] & 0xFF;
    return PJD_OK;
}

int pjd_batch_download_packed(pjd_batch *b, uint8_t *host, uint64_t capacity, int32_t *status)
{
    if (!b || !host) return PJD_E_ARG;
    if (!b->decoded) { b->ctx->err = "download before decode"; return PJD_E_STATE; }
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    hipSetDevice(ctx->device);
    if (capacity < P.out_buf_bytes) { ctx->err = "download_packed: buffer smaller than pjd_batch_packed_size"; return PJD_E_ARG; }
    int rc = settle(b);
    if (rc != PJD_OK) return rc;
    // The runtime's copy (SDMA engine) by default.  PJD_DOWNLOAD=kernel: a small copy kernel on the device's download
    // stream storing into the mapped page-locked destination instead -- it keeps the link as busy (tools/pcie_probe.hip)
    // and leaves the engine to the uploads, but stores waiting for the link slow concurrent decode kernels down, and
    // the pipelined batcher measured 9.1 GPix/s with it against 11.8 with the engine (profiles/r02_pcie.md).
    void *mapped = nullptr;
    static const bool by_kernel = [] { const char *e = std::getenv("PJD_DOWNLOAD"); return e && !std::strcmp(e, "kernel"); }();
    hipPointerAttribute_t attr;
    const bool pinned = by_kernel && (P.out_buf_bytes % 16) == 0 && ((uintptr_t)host % 16) == 0 &&
                        hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost &&
                        hipHostGetDevicePointer(&mapped, host, 0) == hipSuccess && mapped;
    if (by_kernel && !pinned) (void)hipGetLastError();
    hipStream_t ds = pinned ? download_stream(ctx->device) : nullptr;
    if (ds) {
        // settle() has synchronised ctx->stream: the pictures are final
        hipEvent_t done;
        HIP_TRY(ctx, hipEventCreateWithFlags(&done, hipEventDisableTiming));
        pjd_launch_copy_out(ds, b->dev.out, mapped, P.out_buf_bytes);
        hipError_t e = hipEventRecord(done, ds);
        if (e == hipSuccess) e = hipEventSynchronize(done);
        (void)hipEventDestroy(done);
        HIP_TRY(ctx, e);
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(host, b->dev.out, P.out_buf_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (status)
        for (size_t i = 0; i < P.images.size(); i++) status[i] = b->h_status[i] & 0xFF;
    return PJD_OK;
}

uint64_t pjd_batch_packed_size(pjd_batch *b) { return b ? b->plan.out_buf_bytes : 0; }

uint64_t pjd_batch_output_offset(pjd_batch *b, int image)
{
    if (!b || image < 0 || (size_t)image >= b->plan.images.size()) return 0;
    return b->plan.images[image].out_off;
}

void *pjd_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void pjd_host_free(void *p) { if (p) hipHostFree(p); }

int pjd_batch_get_info(pjd_batch *b, pjd_batch_info *info)
{
    if (!b || !info) return PJD_E_ARG;
    PjdPlan &P = b->plan;
    hipSetDevice(b->ctx->device);
    std::memset(info, 0, sizeof *info);
    info->n_images = (int32_t)P.images.size();
    info->pixels = P.pixels; info->ecs_bytes = P.ecs_bytes; info->out_bytes = P.out_bytes;
    info->coef_bytes = P.n_ent * 2 + P.n_words * 4 + P.dense_du * 128;
    info->n_data_units = P.n_du;
    info->n_subsequences = P.subs.size();
    info->device_bytes = b->device_bytes;
    info->n_sequential = (int32_t)b->seq_list.size();
    info->n_fallback = b->n_fallback;
    info->exact_fallback_ms = b->exact_fallback_ms;
    info->n_entropy_errors = b->n_entropy_errors;
    info->sub_bytes = P.sub_bytes;
    info->plan_mode = (uint32_t)P.plan_mode;
    info->n_table_sets = (uint32_t)P.tsets.size();
    info->huff_lds_bytes = (uint32_t)(P.max_lut_bytes + PJD_HUFF_WAVES * (PJD_WAVE_LDS + PJD_PHASE_LDS) + 16);
    info->n_huff_waves = P.hwaves.size();
    // on the batch's own stream into page-locked memory: a plain hipMemcpy would wait for every other stream of the
    // device (it made the slots of the pipelined batcher run in lockstep, profiles/r02_pcie.md)
    unsigned long long *st = b->h_stats;
    if (b->decoded && hipMemcpyAsync(st, b->dev.stats, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->ctx->stream) == hipSuccess &&
        hipStreamSynchronize(b->ctx->stream) == hipSuccess) {
        info->sync_rounds = st[0]; info->sync_lane_passes = st[1]; info->fix_rounds = st[2]; info->fix_lane_passes = st[3];
        info->walks = st[PJD_STAT_WALKS]; info->walk_lanes = st[PJD_STAT_WALKS + 1];
        for (int r = 0; r < PJD_FLAG_REASONS && r < 8; r++) info->flag_waves[r] = st[PJD_STAT_FLAG0 + r];
        info->n_entries = st[PJD_STAT_ENTRIES];               // entries the lanes emitted in the last decode
        info->n_steps = st[PJD_STAT_STEPS];                   // ... in this many steps of the write pass
        info->lane_fill_x1024 = (uint32_t)st[PJD_STAT_FILL];  // the fullest lane region
    }
    info->n_huff_workgroups = P.hwgs.size();
    if (b->dev.dbg && b->decoded) {          // PJD_DEBUG_STATS: wave timeline of the last decode (units of 10 ns)
        const size_t nw = P.hwaves.size();
        std::vector<uint32_t> d(nw * 32);
        if (hipMemcpy(d.data(), b->dev.dbg, d.size() * 4, hipMemcpyDeviceToHost) == hipSuccess && nw) {
            if (const char *dump = std::getenv("PJD_DEBUG_DUMP")) {          // the raw timeline (32 words per wave) for offline analysis
                if (FILE *f = std::fopen(dump, "wb")) { std::fwrite(d.data(), 4, d.size(), f); std::fclose(f); }
            }
            uint32_t t0 = d[0];
            for (size_t k = 0; k < nw; k++) if ((int32_t)(d[k * 32] - t0) < 0) t0 = d[k * 32];
            double sum[6] = {0}; uint32_t mx[6] = {0}; size_t worst = 0; uint32_t worst_end = 0;
            for (size_t k = 0; k < nw; k++) {
                const uint32_t *e = &d[k * 32];
                uint32_t end = e[0] - t0;
                for (int q = 1; q <= 5; q++) { sum[q] += e[q]; if (e[q] > mx[q]) mx[q] = e[q]; end += e[q]; }
                sum[0] += e[0] - t0; if (e[0] - t0 > mx[0]) mx[0] = e[0] - t0;
                if (end > worst_end) { worst_end = end; worst = k; }
            }
            const double n = (double)nw;
            {   // shader clock held during the kernel: cycles per 10 ns of every wave (bits 8.. of word 7), median
                std::vector<uint32_t> cl;
                for (size_t k = 0; k < nw; k++) { cl.push_back(d[k * 32 + 7] >> 8); d[k * 32 + 7] &= 0xffu; }
                std::sort(cl.begin(), cl.end());
                std::fprintf(stderr, "[pjd waves] shader clock while the waves ran: median %.2f GHz (min %.2f, max %.2f)\n", cl[nw / 2] / 160.0, cl.front() / 160.0, cl.back() / 160.0);
            }
            std::fprintf(stderr, "[pjd waves] n %zu | mean(us): start %.1f passA %.1f rounds %.1f stitch %.1f scan %.1f write+verify %.1f | max(us): %.1f %.1f %.1f %.1f %.1f %.1f\n",
                         nw, sum[0] / n / 100, sum[1] / n / 100, sum[2] / n / 100, sum[3] / n / 100, sum[4] / n / 100, sum[5] / n / 100,
                         mx[0] / 100.0, mx[1] / 100.0, mx[2] / 100.0, mx[3] / 100.0, mx[4] / 100.0, mx[5] / 100.0);
            const uint32_t *e = &d[worst * 32];
            std::fprintf(stderr, "[pjd waves] last to finish: wave %zu (image %u, %u lanes) start %.1f passA %.1f rounds %.1f stitch %.1f scan %.1f write %.1f -> end %.1f us\n",
                         worst, e[6], e[7], (e[0] - t0) / 100.0, e[1] / 100.0, e[2] / 100.0, e[3] / 100.0, e[4] / 100.0, e[5] / 100.0, worst_end / 100.0);
            // the waves of that image
            for (size_t k = 0; k < nw; k++)
                if (d[k * 32 + 6] == e[6] && d[k * 32 + 7] != 0) {
                    std::fprintf(stderr, "[pjd waves]   wave %zu: start %.1f passA %.1f rounds %.1f stitch %.1f scan %.1f write %.1f\n", k,
                                 (d[k * 32] - t0) / 100.0, d[k * 32 + 1] / 100.0, d[k * 32 + 2] / 100.0, d[k * 32 + 3] / 100.0, d[k * 32 + 4] / 100.0, d[k * 32 + 5] / 100.0);
                    std::fprintf(stderr, "[pjd waves]     rounds (lanes:us):");
                    for (int r = 0; r < 24 && d[k * 32 + 8 + r]; r++) {        // a walk (pjd_k_huffman.hip, walk_lane) is printed as w<lanes walked>
                        const uint32_t v = d[k * 32 + 8 + r];
                        std::fprintf(stderr, (v >> 31) ? " w%u:%.1f" : " %u:%.1f", (v >> 24) & 0x7fu, (v & 0xffffff) / 100.0);
                    }
                    std::fprintf(stderr, "\n");
                }
        }
    }
    return PJD_OK;
}

int pjd_plan_info(const pjd_image_desc *images, int n_images, int out_format, pjd_batch_info *info)
{
    if (!info) return PJD_E_ARG;
    PjdPlan P;
    std::string err;
    int mode = PJD_PLAN_LATENCY;           // as pjd_open: the environment's plan mode
    if (const char *pm = std::getenv("PJD_PLAN_MODE")) mode = (pm[0] == 't' || pm[0] == '1') ? PJD_PLAN_THROUGHPUT : PJD_PLAN_LATENCY;
    int rc = pjd_make_plan(images, n_images, out_format, P, err, 0, mode);
    if (rc != PJD_OK) return rc;
    std::memset(info, 0, sizeof *info);
    info->n_images = (int32_t)P.images.size();
    info->pixels = P.pixels; info->ecs_bytes = P.ecs_bytes; info->out_bytes = P.out_bytes;
    info->coef_bytes = P.n_ent * 2 + P.n_words * 4 + P.dense_du * 128;      // as pjd_batch_get_info: lane streams + transposed words + dense scratch
    info->n_data_units = P.n_du;
    info->n_subsequences = P.subs.size();
    info->n_sequential = (int32_t)P.seq_images.size();
    info->sub_bytes = P.sub_bytes;
    info->plan_mode = (uint32_t)P.plan_mode;
    info->n_table_sets = (uint32_t)P.tsets.size();
    info->huff_lds_bytes = (uint32_t)(P.max_lut_bytes + PJD_HUFF_WAVES * (PJD_WAVE_LDS + PJD_PHASE_LDS) + 16);
    info->n_huff_waves = P.hwaves.size();
    info->n_huff_workgroups = P.hwgs.size();
    return PJD_OK;
}

int pjd_plan_step_bits(const pjd_image_desc *image, uint32_t *step_bits_x256)
{
    if (!image || !step_bits_x256) return PJD_E_ARG;
    PjdPlan P;
    std::string err;
    int rc = pjd_make_plan(image, 1, PJD_OUT_RGB8, P, err);
    if (rc != PJD_OK) return rc;
    if (P.images.empty() || P.tset_step_bits.empty() || (P.images[0].flags & PJD_IF_SEQUENTIAL)) return PJD_E_ARG;   // no lane streams for this picture
    *step_bits_x256 = P.tset_step_bits[P.images[0].tset];
    return PJD_OK;
}

uint64_t pjd_batch_output_size(pjd_batch *b, int image)
{
    if (!b || image < 0 || (size_t)image >= b->plan.host.size()) return 0;
    return b->plan.host[image].out_bytes;
}

void *pjd_batch_device_output(pjd_batch *b, int image)
{
    if (!b || image < 0 || (size_t)image >= b->plan.host.size()) return nullptr;
    return b->dev.out + b->plan.images[image].out_off;
}

void *pjd_batch_device_status(pjd_batch *b) { return b ? (void *)b->dev.status : nullptr; }

uint64_t pjd_coefficients_size(uint32_t width, uint32_t height, uint8_t h_samp, uint8_t v_samp)
{
    // reference src/jpeg_scanner.cpp:257-262 (mcu_*_real) and src/decoder_host.cpp:125-128 (DPUs needed, 100 positions each)
    uint64_t w = (width + 7) / 8, h = (height + 7) / 8;
    if (h_samp == 2 && (w & 1)) w++;
    if (v_samp == 2 && (h & 1)) h++;
    const uint64_t pw = (w + 1) / 2 * 2, ph = (h + 1) / 2 * 2;
    return (pw * ph + 99) / 100 * 19200;
}

int pjd_batch_download_coefficients(pjd_batch *b, int image, int16_t *out, uint64_t capacity_int16)
{
    if (!b || !out || image < 0 || (size_t)image >= b->plan.images.size()) return PJD_E_ARG;
    if (!b->decoded) { b->ctx->err = "download_coefficients before decode"; return PJD_E_STATE; }
    pjd_ctx *ctx = b->ctx;
    PjdPlan &P = b->plan;
    hipSetDevice(ctx->device);
    int rc = settle(b);
    if (rc != PJD_OK) return rc;
    const PjdDevImage &g = P.images[image];
    const uint64_t n16 = pjd_coefficients_size(g.width, g.height, (uint8_t)g.hs, (uint8_t)g.vs);
    if (capacity_int16 < n16) { ctx->err = "download_coefficients: buffer smaller than pjd_coefficients_size"; return PJD_E_ARG; }
    hipStream_t s = ctx->stream;
    int16_t *d_out = nullptr, *scratch = nullptr;
    uint32_t *d_list = nullptr; uint64_t *d_base = nullptr;
    auto cleanup = [&] { hipFree(d_out); hipFree(scratch); hipFree(d_list); hipFree(d_base); };
    if (hipMalloc((void **)&d_out, n16 * sizeof(int16_t)) != hipSuccess) { ctx->err = "hipMalloc failed (coefficients)"; return PJD_E_NOMEM; }
    pjd_launch_zero(s, d_out, n16 * sizeof(int16_t));                  // 19200 int16 per DPU: a multiple of 16 bytes
    const uint32_t n_du = (g.last_mcu - g.first_mcu) * g.dus_per_mcu, first_du = g.first_mcu * g.dus_per_mcu;
    const bool routed = P.host[image].sequential;
    const bool fell_back = !routed && (b->h_status[image] & PJD_STW_NEEDS_EXACT);
    hipError_t e = hipSuccess;
    if (routed) {
        pjd_launch_coefdump_dense(s, b->dev, (uint32_t)image, b->dev.coef + g.dense_base * 64, first_du, n_du, d_out);
    } else if (fell_back) {
        // the scratch settle() used is gone: run the exact kernel for this one image again (its status word does not change)
        const uint32_t one = (uint32_t)image; const uint64_t zero = 0;
        if (hipMalloc((void **)&scratch, (size_t)n_du * 64 * sizeof(int16_t)) != hipSuccess || hipMalloc((void **)&d_list, sizeof one) != hipSuccess ||
            hipMalloc((void **)&d_base, sizeof zero) != hipSuccess) { cleanup(); ctx->err = "hipMalloc failed (coefficients scratch)"; return PJD_E_NOMEM; }
        pjd_launch_zero(s, scratch, (size_t)n_du * 64 * sizeof(int16_t));
        e = hipMemcpyAsync(d_list, &one, sizeof one, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_base, &zero, sizeof zero, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            PjdDevBatch dv = b->dev;
            dv.coef = scratch;
            pjd_launch_huff_sequential(s, dv, d_list, d_base, 1);
            pjd_launch_coefdump_dense(s, dv, (uint32_t)image, scratch, first_du, n_du, d_out);
        }
    } else {
        pjd_launch_coefdump_lanes(s, b->dev, (uint32_t)image, g.n_iwg, d_out);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n16 * sizeof(int16_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    cleanup();
    if (e != hipSuccess) { ctx->err = std::string("download_coefficients: ") + hipGetErrorString(e); return PJD_E_HIP; }
    return PJD_OK;
}

int pjd_decode_batch(pjd_ctx *ctx, const pjd_image_desc *images, int n_images, int out_format,
                     uint8_t *const *out, int32_t *status)
{
    pjd_batch *b = nullptr;
    int rc = pjd_batch_create(ctx, images, n_images, out_format, &b);
    if (rc != PJD_OK) return rc;
    rc = pjd_batch_upload(b);
    if (rc == PJD_OK) rc = pjd_batch_decode(b);
    if (rc == PJD_OK) rc = pjd_batch_download(b, out, status);
    pjd_batch_destroy(b);
    return rc;
}

int pjd_exec_dpu_payload(pjd_ctx *ctx, const uint32_t *metadata, int16_t *mcus, int n_dpus)
{
    if (!ctx || !metadata || !mcus || n_dpus < 0) return PJD_E_ARG;
    if (n_dpus == 0) return PJD_OK;
    for (int d = 0; d < n_dpus; d++) {
        const uint32_t *m = metadata + (size_t)d * 276;
        const uint32_t V = m[5] & 255, H = m[6] & 255;
        if (m[19] != 100 || m[4] > 3 || (V != 1 && V != 2) || (H != 1 && H != 2)) { ctx->err = "DPU metadata outside the supported envelope"; return PJD_E_ARG; }
        for (uint32_t c = 0; c < m[4]; c++) if ((m[7 + c] & 255) > 3) { ctx->err = "DPU metadata: quantisation table id > 3"; return PJD_E_ARG; }
    }
    hipSetDevice(ctx->device);
    uint32_t *dm = nullptr; int16_t *dc = nullptr;
    const size_t mb = (size_t)n_dpus * 276 * 4, cb = (size_t)n_dpus * 19200 * 2;
    HIP_TRY(ctx, hipMalloc((void **)&dm, mb));
    if (hipMalloc((void **)&dc, cb) != hipSuccess) { hipFree(dm); ctx->err = "hipMalloc(mcus)"; return PJD_E_NOMEM; }
    int rc = PJD_OK;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == PJD_OK) { ctx->err = std::string(what) + ": " + hipGetErrorString(e); rc = PJD_E_HIP; } };
    chk(hipMemcpyAsync(dm, metadata, mb, hipMemcpyHostToDevice, ctx->stream), "copy(metadata_buffer)");
    chk(hipMemcpyAsync(dc, mcus, cb, hipMemcpyHostToDevice, ctx->stream), "copy(mcus)");
    if (rc == PJD_OK) { pjd_launch_dpu_payload(ctx->stream, dm, dc, n_dpus); chk(hipGetLastError(), "exec"); }
    chk(hipMemcpyAsync(mcus, dc, cb, hipMemcpyDeviceToHost, ctx->stream), "copy back");
    chk(hipStreamSynchronize(ctx->stream), "sync");
    hipFree(dm); hipFree(dc);
    return rc;
}

}  // extern "C"
