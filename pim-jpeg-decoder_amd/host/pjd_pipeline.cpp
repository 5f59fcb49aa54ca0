// pjd_pipeline.cpp -- the pipelined batcher behind include/pjd_pipeline.h (libpjdpipe.so).
//
// Stands where the reference has its producer thread (mcu_prepare, reference src/decoder_host.cpp:104-211),
// its consumer thread (offloading, :213-350) and the std::queue<Batch> between them (:25-38):
//
//   scan workers --(batch complete)--> ready queue of the batch's device --> that device's GPU slots
//                --(pictures)--> sink queue --> sink workers
//
// A batch is `batch_images` CONSECUTIVE inputs, so its composition does not depend on thread timing.
// Every GPU slot owns a pjd_ctx, i.e. its own HIP stream and its own buffer pool: while one slot's
// kernels run, another slot's bitstreams go up and a third slot's pictures come down.  Pictures leave
// the GPU in one packed copy into page-locked memory (pjd_batch_download_packed).
//
// With several devices (the reference spreads pictures over all its DPUs, src/decoder_host.cpp:225,262-300) each
// device has its own slots and its own ready queue; batches are dealt to devices up front by pjd_pipe_assign
// (longest first onto the least loaded device, by input bytes), and a device whose queue has run dry takes
// batches from the others once scanning is over.
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pjd_host.h"
#include "../../include/pjd_pipeline.h"

namespace {

double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

struct Input {
    const char *path = nullptr;          // file input
    const uint8_t *data = nullptr;       // memory input
    uint64_t len = 0;
    const char *name = "";
    pjd_scanned *sc = nullptr;
    int scan_rc = 2;
};

struct Job {                             // one batch: inputs [first, first + count)
    int first = 0, count = 0;
    int dev = 0;                         // index into Pipe::devs the batch was dealt to
    uint64_t cost = 0;                   // input bytes
    std::atomic<int> scanned{0};
};

struct SinkTask {
    int index;
    int status;
    const uint8_t *data;
    uint64_t len;
    std::atomic<int> *latch;             // counts the tasks of one batch still running (may be null)
};

template <class T>
struct Queue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> l(m); q.push_back(std::move(v)); } cv.notify_one(); }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
    bool try_pop(T &out)
    {
        std::lock_guard<std::mutex> l(m);
        if (q.empty()) return false;
        out = std::move(q.front());
        q.pop_front();
        return true;
    }
    bool pop(T &out)
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        out = std::move(q.front());
        q.pop_front();
        return true;
    }
};

// What a GPU slot keeps between runs: its context (stream + buffer pools) and its page-locked output buffer.
// Page-locking ~0.7 GB costs more than decoding a whole batch, so slots are parked here when a run ends and
// picked up by the next run on the same device; pjd_pipe_release() frees them.
struct SlotRes {
    int device = -1;
    pjd_ctx *ctx = nullptr;
    uint8_t *pinned = nullptr;
    uint64_t pinned_cap = 0;
};
std::mutex g_slot_m;
std::vector<SlotRes> g_parked;

SlotRes take_slot(int device)
{
    {
        std::lock_guard<std::mutex> l(g_slot_m);
        for (size_t k = 0; k < g_parked.size(); k++)
            if (g_parked[k].device == device) {
                SlotRes r = g_parked[k];
                g_parked.erase(g_parked.begin() + k);
                return r;
            }
    }
    SlotRes r;
    r.device = device;
    if (pjd_open(device, &r.ctx) != PJD_OK) r.ctx = nullptr;
    return r;
}

void park_slot(const SlotRes &r)
{
    if (!r.ctx) { pjd_host_free(r.pinned); return; }
    std::lock_guard<std::mutex> l(g_slot_m);
    g_parked.push_back(r);
}

struct Pipe {
    pjd_pipe_opts o;
    std::vector<Input> in;
    std::vector<std::unique_ptr<Job>> jobs;
    std::atomic<int> next_input{0};
    std::vector<int> devs;               // HIP ordinals in use (after run() opened them: the ones that opened)
    std::vector<int> entry_of;           // per device in use: its index in the caller's `devices` list (statistics are reported per entry)
    std::vector<std::unique_ptr<Queue<int>>> ready;   // per device: job indices whose inputs are all scanned
    Queue<SinkTask> sinkq;
    std::mutex stat_m, latch_m;
    std::condition_variable latch_cv;
    pjd_pipe_stats st{};
    std::atomic<int> slots_open{0};
    std::mutex ahead_m;
    std::condition_variable ahead_cv;
    int finished_jobs = 0, max_ahead = 8;   // batches scanned (or being scanned) beyond the ones already done with
    bool trace = std::getenv("PJD_PIPE_TRACE") != nullptr;
    double t_run0 = 0;

    void add(double pjd_pipe_stats::*f, double v) { std::lock_guard<std::mutex> l(stat_m); st.*f += v; }

    void scan_worker()
    {
        double t_scan = 0;
        uint64_t in_bytes = 0;
        for (;;) {
            const int i = next_input.fetch_add(1);
            if (i >= (int)in.size()) break;
            {   // bounded look-ahead: scanned inputs (each holds a copy of its bitstream) wait for the GPU, not the other way round
                std::unique_lock<std::mutex> l(ahead_m);
                ahead_cv.wait(l, [&] { return i / o.batch_images < finished_jobs + max_ahead; });
            }
            Input &x = in[i];
            const double t0 = now_s();
            if (x.path) {
                struct stat sb;
                if (stat(x.path, &sb) == 0) in_bytes += (uint64_t)sb.st_size;
                x.scan_rc = pjd_scan_file_ex(x.path, o.scan_options, &x.sc);
            } else {
                in_bytes += x.len;
                x.scan_rc = pjd_scan_memory_ex(x.data, x.len, x.name, o.scan_options, &x.sc);
            }
            t_scan += now_s() - t0;
            Job &j = *jobs[i / o.batch_images];
            if (j.scanned.fetch_add(1) + 1 == j.count) ready[j.dev]->push(i / o.batch_images);
        }
        std::lock_guard<std::mutex> l(stat_m);
        st.scan_s += t_scan;
        st.in_bytes += in_bytes;
    }

    void emit(int index, int status, const uint8_t *data, uint64_t len, std::atomic<int> *latch)
    {
        if (!o.sink) return;
        if (latch) latch->fetch_add(1);
        sinkq.push(SinkTask{index, status, data, len, latch});
    }

    // next batch for a slot of device `d`: its own queue first; when that is closed and empty, the others'
    bool next_job(int d, int &j, bool &stolen)
    {
        stolen = false;
        if (ready[d]->pop(j)) return true;
        for (size_t k = 1; k < ready.size(); k++)
            if (ready[(d + k) % ready.size()]->try_pop(j)) { stolen = true; return true; }
        return false;
    }

    void slot_worker(int d, SlotRes res)
    {
        pjd_ctx *ctx = res.ctx;
        uint8_t *&pinned = res.pinned;
        uint64_t &pinned_cap = res.pinned_cap;
        std::atomic<int> latch{0};
        int j;
        bool stolen;
        while (next_job(d, j, stolen)) {
            Job &job = *jobs[j];
            const double t_begin = now_s();
            std::vector<int> idx;                          // inputs of this batch the scanner accepted
            std::vector<pjd_image_desc> descs;
            uint64_t rejected = 0;
            for (int i = job.first; i < job.first + job.count; i++) {
                Input &x = in[i];
                if (x.scan_rc == 0 && x.sc) { idx.push_back(i); descs.push_back(*pjd_scanned_desc(x.sc)); }
                else { rejected++; emit(i, -1, nullptr, 0, &latch); }
            }
            double t_create = 0, t_up = 0, t_exec = 0, t_down = 0;
            uint64_t pixels = 0, ecs = 0, outb = 0, decoded = 0, exact = 0;
            bool failed = false;
            if (!idx.empty()) {
                pjd_batch *b = nullptr;
                int rc = ctx ? PJD_OK : PJD_E_NODEVICE;
                double t0 = now_s();
                if (rc == PJD_OK) rc = pjd_batch_create(ctx, descs.data(), (int)descs.size(), o.out_format, &b);
                t_create = now_s() - t0; t0 = now_s();
                if (rc == PJD_OK) rc = pjd_batch_upload(b);
                t_up = now_s() - t0; t0 = now_s();
                if (rc == PJD_OK) rc = pjd_batch_decode(b);
                if (rc == PJD_OK) rc = pjd_batch_sync(b);
                t_exec = now_s() - t0; t0 = now_s();
                std::vector<int32_t> status(idx.size(), 0);
                if (rc == PJD_OK) {
                    const uint64_t need = pjd_batch_packed_size(b);
                    if (need > pinned_cap) {
                        pjd_host_free(pinned);
                        pinned_cap = need + need / 4;
                        pinned = (uint8_t *)pjd_host_alloc(pinned_cap);
                        if (!pinned) { pinned_cap = 0; rc = PJD_E_NOMEM; }
                    }
                    if (rc == PJD_OK) rc = pjd_batch_download_packed(b, pinned, pinned_cap, status.data());
                }
                t_down = now_s() - t0;
                if (rc == PJD_OK) {
                    pjd_batch_info info;
                    if (pjd_batch_get_info(b, &info) == PJD_OK) { pixels = info.pixels; ecs = info.ecs_bytes; outb = info.out_bytes; exact = (uint64_t)(info.n_sequential + info.n_fallback); }
                    for (size_t k = 0; k < idx.size(); k++)
                        emit(idx[k], status[k], pinned + pjd_batch_output_offset(b, (int)k), pjd_batch_output_size(b, (int)k), &latch);
                    decoded = idx.size();
                } else {
                    failed = true;
                    for (size_t k = 0; k < idx.size(); k++) emit(idx[k], -2, nullptr, 0, &latch);
                }
                if (b) pjd_batch_destroy(b);
            }
            // the pinned buffer and the scan logs are referenced by the sink tasks of this batch
            const double t_sink0 = now_s();
            {
                std::unique_lock<std::mutex> l(latch_m);
                latch_cv.wait(l, [&] { return latch.load() == 0; });
            }
            if (trace)                                         // PJD_PIPE_TRACE=1: one line per batch, times in ms since the run began
                std::fprintf(stderr, "[pjdpipe] batch %d dev %d start %.2f create %.2f upload %.2f exec %.2f download %.2f other %.2f sink-wait %.2f end %.2f%s\n",
                             j, devs[d], (t_begin - t_run0) * 1e3, t_create * 1e3, t_up * 1e3, t_exec * 1e3, t_down * 1e3,
                             (t_sink0 - t_begin - t_create - t_up - t_exec - t_down) * 1e3, (now_s() - t_sink0) * 1e3, (now_s() - t_run0) * 1e3,
                             stolen ? " (stolen)" : "");
            for (int i = job.first; i < job.first + job.count; i++)
                if (in[i].sc) { pjd_scanned_free(in[i].sc); in[i].sc = nullptr; }
            { std::lock_guard<std::mutex> l(ahead_m); finished_jobs++; }
            ahead_cv.notify_all();
            std::lock_guard<std::mutex> l(stat_m);
            st.create_s += t_create; st.upload_s += t_up; st.exec_s += t_exec; st.download_s += t_down;
            st.n_batches++; st.n_batch_failures += failed ? 1 : 0;
            st.n_decoded += decoded; st.n_rejected += rejected;
            st.pixels += pixels; st.ecs_bytes += ecs; st.out_bytes += outb;
            st.n_exact_images += exact;
            st.device_batches[entry_of[d]]++; st.device_in_bytes[entry_of[d]] += job.cost; st.n_stolen += stolen ? 1 : 0;   // per entry of opts.devices
        }
        park_slot(res);
        if (trace) std::fprintf(stderr, "[pjdpipe] slot of dev %d leaves %.2f\n", devs[d], (now_s() - t_run0) * 1e3);
    }

    void sink_worker()
    {
        double t_sink = 0;
        SinkTask t;
        while (sinkq.pop(t)) {
            const Input &x = in[t.index];
            const double t0 = now_s();
            const char *log = x.sc ? pjd_scanned_log(x.sc) : "";
            std::string opened;
            if (x.scan_rc == 2) {                          // the reference's message for an unreadable file (jpeg_scanner.cpp:351)
                opened = std::string(x.name) + ": Error - Error opening input file\n" + x.name + ": Error - Invalid JPEG\n";
                log = opened.c_str();
            }
            o.sink(o.sink_user, t.index, x.name, log, t.status, t.data, t.len);
            t_sink += now_s() - t0;
            if (t.latch && t.latch->fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> l(latch_m);
                latch_cv.notify_all();
            }
        }
        add(&pjd_pipe_stats::sink_s, t_sink);
    }

    int run()
    {
        const double t0 = now_s();
        t_run0 = t0;
        const int n = (int)in.size();
        const int nb = (n + o.batch_images - 1) / o.batch_images;
        for (int k = 0; k < nb; k++) {
            jobs.emplace_back(new Job);
            jobs[k]->first = k * o.batch_images;
            jobs[k]->count = (k + 1) * o.batch_images <= n ? o.batch_images : n - k * o.batch_images;
        }
        st.n_inputs = (uint64_t)n;
        // open the slots first: a device that does not open is left out of the deal
        std::vector<std::vector<SlotRes>> res;
        {
            std::vector<int> live;
            for (size_t di = 0; di < devs.size(); di++) {
                const int dev = devs[di];
                std::vector<SlotRes> r;
                for (int k = 0; k < o.slots; k++) {
                    SlotRes s = take_slot(dev);
                    if (!s.ctx) { park_slot(s); break; }
                    // several slots per device decode at once: plan for pictures per second; one slot decodes its batches alone
                    pjd_set_plan_mode(s.ctx, o.slots > 1 ? PJD_PLAN_THROUGHPUT : PJD_PLAN_LATENCY);
                    r.push_back(s);
                }
                if (!r.empty()) { live.push_back(dev); entry_of.push_back((int)di); res.push_back(std::move(r)); }
            }
            if (live.empty()) {                            // no device: one slot without a context reports every batch as failed
                SlotRes none;
                none.device = devs[0];
                res.push_back(std::vector<SlotRes>(1, none));
                live.push_back(devs[0]);
                entry_of.push_back(0);
            } else {
                for (const std::vector<SlotRes> &r : res) slots_open.fetch_add((int)r.size());
                st.n_devices = live.size();
            }
            devs = live;
        }
        for (size_t d = 0; d < devs.size(); d++) ready.emplace_back(new Queue<int>);
        // deal the batches: cost = input bytes (stat for files; a file that cannot be read costs nothing)
        {
            std::vector<uint64_t> cost((size_t)nb, 0);
            for (int k = 0; k < nb; k++) {
                for (int i = jobs[k]->first; i < jobs[k]->first + jobs[k]->count; i++) {
                    struct stat sb;
                    if (in[i].path) cost[k] += stat(in[i].path, &sb) == 0 ? (uint64_t)sb.st_size : 0;
                    else cost[k] += in[i].len;
                }
                jobs[k]->cost = cost[k];
            }
            std::vector<int32_t> dev_of((size_t)nb, 0);
            if (nb > 0) pjd_pipe_assign(cost.data(), nb, (int)devs.size(), dev_of.data());
            for (int k = 0; k < nb; k++) jobs[k]->dev = dev_of[k];
        }
        {
            size_t total_slots = 0;
            for (const std::vector<SlotRes> &r : res) total_slots += r.size();
            max_ahead = (int)(total_slots + 3);
        }
        std::vector<std::thread> scanners, slots, sinks;
        for (int k = 0; k < o.sink_threads && o.sink; k++) sinks.emplace_back([this] { sink_worker(); });
        for (size_t d = 0; d < devs.size(); d++)
            for (const SlotRes &r : res[d]) slots.emplace_back([this, d, r] { slot_worker((int)d, r); });
        for (int k = 0; k < o.scan_threads; k++) scanners.emplace_back([this] { scan_worker(); });
        for (std::thread &t : scanners) t.join();
        if (trace) std::fprintf(stderr, "[pjdpipe] scanners done %.2f\n", (now_s() - t0) * 1e3);
        for (auto &q : ready) q->close();                  // every job has been pushed by now
        for (std::thread &t : slots) t.join();
        if (trace) std::fprintf(stderr, "[pjdpipe] slots joined %.2f\n", (now_s() - t0) * 1e3);
        sinkq.close();
        for (std::thread &t : sinks) t.join();
        st.wall_s = now_s() - t0;
        if (trace) std::fprintf(stderr, "[pjdpipe] run done %.2f\n", st.wall_s * 1e3);
        return slots_open.load() > 0 || n == 0 ? PJD_OK : PJD_E_NODEVICE;
    }
};

int run_pipe(Pipe &p, const pjd_pipe_opts *opts, pjd_pipe_stats *stats)
{
    if (!opts) return PJD_E_ARG;
    p.o = *opts;
    if (p.o.batch_images <= 0) p.o.batch_images = 1024;
    if (p.o.scan_threads <= 0) p.o.scan_threads = 4;
    if (p.o.slots <= 0) p.o.slots = 3;
    if (p.o.sink_threads <= 0) p.o.sink_threads = 4;
    if (p.o.out_format != PJD_OUT_BMP && p.o.out_format != PJD_OUT_RGB8) return PJD_E_ARG;
    if (p.o.devices && p.o.n_devices > 0) {
        if (p.o.n_devices > PJD_PIPE_MAX_DEVICES) return PJD_E_ARG;
        // PJD_PIPE_ALLOW_DUP_DEVICES=1 (tests on a one-GPU box): the same ordinal may be listed several times and then counts as
        // several devices, so that dealing, per-device queues and stealing run for real
        const bool dup_ok = std::getenv("PJD_PIPE_ALLOW_DUP_DEVICES") != nullptr;
        for (int k = 0; k < p.o.n_devices; k++) {
            const int d = p.o.devices[k];
            if (d < 0 || (!dup_ok && std::find(p.devs.begin(), p.devs.end(), d) != p.devs.end())) return PJD_E_ARG;   // negative / listed twice
            p.devs.push_back(d);
        }
    } else {
        p.devs.push_back(p.o.device);
    }
    const int rc = p.run();
    if (stats) *stats = p.st;
    return rc;
}

}  // namespace

extern "C" {

int pjd_pipe_assign(const uint64_t *cost, int n, int n_devices, int32_t *device_of)
{
    if (n < 0 || n_devices <= 0 || (n > 0 && (!cost || !device_of))) return PJD_E_ARG;
    std::vector<int> order((size_t)n);
    for (int k = 0; k < n; k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::vector<uint64_t> load((size_t)n_devices, 0);
    for (int k : order) {
        const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());   // first of the least loaded
        device_of[k] = d;
        load[d] += cost[k] ? cost[k] : 1;                  // empty batches still take a turn
    }
    return PJD_OK;
}

void pjd_pipe_release(void)
{
    std::lock_guard<std::mutex> l(g_slot_m);
    for (SlotRes &r : g_parked) { pjd_host_free(r.pinned); if (r.ctx) pjd_close(r.ctx); }
    g_parked.clear();
}

int pjd_pipe_run_files(const char *const *paths, int n, const pjd_pipe_opts *opts, pjd_pipe_stats *stats)
{
    if (n < 0 || (n > 0 && !paths)) return PJD_E_ARG;
    Pipe p;
    p.in.resize((size_t)n);
    for (int i = 0; i < n; i++) { p.in[i].path = paths[i]; p.in[i].name = paths[i]; }
    return run_pipe(p, opts, stats);
}

int pjd_pipe_run_memory(const uint8_t *const *data, const uint64_t *len, const char *const *names, int n,
                        const pjd_pipe_opts *opts, pjd_pipe_stats *stats)
{
    if (n < 0 || (n > 0 && (!data || !len))) return PJD_E_ARG;
    Pipe p;
    p.in.resize((size_t)n);
    for (int i = 0; i < n; i++) { p.in[i].data = data[i]; p.in[i].len = len[i]; p.in[i].name = names && names[i] ? names[i] : ""; }
    return run_pipe(p, opts, stats);
}

}  // extern "C"
