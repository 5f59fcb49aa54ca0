// decoder_main.cpp -- `bin/decoder <jpeg>...`: the reference's CLI on the MI355X path.
//
// Same contract as the reference's main() (reference src/decoder_host.cpp:352-451):
//   * positional JPEG paths; inputs are processed in ascending file-size order (:360);
//   * "<stem>.bmp" is written next to every decodable input (:328-330);
//   * parse errors go to stdout as "<file>: Error - ..." followed by "<file>: Error - Invalid JPEG";
//   * Huffman errors are printed and the (partial) picture is still written (:181);
//   * exit code 0, or 1 with "Error - Invalid arguments" when no file is given (:353-356);
//   * a "Profiles:" block with the same rows (:379-394).
// What differs by construction: "<n> dpus are allocated" becomes a line about the GPU, the
// producer/consumer pair is a scan stage followed by batched GPU decodes, and the four DPU cycle
// counters become per-kernel milliseconds measured with HIP events.
//
// Extensions (do not change the default behaviour): --device N, --batch M (images per GPU batch),
// --pipeline [--slots S --scan-threads T --write-threads W --devices 0,1,...]: the pipelined batcher of
// include/pjd_pipeline.h (scan, copies, kernels and BMP writes overlap; messages are printed in input
// order at the end instead of interleaved).  --devices spreads the batches over several GPUs of the node,
// as the reference spreads pictures over all allocated DPUs (:225); it implies --pipeline.
// --split [--devices 0,1,...]: every picture is decoded by all listed devices TOGETHER (pjd_split_decode): restart-segment ranges per
// device, the descriptor broadcast with RCCL, rows assembled on the host -- for single pictures larger than one device's share.
// --progressive: progressive (SOF2) files are decoded scan by scan instead of being rejected as the reference rejects them
// (not reference behaviour, parity unpinned).
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/pjd.h"
#include "../../include/pjd_host.h"
#include "../../include/pjd_pipeline.h"

static double now_s()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static std::string bmp_name(const std::string &in)
{
    const size_t pos = in.find_last_of('.');
    return pos == std::string::npos ? in + ".bmp" : in.substr(0, pos) + ".bmp";
}

// ---- --pipeline mode ---------------------------------------------------------------------------
struct PipeOut {
    std::vector<std::string> messages;       // per input, printed in input order afterwards
};

static void pipe_sink(void *user, int index, const char *name, const char *log, int status, const uint8_t *data, uint64_t len)
{
    PipeOut *po = (PipeOut *)user;
    std::string &m = po->messages[(size_t)index];    // one call per index: no lock needed
    m = log;
    if (status == -2) m += std::string(name) + ": Error - GPU batch failed\n";
    if (status > 0) m += std::string(name) + ": " + pjd_status_string(status) + "\n";
    if (data) {
        const std::string out = bmp_name(name);
        if (pjd_write_file(out.c_str(), data, len) != 0) m += out + ": Error - Unable to create BMP file\n";
    }
}

static int run_pipeline(const std::vector<std::string> &files, const std::vector<int32_t> &devices, int batch_images, int slots, int scan_threads,
                        int write_threads, uint32_t scan_options)
{
    std::vector<const char *> paths;
    for (const std::string &f : files) paths.push_back(f.c_str());
    PipeOut po;
    po.messages.resize(files.size());
    pjd_pipe_opts o;
    std::memset(&o, 0, sizeof o);
    o.device = devices[0]; o.devices = devices.data(); o.n_devices = (int32_t)devices.size();
    o.out_format = PJD_OUT_BMP; o.batch_images = batch_images;
    o.slots = slots; o.scan_threads = scan_threads; o.sink_threads = write_threads;
    o.sink = pipe_sink; o.sink_user = &po;
    o.scan_options = scan_options;
    pjd_pipe_stats st;
    const int rc = pjd_pipe_run_files(paths.data(), (int)paths.size(), &o, &st);
    if (rc == PJD_E_NODEVICE) {
        std::cout << "Error - no usable MI355X (gfx950) device (pjd_open returned " << rc << ")\n";
        return 2;
    }
    if (rc != PJD_OK) {
        std::cout << "Error - Invalid arguments\n";
        return 1;
    }
    if (st.n_devices == 1) std::cout << "1 MI355X device is allocated\n";
    else std::cout << st.n_devices << " MI355X devices are allocated\n";
    for (const std::string &m : po.messages) std::cout << m;
    std::cout << "\nProfiles:\n";
    std::cout << "End-to-end execution time: " << st.wall_s << "s\n";
    std::cout << "MCU Offloader execution time (summed over overlapping workers): \n";
    std::cout << " - JPEG scan time: " << st.scan_s << "s\n";
    std::cout << " - batch planning time: " << st.create_s << "s\n";
    std::cout << " - CPU-to-GPU transfer time: " << st.upload_s << "s\n";
    std::cout << " - GPU execution time: " << st.exec_s << "s\n";
    std::cout << " - GPU-to-CPU transfer time: " << st.download_s << "s\n";
    std::cout << " - BMP write time: " << st.sink_s << "s\n";
    std::cout << " - Total " << st.n_batches << " calls, " << st.n_decoded << " pictures, " << st.pixels / 1e6 << " MPixels\n";
    for (size_t d = 0; d < devices.size() && devices.size() > 1; d++)
        std::cout << " - device " << devices[d] << ": " << st.device_batches[d] << " calls, " << st.device_in_bytes[d] / 1e6 << " MB of input\n";
    return 0;
}

// ---- --split mode: every picture is decoded by ALL listed devices together (pjd_split_decode) -------------------------------
// For pictures larger than one device's share: the reference spreads every picture over all its DPUs (decoder_host.cpp:125-149,
// 225); here a picture with restart intervals is cut into restart-segment ranges, one per device, the descriptor is broadcast
// with RCCL and the rows are assembled on the host.  Pictures that cannot be cut are decoded by the first device.
static int run_split(const std::vector<std::string> &files, const std::vector<int32_t> &devices, uint32_t scan_options)
{
    double t_total = now_s(), t_scan = 0, t_bc = 0, t_up = 0, t_exec = 0, t_down = 0, t_bmp = 0;
    int calls = 0, rccl_calls = 0;
    bool announced = false;
    for (const std::string &f : files) {
        double t0 = now_s();
        pjd_scanned *s = nullptr;
        const int sr = pjd_scan_file_ex(f.c_str(), scan_options, &s);
        t_scan += now_s() - t0;
        if (sr == 2) { std::cout << f << ": Error - Error opening input file\n" << f << ": Error - Invalid JPEG\n"; continue; }
        std::cout << pjd_scanned_log(s);
        if (sr != 0) { pjd_scanned_free(s); continue; }
        const pjd_image_desc *d = pjd_scanned_desc(s);
        std::vector<uint8_t> out(pjd_output_size(d->width, d->height, PJD_OUT_BMP));
        int32_t status = 0;
        pjd_split_stats st;
        const int rc = pjd_split_decode(d, devices.data(), (int)devices.size(), PJD_OUT_BMP, out.data(), out.size(), &status, &st);
        if (rc == PJD_E_NODEVICE) { std::cout << "Error - no usable MI355X (gfx950) device (pjd_open returned " << rc << ")\n"; pjd_scanned_free(s); return 2; }
        if (rc == PJD_E_ARG && !announced) { std::cout << "Error - Invalid arguments\n"; pjd_scanned_free(s); return 1; }
        if (!announced) { std::cout << devices.size() << " MI355X devices are allocated (one picture is split over " << devices.size() << " devices)\n"; announced = true; }
        if (rc != PJD_OK) { std::cout << f << ": Error - GPU decode failed (" << rc << ")\n"; pjd_scanned_free(s); continue; }
        calls++; rccl_calls += st.rccl_used;
        t_bc += st.broadcast_s; t_up += st.upload_s; t_exec += st.exec_s; t_down += st.download_s;
        if (status != PJD_ST_OK) std::cout << f << ": " << pjd_status_string(status) << "\n";
        std::cout << f << ": " << st.n_segments << " restart segments, split over " << st.n_ranks << " devices"
                  << (st.rccl_used ? ", descriptor broadcast by RCCL" : "") << (st.redone_whole ? ", decoded again on one device" : "") << "\n";
        t0 = now_s();
        const std::string o = bmp_name(f);
        if (pjd_write_file(o.c_str(), out.data(), out.size()) != 0) std::cout << o << ": Error - Unable to create BMP file" << std::endl;
        t_bmp += now_s() - t0;
        pjd_scanned_free(s);
    }
    pjd_split_release();
    t_total = now_s() - t_total;
    std::cout << "\nProfiles:\n";
    std::cout << "End-to-end execution time: " << t_total << "s\n";
    std::cout << "MCU Offloader execution time (total; per picture the slowest device counts): \n";
    std::cout << " - JPEG scan time: " << t_scan << "s\n";
    std::cout << " - descriptor broadcast time: " << t_bc << "s (" << rccl_calls << " by RCCL)\n";
    std::cout << " - CPU-to-GPU transfer time: " << t_up << "s\n";
    std::cout << " - GPU execution time: " << t_exec << "s\n";
    std::cout << " - GPU-to-CPU transfer time: " << t_down << "s\n";
    std::cout << " - BMP write time: " << t_bmp << "s\n";
    std::cout << " - Total " << calls << " calls\n";
    return 0;
}

int main(int argc, char **argv)
{
    int device = 0;
    size_t batch_images = 1024;
    bool pipeline = false, split = false;
    uint32_t scan_options = 0;
    int slots = 0, scan_threads = 0, write_threads = 0;
    std::vector<std::string> files;
    std::vector<int32_t> devices;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char *p = argv[++i]; *p;) {
                char *end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) { devices.clear(); devices.push_back(-1); break; }     // not a number: rejected below
                devices.push_back((int32_t)v);
                p = *end == ',' ? end + 1 : end;
            }
            pipeline = true;
        }
        else if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch_images = (size_t)std::atoll(argv[++i]);
        else if (!std::strcmp(argv[i], "--pipeline")) pipeline = true;
        else if (!std::strcmp(argv[i], "--split")) split = true;
        else if (!std::strcmp(argv[i], "--progressive")) scan_options |= PJD_SCAN_PROGRESSIVE;
        else if (!std::strcmp(argv[i], "--slots") && i + 1 < argc) slots = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--scan-threads") && i + 1 < argc) scan_threads = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--write-threads") && i + 1 < argc) write_threads = std::atoi(argv[++i]);
        else files.push_back(argv[i]);
    }
    if (files.empty()) {
        std::cout << "Error - Invalid arguments\n";
        return 1;
    }
    // ascending file size, like sort_by_size (decoder_host.cpp:46-61)
    std::vector<std::pair<long long, std::string>> sized;
    for (const std::string &f : files) {
        struct stat st;
        sized.emplace_back(stat(f.c_str(), &st) == 0 ? (long long)st.st_size : 0LL, f);
    }
    std::stable_sort(sized.begin(), sized.end(), [](const std::pair<long long, std::string> &a, const std::pair<long long, std::string> &b) { return a.first < b.first; });
    if (split) {
        std::vector<std::string> ordered;
        for (const auto &p : sized) ordered.push_back(p.second);
        if (devices.empty()) devices.push_back(device);
        for (int32_t d : devices) if (d < 0) { std::cout << "Error - Invalid arguments\n"; return 1; }
        return run_split(ordered, devices, scan_options);
    }
    if (pipeline) {
        std::vector<std::string> ordered;
        for (const auto &p : sized) ordered.push_back(p.second);
        if (devices.empty()) devices.push_back(device);
        return run_pipeline(ordered, devices, (int)batch_images, slots, scan_threads, write_threads, scan_options);
    }

    pjd_ctx *ctx = nullptr;
    int rc = pjd_open(device, &ctx);
    if (rc != PJD_OK) {
        std::cout << "Error - no usable MI355X (gfx950) device (pjd_open returned " << rc << ")\n";
        return 2;
    }
    std::cout << "1 MI355X device is allocated\n";

    double t_total = now_s(), t_scan = 0, t_upload = 0, t_exec = 0, t_download = 0, t_bmp = 0;
    pjd_timings last_timings;
    std::memset(&last_timings, 0, sizeof last_timings);
    int calls = 0;

    size_t next = 0;
    while (next < sized.size()) {
        // ---- scan stage (replaces mcu_prepare's read_JPEG loop, decoder_host.cpp:118-123)
        double t0 = now_s();
        std::vector<pjd_scanned *> scanned;
        std::vector<std::string> names;
        while (next < sized.size() && scanned.size() < batch_images) {
            const std::string &f = sized[next++].second;
            pjd_scanned *s = nullptr;
            int sr = pjd_scan_file_ex(f.c_str(), scan_options, &s);
            if (sr == 2) {
                std::cout << f << ": Error - Error opening input file\n" << f << ": Error - Invalid JPEG\n";
                continue;
            }
            std::cout << pjd_scanned_log(s);
            if (sr != 0) { pjd_scanned_free(s); continue; }
            scanned.push_back(s);
            names.push_back(f);
        }
        t_scan += now_s() - t0;
        if (scanned.empty()) continue;

        std::vector<pjd_image_desc> descs;
        for (pjd_scanned *s : scanned) descs.push_back(*pjd_scanned_desc(s));
        pjd_batch *b = nullptr;
        t0 = now_s();
        rc = pjd_batch_create(ctx, descs.data(), (int)descs.size(), PJD_OUT_BMP, &b);
        if (rc == PJD_OK) rc = pjd_batch_upload(b);
        t_upload += now_s() - t0;
        if (rc != PJD_OK) {
            std::cout << "Error - GPU batch setup failed: " << pjd_last_error(ctx) << "\n";
            for (pjd_scanned *s : scanned) pjd_scanned_free(s);
            if (b) pjd_batch_destroy(b);
            continue;
        }
        t0 = now_s();
        rc = pjd_batch_decode_timed(b, &last_timings);
        if (rc == PJD_OK) rc = pjd_batch_sync(b);
        t_exec += now_s() - t0;
        calls++;

        std::vector<std::vector<uint8_t>> outs(descs.size());
        std::vector<uint8_t *> ptrs(descs.size());
        std::vector<int32_t> status(descs.size(), 0);
        for (size_t i = 0; i < descs.size(); i++) {
            outs[i].resize(pjd_batch_output_size(b, (int)i));
            ptrs[i] = outs[i].data();
        }
        t0 = now_s();
        if (rc == PJD_OK) rc = pjd_batch_download(b, ptrs.data(), status.data());
        t_download += now_s() - t0;
        if (rc != PJD_OK) std::cout << "Error - GPU decode failed: " << pjd_last_error(ctx) << "\n";

        t0 = now_s();
        for (size_t i = 0; i < descs.size() && rc == PJD_OK; i++) {
            if (status[i] != PJD_ST_OK) std::cout << names[i] << ": " << pjd_status_string(status[i]) << "\n";
            const std::string out = bmp_name(names[i]);
            if (pjd_write_file(out.c_str(), outs[i].data(), outs[i].size()) != 0)
                std::cout << out << ": Error - Unable to create BMP file" << std::endl;
        }
        t_bmp += now_s() - t0;
        pjd_batch_destroy(b);
        for (pjd_scanned *s : scanned) pjd_scanned_free(s);
    }
    t_total = now_s() - t_total;

    std::cout << "\nProfiles:\n";
    std::cout << "End-to-end execution time: " << t_total << "s\n";
    std::cout << "MCU Offloader execution time (total): \n";
    std::cout << " - JPEG scan time: " << t_scan << "s\n";
    std::cout << " - CPU-to-GPU transfer time: " << t_upload << "s\n";
    std::cout << " - GPU execution time: " << t_exec << "s\n";
    for (int k = 0; k < last_timings.n; k++)
        std::cout << " - GPU execution - " << last_timings.name[k] << ": " << last_timings.ms[k] << " ms (last batch)\n";
    std::cout << " - GPU-to-CPU transfer time: " << t_download << "s\n";
    std::cout << " - BMP write time: " << t_bmp << "s\n";
    std::cout << " - Total " << calls << " calls\n";
    pjd_close(ctx);
    return 0;
}
