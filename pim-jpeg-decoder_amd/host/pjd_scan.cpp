// pjd_scan.cpp -- host-side JPEG container scanner and BMP helpers (libpjdhost.so).
//
// Mirrors the reference's read_JPEG (reference src/jpeg_scanner.cpp:345-436 and the segment
// readers :6-343): same accepted / rejected set, same field values, same messages.  It is a
// memory scanner -- the file is read (or handed over) whole, marker segments are parsed by
// offset, and the entropy-coded segment is copied with memchr()-found runs instead of one
// std::ifstream::get() per byte.  Unlike the reference it keeps the byte offset (in the
// destuffed output) at which every restart segment starts; the GPU decoder needs them to
// treat restart segments as independent streams.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pjd_host.h"

namespace {

// zigzag index -> natural index as the reference has it, entry 48 = 38
// (reference src/headers/common.h:9-18)
const uint8_t kZigzag[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    38, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// Byte source with the end-of-file behaviour of std::ifstream::get(): past the end it
// returns -1 (0xFF once narrowed to a byte) and the stream turns bad for good.
struct Bytes {
    const uint8_t *p;
    uint64_t n, pos = 0;
    bool bad = false;
    int get()
    {
        if (bad || pos >= n) { bad = true; return -1; }
        return p[pos++];
    }
    uint32_t be16() { int hi = get(), lo = get(); return (uint32_t)(hi * 256 + lo); }     // (past the end: -1 * 256 + -1, as the reference's (get() << 8) + get() yields)
};

}  // namespace

struct pjd_scanned {
    pjd_image_desc d{};
    bool valid = true;
    std::string name, log;
    std::vector<uint8_t> ecs;
    std::vector<uint64_t> segs;
    uint8_t frame_type = 0;
    bool zero_based = false, in_frame[3] = {false, false, false}, in_scan[3] = {false, false, false};
    uint32_t mcu_w = 0, mcu_h = 0, mcu_w_real = 0, mcu_h_real = 0;
    // PJD_SCAN_PROGRESSIVE: the scans of a progressive frame (descriptors + their destuffed bytes)
    uint32_t options = 0;
    uint8_t last_ss = 0, last_se = 63, last_ah = 0, last_al = 0, last_ncs = 0, last_comp[3] = {0, 0, 0};
    std::vector<pjd_scan_desc> scans;
    std::vector<std::vector<uint8_t>> scan_ecs;

    void say(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        log += name;
        log += buf;
    }
    void reject(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        std::vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        log += name;
        log += buf;
        valid = false;
    }
};

namespace {

// reference src/jpeg_scanner.cpp:187-285
void frame_header(Bytes &in, pjd_scanned &s)
{
    pjd_image_desc &d = s.d;
    if (d.num_components != 0) return s.reject(": Error - Multiple SOFs detected\n");
    const uint32_t len = in.be16();
    const uint8_t precision = (uint8_t)in.get();
    if (precision != 8) return s.reject(": Error - Invalid precision: %u\n", (unsigned)precision);
    d.height = in.be16();
    d.width = in.be16();
    if (d.height == 0 || d.width == 0) return s.reject(": Error - Invalid dimensions\n");
    s.mcu_h = s.mcu_h_real = (d.height + 7) / 8;
    s.mcu_w = s.mcu_w_real = (d.width + 7) / 8;
    d.num_components = (uint8_t)in.get();
    if (d.num_components == 4) return s.reject(": Error - CMYK color mode not supported\n");
    if (d.num_components == 0) return s.reject(": Error - Number of color components must not be zero\n");
    for (unsigned k = 0; k < d.num_components; k++) {
        uint8_t id = (uint8_t)in.get();
        if (id == 0 && k == 0) s.zero_based = true;
        if (s.zero_based) id++;
        if (id == 4 || id == 5) return s.reject(": Error - YIQ color mode not supported\n");
        if (id == 0 || id > d.num_components || id > 3) return s.reject(": Error - Invalid component ID:%u\n", (unsigned)id);
        const int c = id - 1;
        if (s.in_frame[c]) return s.reject(": Duplicate color component ID\n");
        s.in_frame[c] = true;
        const uint8_t sf = (uint8_t)in.get();
        d.comp_h[c] = sf >> 4;
        d.comp_v[c] = sf & 15;
        if (id == 1) {
            if ((d.comp_h[c] != 1 && d.comp_h[c] != 2) || (d.comp_v[c] != 1 && d.comp_v[c] != 2))
                return s.reject(": Error - Sampling factors not supported\n");
            if (d.comp_h[c] == 2 && (s.mcu_w & 1)) s.mcu_w_real++;
            if (d.comp_v[c] == 2 && (s.mcu_h & 1)) s.mcu_h_real++;
            d.h_samp = d.comp_h[c];
            d.v_samp = d.comp_v[c];
        } else if (d.comp_h[c] != 1 || d.comp_v[c] != 1) {
            return s.reject(": Error - Sampling factors not supported\n");
        }
        d.comp_qt[c] = (uint8_t)in.get();
        if (d.comp_qt[c] > 3) return s.reject(": Error - Invalid quantization table ID in frame components\n");
    }
    if (len - 8 - 3u * d.num_components != 0) s.reject(": Error - SOF invalid\n");
}

// reference src/jpeg_scanner.cpp:287-321
void quant_tables(Bytes &in, pjd_scanned &s)
{
    int left = (int)in.be16() - 2;
    while (left > 0) {
        const uint8_t info = (uint8_t)in.get();
        left -= 1;
        const uint8_t id = info & 15;
        if (id > 3) return s.reject(": Error Invalid quantization table ID: %u\n", (unsigned)id);
        s.d.qt_set[id] = 1;
        const bool wide = (info >> 4) != 0;
        for (int k = 0; k < 64; k++) {
            const uint32_t q = wide ? in.be16() : (uint32_t)in.get();
            s.d.qt[id][kZigzag[k]] = q;
            if (k == 48) s.d.qt_slot48[id] = q;            // lost in `qt` (entry 52 lands on the same position); PJD_F_STANDARD_ZIGZAG
        }
        left -= wide ? 128 : 64;
    }
    if (left != 0) s.reject(": Error - DQT invalid\n");
}

// reference src/jpeg_scanner.cpp:140-185
void huffman_tables(Bytes &in, pjd_scanned &s)
{
    int left = (int)in.be16() - 2;
    while (left > 0) {
        const uint8_t info = (uint8_t)in.get();
        const uint8_t id = info & 15;
        if (id > 3) return s.reject(": Error - Invalid Huffman table ID: %u\n", (unsigned)id);
        pjd_huff_table &t = (info >> 4) ? s.d.ac[id] : s.d.dc[id];
        t.set = 1;
        t.offsets[0] = 0;
        uint32_t total = 0;
        for (int k = 1; k <= 16; k++) { total += (uint32_t)in.get(); t.offsets[k] = (uint8_t)total; }
        if (total > 162) return s.reject(": : Error - Too many symbols in Huffman table\n");
        for (uint32_t k = 0; k < total; k++) t.symbols[k] = (uint8_t)in.get();
        left -= 17 + (int)total;
    }
    if (left != 0) s.reject(": Error - DHT invalid\n");
}

// reference src/jpeg_scanner.cpp:6-138
void scan_header(Bytes &in, pjd_scanned &s)
{
    pjd_image_desc &d = s.d;
    if (d.num_components == 0) return s.reject(": Error - SOS detected before SOF\n");
    const uint32_t len = in.be16();
    for (unsigned k = 0; k < d.num_components; k++) s.in_scan[k] = false;
    const unsigned ncs = (uint8_t)in.get();
    if (ncs == 0) return s.reject(": Error - Scan must include at least 1 component\n");
    for (unsigned k = 0; k < ncs; k++) {
        uint8_t id = (uint8_t)in.get();
        if (s.zero_based) id++;
        if (id == 0 || id > d.num_components) return s.reject(": Error - Invalid color component ID: %u\n", (unsigned)id);
        const int c = id - 1;
        if (!s.in_frame[c]) return s.reject(": Error - Invalid color component ID: %u\n", (unsigned)id);
        if (s.in_scan[c]) return s.reject(": Error - Duplicate color component ID\n");
        s.in_scan[c] = true;
        const uint8_t ids = (uint8_t)in.get();
        d.comp_dc[c] = ids >> 4;
        d.comp_ac[c] = ids & 15;
        if (d.comp_dc[c] > 3) return s.reject(": Error - Invalid Huffman DC table ID: %u\n", (unsigned)d.comp_dc[c]);
        if (d.comp_ac[c] > 3) return s.reject(": Error - Invalid Huffman AC table ID: %u\n", (unsigned)d.comp_ac[c]);
    }
    const unsigned ss = (uint8_t)in.get(), se = (uint8_t)in.get();
    const uint8_t sa = (uint8_t)in.get();
    const unsigned ah = sa >> 4, al = sa & 15;
    s.last_ss = (uint8_t)ss; s.last_se = (uint8_t)se; s.last_ah = (uint8_t)ah; s.last_al = (uint8_t)al;
    s.last_ncs = 0;
    for (unsigned c = 0; c < d.num_components && s.last_ncs < 3; c++) if (s.in_scan[c]) s.last_comp[s.last_ncs++] = (uint8_t)c;
    if (s.frame_type == 0xC0) {
        if (ss != 0 || se != 63) return s.reject(": Error - Invalid spectral selection\n");
        if (ah != 0 || al != 0) return s.reject(": Error - Invalid successive approximation\n");
    } else if (s.frame_type == 0xC2) {
        if (ss > se) return s.reject(": Error - Invalid spectral selection (start greater than end)\n");
        if (se > 63) return s.reject(": Error - Invalid spectral selection (end greater than 63)\n");
        if (ss == 0 && se != 0) return s.reject(": Error - Invalid spectral selection (contains DC and AC)\n");
        if (ss != 0 && ncs != 1) return s.reject(": Error - Invalid spectral selection (AC scan contains multiple components)\n");
        if (ah != 0 && al != ah - 1) return s.reject(": Error - Invalid succesive approximation\n");
    }
    for (unsigned c = 0; c < d.num_components; c++) {
        if (!s.in_scan[c]) continue;
        if (!d.qt_set[d.comp_qt[c]]) return s.reject(": Error - Color component using uninitialized quantization table\n");
        if (ss == 0 && !d.dc[d.comp_dc[c]].set) return s.reject(": Error - Color component using uninitialized Huffman DC table\n");
        if (se > 0 && !d.ac[d.comp_ac[c]].set) return s.reject(": Error - Color component using uninitialized Huffman AC table\n");
    }
    if (len - 6 - 2 * ncs != 0) s.reject(": Error - SOS invalid\n");
}

void skip_segment(Bytes &in)
{
    const uint32_t cnt = in.be16() - 2;    // unsigned like the reference (jpeg_scanner.cpp:333-343)
    if (in.bad) return;
    if (cnt > in.n - in.pos) { in.pos = in.n; in.bad = true; return; }   // same end state as 'cnt' failing get()s
    in.pos += cnt;
}

// Bitstream buffers of freed pjd_scanned objects, kept for the next scans (at most 512 MiB): a batcher scans and frees
// ~100 KB per picture around the clock, and handing that to malloc/free makes every scan fault fresh pages in and ends
// a run with the allocator returning gigabytes to the system (100 ms at the end of a 16-batch run, profiles/r02_pcie.md).
struct EcsCache {
    std::mutex m;
    std::vector<std::vector<uint8_t>> free_list;
    size_t bytes = 0;
    static constexpr size_t kMaxBytes = (size_t)512 << 20;
    void take(std::vector<uint8_t> &v, size_t want)
    {
        std::lock_guard<std::mutex> l(m);
        // newest first; a buffer that is too small is still taken (reserve() below grows it once)
        if (free_list.empty()) return;
        v.swap(free_list.back());
        free_list.pop_back();
        bytes -= v.capacity();
        (void)want;
    }
    void give(std::vector<uint8_t> &v)
    {
        if (v.capacity() == 0) return;
        std::lock_guard<std::mutex> l(m);
        if (bytes + v.capacity() > kMaxBytes) return;          // the vector frees itself
        bytes += v.capacity();
        free_list.emplace_back();
        free_list.back().swap(v);
    }
};
EcsCache g_ecs_cache;

// Entropy-coded data (reference src/jpeg_scanner.cpp:405-433), run-at-a-time.  Returns false
// after an error was logged.
bool entropy_segment(Bytes &in, pjd_scanned &s)
{
    const uint8_t *p = in.p;
    const uint64_t n = in.n;
    uint64_t i = in.pos;
    if (s.ecs.capacity() == 0) g_ecs_cache.take(s.ecs, n - i + 16);
    s.ecs.clear();
    s.ecs.reserve(n - i + 16);
    s.segs.assign(1, 0);
    for (;;) {
        const uint8_t *ff = i < n ? (const uint8_t *)std::memchr(p + i, 0xFF, n - i) : nullptr;
        if (!ff) {
            // no further 0xFF: the reference runs off the end of the file
            s.reject(": Error - File ended prematurely\n");
            return false;
        }
        const uint64_t k = (uint64_t)(ff - p);
        s.ecs.insert(s.ecs.end(), p + i, p + k);
        // p[k] == 0xFF; classify by the byte(s) that follow, skipping FF fill bytes
        uint64_t j = k + 1;
        while (j < n && p[j] == 0xFF) j++;
        if (j >= n) { s.reject(": Error - File ended prematurely\n"); return false; }
        const uint8_t m = p[j];
        if (m == 0xD9) { in.pos = j + 1; return true; }
        if (m == 0x00) { s.ecs.push_back(0xFF); i = j + 1; continue; }
        if (m >= 0xD0 && m <= 0xD7) { s.segs.push_back(s.ecs.size()); i = j + 1; continue; }
        s.reject(": Error - Invalid marker during compressed data scan: 0x%x\n", (unsigned)m);
        return false;
    }
}

// Entropy-coded data of ONE scan of a progressive frame: like entropy_segment, but the scan ends at the next marker that is
// neither a restart marker nor stuffing; that marker is left to the caller (in.pos on its 0xFF).
bool scan_bytes(Bytes &in, pjd_scanned &s, std::vector<uint8_t> &out)
{
    const uint8_t *p = in.p;
    const uint64_t n = in.n;
    uint64_t i = in.pos;
    out.clear();
    for (;;) {
        const uint8_t *ff = i < n ? (const uint8_t *)std::memchr(p + i, 0xFF, n - i) : nullptr;
        if (!ff) { s.reject(": Error - File ended prematurely\n"); return false; }
        const uint64_t k = (uint64_t)(ff - p);
        out.insert(out.end(), p + i, p + k);
        uint64_t j = k + 1;
        while (j < n && p[j] == 0xFF) j++;
        if (j >= n) { s.reject(": Error - File ended prematurely\n"); return false; }
        const uint8_t m = p[j];
        if (m == 0x00) { out.push_back(0xFF); i = j + 1; continue; }
        if (m >= 0xD0 && m <= 0xD7) { i = j + 1; continue; }
        in.pos = j - 1;                                       // the marker's 0xFF
        return true;
    }
}

// PJD_SCAN_PROGRESSIVE: the scans of a progressive frame, first SOS header already parsed (ITU T.81 B.2.3; what the reference's
// single-scan progressive branches would need for every scan, src/jpeg_scanner.cpp:521-704).  Tables and restart intervals may
// be redefined between scans: every scan descriptor carries the tables in force when it starts.
void progressive_scans(Bytes &in, pjd_scanned &s)
{
    pjd_image_desc &d = s.d;
    for (;;) {
        pjd_scan_desc sc;
        std::memset(&sc, 0, sizeof sc);
        sc.n_comp = s.last_ncs;
        for (unsigned k = 0; k < sc.n_comp; k++) {
            const unsigned c = s.last_comp[k];
            sc.comp[k] = (uint8_t)c;
            sc.table[k] = s.last_ss == 0 ? d.dc[d.comp_dc[c]] : d.ac[d.comp_ac[c]];
        }
        sc.ss = s.last_ss; sc.se = s.last_se; sc.ah = s.last_ah; sc.al = s.last_al;
        sc.restart_interval = d.restart_interval;
        if (sc.al > 13) return s.reject(": Error - Invalid successive approximation\n");
        s.scan_ecs.emplace_back();
        if (!scan_bytes(in, s, s.scan_ecs.back())) return;
        s.scans.push_back(sc);
        // markers up to the next SOS or EOI
        for (;;) {
            const int a = in.get(), b = in.get();
            if (in.bad || a != 0xFF) return s.reject(": Error - Expected a marker\n");
            uint8_t cur = (uint8_t)b;
            while (cur == 0xFF) cur = (uint8_t)in.get();
            if (cur == 0xD9) goto done;
            if (cur == 0xC4) huffman_tables(in, s);
            else if (cur == 0xDB) quant_tables(in, s);
            else if (cur == 0xDD) { const uint32_t l = in.be16(); d.restart_interval = in.be16(); if (l - 4 != 0) s.reject(": Error - DRI invalid\n"); }
            else if (cur == 0xDA) { scan_header(in, s); break; }
            else if ((cur >= 0xE0 && cur <= 0xEF) || cur == 0xFE) skip_segment(in);
            else return s.reject(": Error - Invalid marker during compressed data scan: 0x%x\n", (unsigned)cur);
            if (!s.valid || in.bad) { if (s.valid) s.reject(": Error - File ended prematurely\n"); return; }
        }
        if (!s.valid) return;
        if (s.scans.size() > 1024) return s.reject(": Error - Too many scans\n");
    }
done:
    for (size_t k = 0; k < s.scans.size(); k++) { s.scans[k].ecs = s.scan_ecs[k].data(); s.scans[k].ecs_len = s.scan_ecs[k].size(); }
    d.scans = s.scans.data();
    d.n_scans = (uint32_t)s.scans.size();
    d.flags |= PJD_F_PROGRESSIVE;
    d.ecs = nullptr; d.ecs_len = 0; d.seg_offsets = nullptr; d.n_segments = 0;
    d.restart_interval = 0;                                   // per scan (pjd_scan_desc)
}

void scan_all(const uint8_t *data, uint64_t len, pjd_scanned &s)
{
    Bytes in{data, len};
    pjd_image_desc &d = s.d;
    d.h_samp = d.v_samp = 1;
    for (int c = 0; c < 3; c++) d.comp_h[c] = d.comp_v[c] = 1;

    uint8_t last = (uint8_t)in.get(), cur = (uint8_t)in.get();
    if (last != 0xFF || cur != 0xD8) { s.valid = false; return; }     // silently invalid, as the reference
    last = (uint8_t)in.get(); cur = (uint8_t)in.get();
    bool at_scan = false;
    while (s.valid) {
        if (in.bad || last != 0xFF) {
            if (in.bad) s.say(": Error - File ended prematurely\n");
            if (last != 0xFF) s.say(": Error - Expected a marker\n");
            s.valid = false;
            return;
        }
        if (cur == 0xC0 || cur == 0xC2) { s.frame_type = cur; frame_header(in, s); }
        else if (cur == 0xDB) quant_tables(in, s);
        else if (cur == 0xC4) huffman_tables(in, s);
        else if (cur == 0xDA) { scan_header(in, s); at_scan = true; break; }
        else if (cur == 0xDD) {
            const uint32_t l = in.be16();
            d.restart_interval = in.be16();
            if (l - 4 != 0) s.reject(": Error - DRI invalid\n");
        }
        else if ((cur >= 0xE0 && cur <= 0xEF) || cur == 0xFE || (cur >= 0xF0 && cur <= 0xFD) || cur == 0xDC || cur == 0xDE || cur == 0xDF)
            skip_segment(in);
        else if (cur == 0x01) { /* TEM */ }
        else if (cur == 0xFF) { cur = (uint8_t)in.get(); continue; }
        else s.say(": Error - Unknown marker: 0x%x\n", (unsigned)cur);
        last = (uint8_t)in.get(); cur = (uint8_t)in.get();
    }
    if (!s.valid || !at_scan) return;
    // The reference reads one byte ahead before its loop and tests the stream at the top of
    // every iteration; an SOS header that ends exactly at end-of-file is "ended prematurely".
    if (in.bad) { s.reject(": Error - File ended prematurely\n"); return; }
    if (s.frame_type == 0xC2 && (s.options & PJD_SCAN_PROGRESSIVE)) { progressive_scans(in, s); return; }
    if (!entropy_segment(in, s)) return;
    if (s.frame_type == 0xC2) {
        // A progressive frame whose first scan runs to EOI would enter the reference's
        // progressive branches (jpeg_scanner.cpp:521-704), which are outside this path.
        s.reject(": Error - Progressive JPEG not supported by the GPU path\n");
        return;
    }
    d.ecs = s.ecs.data();
    d.ecs_len = s.ecs.size();
    d.seg_offsets = s.segs.data();
    d.n_segments = (uint32_t)s.segs.size();
}

}  // namespace

extern "C" {

int pjd_scan_memory(const uint8_t *data, uint64_t len, const char *name, pjd_scanned **out)
{
    return pjd_scan_memory_ex(data, len, name, 0, out);
}

int pjd_scan_file(const char *path, pjd_scanned **out) { return pjd_scan_file_ex(path, 0, out); }

int pjd_scan_memory_ex(const uint8_t *data, uint64_t len, const char *name, uint32_t options, pjd_scanned **out)
{
    pjd_scanned *s = new pjd_scanned;
    s->name = name ? name : "";
    s->options = options;
    scan_all(data, len, *s);
    if (!s->valid) s->say(": Error - Invalid JPEG\n");
    *out = s;
    return s->valid ? 0 : 1;
}

int pjd_scan_file_ex(const char *path, uint32_t options, pjd_scanned **out)
{
    *out = nullptr;
    FILE *f = std::fopen(path, "rb");
    if (!f) return 2;      // reference: "<file>: Error - Error opening input file" then "Invalid JPEG"
    std::vector<uint8_t> buf;
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz > 0) { buf.resize((size_t)sz); if (std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) buf.clear(); }
    std::fclose(f);
    return pjd_scan_memory_ex(buf.data(), buf.size(), path, options, out);
}

const pjd_image_desc *pjd_scanned_desc(const pjd_scanned *s) { return &s->d; }
const char *pjd_scanned_log(const pjd_scanned *s) { return s->log.c_str(); }
int pjd_scanned_valid(const pjd_scanned *s) { return s->valid ? 1 : 0; }
void pjd_scanned_free(pjd_scanned *s)
{
    if (!s) return;
    g_ecs_cache.give(s->ecs);
    delete s;
}

void pjd_scanned_metadata(const pjd_scanned *s, uint32_t *m)
{
    const pjd_image_desc &d = s->d;
    std::memset(m, 0, 276 * sizeof(uint32_t));
    m[0] = s->mcu_h; m[1] = s->mcu_w; m[2] = s->mcu_h_real; m[3] = s->mcu_w_real;
    m[4] = d.num_components; m[5] = d.v_samp; m[6] = d.h_samp;
    for (unsigned c = 0; c < d.num_components; c++) {
        m[7 + c] = d.comp_qt[c];
        m[7 + d.num_components + c] = d.comp_h[c];
        m[7 + 2 * d.num_components + c] = d.comp_v[c];
    }
    m[17] = d.height; m[18] = d.width; m[19] = 100;
    for (int t = 0; t < 4 && d.qt_set[t]; t++)
        for (int k = 0; k < 64; k++) m[20 + 64 * t + k] = d.qt[t][k];
}

void pjd_rgb_to_bmp(const uint8_t *rgb, uint32_t w, uint32_t h, uint8_t *out)
{
    const uint32_t stride = w * 3 + w % 4;          // the reference pads by W % 4 (bmp_writer.cpp:28)
    const uint32_t size = 26 + h * stride;
    std::memset(out, 0, size);
    out[0] = 'B'; out[1] = 'M';
    for (int k = 0; k < 4; k++) out[2 + k] = (uint8_t)(size >> (8 * k));
    out[10] = 0x1A; out[14] = 12;
    out[18] = (uint8_t)w; out[19] = (uint8_t)(w >> 8);
    out[20] = (uint8_t)h; out[21] = (uint8_t)(h >> 8);
    out[22] = 1; out[24] = 24;
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *row = out + 26 + (size_t)(h - 1 - y) * stride;
        const uint8_t *src = rgb + (size_t)y * w * 3;
        for (uint32_t x = 0; x < w; x++) { row[3 * x] = src[3 * x + 2]; row[3 * x + 1] = src[3 * x + 1]; row[3 * x + 2] = src[3 * x]; }
    }
}

int pjd_write_file(const char *path, const uint8_t *data, uint64_t len)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) return -1;
    const size_t w = len ? std::fwrite(data, 1, (size_t)len, f) : 0;
    std::fclose(f);
    return w == len ? 0 : -1;
}

}  // extern "C"
