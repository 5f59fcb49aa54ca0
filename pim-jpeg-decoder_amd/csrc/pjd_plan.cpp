// pjd_plan.cpp -- host-side batch planner (no HIP calls; runs anywhere).
//
// Turns the parsed-JPEG descriptors (the fields of the reference `Header`,
// reference src/headers/jpeg.h:146-179) into the flat arrays the gfx950 kernels
// walk: per-image geometry, restart segments, Huffman subsequences (one decode
// lane each), Huffman workgroups, IDCT/colour workgroups, packed table sets.
#include "pjd_plan.h"

#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <map>

namespace {

uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

bool same_table(const pjd_huff_table &a, const pjd_huff_table &b)
{
    return std::memcmp(a.offsets, b.offsets, 17) == 0 && std::memcmp(a.symbols, b.symbols, 162) == 0;
}

std::string fmt(const char *f, int i, long a = 0, long b = 0)
{
    char buf[256];
    std::snprintf(buf, sizeof buf, f, i, a, b);
    return buf;
}

}  // namespace

extern "C" uint64_t pjd_output_size(uint32_t width, uint32_t height, int out_format)
{
    if (out_format == PJD_OUT_BMP)   // reference src/bmp_writer.cpp:28-29 (its own padding rule: W % 4)
        return 26ull + (uint64_t)height * ((uint64_t)width * 3 + width % 4);
    return (uint64_t)width * height * 3;
}

// ---------------------------------------------------------------------------------------------
// Fewest bits of stream per STEP of the write pass that a table set can be made to sustain (x 256) -- what sizes the lane regions.
// A step is one symbol or the pair the decode tables hold (pjd_internal.h): first symbol whole inside 9 bits (code + value bits, at
// most 8), the second symbol's code inside the rest.  Model: a graph whose edges are steps weighted by the bits they consume --
//   node D   the next symbol is a unit's DC symbol            node A_L   the next symbol is an AC symbol with a code of >= L bits
// (a symbol that stayed single because the code after it did not fit the 9 bits leaves that code, which is long, for the next step);
// every unit is a cycle through D, so over whole units bits >= steps x (minimum mean weight of a cycle): Karp's algorithm on 17 nodes.
// A unit may also end on any run/size symbol (it fills slot 63), which the graph allows everywhere: conservative.
// `combos`: the (DC table, AC table, DC pairs allowed) of the components.
struct SymBits { uint32_t len, bits; bool eob, valid; };
static int table_symbols(const pjd_huff_table &t, bool ac, SymBits *v)
{
    int n = 0;
    for (int len = 1; len <= 16; len++)
        for (uint32_t q = t.offsets[len - 1]; q < t.offsets[len] && q < 162; q++) {
            const uint32_t sym = t.symbols[q];
            // the reference's "no symbol" and the out-of-range sizes are errors, but the write pass decodes on to the lane's end before
            // anyone looks: such an entry consumes its code alone and never pairs (lut_entry, pjd_k_huffman.hip)
            const bool valid = sym != 0xFF && (ac ? (sym & 15u) <= 10 : sym <= 11);
            v[n++] = {(uint32_t)len, (uint32_t)len + (valid ? (ac ? (sym & 15u) : sym) : 0u), valid && ac && sym == 0, valid};
        }
    return n;
}
static uint32_t min_step_bits_x256(const std::vector<std::pair<const pjd_huff_table *, const pjd_huff_table *>> &combos, const std::vector<char> &dc_pairs)
{
    enum { V = 17 };                                     // 0: D, L = 1..16: A_L
    const uint32_t INF = 1u << 30;
    uint32_t w[V][V];
    for (int u = 0; u < V; u++) for (int v = 0; v < V; v++) w[u][v] = INF;
    auto lower = [](uint32_t &dst, uint32_t bits) { if (bits < dst) dst = bits; };
    // the code after a symbol that stayed single is at least this long (a pair broken for another reason -- the lane ends -- happens once
    // per lane: PJD_LANE_CAP's slack)
    auto need = [](uint32_t first_bits) { return first_bits >= 9 ? 1 : (int)(10 - first_bits); };
    for (size_t c = 0; c < combos.size(); c++) {
        bool seen = false;                               // Cb and Cr mostly share their tables: once is enough
        for (size_t k = 0; k < c; k++) seen = seen || (combos[k] == combos[c] && dc_pairs[k] == dc_pairs[c]);
        if (seen) continue;
        SymBits dc[162], ac[162];
        const int ndc = table_symbols(*combos[c].first, false, dc), nac = table_symbols(*combos[c].second, true, ac);
        // the cheapest second symbol of a pair whose code fits r bits: any (the unit may end there: an EOB, or slot 63 filled) / one
        // that leaves the unit open
        uint32_t any2[10], open2[10];
        for (int r = 0; r < 10; r++) any2[r] = open2[r] = INF;
        for (int i = 0; i < nac; i++)
            if (ac[i].valid)
                for (uint32_t r = ac[i].len; r < 10; r++) { lower(any2[r], ac[i].bits); if (!ac[i].eob) lower(open2[r], ac[i].bits); }
        for (int i = 0; i < ndc; i++) {
            const SymBits &x = dc[i];
            const bool pairs = dc_pairs[c] && x.valid;
            lower(w[0][pairs ? need(x.bits) : 1], x.bits);
            if (pairs && x.bits <= 8) {
                const uint32_t r = 9 - x.bits;
                if (open2[r] < INF) lower(w[0][1], x.bits + open2[r]);
                if (any2[r] < INF) lower(w[0][0], x.bits + any2[r]);
            }
        }
        // an AC symbol with a code of `len` bits is an edge out of every A_L with L <= len: collect by length, then take suffix minima
        uint32_t by_len[V][V];
        for (int u = 0; u < V; u++) for (int v = 0; v < V; v++) by_len[u][v] = INF;
        for (int i = 0; i < nac; i++) {
            const SymBits &y = ac[i];
            uint32_t *e = by_len[y.len];
            lower(e[0], y.bits);                                             // EOB, or slot 63 filled: the unit ends
            if (y.eob) continue;
            lower(e[y.valid ? need(y.bits) : 1], y.bits);                    // stays single: the next code is long
            if (y.valid && y.bits <= 8) {
                const uint32_t r = 9 - y.bits;
                if (open2[r] < INF) lower(e[1], y.bits + open2[r]);
                if (any2[r] < INF) lower(e[0], y.bits + any2[r]);
            }
        }
        for (int L = 16; L >= 1; L--)
            for (int v = 0; v < V; v++) {
                if (L < 16) lower(by_len[L][v], by_len[L + 1][v]);
                lower(w[L][v], by_len[L][v]);
            }
    }
    // Karp: minimum mean cycle.  dist[k][v] = lightest walk of exactly k edges from anywhere to v
    const uint64_t FAR = (uint64_t)1 << 40;
    uint64_t dist[V + 1][V];
    for (int v = 0; v < V; v++) dist[0][v] = 0;
    for (int k = 1; k <= V; k++) {
        for (int v = 0; v < V; v++) dist[k][v] = FAR;
        for (int u = 0; u < V; u++) {
            if (dist[k - 1][u] >= FAR) continue;
            for (int v = 0; v < V; v++)
                if (w[u][v] < INF && dist[k - 1][u] + w[u][v] < dist[k][v]) dist[k][v] = dist[k - 1][u] + w[u][v];
        }
    }
    double best = 1e9;
    for (int v = 0; v < V; v++) {
        if (dist[V][v] >= FAR) continue;
        double worst = -1e9;
        for (int k = 0; k < V; k++)
            if (dist[k][v] < FAR) worst = std::max(worst, (double)((int64_t)dist[V][v] - (int64_t)dist[k][v]) / (double)(V - k));
        best = std::min(best, worst);
    }
    if (best > 64.0 || best < 1.0) best = best < 1.0 ? 1.0 : 64.0;
    return (uint32_t)(best * 256.0);                    // rounded down
}

int pjd_make_plan(const pjd_image_desc *images, int n, int out_format, PjdPlan &P, std::string &err,
                  uint32_t sub_bytes_override, int plan_mode)
{
    if (n < 0 || (n > 0 && !images)) { err = "null image array"; return PJD_E_ARG; }
    if (out_format != PJD_OUT_RGB8 && out_format != PJD_OUT_BMP) { err = "unknown output format"; return PJD_E_ARG; }
    P = PjdPlan();
    P.out_format = out_format;
    P.plan_mode = plan_mode == PJD_PLAN_THROUGHPUT ? PJD_PLAN_THROUGHPUT : PJD_PLAN_LATENCY;
    P.images.resize(n);
    P.host.resize(n);
    P.qtab.assign((size_t)n * 3 * 64, 0);

    // Subsequence size.  A decoder started at a wrong position falls into step with the true one only when bit position,
    // zigzag slot AND the phase inside the MCU all agree; measured, that takes about two MCUs' worth of stream on average
    // with a long tail (4:2:0: ~160 B at 80 B per MCU, several times that for dense high-quality streams).  Longer lanes mean
    // fewer re-sync passes per byte (less work: what counts with batches in flight), shorter lanes mean shorter passes, rounds and
    // write passes on every chain (what counts for one batch alone).  Round 4, after the second landing pad and the symbol pairs
    // (profiles/r04_experiments.md #18), default batch, stream of 8 / 4.5 / 4 / 3.5 / 3 / 2.5 MCUs per lane (capped at 1024 B):
    // entropy decoder alone 2.18 / 2.45 / 2.05 / 1.87 / 2.04 / 2.19 ms, one batch 2.56 / 2.83 / 2.33 / 2.23 / 2.29 / 2.45 ms, four
    // batches in flight 116.5 / 117.1 / 114.9 / 114.3 / 111.3 / 107.5 GPix/s on the same box (8 MCUs on another: 118.6):
    // and 6 MCUs with the walker threshold of 8 (6 lanes): one batch 2.66, in flight 122.2 against 116.0 for 3.5 MCUs on one box.
    // Neither serves both, so the caller says what it runs (pjd_set_plan_mode): 3.5 MCUs' worth for a batch that is decoded alone
    // (PJD_PLAN_LATENCY, the default: 16 % off the batch's time), 6 MCUs' worth for batches kept in flight (PJD_PLAN_THROUGHPUT:
    // 5 % more pictures per second).  (Round 2, before either change: 8 MCUs was best for both.)  A small batch needs enough
    // lanes to fill 256 CUs.
    {
        uint64_t total = 0, mcus = 0;
        for (int i = 0; i < n; i++) {
            total += images[i].ecs_len;
            const uint32_t hs = images[i].h_samp ? images[i].h_samp : 1, vs = images[i].v_samp ? images[i].v_samp : 1;
            mcus += (uint64_t)((images[i].width + 8 * hs - 1) / (8 * hs)) * ((images[i].height + 8 * vs - 1) / (8 * vs));
        }
        const uint32_t by_total = total >= (64u << 20) ? (uint32_t)PJD_SUB_BYTES_MAX : (total >= (24u << 20) ? 512u : (total >= (2u << 20) ? 256u : 128u));
        static const uint64_t mcus_x2_env = [] { const char *e = std::getenv("PJD_SUB_MCUS_X2"); const int v = e ? std::atoi(e) : 0; return (uint64_t)(v > 0 ? v : 0); }();   // experiments: subsequence = this many half-MCUs of stream
        const uint64_t mcus_x2 = mcus_x2_env ? mcus_x2_env : (P.plan_mode == PJD_PLAN_THROUGHPUT ? 12 : 7);
        uint64_t by_density = mcus ? (total * mcus_x2 / 2 / mcus + 63) / 64 * 64 : 512;
        if (by_density < PJD_SUB_BYTES_MIN) by_density = PJD_SUB_BYTES_MIN;
        if (by_density > PJD_SUB_BYTES_MAX) by_density = PJD_SUB_BYTES_MAX;
        uint32_t sb = by_total < by_density ? by_total : (uint32_t)by_density;
        if (sub_bytes_override) {
            if (sub_bytes_override < PJD_SUB_BYTES_MIN || sub_bytes_override > PJD_SUB_BYTES_MAX || (sub_bytes_override & 63)) {
                err = "sub_bytes override must be a multiple of 64 in [128, 1024]";
                return PJD_E_ARG;
            }
            sb = sub_bytes_override;
        }
        P.sub_bytes = sb;
    }
    const uint32_t SB = P.sub_bytes;
    // experiments: one walker threshold for every picture of the plan (pjd_internal.h); read once per plan
    int walk_max_env = -1;
    if (const char *e = std::getenv("PJD_WALK_MAX")) { const int v = std::atoi(e); walk_max_env = v < 0 ? 0 : v; }

    uint64_t ecs_off = 0, out_off = 0, du_total = 0, dense_seq = 0, lut_off = 0;
    uint32_t sb_max = SB;                          // largest per-image subsequence: sizes the word rows and the lane regions
    std::map<std::string, uint32_t> tset_of;       // raw bytes of a deduplicated table list -> table set
    std::vector<char> tset_parallel;               // per set: the two-level tables fit the parallel decoder
    std::vector<uint32_t> tset_min_bits;           // per set: fewest bits (code + value bits) any of its symbols consumes
    std::vector<uint32_t> &tset_step_bits = P.tset_step_bits;      // per set: fewest bits per step of the write pass, x 256 (min_step_bits_x256)
    uint64_t ent_off = 0;                          // slots of the lane regions handed out so far
    for (int i = 0; i < n; i++) {
        const pjd_image_desc &d = images[i];
        PjdDevImage &g = P.images[i];
        PjdHostImage &h = P.host[i];
        std::memset(&g, 0, sizeof g);

        // ---- envelope: exactly what the reference scanner lets through (jpeg_scanner.cpp:187-285)
        if (d.width == 0 || d.height == 0 || d.width > 65535 || d.height > 65535) { err = fmt("image %d: bad dimensions", i); return PJD_E_ARG; }
        if (d.num_components < 1 || d.num_components > 3) { err = fmt("image %d: num_components must be 1..3", i); return PJD_E_ARG; }
        if ((d.h_samp != 1 && d.h_samp != 2) || (d.v_samp != 1 && d.v_samp != 2)) { err = fmt("image %d: luma sampling must be 1 or 2", i); return PJD_E_ARG; }
        if (d.comp_h[0] != d.h_samp || d.comp_v[0] != d.v_samp) { err = fmt("image %d: comp_h/v[0] must equal h_samp/v_samp", i); return PJD_E_ARG; }
        const bool progressive = (d.flags & PJD_F_PROGRESSIVE) != 0;
        for (int c = 0; c < d.num_components; c++) {
            if (c > 0 && (d.comp_h[c] != 1 || d.comp_v[c] != 1)) { err = fmt("image %d: chroma sampling must be 1x1", i); return PJD_E_ARG; }
            if (d.comp_qt[c] > 3 || d.comp_dc[c] > 3 || d.comp_ac[c] > 3) { err = fmt("image %d: table selector > 3", i); return PJD_E_ARG; }
            if (!d.qt_set[d.comp_qt[c]]) { err = fmt("image %d: component uses an unset table", i); return PJD_E_ARG; }
            if (!progressive && (!d.dc[d.comp_dc[c]].set || !d.ac[d.comp_ac[c]].set)) { err = fmt("image %d: component uses an unset table", i); return PJD_E_ARG; }
        }
        if (progressive) {
            // the scans carry their own tables (pjd_scan_desc); what a scan may be is ITU T.81 G.1.1.1 (the reference checks the same
            // in its SOS reader, src/jpeg_scanner.cpp:88-112)
            if (!d.scans || d.n_scans == 0 || d.n_scans > 4096) { err = fmt("image %d: progressive frame without scans", i); return PJD_E_ARG; }
            if (d.shard_n_segs != 0) { err = fmt("image %d: a progressive frame cannot be sharded", i); return PJD_E_ARG; }
            for (uint32_t k = 0; k < d.n_scans; k++) {
                const pjd_scan_desc &sc = d.scans[k];
                bool ok = sc.n_comp >= 1 && sc.n_comp <= 3 && sc.n_comp <= d.num_components && sc.ss <= sc.se && sc.se <= 63 && sc.al <= 13 && sc.ah <= 13;
                ok = ok && !(sc.ss == 0 && sc.se != 0) && !(sc.ss != 0 && sc.n_comp != 1) && (sc.ah == 0 || sc.al + 1 == sc.ah);
                for (int q = 0; ok && q < sc.n_comp; q++) {
                    ok = sc.comp[q] < d.num_components && (q == 0 || sc.comp[q] > sc.comp[q - 1]);
                    const pjd_huff_table &t = sc.table[q];
                    const bool needs_table = !(sc.ss == 0 && sc.ah != 0);          // a DC refinement scan reads raw bits
                    if (needs_table) {
                        ok = ok && t.set && t.offsets[0] == 0 && t.offsets[16] <= 162;
                        for (int l = 1; ok && l <= 16; l++) ok = t.offsets[l] >= t.offsets[l - 1];
                    }
                }
                if (!ok || (sc.ecs_len > 0 && !sc.ecs) || sc.ecs_len >= (1ull << 29)) { err = fmt("image %d: malformed scan %ld", i, (long)k); return PJD_E_ARG; }
            }
        }
        for (int c = 0; c < d.num_components && !progressive; c++)
            for (int a = 0; a < 2; a++) {
                const pjd_huff_table &t = a ? d.ac[d.comp_ac[c]] : d.dc[d.comp_dc[c]];
                bool ok = t.offsets[0] == 0 && t.offsets[16] <= 162;
                for (int k = 1; k <= 16; k++) ok = ok && t.offsets[k] >= t.offsets[k - 1];
                if (!ok) { err = fmt("image %d: malformed Huffman table offsets", i); return PJD_E_ARG; }
            }
        if (d.ecs_len > 0 && !d.ecs) { err = fmt("image %d: null ecs", i); return PJD_E_ARG; }
        if (d.ecs_len >= (1ull << 29)) { err = fmt("image %d: ecs larger than 512 MiB", i); return PJD_E_ARG; }

        g.width = d.width; g.height = d.height;
        g.ncomp = d.num_components; g.hs = d.h_samp; g.vs = d.v_samp;
        g.n_luma = g.hs * g.vs;
        g.dus_per_mcu = g.n_luma + g.ncomp - 1;
        g.ref_mcu_w = (d.width + 7) / 8; g.ref_mcu_h = (d.height + 7) / 8;
        g.ref_mcu_w_real = g.ref_mcu_w + ((g.hs == 2 && (g.ref_mcu_w & 1)) ? 1 : 0);
        g.mcux = (g.ref_mcu_w + g.hs - 1) / g.hs;
        g.mcuy = (g.ref_mcu_h + g.vs - 1) / g.vs;
        g.n_mcu = g.mcux * g.mcuy;
        g.restart_interval = d.restart_interval;
        g.n_du = g.n_mcu * g.dus_per_mcu;
        if (out_format == PJD_OUT_BMP) { g.flags |= PJD_IF_BMP; g.out_stride = d.width * 3 + d.width % 4; }
        else g.out_stride = d.width * 3;
        if (d.flags & PJD_F_STANDARD_RESTART) g.flags |= PJD_IF_STANDARD_RESTART;
        if (d.flags & PJD_F_STANDARD_ZIGZAG) g.flags |= PJD_IF_STANDARD_ZIGZAG;

        // ---- quantisation tables: the reference copies QT t only while every t' < t is set
        //      (decoder_host.cpp:173-178); later tables read as zero.  Only the low 16 bits of
        //      a product survive the int16 store (decoder_dpu.c:169-172).
        for (int c = 0; c < (int)g.ncomp; c++) {
            bool visible = true;
            for (int t = 0; t <= d.comp_qt[c]; t++) visible = visible && d.qt_set[t];
            for (int k = 0; k < 64; k++)
                P.qtab[((size_t)i * 3 + c) * 64 + k] = visible ? (uint16_t)d.qt[d.comp_qt[c]][k] : 0;
            if ((d.flags & PJD_F_STANDARD_ZIGZAG) && visible) P.qtab[((size_t)i * 3 + c) * 64 + 58] = (uint16_t)d.qt_slot48[d.comp_qt[c]];
        }

        // ---- Huffman tables: dedupe the (up to) 3 DC + 3 AC tables the components reference, then find or make
        //      the table SET (identical lists share decode tables, and their waves can share a workgroup)
        //      A DC table's decode table holds the pairs "DC symbol + the unit's first AC symbol", so it belongs to ONE AC table: two
        //      components with the same DC table and different AC tables get a DC slot each (at most 3 + 3 slots).
        int nt = 0;
        const pjd_huff_table *seen[PJD_MAX_TABLES];
        uint8_t seen_ac[PJD_MAX_TABLES];
        uint8_t pair_ac[PJD_MAX_TABLES];               // of a DC slot: its AC slot
        std::memset(pair_ac, 0xff, sizeof pair_ac);
        for (int c = 0; c < (int)g.ncomp && !progressive; c++) {
            const pjd_huff_table *ta = &d.ac[d.comp_ac[c]], *td = &d.dc[d.comp_dc[c]];
            int sa = -1, sd = -1;
            for (int k = 0; k < nt; k++)
                if (seen_ac[k] == 1 && same_table(*seen[k], *ta)) { sa = k; break; }
            if (sa < 0) { sa = nt++; seen[sa] = ta; seen_ac[sa] = 1; }
            for (int k = 0; k < nt; k++)
                if (seen_ac[k] == 0 && pair_ac[k] == (uint8_t)sa && same_table(*seen[k], *td)) { sd = k; break; }
            if (sd < 0) { sd = nt++; seen[sd] = td; seen_ac[sd] = 0; pair_ac[sd] = (uint8_t)sa; }
            g.tbl_slot[c][0] = (uint8_t)sd;
            g.tbl_slot[c][1] = (uint8_t)sa;
        }
        std::string key;
        key.reserve((size_t)nt * 180 + 8);
        for (int k = 0; k < nt; k++) {
            key.push_back((char)seen_ac[k]);
            key.append(reinterpret_cast<const char *>(seen[k]->offsets), 17);
            key.append(reinterpret_cast<const char *>(seen[k]->symbols), 162);
        }
        // the components' assignment belongs to the set: it decides which DC tables hold pairs and how many steps a lane can take
        key.push_back((char)g.ncomp);
        for (int c = 0; c < (int)g.ncomp; c++) { key.push_back((char)g.tbl_slot[c][0]); key.push_back((char)g.tbl_slot[c][1]); }
        auto found = tset_of.find(key);
        if (progressive) found = tset_of.end();                 // no table set: the scans bring their tables
        else if (found == tset_of.end()) {
            const uint32_t ts = (uint32_t)P.tsets.size();
            found = tset_of.emplace(std::move(key), ts).first;
            PjdDevTset T;
            std::memset(&T, 0, sizeof T);
            T.n_tables = (uint32_t)nt;
            P.tables.resize((size_t)(ts + 1) * PJD_MAX_TABLES, PjdDevHuffRaw());
            // decode-table layout (pjd_internal.h): first-level tables, then one 128-entry second-level table per
            // 9-bit prefix that holds codes longer than 9 bits.  Over-subscribed tables (not a prefix code) and
            // tables whose long codes need more LDS than PJD_LUT_LDS_MAX go to the exact kernel.
            bool ok = true;
            uint32_t set_min_bits = 16;
            uint32_t lut_bytes = (uint32_t)nt * PJD_L1_BYTES;
            for (int k = 0; k < nt; k++) {
                PjdDevHuffRaw &r = P.tables[(size_t)ts * PJD_MAX_TABLES + k];
                std::memcpy(r.offsets, seen[k]->offsets, 17);
                std::memcpy(r.symbols, seen[k]->symbols, 162);
                r.is_ac = seen_ac[k];
                uint32_t code = 0, end10 = 0;
                for (int len = 1; len <= 16; len++) {            // reference generate_codes (jpeg_scanner.cpp:438-448)
                    const uint32_t cnt = (uint32_t)r.offsets[len] - r.offsets[len - 1];
                    // bits the cheapest symbol of this length consumes: code + value bits (an out-of-range size stores no value)
                    for (uint32_t q = r.offsets[len - 1]; q < r.offsets[len] && q < 162; q++) {
                        const uint32_t sym = r.symbols[q], size = seen_ac[k] ? ((sym & 15u) > 10 ? 0u : (sym & 15u)) : (sym > 11 ? 0u : sym);
                        if ((uint32_t)len + size < P.min_sym_bits) P.min_sym_bits = (uint32_t)len + size;
                        if ((uint32_t)len + size < set_min_bits) set_min_bits = (uint32_t)len + size;
                    }
                    if (code + cnt > (1u << len)) ok = false;
                    if (len == PJD_LUT_BITS) end10 = code + cnt;
                    code = (code + cnt) << 1;
                }
                const uint32_t end16 = code >> 1;
                const uint32_t np = 1u << PJD_LUT_BITS;
                uint32_t p0 = end10 < np ? end10 : np, p1 = (end16 + (1u << PJD_L2_BITS) - 1) >> PJD_L2_BITS;
                if (p1 > np) p1 = np;
                if (p1 < p0 || !ok) p1 = p0;
                T.l2_p0[k] = (uint16_t)p0; T.l2_p1[k] = (uint16_t)p1;
                T.l2_off[k] = (uint16_t)(lut_bytes / 2);
                lut_bytes += (p1 - p0) * (2u << PJD_L2_BITS);
                if (lut_bytes > PJD_LUT_LDS_MAX) ok = false;
            }
            std::memcpy(T.pair_ac, pair_ac, sizeof pair_ac);
            {
                std::vector<std::pair<const pjd_huff_table *, const pjd_huff_table *>> combos;
                std::vector<char> dcp;
                for (int c = 0; c < (int)g.ncomp; c++) {
                    combos.push_back({&d.dc[d.comp_dc[c]], &d.ac[d.comp_ac[c]]});
                    dcp.push_back(pair_ac[g.tbl_slot[c][0]] == g.tbl_slot[c][1] ? 1 : 0);
                }
                tset_step_bits.push_back(min_step_bits_x256(combos, dcp));
            }
            T.lut_bytes = ok ? (uint32_t)align_up(lut_bytes, 16) : 0;
            T.lut_off16 = (uint32_t)(lut_off / 16);
            lut_off += T.lut_bytes;
            if (T.lut_bytes > P.max_lut_bytes) P.max_lut_bytes = T.lut_bytes;
            P.tsets.push_back(T);
            tset_parallel.push_back(ok ? 1 : 0);
            tset_min_bits.push_back(set_min_bits < 1 ? 1u : set_min_bits);
        }
        g.tset = progressive ? 0u : found->second;
        const bool tables_parallel_ok = !progressive && tset_parallel[g.tset] != 0;

        // ---- restart segments and routing
        const uint32_t RI = d.restart_interval;
        const bool luma11 = (g.hs == 1 && g.vs == 1);
        const bool std_rule = (d.flags & PJD_F_STANDARD_RESTART) != 0;
        bool sequential = (d.flags & PJD_F_FORCE_SEQUENTIAL) != 0 || !tables_parallel_ok;
        if (progressive) g.flags |= PJD_IF_PROGRESSIVE;
        if (g.n_du + 1 >= (1u << 28)) sequential = true;  // look-back descriptors carry 28-bit unit indices, saturating at n_du + 1
        uint32_t nseg_total = 1;
        if (RI != 0) {
            nseg_total = (g.n_mcu + RI - 1) / RI;
            // the reference's rule (jpeg_scanner.cpp:723) equals the MCU counter only for 1x1 luma
            if (!luma11 && !std_rule) sequential = true;
            if (!d.seg_offsets || d.n_segments != nseg_total) sequential = true;
        }
        if (!sequential && d.seg_offsets && d.n_segments > 0) {
            if (d.seg_offsets[0] != 0) sequential = true;
            for (uint32_t k = 1; k < d.n_segments && !sequential; k++)
                if (d.seg_offsets[k] < d.seg_offsets[k - 1] || d.seg_offsets[k] > d.ecs_len) sequential = true;
        }
        uint32_t seg_lo = 0, seg_hi = nseg_total;
        if (d.shard_n_segs != 0) {
            if (sequential || RI == 0) { err = fmt("image %d: sharding needs restart segments decodable by the parallel path", i); return PJD_E_ARG; }
            if (d.shard_first_seg >= nseg_total || d.shard_first_seg + d.shard_n_segs > nseg_total) { err = fmt("image %d: shard out of range", i); return PJD_E_ARG; }
            seg_lo = d.shard_first_seg; seg_hi = seg_lo + d.shard_n_segs;
        }
        h.sequential = sequential;
        if (sequential) g.flags |= PJD_IF_SEQUENTIAL;

        uint64_t byte_lo = 0, byte_hi = d.ecs_len;
        if (RI != 0 && !sequential) {
            byte_lo = d.seg_offsets[seg_lo];
            byte_hi = (seg_hi < nseg_total) ? d.seg_offsets[seg_hi] : d.ecs_len;
        }
        if (seg_hi == nseg_total) g.flags |= PJD_IF_ENDS_STREAM;   // this image (or shard) decodes the bitstream's last segment
        g.first_mcu = (RI != 0 && !sequential) ? seg_lo * RI : 0;
        g.last_mcu = (RI != 0 && !sequential && seg_hi < nseg_total) ? seg_hi * RI : g.n_mcu;

        h.ecs_src = (d.ecs && !progressive) ? d.ecs + byte_lo : nullptr;
        h.ecs_copy_len = progressive ? 0 : byte_hi - byte_lo;
        g.ecs_len = (uint32_t)h.ecs_copy_len;
        g.ecs_off = ecs_off;
        ecs_off = align_up(ecs_off + h.ecs_copy_len + 48, 16);   // >= 48 zero bytes after every stream (a symbol may start on its last bit)
        if (progressive) {
            g.pscan_base = (uint32_t)P.pscans.size();
            g.n_pscan = d.n_scans;
            for (uint32_t k = 0; k < d.n_scans; k++) {
                const pjd_scan_desc &sc = d.scans[k];
                PjdDevScan ds;
                std::memset(&ds, 0, sizeof ds);
                ds.ecs_off = ecs_off; ds.ecs_len = (uint32_t)sc.ecs_len; ds.restart_interval = sc.restart_interval;
                ds.n_comp = sc.n_comp; ds.ss = sc.ss; ds.se = sc.se; ds.ah = sc.ah; ds.al = sc.al;
                for (int q = 0; q < sc.n_comp; q++) {
                    ds.comp[q] = sc.comp[q];
                    std::memcpy(ds.table[q].offsets, sc.table[q].offsets, 17);
                    std::memcpy(ds.table[q].symbols, sc.table[q].symbols, 162);
                    ds.table[q].is_ac = sc.ss != 0;
                }
                P.pscans.push_back(ds);
                P.host_scans.push_back({sc.ecs, sc.ecs_len, ecs_off});
                ecs_off = align_up(ecs_off + sc.ecs_len + 48, 16);
                P.ecs_bytes += sc.ecs_len;
            }
        }

        g.seg_base = (uint32_t)P.segs.size();
        g.lane_base = (uint32_t)P.subs.size();
        g.hwave_base = (uint32_t)P.hwaves.size();
        // Subsequence size of this image: the batch's, nudged so that the image's lanes fill whole waves (a wave holds lanes of
        // one image; at 832 B an ImageNet-sized picture is ~2.1 waves' worth, i.e. three waves, the last nearly empty).
        uint32_t SBi = SB;
        if (!sequential && !sub_bytes_override) {
            auto seg_len = [&](uint32_t k) {
                const uint64_t b0 = (RI != 0) ? d.seg_offsets[k] : 0, b1 = (RI != 0 && k + 1 < nseg_total) ? d.seg_offsets[k + 1] : d.ecs_len;
                return (uint32_t)(b1 - b0);
            };
            auto lanes_for = [&](uint32_t S) {
                uint64_t nl = 0;
                for (uint32_t k = seg_lo; k < seg_hi; k++) { const uint32_t len = seg_len(k); nl += len ? (len + S - 1) / S : 1; }
                return nl;
            };
            const uint64_t n0 = lanes_for(SB);
            if (n0 > PJD_HUFF_LANES / 2 && n0 % PJD_HUFF_LANES != 0) {
                const uint64_t kw = (n0 + PJD_HUFF_LANES / 2) / PJD_HUFF_LANES;           // nearest number of whole waves
                uint32_t best = SB;
                // smallest multiple of 64 bytes (within -30 % / +45 % of the batch's size) whose lanes fit kw waves
                const uint32_t lo = SB * 7 / 10 / 64 * 64 > PJD_SUB_BYTES_MIN ? SB * 7 / 10 / 64 * 64 : PJD_SUB_BYTES_MIN;
                const uint32_t hi = SB * 29 / 20 < PJD_SUB_BYTES_MAX ? SB * 29 / 20 : PJD_SUB_BYTES_MAX;
                for (uint32_t S = lo; S <= hi; S += 64)
                    if (lanes_for(S) <= kw * PJD_HUFF_LANES) { best = S; break; }
                if (lanes_for(best) <= kw * PJD_HUFF_LANES) SBi = best;
            }
        }
        g.sub_bytes = SBi;
        {   // which re-sync rounds the wave walks (pjd_internal.h)
            // density of what THIS image (or shard: a slice of the stream for a slice of the MCUs) decodes
            const uint64_t n_mcu_dec = g.last_mcu > g.first_mcu ? g.last_mcu - g.first_mcu : 1;
            const uint64_t bpm = (byte_hi - byte_lo) / n_mcu_dec;                         // bytes of stream per MCU
            uint32_t wm = PJD_WALK_LANES(SBi);
            if ((uint64_t)SBi <= PJD_WALK_DENSE_MCUS * bpm && wm < PJD_WALK_DENSE) wm = PJD_WALK_DENSE;
            if (walk_max_env >= 0) wm = (uint32_t)(walk_max_env > 64 ? 64 : walk_max_env);
            g.walk_max = (uint8_t)wm;
        }
        if (SBi > sb_max) sb_max = SBi;
        if (!sequential) {
            for (uint32_t k = seg_lo; k < seg_hi; k++) {
                PjdDevSegment sg;
                uint64_t b0 = (RI != 0) ? d.seg_offsets[k] : 0;
                uint64_t b1 = (RI != 0 && k + 1 < nseg_total) ? d.seg_offsets[k + 1] : d.ecs_len;
                sg.byte_start = (uint32_t)(b0 - byte_lo);
                sg.byte_end = (uint32_t)(b1 - byte_lo);
                uint32_t m0 = (RI != 0) ? k * RI : 0;
                uint32_t m1 = (RI != 0) ? ((k + 1) * RI < g.n_mcu ? (k + 1) * RI : g.n_mcu) : g.n_mcu;
                sg.first_du = m0 * g.dus_per_mcu;
                sg.n_du = (m1 - m0) * g.dus_per_mcu;
                const uint32_t seg_index = (uint32_t)P.segs.size();
                P.segs.push_back(sg);
                uint32_t len = sg.byte_end - sg.byte_start;
                uint32_t nsub = len ? (len + SBi - 1) / SBi : 1;
                for (uint32_t j = 0; j < nsub; j++) {
                    PjdDevSub q;
                    q.byte_start = sg.byte_start + j * SBi;
                    q.seg = seg_index | (j == 0 ? 0x80000000u : 0u);
                    P.subs.push_back(q);
                }
            }
            g.n_seg = (uint32_t)P.segs.size() - g.seg_base;
            g.n_lane = (uint32_t)P.subs.size() - g.lane_base;
            // lane regions of this picture: sized by ITS subsequence and the cheapest symbol of ITS tables (pjd_internal.h)
            g.lane_cap = PJD_LANE_CAP(SBi, tset_step_bits[g.tset]);
            g.ent_base = ent_off;
            ent_off += (uint64_t)g.n_lane * g.lane_cap;
            if (g.lane_cap > P.lane_cap) P.lane_cap = g.lane_cap;
            for (uint32_t s0 = 0; s0 < g.n_lane; s0 += PJD_HUFF_LANES) {
                PjdDevHuffWave w;
                w.image = (uint32_t)i;
                w.first_lane = g.lane_base + s0;
                w.n_lanes = (g.n_lane - s0 < PJD_HUFF_LANES) ? g.n_lane - s0 : PJD_HUFF_LANES;
                w.pad_ = 0;
                // workgroups: consecutive waves that use the same table set, PJD_HUFF_WAVES at most
                if (P.hwgs.empty() || P.hwgs.back().tset != g.tset || P.hwgs.back().n_waves == PJD_HUFF_WAVES) {
                    PjdDevHuffWg wgp;
                    wgp.first_wave = (uint32_t)P.hwaves.size(); wgp.n_waves = 0; wgp.tset = g.tset; wgp.pad_ = 0;
                    P.hwgs.push_back(wgp);
                }
                P.hwgs.back().n_waves++;
                P.hwaves.push_back(w);
            }
            g.n_hwave = (uint32_t)P.hwaves.size() - g.hwave_base;
            P.fast_images.push_back((uint32_t)i);
        } else {
            P.seq_images.push_back((uint32_t)i);
        }

        // ---- dense scratch (exact-kernel images), IDCT workgroups, output
        g.image_index = (uint32_t)i;
        du_total += g.n_du;
        if (sequential) { g.dense_base = dense_seq; dense_seq += g.n_du; }
        g.idct_mcus = PJD_IDCT_MAX_DU / g.dus_per_mcu;
        g.iwg_base = (uint32_t)(sequential ? P.iwgs_dense.size() : P.iwgs.size());
        for (uint32_t m = g.first_mcu; m < g.last_mcu; m += g.idct_mcus) {
            PjdDevIdctWg w;
            w.image = (uint32_t)i;
            w.first_mcu = m;
            w.n_mcu = (g.last_mcu - m < g.idct_mcus) ? g.last_mcu - m : g.idct_mcus;
            w.pad_ = sequential ? (uint32_t)(P.seq_images.size() - 1) : 0;   // dense path: which entry of the scratch-base list
            (sequential ? P.iwgs_dense : P.iwgs).push_back(w);
        }
        g.n_iwg = (uint32_t)(sequential ? P.iwgs_dense.size() : P.iwgs.size()) - g.iwg_base;
        h.out_bytes = pjd_output_size(d.width, d.height, out_format);
        g.out_off = out_off;
        out_off = align_up(out_off + h.out_bytes, 256);

        P.pixels += (uint64_t)d.width * d.height;
        P.ecs_bytes += d.ecs_len;
        P.out_bytes += h.out_bytes;
    }
    // Order of the Huffman workgroups = order in which they start (the kernel hands out indices by ticket).  Pictures whose stream
    // is dense (many bytes per MCU) take the most re-sync rounds and finish last: start them first, so that their chains run beside
    // the bulk of the batch instead of after it (matters with several batches in flight, when workgroups queue for a place on the
    // chip).  Waves only wait for EARLIER waves of their own picture, so whole pictures can be moved freely; pictures that share a
    // workgroup (same table set, consecutive waves) move together.
    if (P.hwgs.size() > 1 && !std::getenv("PJD_KEEP_WG_ORDER")) {
        struct Unit { size_t first, count; double key; };
        std::vector<Unit> units;
        auto density = [&](uint32_t img) { const PjdDevImage &g = P.images[img]; return g.n_mcu ? (double)g.ecs_len / (double)(g.last_mcu - g.first_mcu ? g.last_mcu - g.first_mcu : 1) : 0.0; };
        for (size_t k = 0; k < P.hwgs.size(); k++) {
            const PjdDevHuffWg &w = P.hwgs[k];
            const uint32_t img_first = P.hwaves[w.first_wave].image, img_last = P.hwaves[w.first_wave + w.n_waves - 1].image;
            double key = 0;
            for (uint32_t im = img_first; im <= img_last; im++) key = std::max(key, density(im));
            const bool joins = !units.empty() && P.hwaves[P.hwgs[k - 1].first_wave + P.hwgs[k - 1].n_waves - 1].image == img_first;
            if (joins) { units.back().count++; units.back().key = std::max(units.back().key, key); }
            else units.push_back({k, 1, key});
        }
        std::stable_sort(units.begin(), units.end(), [](const Unit &a, const Unit &b) { return a.key > b.key; });
        std::vector<PjdDevHuffWg> ordered;
        ordered.reserve(P.hwgs.size());
        std::vector<size_t> unit_end;          // positions in the ordered list where a unit ends: no picture straddles them
        for (const Unit &u : units) {
            ordered.insert(ordered.end(), P.hwgs.begin() + (long)u.first, P.hwgs.begin() + (long)(u.first + u.count));
            unit_end.push_back(ordered.size());
        }
        P.hwgs.swap(ordered);
        // Picture groups (pjd_internal.h): cut the start order into runs of about equal numbers of waves, densest pictures first.
        // Only for batches of many small pictures: the group's DC predictors are scanned by one workgroup per picture.
        // The first group -- the densest pictures, whose entropy decode ends last -- is the smallest: every group's entropy decode lasts
        // about as long as its longest chain of re-sync rounds whatever its share of the work, and what a group adds behind that is its
        // own back end.  Measured for one batch alone (profiles/r04_experiments.md): round-3 lanes (8 MCUs' worth): no groups 2.89 ms,
        // two groups cut at 30 % 2.60, thirds 2.59-2.61, 20/55 % 2.66, 12/35/65 % 2.67, six groups 4.1 (more streams than the runtime's
        // four hardware queues: chains queue behind each other); latency-plan lanes (3.5 MCUs' worth): no groups 2.50, 30 % 2.23,
        // 40 / 50 % 2.21 / 2.20, 30/65 % 2.11, 35/70 2.12, 25/60 2.18, 25/50/75 2.12, 30/55/80 2.10: three groups, cut at 30 and 65 %.
        // PJD_GROUPS / PJD_GROUP_CUTS ("20,55": percent of the waves) for experiments.
        uint32_t want = 3;
        std::vector<uint32_t> cuts_pct = {30, 65};
        if (const char *e = std::getenv("PJD_GROUPS")) { const int v = std::atoi(e); want = (uint32_t)(v < 1 ? 1 : (v > PJD_MAX_GROUPS ? PJD_MAX_GROUPS : v)); cuts_pct.clear(); }
        if (const char *e = std::getenv("PJD_GROUP_CUTS")) {
            cuts_pct.clear();
            for (const char *q = e; *q;) { cuts_pct.push_back((uint32_t)std::strtoul(q, const_cast<char **>(&q), 10)); while (*q == ',' || *q == ' ') q++; }
            want = (uint32_t)cuts_pct.size() + 1;
            if (want > PJD_MAX_GROUPS) { want = PJD_MAX_GROUPS; cuts_pct.resize(PJD_MAX_GROUPS - 1); }
        }
        if (cuts_pct.empty()) for (uint32_t k = 1; k < want; k++) cuts_pct.push_back(100u * k / want);
        bool small = P.fast_images.size() >= 64;
        for (uint32_t i : P.fast_images) if (P.images[i].n_lane > 8192) small = false;
        if (small && want > 1) {
            const size_t total_waves = P.hwaves.size();
            size_t h0 = 0, waves = 0;
            std::vector<char> seen(P.images.size(), 0);
            auto close_group = [&](size_t h1) {
                PjdDevGroup g;
                std::memset(&g, 0, sizeof g);
                g.hwg_first = (uint32_t)h0; g.hwg_count = (uint32_t)(h1 - h0);
                g.img_first = (uint32_t)P.group_images.size();
                g.iwg_first = (uint32_t)P.iwg_order.size();
                for (size_t k = h0; k < h1; k++)
                    for (uint32_t w = P.hwgs[k].first_wave; w < P.hwgs[k].first_wave + P.hwgs[k].n_waves; w++) {
                        const uint32_t im = P.hwaves[w].image;
                        if (seen[im]) continue;
                        seen[im] = 1;
                        P.group_images.push_back(im);
                        for (uint32_t q = 0; q < P.images[im].n_iwg; q++) P.iwg_order.push_back(P.images[im].iwg_base + q);
                    }
                g.img_count = (uint32_t)P.group_images.size() - g.img_first;
                g.iwg_count = (uint32_t)P.iwg_order.size() - g.iwg_first;
                P.groups.push_back(g);
                h0 = h1;
            };
            size_t pos = 0;
            for (size_t u = 0; u < unit_end.size(); u++) {
                for (; pos < unit_end[u]; pos++) waves += P.hwgs[pos].n_waves;
                const size_t k = P.groups.size();
                if (k + 1 < want && waves * 100 >= (size_t)cuts_pct[k] * total_waves && u + 1 < unit_end.size()) close_group(pos);
            }
            close_group(P.hwgs.size());
            if (P.groups.size() < 2) { P.groups.clear(); P.group_images.clear(); P.iwg_order.clear(); }
        }
    }
    // the lane-word kernel copies PJD_WORD_ROWS words from every lane's first byte, whatever the lane's length
    P.ecs_buf_bytes = align_up(ecs_off + PJD_SUB_BYTES_MAX + 64, 256);
    P.n_du = du_total;
    P.sub_bytes = sb_max;
    if (P.min_sym_bits < 1 || P.tsets.empty()) P.min_sym_bits = 1;
    P.n_ent = ent_off + 16;
    P.n_words = (uint64_t)P.hwaves.size() * PJD_WORD_ROWS(sb_max) * 64;
    P.dense_du = dense_seq;
    P.out_buf_bytes = align_up(out_off, 256);
    P.n_dcblk = (P.subs.size() + PJD_DC_BLOCK - 1) / PJD_DC_BLOCK;
    P.lut_buf_bytes = align_up(lut_off + 16, 256);
    if (P.ecs_buf_bytes >= (1ull << 40)) { err = "batch bitstream too large"; return PJD_E_ARG; }
    // every byte a lane can make the kernels read lies inside the bitstream buffer: pjd_k_lane_words copies PJD_WORD_ROWS words
    // from each lane's first byte (the decoders then only touch those rows); checked here, once per batch, not assumed
    {
        const uint64_t reach = (uint64_t)PJD_WORD_ROWS(sb_max) * 4;
        for (const PjdDevImage &g : P.images)
            for (uint32_t q = g.lane_base; q < g.lane_base + g.n_lane; q++)
                if (g.ecs_off + P.subs[q].byte_start + reach > P.ecs_buf_bytes) { err = "internal: a Huffman lane would read past the bitstream buffer"; return PJD_E_ARG; }
    }
    if (P.subs.size() >= (1ull << 31) || P.n_ent >= (1ull << 40)) { err = "batch has too many Huffman lanes"; return PJD_E_ARG; }
    return PJD_OK;
}
