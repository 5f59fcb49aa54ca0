// Host-side robustness check (CPU only): the scanner (host/pjd_scan.cpp) and the planner (csrc/pjd_plan.cpp) under
// AddressSanitizer + UndefinedBehaviorSanitizer on mutated copies of the golden fixtures (bytes overwritten, truncations, insertions,
// deletions), with and without PJD_SCAN_PROGRESSIVE; every file the scanner accepts is planned alone and in a batch of three.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iinclude -Ipim-jpeg-decoder_amd/csrc tools/asan_host_fuzz.cpp \
//       -D__host__= -D__device__= pim-jpeg-decoder_amd/host/pjd_scan.cpp pim-jpeg-decoder_amd/csrc/pjd_plan.cpp -o /tmp/asan_host_fuzz && /tmp/asan_host_fuzz tests/golden/*.jpg
// (tests/test_host_sanitizers.py runs a short version; PJD_FUZZ_REPS = passes over the file list, default 40)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../include/pjd.h"
#include "../include/pjd_host.h"
#include "../pim-jpeg-decoder_amd/csrc/pjd_plan.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(uint32_t n) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33) % (n ? n : 1); }

int main(int argc, char **argv)
{
    int reps = 40;
    if (const char *e = std::getenv("PJD_FUZZ_REPS")) reps = std::atoi(e);
    uint64_t scanned = 0, accepted = 0, planned = 0;
    for (int rep = 0; rep < reps; rep++)
        for (int a = 1; a < argc; a++) {
            std::vector<uint8_t> b;
            if (FILE *f = std::fopen(argv[a], "rb")) { uint8_t buf[65536]; size_t n; while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n); std::fclose(f); }
            if (b.size() < 8) continue;
            switch (rnd(5)) {
                case 0: for (uint32_t k = 0, m = 1 + rnd(5); k < m; k++) b[rnd((uint32_t)b.size())] = (uint8_t)rnd(256); break;
                case 1: b.resize(2 + rnd((uint32_t)b.size() - 2)); break;
                case 2: { const uint32_t i = rnd((uint32_t)b.size()), m = 1 + rnd(40); std::vector<uint8_t> ins(m); for (auto &x : ins) x = (uint8_t)rnd(256); b.insert(b.begin() + i, ins.begin(), ins.end()); break; }
                case 3: { const uint32_t i = rnd((uint32_t)b.size() - 1), m = 1 + rnd(64); b.erase(b.begin() + i, b.begin() + (i + m < b.size() ? i + m : b.size())); break; }
                default: break;       // unchanged
            }
            for (uint32_t opt = 0; opt < 2; opt++) {
                pjd_scanned *sc = nullptr;
                pjd_scan_memory_ex(b.data(), b.size(), "fuzz", opt ? PJD_SCAN_PROGRESSIVE : 0, &sc);
                scanned++;
                if (!sc) continue;
                if (pjd_scanned_valid(sc)) {
                    accepted++;
                    const pjd_image_desc *d = pjd_scanned_desc(sc);
                    for (int fmt : {PJD_OUT_RGB8, PJD_OUT_BMP}) {
                        PjdPlan P; std::string err;
                        pjd_image_desc three[3] = {*d, *d, *d};
                        three[1].flags |= PJD_F_FORCE_SEQUENTIAL; three[2].flags |= PJD_F_STANDARD_RESTART;
                        if (pjd_make_plan(d, 1, fmt, P, err, 0) == PJD_OK) planned++;
                        if (pjd_make_plan(three, 3, fmt, P, err, rnd(2) ? 0 : 64u * (2 + rnd(15))) == PJD_OK) planned++;
                    }
                }
                pjd_scanned_free(sc);
            }
        }
    std::printf("asan_host_fuzz: %llu scans, %llu accepted, %llu plans, no sanitizer report\n", (unsigned long long)scanned, (unsigned long long)accepted, (unsigned long long)planned);
    return 0;
}
