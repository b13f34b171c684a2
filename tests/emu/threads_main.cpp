// tests only: the coding-unit pipeline with one CPU thread per wave and real waits (coder_core.hpp built with
// -DICER_WAVE_EMU -DICER_WAVE_THREADS), optionally under ThreadSanitizer.  Reads a raw sign-magnitude plane,
// codes it `reps` times as one unit and prints "<bits> <fnv1a of the payload>" per repetition; the Python test
// compares with the oracle.
//   threads_main <plane.raw> <w> <h> <subband> <lsb> <reps> [stop_us [cap_div]]
// stop_us > 0: progressive-mode stop -- another thread raises the "quota spent" flag after a random 0..stop_us
// microseconds, the unit must wind down (prints "-3 0") or finish normally if it was faster.  cap_div > 1: payload slot
// of 1 / cap_div of the safe size, the unit must give up with "slot too small" (prints "-5 0") unless it fits.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <random>
#include <thread>
#include <vector>

#define ICER_WAVE_EMU
#define ICER_WAVE_THREADS
#include "../../icer_compression_amd/csrc/coder_core.hpp"
#include "coder_emu_drivers.hpp"

using namespace icer;

static CoderShared g_sh;

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const size_t w = (size_t)atol(argv[2]), h = (size_t)atol(argv[3]);
    const int sb = atoi(argv[4]), lsb = atoi(argv[5]), reps = atoi(argv[6]);
    const int stop_us = argc > 7 ? atoi(argv[7]) : 0, cap_div = argc > 8 ? atoi(argv[8]) : 1;
    if (const char *sh = getenv("ICER_EMU_SHAPE")) { const bool small = atoi(sh) == 1; g_emu_npw = small ? 1u : 2u; g_emu_ngw = small ? 0u : 2u; }   // workgroup shape: 1 = 8 waves, 2 = 11
    std::mt19937 rng(12345);
    std::vector<uint16_t> plane(w * h);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(plane.data(), 2, w * h, f) != w * h) return 3;
    fclose(f);
    for (int r = 0; r < reps; r++) {
        memset(&g_sh, 0xA5, sizeof g_sh);
        build_coder_tables(&g_sh.tab);
        UnitArgs a;
        a.seg = plane.data(); a.stride = (uint32_t)w; a.w = (uint32_t)w; a.h = (uint32_t)h;
        a.subband = sb; a.lsb = lsb;
        a.cap_words = (uint32_t)((w * h * 3 + 64 + 3) / 4) / (uint32_t)(cap_div > 1 ? cap_div : 1);
        std::vector<uint32_t> words(a.cap_words + 1, 0);
        a.out_words = words.data();
        a.timers = nullptr;
        a.done_bytes = nullptr; a.prio_index = 0; a.early_quota = 0;
        EmuEvents evs;
        evs.build(a);
        uint32_t stop_flag = 0;
        std::thread clock;
        if (stop_us > 0) {
            a.done_bytes = &stop_flag; a.early_quota = 1;
            const int wait_us = (int)(rng() % (unsigned)stop_us);
            clock = std::thread([&stop_flag, wait_us] {
                std::this_thread::sleep_for(std::chrono::microseconds(wait_us));
                __atomic_store_n(&stop_flag, 1u, __ATOMIC_RELAXED);
            });
        }
        const uint32_t bits = code_unit_threads(g_sh, a);
        if (clock.joinable()) clock.join();
        const uint32_t ab = __atomic_load_n(&g_sh.abort, __ATOMIC_RELAXED);
        if (bits == kUnitFailed) { printf("-10 0\n"); continue; }
        if (ab == 3u) { printf("-3 0\n"); continue; }
        if (bits == kUnitTooBig) { printf("-5 0\n"); continue; }
        uint64_t hsh = 1469598103934665603ull;
        const uint8_t *p = (const uint8_t *)words.data();
        for (size_t i = 0; i < ((size_t)bits + 7) / 8; i++) { hsh ^= p[i]; hsh *= 1099511628211ull; }
        printf("%u %016llx\n", bits, (unsigned long long)hsh);
    }
    return 0;
}
