// tests only: the coding-unit pipeline with one CPU thread per wave and real waits (coder_core.hpp built with
// -DICER_WAVE_EMU -DICER_WAVE_THREADS), optionally under ThreadSanitizer.  Reads a raw sign-magnitude plane,
// codes it `reps` times as one unit and prints "<bits> <fnv1a of the payload>" per repetition; the Python test
// compares with the oracle.
//   threads_main <plane.raw> <w> <h> <subband> <lsb> <reps>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define ICER_WAVE_EMU
#define ICER_WAVE_THREADS
#include "../../icer_compression_amd/csrc/coder_core.hpp"

using namespace icer;

static CoderShared g_sh;

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const size_t w = (size_t)atol(argv[2]), h = (size_t)atol(argv[3]);
    const int sb = atoi(argv[4]), lsb = atoi(argv[5]), reps = atoi(argv[6]);
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
        a.cap_words = (uint32_t)((w * h * 3 + 64 + 3) / 4);
        std::vector<uint32_t> words(a.cap_words + 1, 0);
        a.out_words = words.data();
        a.timers = nullptr;
        a.done_bytes = nullptr; a.prio_index = 0; a.early_quota = 0;
        const uint32_t bits = code_unit_threads(g_sh, a);
        if (bits == kUnitTooBig || bits == kUnitFailed) { printf("%d 0\n", bits == kUnitFailed ? -10 : -5); continue; }
        uint64_t hsh = 1469598103934665603ull;
        const uint8_t *p = (const uint8_t *)words.data();
        for (size_t i = 0; i < ((size_t)bits + 7) / 8; i++) { hsh ^= p[i]; hsh *= 1099511628211ull; }
        printf("%u %016llx\n", bits, (unsigned long long)hsh);
    }
    return 0;
}
