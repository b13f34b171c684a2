// TESTS ONLY: the barrier-only workgroup coder (csrc/coder_wg.hpp) as a CPU lane-loop build (-DICER_WAVE_EMU, see
// csrc/wave.hpp): the waves of a workgroup become an array of register states, a barrier-closed region a loop over
// them.  Compared with the oracle by tests/test_emu_wg.py.  Not part of the product library.
#define ICER_WAVE_EMU 1
#ifndef ICER_WG_WAVES
#define ICER_WG_WAVES 16
#endif
#include "../../icer_compression_amd/csrc/coder_wg.hpp"
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace icer;

static wg::Shared g_sh;
static int g_no_table = 0;
extern "C" void emu_wg_use_table(int on) { g_no_table = !on; }

extern "C" int emu_wg_waves(void) { return (int)wg::kWgWaves; }
// the exact small divisions of blank_run (float reciprocal + correction)
extern "C" unsigned emu_wg_floor_div(unsigned num, unsigned den) { return wg::floor_div_small(num, den); }
extern "C" unsigned emu_wg_ceil_div(unsigned num, unsigned den) { return wg::ceil_div_small(num, den); }
// 0: the waves run every region in order, 1: backwards, >= 2: shuffled (seed)
extern "C" void emu_wg_set_order(unsigned order) { wg::g_wg_order = order; wg::g_wg_order_state = order; }
extern "C" void emu_wg_stats(unsigned long long *out, int reset) { for (int i = 0; i < 8; i++) { out[i] = wg::g_wg_stats[i]; if (reset) wg::g_wg_stats[i] = 0; } }
extern "C" int emu_wg_assert_line(void) { const int l = wg::g_wg_assert_line; wg::g_wg_assert_line = 0; return l; }

// one coding unit; returns the payload length in bits, -5 slot too small, -3 stopped (progressive mode)
extern "C" long emu_wg_code_unit(const uint16_t *seg, size_t w, size_t h, size_t stride, int subband, int lsb,
                                 uint8_t *out, size_t cap_bytes, int stop_flag)
{
    memset(&g_sh, 0xA5, sizeof g_sh);             // the kernel must not depend on LDS contents
    build_coder_tables(&g_sh.tab);
    wg::UnitArgs a;
    a.seg = seg; a.stride = (uint32_t)stride; a.w = (uint32_t)w; a.h = (uint32_t)h;
    a.subband = subband; a.lsb = lsb;
    a.cap_words = (uint32_t)(cap_bytes / 4);
    std::vector<uint32_t> words(a.cap_words + 1, 0);
    a.out_words = words.data();
    uint32_t flag = stop_flag ? 1u : 0u;
    a.done_bytes = &flag; a.prio_index = 0; a.early_quota = stop_flag ? 1u : 0u; a.timers = nullptr;
    // the family's chunk table (chunk_sig_kernel in the product); g_no_table: without, the general path only
    std::vector<uint8_t> sig((w * h + 63) / 64 + 4);
    for (uint32_t j = 0; j < (uint32_t)((w * h + 63) / 64); j++) sig[j] = (uint8_t)wg::chunk_blank_plane(seg, (uint32_t)stride, (uint32_t)w, (uint32_t)h, j);
    a.sig = g_no_table ? nullptr : sig.data();
    static wg::Wave regs[ICER_WG_WAVES];
    memset(regs, 0x5A, sizeof regs);
    { wg::Wave &R = regs[0]; (void)R; wg::unit_state_init(g_sh, a); }
    const uint32_t bits = wg::code_unit_wg(g_sh, a, regs);
    long res = bits == wg::kUnitTooBig ? -5 : bits == wg::kUnitStopped ? -3 : (long)bits;
    if (res >= 0) memcpy(out, words.data(), (size_t)(bits + 7) / 8);
    return res;
}
