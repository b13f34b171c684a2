// TESTS ONLY: builds the kernel cores (csrc/coder_core.hpp, assemble_core.hpp, dwt_tile.hpp) and the
// host planner (csrc/plan.hpp) as a CPU lane-loop emulation (-DICER_WAVE_EMU, see csrc/wave.hpp), so
// the wave-parallel algorithms can be compared with the oracle in a container without a GPU.
// Not part of the product library; nothing here is reachable from libicer_hip.so.
#define ICER_WAVE_EMU 1
#include "../../icer_compression_amd/csrc/assemble_core.hpp"
#include "../../icer_compression_amd/csrc/coder_core.hpp"
#include "coder_emu_drivers.hpp"
#include "../../icer_compression_amd/csrc/coder_wg.hpp"
#include "../../icer_compression_amd/csrc/dwt_tile.hpp"
#include "../../icer_compression_amd/csrc/plan.hpp"
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace icer;

static CoderShared g_sh;
// which coder the whole-frame pipeline below uses: 0 = the eight-wave pipeline (coder_core.hpp), 1 = the workgroup-window
// coder (coder_wg.hpp; `order`: 0 the waves run a region in order, 1 backwards, >= 2 shuffled)
static int g_use_wg = 0;
static wg::Shared g_wsh;
extern "C" void emu_set_coder(int use_wg, unsigned order) { g_use_wg = use_wg; wg::g_wg_order = order; wg::g_wg_order_state = order; }
// shape of the emulated pipeline workgroup: pixel waves, golomb workers (1 or 2 each)
extern "C" void emu_set_shape(unsigned npw, unsigned ngw) { g_emu_npw = npw; g_emu_ngw = ngw; }
extern "C" int emu_wg_assert_line(void) { const int l = wg::g_wg_assert_line; wg::g_wg_assert_line = 0; return l; }
static uint32_t wg_unit(const UnitArgs &a0)
{
    memset(&g_wsh, 0xA5, sizeof g_wsh);
    build_coder_tables(&g_wsh.tab);
    wg::UnitArgs a;
    a.seg = a0.seg; a.stride = a0.stride; a.w = a0.w; a.h = a0.h; a.subband = a0.subband; a.lsb = a0.lsb;
    a.out_words = a0.out_words; a.cap_words = a0.cap_words;
    a.done_bytes = nullptr; a.prio_index = 0; a.early_quota = 0; a.timers = nullptr;
    std::vector<uint8_t> sig(((size_t)a.w * a.h + 63) / 64 + 4);
    for (uint32_t j = 0; j < (a.w * a.h + 63u) / 64u; j++) sig[j] = (uint8_t)wg::chunk_blank_plane(a.seg, a.stride, a.w, a.h, j);
    a.sig = sig.data();
    static wg::Wave regs[ICER_WG_WAVES];
    memset(regs, 0x5A, sizeof regs);
    wg::unit_state_init(g_wsh, a);
    return wg::code_unit_wg(g_wsh, a, regs);
}
unsigned long long g_emu_chunks[4] = {0, 0, 0, 0};
extern "C" void emu_chunk_stats(unsigned long long *out, int reset) { for (int i = 0; i < 4; i++) { out[i] = g_emu_chunks[i]; if (reset) g_emu_chunks[i] = 0; } }

extern "C" long emu_code_unit(const uint16_t *seg, size_t w, size_t h, size_t stride, int subband, int lsb,
                              uint8_t *out, size_t cap_bytes)
{
    memset(&g_sh, 0xA5, sizeof g_sh);             // the kernel must not depend on LDS contents
    build_coder_tables(&g_sh.tab);
    UnitArgs a;
    a.seg = seg; a.stride = (uint32_t)stride; a.w = (uint32_t)w; a.h = (uint32_t)h;
    a.subband = subband; a.lsb = lsb;
    a.cap_words = (uint32_t)(cap_bytes / 4);
    std::vector<uint32_t> words(a.cap_words + 1, 0);
    a.out_words = words.data();
    a.timers = nullptr;
    EmuEvents evs;
    evs.build(a);
    uint32_t bits = code_unit_emu(g_sh, a);
    long res = (bits == kUnitTooBig) ? -5 : (long)bits;
    if (res >= 0) memcpy(out, words.data(), (size_t)(bits + 7) / 8);
    return res;
}

// the unit cut into n_sub sub-ranges, one emulated workgroup each, spliced (code_unit_emu_split); *matches = workgroups that
// stopped at another one's snapshot
extern "C" long emu_code_unit_split(const uint16_t *seg, size_t w, size_t h, size_t stride, int subband, int lsb,
                                    uint8_t *out, size_t cap_bytes, uint32_t n_sub, uint32_t order, uint32_t *matches)
{
    memset(&g_sh, 0xA5, sizeof g_sh);
    build_coder_tables(&g_sh.tab);
    UnitArgs a;
    a.seg = seg; a.stride = (uint32_t)stride; a.w = (uint32_t)w; a.h = (uint32_t)h;
    a.subband = subband; a.lsb = lsb;
    a.cap_words = (uint32_t)(cap_bytes / 4);
    std::vector<uint32_t> words(a.cap_words + 1, 0xDEADBEEFu);
    a.out_words = words.data();
    a.timers = nullptr;
    a.done_bytes = nullptr; a.prio_index = 0; a.early_quota = 0;
    EmuEvents evs;
    evs.build(a);
    const uint32_t bits = code_unit_emu_split(g_sh, a, n_sub, order, matches);
    const long res = bits == kUnitTooBig ? -5 : bits == kUnitFailed ? -10 : (long)bits;
    if (res >= 0) memcpy(out, words.data(), (size_t)(bits + 7) / 8);
    return res;
}

// the same under the random wave scheduler (code_unit_emu_random); -10 = the waves dead-locked
extern "C" long emu_code_unit_random(const uint16_t *seg, size_t w, size_t h, size_t stride, int subband, int lsb,
                                     uint8_t *out, size_t cap_bytes, uint32_t seed)
{
    memset(&g_sh, 0xA5, sizeof g_sh);
    build_coder_tables(&g_sh.tab);
    UnitArgs a;
    a.seg = seg; a.stride = (uint32_t)stride; a.w = (uint32_t)w; a.h = (uint32_t)h;
    a.subband = subband; a.lsb = lsb;
    a.cap_words = (uint32_t)(cap_bytes / 4);
    std::vector<uint32_t> words(a.cap_words + 1, 0);
    a.out_words = words.data();
    a.timers = nullptr;
    EmuEvents evs;
    evs.build(a);
    const uint32_t bits = code_unit_emu_random(g_sh, a, seed);
    const long res = bits == kUnitTooBig ? -5 : bits == kUnitFailed ? -10 : (long)bits;
    if (res >= 0) memcpy(out, words.data(), (size_t)(bits + 7) / 8);
    return res;
}

// forward DWT with the structure of the product: one fused LDS-tile pass per stage (csrc/dwt_tile.hpp), the LL
// band handed from stage to stage through a side buffer, the three detail bands written in place
static int emu_dwt_lim(uint16_t *img, size_t w, size_t h, int stages, int filt, int32_t lim, int sm);
static int g_dwt_fast = 1;                 // interior tiles take the fast phases (0: every tile the generic ones)
static unsigned long long g_dwt_fast_tiles = 0;
extern "C" void emu_dwt_fast(int on) { g_dwt_fast = on; }
extern "C" unsigned long long emu_dwt_fast_tiles(void) { return g_dwt_fast_tiles; }
extern "C" int emu_dwt(uint16_t *img, size_t w, size_t h, int stages, int filt) { return emu_dwt_lim(img, w, h, stages, filt, 32767, 0); }
// sm: how the detail bands are stored (DwtStageArgs::sm; 0 = plain two's complement, what oracle.dwt returns)
static int emu_dwt_lim(uint16_t *img, size_t w, size_t h, int stages, int filt, int32_t lim, int sm)
{
    if (dim_low(w, stages) < 3 || dim_low(h, stages) < 3) return kTooManyStages;
    std::vector<int16_t> src((int16_t *)img, (int16_t *)img + w * h), tmp(w * h);
    int16_t *coef = (int16_t *)img;
    static DwtTileShared sh;
    DwtStageArgs a;
    a.lim = lim;
    a.sm = sm;
    a.f = filter_taps(filt);
    a.coef = coef; a.coef_stride = (uint32_t)w;
    a.src = src.data(); a.src_stride = (uint32_t)w;
    size_t cw = w, ch = h, off = 0;
    bool ovf = false;
    for (int s = 0; s < stages; s++) {
        const int nlw = (int)((cw + 1) / 2), nlh = (int)((ch + 1) / 2);
        a.cw = (int)cw; a.ch = (int)ch;
        if (s == stages - 1) { a.ll = coef; a.ll_stride = (uint32_t)w; }
        else { a.ll = tmp.data() + off; a.ll_stride = (uint32_t)nlw; }
        for (int ty = 0; ty < (nlh + kTileKY - 1) / kTileKY; ty++)
            for (int tx = 0; tx < (nlw + kTileKX - 1) / kTileKX; tx++) {
                memset(&sh, 0x5A, sizeof sh);
                if (g_dwt_fast && dwt_tile_is_interior(a, tx, ty)) {     // (as dwt_tile_kernel does)
                    static DwtFastShared fs;
                    memset(&fs, 0x5A, sizeof fs);
                    for (int t = 0; t < kTileThreads; t++) ovf |= dwt_fast_rows_step1(fs, a, tx, ty, t);
                    for (int t = 0; t < kTileThreads; t++) ovf |= dwt_fast_rows_step2(fs, a, t);
                    for (int t = 0; t < kTileThreads; t++) ovf |= dwt_fast_cols_step1(fs, a, t);
                    for (int t = 0; t < kTileThreads; t++) ovf |= dwt_fast_cols_step2(fs, a, tx, ty, t);
                    g_dwt_fast_tiles++;
                    continue;
                }
                for (int t = 0; t < kTileThreads; t++) dwt_tile_load(sh, a, tx, ty, t);
                for (int t = 0; t < kTileThreads; t++) ovf |= dwt_tile_rows_step1(sh, a, tx, ty, t);
                for (int t = 0; t < kTileThreads; t++) ovf |= dwt_tile_rows_step2(sh, a, tx, ty, t);
                for (int t = 0; t < kTileThreads; t++) ovf |= dwt_tile_cols_step1(sh, a, tx, ty, t);
                for (int t = 0; t < kTileThreads; t++) ovf |= dwt_tile_cols_step2(sh, a, tx, ty, t);
            }
        a.src = a.ll; a.src_stride = a.ll_stride;
        off += (size_t)nlw * nlh;
        cw = nlw; ch = nlh;
    }
    return ovf ? kIntegerOverflow : kOk;
}

// whole-frame pipeline with the structure of api.hip::enqueue, on host memory
extern "C" int emu_compress_bits(uint16_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                                 int segments, size_t quota, unsigned bits_per_pixel, uint8_t *out, size_t *size_used,
                                 int *bound_overflow, int sample_bits);
extern "C" int emu_compress(uint16_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                            int segments, size_t quota, unsigned bits_per_pixel, uint8_t *out, size_t *size_used,
                            int *bound_overflow)
{
    return emu_compress_bits(planes, channels, w, h, stages, filt, segments, quota, bits_per_pixel, out, size_used, bound_overflow, 16);
}
// sample_bits = 8: the uint8 twins; `planes` then hold the int8 samples sign-extended (what widen_s8_kernel produces)
extern "C" int emu_compress_bits(uint16_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                                 int segments, size_t quota, unsigned bits_per_pixel, uint8_t *out, size_t *size_used,
                                 int *bound_overflow, int sample_bits)
{
    *size_used = 0;
    *bound_overflow = 0;
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments, sample_bits);
    if (rc) return rc;
    for (int c = 0; c < channels; c++)
        if ((rc = emu_dwt_lim(planes[c], w, h, stages, filt, sample_bits == 8 ? 127 : 32767, sample_bits)) != kOk) return rc;
    const size_t llw = dim_low(w, stages), llh = dim_low(h, stages);
    uint16_t means[3];
    for (int c = 0; c < channels; c++) {
        unsigned long long sum = 0;
        for (size_t r = 0; r < llh; r++)
            for (size_t x = 0; x < llw; x++) sum += planes[c][r * w + x] & (sample_bits == 8 ? 0xFFu : 0xFFFFu);   // ll_sum_kernel
        means[c] = sample_bits == 8 ? (uint16_t)(uint8_t)(sum / (llw * llh)) : (uint16_t)(sum / (llw * llh));     // ll_mean_kernel
    }
    for (int c = 0; c < channels; c++)
        if (means[c] > (sample_bits == 8 ? 127 : 32767)) return kIntegerOverflow;
    for (int c = 0; c < channels; c++)
        for (size_t r = 0; r < llh; r++)
            for (size_t x = 0; x < llw; x++) {   // finalize_ll_kernel (the detail bands: sign-magnitude words since the DWT store)
                const int16_t v = (int16_t)((int16_t)planes[c][r * w + x] - (int16_t)means[c]);
                planes[c][r * w + x] = (uint16_t)to_coder_word(v, sample_bits);
            }

    assign_slots(&plan, quota, bits_per_pixel);
    std::vector<uint8_t> slots(plan.slot_bytes + 8);
    const uint32_t n_units = (uint32_t)plan.units.size();
    std::vector<uint32_t> unit_bits(n_units);
    memset(&g_sh, 0x5A, sizeof g_sh);
    build_coder_tables(&g_sh.tab);
    build_crc_table(g_sh);
    for (uint32_t wi = 0; wi < n_units; wi++) {
        const uint32_t ui = plan.work_order[wi];
        const UnitDesc &u = plan.units[ui];
        uint32_t *slot_words = (uint32_t *)(slots.data() + u.slot_off);
        UnitArgs a;
        a.seg = planes[u.chan] + (size_t)u.y0 * w + u.x0;
        a.stride = (uint32_t)w; a.w = u.w; a.h = u.h; a.subband = (int)u.subband; a.lsb = (int)u.lsb;
        a.out_words = slot_words + kHeaderBytes / 4;
        a.cap_words = u.cap_words;
        a.timers = nullptr;
        EmuEvents evs;
        if (!g_use_wg) evs.build(a);
        const uint32_t bits = g_use_wg ? wg_unit(a) : code_unit_emu(g_sh, a);
        if (bits != kUnitTooBig) {
            FinishArgs f;
            f.slot_words = slot_words; f.bits = bits; f.mean = means[u.chan];
            f.level = u.level; f.subband = u.subband; f.seg = u.seg; f.lsb = u.lsb; f.chan = u.chan;
            f.image_w = (uint32_t)w; f.image_h = (uint32_t)h;
            finish_unit_wave(g_sh, f);
        }
        unit_bits[ui] = bits;
    }
    std::vector<uint64_t> final_off(n_units);
    uint32_t kept;
    uint64_t used;
    rc = scan_frame_wave(unit_bits.data(), plan.final_order.data(), n_units, quota, final_off.data(), &kept, &used);
    if (kept < n_units && unit_bits[kept] == kUnitTooBig && plan.units[kept].cap_is_bound) *bound_overflow = 1;
    for (uint32_t ui = 0; ui < n_units; ui++) {
        if (final_off[ui] == ~0ull) continue;
        const size_t len = kHeaderBytes + ((unit_bits[ui] + 7u) >> 3);
        memcpy(out + final_off[ui], slots.data() + plan.units[ui].slot_off, len);
    }
    *size_used = (size_t)used;
    return rc;
}

// planner taps for the host-logic tests
extern "C" int emu_plan_units(size_t w, size_t h, int channels, int stages, int segments, uint32_t *out /* n*9 */, int cap)
{
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments);
    if (rc) return rc;
    int n = (int)plan.units.size();
    for (int i = 0; i < n && i < cap; i++) {
        const UnitDesc &u = plan.units[i];
        uint32_t *o = out + (size_t)i * 9;
        o[0] = u.x0; o[1] = u.y0; o[2] = u.w; o[3] = u.h; o[4] = u.chan; o[5] = u.level; o[6] = u.subband; o[7] = u.lsb; o[8] = u.seg;
    }
    return n;
}

// the launch orders: work_order (every launch) and, with split_chunks > 0, split_launch (a lone frame: units and sub-range workgroups);
// per entry: unit index, family index of the unit, 1 if the entry is a sub-range workgroup.  Returns the number of entries.
extern "C" int emu_plan_orders(size_t w, size_t h, int channels, int stages, int segments, uint32_t split_chunks, uint32_t *out /* n*3 */, int cap,
                               uint32_t *n_families)
{
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments);
    if (rc) return rc;
    assign_slots(&plan, 2 * w * h * (size_t)channels, 3, split_chunks);
    *n_families = plan.n_families;
    const std::vector<uint32_t> &ord = split_chunks ? plan.split_launch : plan.work_order;
    int n = (int)ord.size();
    for (int i = 0; i < n && i < cap; i++) {
        const uint32_t e = ord[(size_t)i];
        const bool sub = split_chunks && (e >> 31);
        const uint32_t ui = sub ? plan.subs[e & 0x7FFFFFFFu].unit : e;
        out[3 * (size_t)i] = ui; out[3 * (size_t)i + 1] = plan.units[ui].family; out[3 * (size_t)i + 2] = sub ? 1u : 0u;
    }
    return n;
}

// the work list of chunk_sig_kernel: per entry unit index, block, the unit's chunk-table offset and chunk count, its plane
extern "C" int emu_plan_sig_blocks(size_t w, size_t h, int channels, int stages, int segments, uint32_t *out /* n*5 */, int cap, size_t *sig_bytes)
{
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments);
    if (rc) return rc;
    *sig_bytes = plan.sig_bytes;
    int n = (int)(plan.sig_blocks.size() / 2);
    for (int i = 0; i < n && i < cap; i++) {
        const uint32_t ui = plan.sig_blocks[2 * (size_t)i], blk = plan.sig_blocks[2 * (size_t)i + 1];
        const UnitDesc &u = plan.units[ui];
        uint32_t *o = out + (size_t)i * 5;
        o[0] = ui; o[1] = blk; o[2] = u.sig_off; o[3] = (u.w * u.h + 63u) / 64u; o[4] = u.lsb;
    }
    return n;
}

// the sub-range size a lone frame of this geometry gets on a chip of n_cus compute units (plan.hpp auto_split_chunks) and the extra workgroups it makes
extern "C" int emu_auto_split(size_t w, size_t h, int channels, int stages, int segments, int n_cus, uint32_t *chunks, uint32_t *extra)
{
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments);
    if (rc) return rc;
    *chunks = auto_split_chunks(plan.units, n_cus, kPlanes);
    assign_slots(&plan, 2 * w * h * (size_t)channels, 3, *chunks);
    *extra = (uint32_t)plan.subs.size();
    return 0;
}

// the workgroup -> (frame, launch position) map of a batch's pipeline kernel (plan.hpp position_major)
extern "C" void emu_position_major(uint32_t b, uint32_t per_frame, uint32_t n_frames, uint32_t *frame, uint32_t *lpos)
{
    position_major(b, per_frame, n_frames, frame, lpos);
}

// every unit's family: per unit family index, chunk-table offset, x0, y0, w, h (the units of a family must share the rectangle)
extern "C" int emu_plan_families(size_t w, size_t h, int channels, int stages, int segments, uint32_t *out /* n*6 */, int cap, uint32_t *n_families)
{
    Plan plan;
    int rc = build_plan(&plan, w, h, channels, stages, segments);
    if (rc) return rc;
    *n_families = plan.n_families;
    int n = (int)plan.units.size();
    for (int i = 0; i < n && i < cap; i++) {
        const UnitDesc &u = plan.units[(size_t)i];
        uint32_t *o = out + (size_t)i * 6;
        o[0] = u.family; o[1] = u.sig_off; o[2] = u.x0; o[3] = u.y0; o[4] = u.w; o[5] = u.h;
    }
    return n;
}

// statistics tap (tools/event_stats.py): the events of one coding unit by coder bin -- the pixel and count stages alone, chunk by chunk.
// hist[bin] = magnitude-bit events, hist[17 + bin] = sign events, hist[34] = chunks, hist[35] = blank chunks,
// hist[36] = chunks without an event of bins 1..7, hist[37] = chunks with a sign event in a Golomb bin
extern "C" void emu_unit_bin_hist(const uint16_t *seg, size_t w, size_t h, size_t stride, int subband, int lsb, unsigned long long *hist)
{
    memset(&g_sh, 0xA5, sizeof g_sh);
    build_coder_tables(&g_sh.tab);
    UnitArgs a;
    a.seg = seg; a.stride = (uint32_t)stride; a.w = (uint32_t)w; a.h = (uint32_t)h; a.subband = subband; a.lsb = lsb;
    a.cap_words = 0; a.out_words = nullptr; a.timers = nullptr;
    EmuEvents evs;
    evs.build(a);
    unit_state_init(g_sh);
    PixelWave pw;
    CountWave cs;
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    for (uint32_t j = 0; j < nchunks; j++) {
        g_sh.b_done = j;
        pixel_wave_run(g_sh, a, pw, j, j + 1, 0, 1);
        count_wave_run(g_sh, a, cs, j, j + 1, 1);
        const EventSlot &q = g_sh.eq[j % kQueueDepth];
        bool v2v = false, sg = false;
        for (int l = 0; l < 64; l++) {
            if (q.ev1[l] & 0x80) { hist[q.ev1[l] & 31]++; if ((q.ev1[l] & 31) >= 1 && (q.ev1[l] & 31) <= 7) v2v = true; }
            if (q.ev2[l] & 0x80) { hist[17 + (q.ev2[l] & 31)]++; if ((q.ev2[l] & 31) >= 1 && (q.ev2[l] & 31) <= 7) v2v = true; if ((q.ev2[l] & 31) >= 8) sg = true; }
        }
        hist[34]++; hist[35] += q.blank ? 1 : 0; hist[36] += v2v ? 0 : 1; hist[37] += sg ? 1 : 0;
    }
}

// raw copy of the product's coder tables (csrc/icer_tables.hpp) for tests/test_tables.py
extern "C" size_t emu_get_tables(void *dst, size_t cap)
{
    CoderTables t;
    build_coder_tables(&t);
    if (cap >= sizeof t) memcpy(dst, &t, sizeof t);
    return sizeof t;
}
extern "C" int emu_pick_bin(uint32_t zero, uint32_t total)
{
    static CoderTables t;
    static bool built = false;
    if (!built) { build_coder_tables(&t); built = true; }
    return (int)pick_bin(t.binlut, zero, total);
}
