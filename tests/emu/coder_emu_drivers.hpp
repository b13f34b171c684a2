// coder_emu_drivers.hpp -- TESTS ONLY.  Drivers that run the eight wave roles of csrc/coder_core.hpp on the CPU: one
// thread per wave with real waits (ICER_WAVE_THREADS, tests/emu/threads_main.cpp), all waves interleaved on one thread
// as far ahead as the protocol allows (code_unit_emu), and under a random scheduler (code_unit_emu_random).  Included
// after coder_core.hpp by the test translation units; nothing here is part of the product library.
#pragma once
#include "../../icer_compression_amd/csrc/coder_core.hpp"
#include "../../icer_compression_amd/csrc/assemble_core.hpp"
#include <string.h>
#include <vector>

namespace icer {

// tests only: what family_events_kernel leaves for ONE unit -- the event bytes of its bit plane (events.hpp) and the chunk table of its
// family --, made pixel by pixel from the same per-pixel functions the kernel calls.  Blank chunks keep a poison byte: the pixel wave
// must not read them.
struct EmuEvents {
    std::vector<uint8_t> ev, sig;
    void build(UnitArgs &a)
    {
        const uint32_t w = a.w, h = a.h, npix = w * h, nchunks = (npix + 63u) / 64u, p = (uint32_t)a.lsb;
        const bool is_hl = a.subband == kHL, is_hh = a.subband == kHH;
        uint8_t tab[48];
        for (uint32_t i = 0; i < 45u; i++) tab[i] = (uint8_t)ev_ctx_entry(is_hh, i);
        ev.assign((size_t)nchunks * 64u, 0xCD);
        sig.assign(nchunks + 64u, 0xEE);
        auto at = [&](uint32_t r, uint32_t c) { return (uint32_t)a.seg[(size_t)r * a.stride + c]; };
        for (uint32_t j = 0; j < nchunks; j++) {
            uint32_t tmax = 0;
            uint8_t bytes[64];
            for (uint32_t l = 0; l < 64u; l++) {
                const uint32_t np = j * 64u + l;
                if (np >= npix) { bytes[l] = (uint8_t)kEvNone; tmax = 255; continue; }
                const uint32_t r = np / w, c = np - r * w;
                const bool hW = c > 0, hE = c + 1 < w, hN = r > 0, hS = r + 1 < h;
                const PixelLens L = pixel_lens(at(r, c), hW ? at(r, c - 1) : 0u, hE ? at(r, c + 1) : 0u, hN ? at(r - 1, c) : 0u, hS ? at(r + 1, c) : 0u,
                                               (hN && hW) ? at(r - 1, c - 1) : 0u, (hN && hE) ? at(r - 1, c + 1) : 0u,
                                               (hS && hW) ? at(r + 1, c - 1) : 0u, (hS && hE) ? at(r + 1, c + 1) : 0u);
                if (tmax != 255u && L.t > tmax) tmax = L.t;
                bytes[l] = (uint8_t)event_byte(L, p, is_hl, is_hh, tab);
            }
            sig[j] = (uint8_t)tmax;
            if (tmax > p) memcpy(&ev[(size_t)j * 64u], bytes, 64);
        }
        a.ev = ev.data();
        a.sig = sig.data();
    }
};

// the shape of the emulated workgroup: pixel waves (1 or 2) and golomb workers (0 = one golomb wave without a state wave,
// 2 = state wave + two workers), as code_units_kernel<8> / <11>
static uint32_t g_emu_npw = 2, g_emu_ngw = 2;

#if defined(ICER_WAVE_EMU) && defined(ICER_WAVE_THREADS)
// tests only: one CPU thread per wave, the roles started exactly like code_units_kernel starts them
static inline uint32_t code_unit_threads(CoderShared &s, const UnitArgs &a)
{
    unit_state_init(s);
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    s.nchunks = nchunks;
    uint32_t bits = kUnitTooBig;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    std::thread t[8], tp[4];
    for (uint32_t k = 1; k < g_emu_npw; k++) tp[k] = std::thread([&s, &a, nchunks, k] { PixelWave pw; pixel_wave_run(s, a, pw, 0, nchunks, k, g_emu_npw); });
    t[0] = std::thread([&] { PixelWave pw; pixel_wave_run(s, a, pw, 0, nchunks, 0, g_emu_npw); });
    t[1] = std::thread([&] { CountWave cs; count_wave_run(s, a, cs, 0, nchunks, g_emu_npw); });
    t[2] = std::thread([&] { compact_wave_run(s, a, 0, nchunks); });
    t[3] = std::thread([&] { WalkWave ww; walk_wave_init(s, ww); walk_wave_run(s, a, ww, nchunks, ~0u); });
    t[4] = std::thread([&] { GolombWave gw; golomb_wave_init(gw); if (g_emu_ngw) golomb_state_run(s, a, gw, nchunks, ~0u); else golomb_wave_run(s, a, gw, nchunks, ~0u, 0, 0); });
    std::thread tg[4];
    for (uint32_t k = 0; k < g_emu_ngw; k++) tg[k] = std::thread([&s, &a, nchunks, k] { GolombWave gw; golomb_wave_init(gw); golomb_wave_run(s, a, gw, nchunks, ~0u, k, g_emu_ngw); });
    t[5] = std::thread([&] { RecordsWave rw; records_wave_run(s, a, rw, ~0u); });
    t[6] = std::thread([&] { drain_wave_run(s, a, ~0u); });
    t[7] = std::thread([&] { bits = merge_wave_run(s, a, 0, nchunks) ? merge_wave_finish(s, a) : kUnitTooBig; });
    for (auto &th : t) th.join();
    for (uint32_t k = 1; k < g_emu_npw; k++) tp[k].join();
    for (uint32_t k = 0; k < g_emu_ngw; k++) tg[k].join();
    if (__atomic_load_n(&s.abort, __ATOMIC_RELAXED) == 2u) bits = kUnitFailed;
    return bits;
}
#elif defined(ICER_WAVE_EMU)
// tests only: the seven waves interleaved on one CPU thread.  Each wave runs as far ahead as the queues and
// the speculation rule allow, so slot reuse and the discard/reload protocol are exercised, not just the
// lock-step order.
static inline uint32_t code_unit_emu(CoderShared &s, const UnitArgs &a)
{
    unit_state_init(s);
    PixelWave pw[4];
    CountWave cs;
    WalkWave ww;
    GolombWave gs, gw[4];
    RecordsWave rw;
    walk_wave_init(s, ww);
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    golomb_wave_init(gs);
    for (auto &g : gw) golomb_wave_init(g);
    uint32_t jp = 0, ja = 0, jc = 0, jb = 0;
    while (jb < nchunks) {
        while (jp < nchunks && jp < s.a_done + kQueueDepth) { pixel_wave_run(s, a, pw[jp % g_emu_npw], jp, jp + 1, jp % g_emu_npw, g_emu_npw); jp++; }
        while (ja < jp && ja < s.b_done + kQueueDepth) { count_wave_run(s, a, cs, ja, ja + 1, g_emu_npw); ja++; }
        while (jc < ja) { compact_wave_run(s, a, jc, jc + 1); jc++; }
        // both speculating waves run as far ahead as events allow (and roll back when told to)
        walk_wave_run(s, a, ww, nchunks, kQueueDepth);
        if (g_emu_ngw) golomb_state_run(s, a, gs, nchunks, kQueueDepth); else golomb_wave_run(s, a, gs, nchunks, kQueueDepth, 0, 0);
        for (uint32_t k = 0; k < g_emu_ngw; k++) golomb_wave_run(s, a, gw[k], nchunks, kQueueDepth, k, g_emu_ngw);
        records_wave_run(s, a, rw, kQueueDepth);
        if (!merge_wave_run(s, a, jb, jb + 1)) return kUnitTooBig;
        jb++;
        drain_wave_run(s, a, jb & 1u);              // the drain lags behind the merge wave on purpose
        if (s.abort) return kUnitTooBig;
    }
    return merge_wave_finish(s, a);
}

// tests only: ONE workgroup of a split unit (coder_core.hpp "Sub-ranges"): the counts-only prefix pass over [0, j0) with
// eight pixel waves, then the pipeline from j0 -- the waves interleaved as in code_unit_emu -- until its state equals a
// later sub-range's snapshot or the unit ends.  Leaves its SubRecord.
static inline void code_subrange_emu(CoderShared &s, const UnitArgs &a)
{
    const SubLayout &L = *a.sub;
    const uint32_t j0 = L.first[L.index];
    unit_state_init(s);
    PixelWave pw[8];
    CountWave cs;
    if (j0) {
        uint32_t jc0 = 0;
        for (uint32_t j = 0; j < j0; j++) {
            pixel_prefix_run(s, a, pw[j % 7u], j, j + 1, j % 7u, 7u);
            if (j % 5u == 4u || j + 1 == j0) { count_prefix_run(s, a, cs, jc0, j + 1, 7u); jc0 = j + 1; }     // (the count wave takes several chunks per look)
        }
        unit_state_init(s, j0);
    }
    PixelWave qw[4];
    WalkWave ww;
    GolombWave gs, gw[4];
    RecordsWave rw;
    rw.next = j0;
    walk_wave_init(s, ww, j0);
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    golomb_wave_init(gs, j0);
    for (auto &g : gw) golomb_wave_init(g, j0);
    uint32_t jp = j0, ja = j0, jc = j0, jb = j0;
    uint32_t bits = 0;
    bool ended = false;
    while (jb < nchunks) {
        while (jp < nchunks && jp < s.a_done + kQueueDepth) { pixel_wave_run(s, a, qw[jp % g_emu_npw], jp, jp + 1, jp % g_emu_npw, g_emu_npw); jp++; }
        while (ja < jp && ja < s.b_done + kQueueDepth) { count_wave_run(s, a, cs, ja, ja + 1, g_emu_npw); ja++; }
        while (jc < ja) { compact_wave_run(s, a, jc, jc + 1); jc++; }
        walk_wave_run(s, a, ww, nchunks, kQueueDepth);
        if (g_emu_ngw) golomb_state_run(s, a, gs, nchunks, kQueueDepth); else golomb_wave_run(s, a, gs, nchunks, kQueueDepth, 0, 0);
        for (uint32_t k = 0; k < g_emu_ngw; k++) golomb_wave_run(s, a, gw[k], nchunks, kQueueDepth, k, g_emu_ngw);
        records_wave_run(s, a, rw, kQueueDepth);
        const uint32_t how = merge_wave_run(s, a, jb, jb + 1);
        if (how == kMergeMatched) return;
        if (how == kMergeAbandoned) { bits = kUnitTooBig; ended = true; break; }
        jb++;
        drain_wave_run(s, a, jb & 1u);
        if (s.abort) { bits = kUnitTooBig; ended = true; break; }
    }
    if (!ended) bits = merge_wave_finish(s, a);
    SubRecord &r = L.rec[L.index];
    r.end_chunk = nchunks; r.end_bits = bits; r.match_sub = 0; r.match_snap = 0; r.done = 1;
}

// tests only: a unit cut into n_sub sub-ranges.  order 0: the workgroups run last sub-range first (every snapshot a
// workgroup could meet exists when it gets there); 1: first sub-range first (no snapshot exists yet: nobody matches,
// workgroup 0 codes the whole unit and the others' work is discarded -- the GPU's worst case); 2: odd ones first.
static inline uint32_t code_unit_emu_split(CoderShared &s, const UnitArgs &a0, uint32_t n_sub, uint32_t order, uint32_t *matches)
{
    const uint32_t nchunks = (a0.w * a0.h + 63u) / 64u;
    std::vector<Snapshot> snaps((size_t)n_sub * kMaxSnaps);
    std::vector<uint32_t> valid((size_t)n_sub * kMaxSnaps, 0u);
    std::vector<SubRecord> rec(n_sub);
    memset(rec.data(), 0, rec.size() * sizeof(SubRecord));
    std::vector<std::vector<uint32_t>> priv(n_sub);
    uint32_t *words[kMaxSubs];
    words[0] = a0.out_words;
    SubLayout L;
    L.n_sub = n_sub;
    for (uint32_t i = 0; i <= n_sub; i++) L.first[i] = (uint32_t)((uint64_t)nchunks * i / n_sub);
    L.snaps = snaps.data(); L.snap_valid = valid.data(); L.rec = rec.data();
    std::vector<uint32_t> seq;
    for (uint32_t i = 0; i < n_sub; i++) seq.push_back(order == 0 ? n_sub - 1 - i : i);
    if (order == 2) { seq.clear(); for (uint32_t i = 1; i < n_sub; i += 2) seq.push_back(i); for (uint32_t i = 0; i < n_sub; i += 2) seq.push_back(i); }
    for (uint32_t i : seq) {
        UnitArgs a = a0;
        L.index = i;
        a.sub = &L;
        if (i) {
            priv[i].assign((size_t)a0.cap_words + 64, 0xDEADBEEFu);
            words[i] = priv[i].data();
            a.out_words = priv[i].data();
            a.cap_words = a0.cap_words;
        }
        memset(&s.stage, 0xA5, sizeof s.stage);
        code_subrange_emu(s, a);
    }
    if (matches) { *matches = 0; for (uint32_t i = 0; i < n_sub; i++) if (rec[i].match_sub) (*matches)++; }
    return splice_unit_wave(n_sub, rec.data(), snaps.data(), words, a0.cap_words);
}

// tests only: the same eight waves under a RANDOM scheduler -- at every step one wave is picked at random and runs one
// chunk if its inputs are there (exactly the conditions the GPU waits for), so waves lag and lead each other in ways
// the deterministic order above never produces.  A state in which no wave can move is a protocol deadlock: reported as
// kUnitFailed.
static inline uint32_t code_unit_emu_random(CoderShared &s, const UnitArgs &a, uint32_t seed)
{
    unit_state_init(s);
    PixelWave pw[4];
    CountWave cs;
    WalkWave ww;
    GolombWave gs, gw[4];
    RecordsWave rw;
    walk_wave_init(s, ww);
    golomb_wave_init(gs);
    for (auto &g : gw) golomb_wave_init(g);
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    uint32_t jpw[4] = {0, 1, 2, 3};                  // next chunk of every pixel wave
    (void)jpw;
    uint32_t ja = 0, jc = 0, jb = 0, idle = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ seed;
    while (jb < nchunks) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t pick = (uint32_t)(rng >> 33) % 16u;
        bool moved = false;
        switch (pick) {
        case 0: case 8: case 9: case 10: {
            const uint32_t k = pick == 0 ? 0u : pick - 7u;           // the pixel waves lag and lead each other too
            if (k >= g_emu_npw) break;
            uint32_t &jp = jpw[k];
            if (jp < nchunks && jp < s.a_done + kQueueDepth) { pixel_wave_run(s, a, pw[k], jp, jp + 1, k, g_emu_npw); jp += g_emu_npw; moved = true; }
        } break;
        case 1: if (ja < s.p_done[ja % g_emu_npw] && ja < s.b_done + kQueueDepth) { count_wave_run(s, a, cs, ja, ja + 1, g_emu_npw); ja++; moved = true; } break;
        case 2: if (jc < s.a_done) { compact_wave_run(s, a, jc, jc + 1); jc++; moved = true; } break;
        case 3: moved = walk_wave_run(s, a, ww, nchunks, 1u) != 0; break;
        case 4: moved = (g_emu_ngw ? golomb_state_run(s, a, gs, nchunks, 1u) : golomb_wave_run(s, a, gs, nchunks, 1u, 0, 0)) != 0; break;
        case 11: case 12: { const uint32_t k = pick - 11u; if (k < g_emu_ngw) moved = golomb_wave_run(s, a, gw[k], nchunks, 1u, k, g_emu_ngw) != 0; } break;
        case 5: { const uint32_t before = rw.next, g = rw.gen; records_wave_run(s, a, rw, 1u); moved = rw.next != before || rw.gen != g; } break;
        case 6: {
            const RecSlot &rq = s.rq[jb % kQueueDepth];
            const uint32_t tag = chunk_tag(jb, s.exact_seq);
            if (rq.gtag == tag && rq.rtag == tag) {
                if (!merge_wave_run(s, a, jb, jb + 1)) return kUnitTooBig;
                jb++;
                moved = true;
            }
        } break;
        case 7: case 15: { const uint32_t before = s.popped; drain_wave_run(s, a, 1u); moved = s.popped != before; } break;
        default: break;
        }
        if (s.abort) return kUnitTooBig;
        idle = moved ? 0u : idle + 1u;
        if (idle > 100000u) return kUnitFailed;
    }
    return merge_wave_finish(s, a);
}
#endif


}  // namespace icer
