// decoder_emu.cpp -- TEST ONLY.  The decoder's device functions (decoder_core.hpp) and host planner (decoder_plan.hpp)
// compiled by g++ and driven the way decoder.hip drives them on the GPU: one "thread" per candidate, per chain, per
// line.  Lets tests/test_emu_decoder.py check the device code against the decoder oracle without a GPU.
#include "../../icer_compression_amd/csrc/decoder_wave.hpp"      // (lane-loop build of the SPMD macros: -DICER_WAVE_EMU)
#include "../../icer_compression_amd/csrc/decoder_planes.hpp"
#include "../../icer_compression_amd/csrc/decoder_core.hpp"
#include "../../icer_compression_amd/csrc/decoder_plan.hpp"
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace icer;

static int g_lockstep = 0;
static unsigned long long g_stats[4];

// The planes of a chain side by side, in lock step (what a group of lanes -- one per packet -- would do): every
// iteration each plane whose upper neighbour is far enough ahead (plane_ready) decodes one sample.  Checks the
// dependency rule and the roll-back after a failing plane (decoder_core.hpp) against the serial order.
static void decode_chain_lockstep(uint16_t *plane, size_t stride, const ChainDesc &c, int subband, const uint8_t *stream,
                                  uint32_t stream_len, const DecoderTables &t, int planes, int sign_bit, unsigned long long *stats)
{
    PlaneDecoder pd[kPlanes];
    PlaneStorage store[kPlanes];
    bool chain_open = true;
    for (int j = 0; j < planes; j++) {
        const int lsb = planes - 1 - j;
        const uint32_t at = c.pkt[lsb];
        plane_attach_local(pd[j], store[j]);
        pd[j].status = 2; pd[j].done = 0; pd[j].lsb = lsb;
        if (at == kNoPacket) chain_open = false;
        if (!chain_open) continue;
        entropy_init(pd[j].d, stream, stream_len, at + (uint32_t)kHeaderBytes, packet_bits(stream, at));
        plane_begin(pd[j], lsb, sign_bit, c.w, c.h);
    }
    uint16_t *seg = plane + c.first;
    for (;;) {
        int st[kPlanes]; uint32_t dn[kPlanes];
        for (int j = 0; j < planes; j++) { st[j] = pd[j].status; dn[j] = pd[j].done; }     // as of the iteration's start
        bool any = false;
        int active = 0;
        for (int jj = 0; jj < planes; jj++) {
            // (mode 2 walks the planes bottom-up inside an iteration: lanes of a wave have no order among themselves,
            // so the result must not depend on it)
            const int j = g_lockstep == 2 ? planes - 1 - jj : jj;
            const bool go = j == 0 ? pd[0].status == 1 : plane_ready(pd[j], st[j - 1], dn[j - 1], c.w, c.h);
            if (!go) continue;
            plane_step(pd[j], seg, c.w, c.h, stride, subband, sign_bit, t);
            any = true; active++;
        }
        if (!any) break;
        if (stats) { stats[0]++; stats[1] += (unsigned long long)active; }
    }
    for (int j = 0; j < planes; j++)
        if (pd[j].status < 0) {                                  // the highest failing plane
            bool below = false;
            for (int k = j + 1; k < planes; k++) below |= pd[k].status != 2 && pd[k].done > 0;
            if (below) { chain_rollback(seg, c.w, c.h, stride, pd[j].lsb, sign_bit); if (stats) stats[2]++; }
            break;
        }
}

// mode 4: the wave-per-plane kernel of decoder_planes.hpp.  Its waves only meet through counters; here they are stepped in
// turns -- round-robin, or in an order drawn from g_order_seed (any order must give the same image) -- and a full round
// without progress is a lock-up (reported through stats[3]).  Chains with a packet too short for the fast entropy path go
// through the lane-per-plane kernel, as in decoder.hip.
static unsigned g_order_seed = 0;
extern "C" void emu_decoder_order_seed(unsigned s) { g_order_seed = s; }
static bool chain_is_fast(const ChainDesc &c, const uint8_t *stream, uint32_t stream_len, const DecoderTables &t, int planes)
{
    if (!t.lut_ok || stream_len < 4u) return false;
    for (int lsb = planes - 1; lsb >= 0 && c.pkt[lsb] != kNoPacket; lsb--)
        if (packet_bits(stream, c.pkt[lsb]) < kFastPacketBits) return false;
    return c.w > 0 && c.h > 0;
}
static void decode_chain_planes(uint16_t *plane, size_t stride, const ChainDesc &c, const uint8_t *stream, uint32_t stream_len,
                                const DecoderTables &t, int planes, int sign_bit, unsigned long long *stats)
{
    std::vector<uint8_t> lds(pw_lds_bytes(c.w, planes), 0);
    PwShared &sh = *reinterpret_cast<PwShared *>(lds.data());
    uint16_t *zero_row = reinterpret_cast<uint16_t *>(lds.data() + sizeof(PwShared));
    uint16_t *ring = zero_row + pw_ring_pitch(c.w);
    uint32_t nrun = 0;
    while ((int)nrun < planes && c.pkt[planes - 1 - (int)nrun] != kNoPacket) nrun++;
    std::vector<PlaneWave> pw(nrun);
    for (uint32_t j = 0; j < nrun; j++) pw_init(pw[j], j, nrun, c, planes, sign_bit, plane, stride, stream, stream_len, &t);
    unsigned rng = g_order_seed * 2654435761u + 12345u;
    for (;;) {
        bool progress = false, all_done = true;
        for (uint32_t k = 0; k < nrun; k++) {
            uint32_t j = k;
            if (g_order_seed) { rng = rng * 1664525u + 1013904223u; j = (rng >> 16) % nrun; }
            const int st = pw_step(pw[j], sh, zero_row, ring);
            if (st == 1) { progress = true; if (stats) { stats[0]++; } }
        }
        for (uint32_t j = 0; j < nrun; j++) all_done = all_done && pw[j].r >= pw[j].h;
        if (all_done) break;
        if (!progress && !g_order_seed) { if (stats) stats[3]++; break; }      // (a random order may simply not have picked the runnable wave)
    }
    if (stats) stats[1] += (unsigned long long)c.w * c.h * nrun;
}

// mode 1 / 2: decode chains with the lock-step schedule (planes top-down / bottom-up inside an iteration); mode 3: the
// wave kernel of decoder_wave.hpp (LDS row ring).  stats: iterations, samples decoded, roll-backs, chains with rows left
extern "C" void emu_decoder_mode(int lockstep) { g_lockstep = lockstep; g_stats[0] = g_stats[1] = g_stats[2] = g_stats[3] = 0; }
// 1: the code-word table of entropy_decode_fast covers the code (so mode 3 really runs plane_decision's fast path)
extern "C" int emu_decoder_lut_ok(void)
{
    CoderTables ct;
    build_coder_tables(&ct);
    static DecoderTables dt;
    build_decoder_tables(&dt, ct);
    return (int)dt.lut_ok;
}
extern "C" void emu_decoder_stats(unsigned long long *out) { for (int i = 0; i < 4; i++) out[i] = g_stats[i]; }
// the wave-per-plane kernel's zero runs since the process started: [runs taken, decisions they stood for]
extern "C" void emu_decoder_run_stats(unsigned long long *out) { out[0] = icer::g_pw_run_stats[0]; out[1] = icer::g_pw_run_stats[1]; }

// planes[c]: >= bufsize uint16 words; for sample_bits = 8 the low byte of each word is the uint8 result.
extern "C" int emu_decompress(uint16_t *const planes[], int channels, size_t *w, size_t *h, size_t bufsize,
                              const uint8_t *data, size_t len, int stages, int filt, unsigned segments, int sample_bits)
{
    uint32_t crc_tab[256];
    build_crc32_table(crc_tab);
    // candidate kernel: one thread per byte offset; payload kernel: one thread per candidate
    std::vector<PacketCandidate> cands;
    for (uint32_t off = 0; off < (uint32_t)len; off++) {
        PacketCandidate c;
        if (header_candidate(crc_tab, data, (uint32_t)len, off, &c)) cands.push_back(c);
    }
    for (PacketCandidate &c : cands) check_payload(crc_tab, data, &c);
    DecodePlan pl;
    plan_decode(&pl, cands, channels, stages, segments, sample_bits, *w, *h, bufsize);
    *w = pl.w; *h = pl.h;
    if (pl.rc == kInvalidInput || pl.rc == kTooManyStages || pl.rc == kByteQuotaExceeded) return pl.rc;
    const size_t W = pl.w, H = pl.h;
    for (int c = 0; c < channels; c++) memset(planes[c], 0, sizeof(uint16_t) * W * H);
    CoderTables ct;
    build_coder_tables(&ct);
    DecoderTables dt;
    build_decoder_tables(&dt, ct);
    const int nplanes = sample_bits == 8 ? kPlanes8 : kPlanes, sign_bit = sample_bits == 8 ? 7 : 15;
    // chain kernel: one thread per chain
    size_t ring_elems = 2;
    for (const ChainDesc &c : pl.chains) ring_elems = std::max(ring_elems, ring_elems_for(c.w, nplanes));
    std::vector<uint16_t> ring(ring_elems);
    std::vector<uint8_t> state(kStateBytes);
    for (size_t i = 0; i < pl.chains.size(); i++) {
        if (g_lockstep == 4 && chain_is_fast(pl.chains[i], data, (uint32_t)len, dt, nplanes)) decode_chain_planes(planes[pl.chains[i].chan], W, pl.chains[i], data, (uint32_t)len, dt, nplanes, sign_bit, g_stats);
        else if (g_lockstep == 3 || g_lockstep == 4) decode_chain_wave(ring.data(), planes[pl.chains[i].chan], W, pl.chains[i], (int)pl.chains[i].subband, data, (uint32_t)len, dt, nplanes, sign_bit, g_stats, state.data());
        else if (g_lockstep) decode_chain_lockstep(planes[pl.chains[i].chan], W, pl.chains[i], (int)pl.chains[i].subband, data, (uint32_t)len, dt, nplanes, sign_bit, g_stats);
        else {
            PlaneDecoder job;
            PlaneStorage own;
            plane_attach_local(job, own);
            decode_chain(job, planes[pl.chains[i].chan], W, pl.chains[i], (int)pl.chains[i].subband, data, (uint32_t)len, dt, nplanes, sign_bit);
        }
    }
    if (!pl.transform) return pl.rc;
    const FilterTaps taps = filter_taps(filt);
    std::vector<int16_t> tmp(W * H);
    std::vector<uint32_t> pos_col, pos_row;
    for (int c = 0; c < channels; c++) {
        int16_t *s = (int16_t *)planes[c];
        for (size_t i = 0; i < W * H; i++) s[i] = from_sign_magnitude(planes[c][i], sign_bit);
        const size_t llw = dim_low(W, stages), llh = dim_low(H, stages);
        for (size_t r = 0; r < llh; r++)
            for (size_t x = 0; x < llw; x++) s[r * W + x] = add_ll_mean(s[r * W + x], pl.mean[c], sample_bits);
        for (const DecodeLevel &lv : pl.levels) {
            pos_col.resize(lv.ch); pos_row.resize(lv.cw);
            interleave_positions(lv.ch, sample_bits, pos_col.data());
            interleave_positions(lv.cw, sample_bits, pos_row.data());
            for (uint32_t x = 0; x < lv.cw; x++) idwt_line(s + x, tmp.data() + x, lv.ch, W, taps, sample_bits, pos_col.data());
            for (uint32_t r = 0; r < lv.ch; r++) idwt_line(tmp.data() + (size_t)r * W, s + (size_t)r * W, lv.cw, 1, taps, sample_bits, pos_row.data());
        }
        for (size_t i = 0; i < W * H; i++) {                    // icer_remove_negative_*, icer_util.c:70-91
            if (s[i] < 0) s[i] = 0;
            if (sample_bits == 8) planes[c][i] &= 0xFFu;
        }
    }
    return pl.rc;
}
