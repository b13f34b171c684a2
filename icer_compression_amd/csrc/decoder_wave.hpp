// decoder_wave.hpp -- one wavefront per chain, one LANE per packet: the bit planes of a segment decoded side by side
// (decoder_core.hpp, plane_ready) over a ring of rows in LDS.  Written with the SPMD macros of wave.hpp, so the same
// source runs in the CPU lane-loop build (tests/emu/decoder_emu.cpp) where its ring / retire / roll-back logic is
// checked against the decoder oracle.
//
// STATUS: the default decode kernel (decoder.hip uses it whenever the segment rows fit the LDS ring; ICER_DEC_WAVE=0
// selects the thread-per-chain kernel).  Bit-exact on the CPU build and on an MI355X (tests/test_gpu_decoder.py);
// 1.28 s for the 160 chains of the 4096 x 4096 headline frame (profiles/archive/r02_decoder_v1_rocprof.md), HISTORY.md 6b (summary: DESIGN.md 8).
//
// Lane j < planes decodes plane planes-1-j.  All lanes are in one wavefront, so there are no waits: every iteration a
// lane either decodes its next sample or sits out (its upper neighbour is not far enough ahead, or the ring has no room).
// The samples live in a ring of rows (ring_rows_for): the most advanced plane reads row r + 1, the least advanced row r - 1,
// and consecutive planes are about one row apart, so the live rows (planes + 2 of them) fit the ring; a row that no
// running plane can still touch is written back to the channel plane by the whole wave and its slot is zeroed for the row
// `rows` further down.
#pragma once
#include "wave.hpp"

#include "decoder_core.hpp"

// global-memory fence of the (rare) roll-back pass, which reads back samples other lanes of this wave have stored
#ifdef ICER_WAVE_EMU
#define ICER_DEC_GLOBAL_FENCE()
#else
#define ICER_DEC_GLOBAL_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent")
#endif

namespace icer {

// Rows of a chain's ring.  The live rows are planes + 2 (the lowest plane still reads its row - 1, the highest its row
// + 1, consecutive planes are a row and two samples apart); the schedule cannot lock up -- every plane waiting for the one
// above and the top one for a ring slot -- when (rows - planes - 2) * w >= 2 * (planes - 1), and one row more keeps the
// top plane from waiting on every retired row.  Every row costs LDS, which decides how many chains a compute unit holds.
ICER_HD uint32_t ring_rows_for(uint32_t w, int planes)
{
    const uint32_t p = (uint32_t)planes, ww = w ? w : 1u;
    return p + 3u + (2u * (p - 1u) + ww - 1u) / ww;
}
ICER_HD uint32_t ring_pitch_for(uint32_t w) { return w < 2u ? 2u : (w + 1u) & ~1u; }          // (even, >= w)
ICER_HD size_t ring_elems_for(uint32_t w, int planes) { return (size_t)ring_rows_for(w, planes) * ring_pitch_for(w); }
constexpr uint32_t kBurst = 16;             // samples a plane may take per look at its neighbours
constexpr uint32_t kStateColumns = 10;      // kPlanes + 1: a chunk of per-bin / per-context arrays for every plane, one for the idle lanes
static_assert(kStateColumns > (uint32_t)kPlanes, "a chunk per plane and one spare");
constexpr size_t kStateBytes = ((size_t)kStateColumns * kPlaneChunkBytes + 15u) & ~(size_t)15;

struct RingImage {
    uint16_t *ring; uint32_t pitch, rows;
    ICER_HD uint32_t at(uint32_t r, uint32_t c) const { return ring[(r % rows) * pitch + c]; }
    ICER_HD void put(uint32_t r, uint32_t c, uint32_t v) { ring[(r % rows) * pitch + c] = (uint16_t)v; }
    ICER_HD uint32_t row_at(uint32_t r) const { return (r % rows) * pitch; }
    ICER_HD uint32_t row_after(uint32_t row) const { return row + pitch == rows * pitch ? 0u : row + pitch; }
    ICER_HD uint32_t at_row(uint32_t row, uint32_t c) const { return ring[row + c]; }
    ICER_HD void put_row(uint32_t row, uint32_t c, uint32_t v) { ring[row + c] = (uint16_t)v; }
};

// `ring`: ring_elems_for(c.w, planes) words of LDS; `plane`: the channel plane (zero where not yet decoded);
// `state_block`: kStateBytes of LDS for the lanes' per-bin / per-context arrays.
// stats (tests): [0] iterations, [1] samples decoded, [2] roll-backs, [3] chains that ended with rows not retired.
ICER_DEV void decode_chain_wave(uint16_t *ring, uint16_t *plane, size_t stride, const ChainDesc &c,
                                int subband, const uint8_t *stream, uint32_t stream_len, const DecoderTables &t,
                                int planes, int sign_bit, unsigned long long *stats, uint8_t *state_block)
{
    DECL_LANE;
    const uint32_t w = c.w, h = c.h;
    const uint32_t pitch = ring_pitch_for(w), rows = ring_rows_for(w, planes);
    uint16_t *seg = plane + c.first;
    // planes that can run: from the top down to the first missing packet (icer_partition.c:431)
    int nrun = 0;
    while (nrun < planes && c.pkt[planes - 1 - nrun] != kNoPacket) nrun++;
    LANEVAR(PlaneDecoder, pd);
    FOR_LANES
    {
        // (lanes that run no plane share a column they never touch)
        plane_attach_chunk(LV(pd), state_block + (size_t)(lane < nrun ? lane + 1 : 0) * kPlaneChunkBytes);
        for (uint32_t i = (uint32_t)lane; i < rows * pitch; i += 64) ring[i] = 0;
        LV(pd).status = 2; LV(pd).done = 0; LV(pd).r = 0; LV(pd).c = 0; LV(pd).lsb = 0;
        if (lane < nrun) {
            const int lsb = planes - 1 - lane;
            const uint32_t at = c.pkt[lsb];
            entropy_init(LV(pd).d, stream, stream_len, at + (uint32_t)kHeaderBytes, packet_bits(stream, at));
            plane_begin(LV(pd), lsb, sign_bit, w, h);
            if (LV(pd).d.total_bits >= kFastPacketBits && t.lut_ok) plane_fast_begin(LV(pd));
        }
    }
    uint32_t retired = 0, retire_at = 0;                     // rows written back so far, and the ring slot of the next one (wave-uniform)
    for (;;) {
        WAVE_SYNC();
        // where every plane stands at the start of the iteration; its upper neighbour's state comes from the lane below
        LANEVAR(uint32_t, st); LANEVAR(uint32_t, dn); LANEVAR(uint32_t, up_lane);
        FOR_LANES
        {
            LV(st) = (uint32_t)LV(pd).status; LV(dn) = LV(pd).done; LV(up_lane) = lane > 0 ? (uint32_t)lane - 1u : 0u;
        }
        // a plane under a failed or cancelled one never runs (again).  (A finished plane has only finished planes above it.)
        const uint64_t dead = BALLOT(lane < nrun && ((int)LV(st) < 0 || LV(st) == 2u));
        FOR_LANES
        {
            if (lane < nrun && LV(st) == 1u && (dead & ((1ull << lane) - 1ull)) != 0ull) { LV(st) = 2u; LV(pd).status = 2; }
        }
        LANEVAR(uint32_t, ast); LANEVAR(uint32_t, adn);
        WAVE_GATHER(ast, st, up_lane);
        WAVE_GATHER(adn, dn, up_lane);
        // how many samples each plane may take before the next look (kBurst at most): its upper neighbour has to stay
        // w + 2 samples ahead of the last one -- plane_needs(), rounded up -- and the row below must have its ring slot
        LANEVAR(uint32_t, grant); LANEVAR(uint32_t, low);
        FOR_LANES
        {
            uint32_t g = 0;
            const bool running = lane < nrun && LV(st) == 1u;
            if (running) {
                const int above_status = lane > 0 ? (int)LV(ast) : kOk;
                g = kBurst;
                if (above_status == 1) {
                    const uint32_t need = LV(dn) + w + 2u;                        // for the first sample of the burst
                    g = LV(adn) >= need ? (LV(adn) - need + 1u < kBurst ? LV(adn) - need + 1u : kBurst) : 0u;
                } else if (above_status != kOk) g = 0;
                const uint32_t room_end = (retired + rows - 1u) * w;       // samples of rows whose row below has a slot
                const uint32_t room = room_end > LV(dn) ? room_end - LV(dn) : 0u;
                if (room < g) g = room;
            }
            LV(grant) = g;
            LV(low) = LV(pd).r;
        }
        const uint64_t G = BALLOT(LV(grant) != 0u);
        // rows below `limit` are dead: a running plane in row r still reads row r - 1; the lowest running plane is the
        // one furthest behind
        const uint64_t running_now = BALLOT(lane < nrun && LV(st) == 1u);
        uint32_t limit = h;
        if (running_now) { const uint32_t rr = READLANE(low, 63 - clz64(running_now)); limit = rr > 0u ? rr - 1u : 0u; }
        const bool retire = retired < limit;
        if (G == 0ull && !retire) break;
        uint32_t steps = 0;
        // a burst: up to `grant` samples per plane, one decision (a magnitude bit or a sign) per round
        LANEVAR(uint32_t, stop_at);
        FOR_LANES
        {
            LV(stop_at) = LV(pd).done + LV(grant);
        }
        for (uint32_t k = 0; k < 2u * kBurst; k++) {
            const uint64_t A = BALLOT(LV(pd).done < LV(stop_at) && LV(pd).status == 1);
            if (A == 0ull) break;
            FOR_LANES
            {
                if (LV(pd).done < LV(stop_at) && LV(pd).status == 1) {
                    RingImage img{ring, pitch, rows};
                    plane_decision(LV(pd), img, w, h, subband, sign_bit, t);
                }
            }
            WAVE_SYNC();
            steps += (uint32_t)popc64(A);
        }
        // (rows that became dead before this burst; at most as many as the burst can open up again)
        for (uint32_t k = 0; k < kBurst / 4u + 1u && retired < limit; k++) {
            uint16_t *slot = ring + retire_at;
            FOR_LANES
            {
                for (uint32_t x = (uint32_t)lane; x < w; x += 64) { seg[(size_t)retired * stride + x] = slot[x]; slot[x] = 0; }
            }
            retired++;
            retire_at = retire_at + pitch == rows * pitch ? 0u : retire_at + pitch;
        }
        WAVE_SYNC();
        if (stats) { stats[0]++; stats[1] += (unsigned long long)steps; }
    }
    if (stats && retired != h) stats[3]++;
    // a failed plane: the planes below it are taken back (chain_rollback), on the written-back samples
    LANEVAR(uint32_t, st2); LANEVAR(uint32_t, dn2);
    FOR_LANES
    {
        LV(st2) = (uint32_t)LV(pd).status; LV(dn2) = LV(pd).done;
    }
    int failed = -1;
    bool below = false;
    for (int j = 0; j < nrun; j++) {
        const int sj = (int)READLANE(st2, j);
        if (failed < 0) { if (sj < 0) failed = j; }
        else below = below || READLANE(dn2, j) > 0u;
    }
    if (failed >= 0 && below) {
        const int failed_lsb = planes - 1 - failed;
        const uint32_t keep = ((1u << sign_bit) - 1u) & ~((1u << failed_lsb) - 1u);
        ICER_DEC_GLOBAL_FENCE();
        FOR_LANES
        {
            for (uint32_t i = (uint32_t)lane; i < w * h; i += 64) {
                uint16_t *p = seg + (size_t)(i / w) * stride + (i % w);
                const uint32_t v = *p, m = v & keep;
                *p = (uint16_t)(m ? (m | (v & (1u << sign_bit))) : 0u);
            }
        }
        if (stats) stats[2]++;
    }
}

}  // namespace icer
