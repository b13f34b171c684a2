// coder_core.hpp -- one wavefront codes one ICER coding unit (channel, level, subband, plane, segment).
//
// Replaces, for the uint16 path, the reference's per-segment chain
//   icer_compress_bitplane_uint16   lib_icer/src/icer_context_modeller.c:312-457
//   icer_encode_bit / icer_compute_bin   icer_encoding.c:37-112, icer_util.c:48-56
//   icer_popbuf_while_avail / icer_flush_encode   icer_encoding.c:114-189
// and must emit the identical payload bits.
//
// Structure per 64-pixel chunk (raster order inside the segment):
//   phase 1  (64 lanes)  pixel category, magnitude bit, 8-neighbour context, sign context
//   phase 2  (64 lanes)  adaptive counts seen by every event: ballot + v_mbcnt ranks per context,
//                        including the single rescale a context can cross inside a chunk
//   phase 3  (64 lanes)  probability fold + bin selection (16 compare/accumulate steps)
//   phase 4              interleaved entropy coder.  Code words are delimited per bin, allocated to
//                        the 2048-word ring in order of their first event and emitted in that order.
//                        Fast path (the ring cannot fill up inside this chunk): Golomb bins 8..16 in
//                        closed form on ballot masks (run length since the bin's last one, mod m),
//                        bins 1..7 by one walker lane per bin over that bin's events, ring slots from
//                        a prefix count of word-start flags, finished words drained 64 at a time
//                        with a prefix sum of their lengths.  Exact path (ring nearly full, so the
//                        forced flush of the oldest open word, E5, may trigger; also the end-of-unit
//                        flush): one lane replays the reference state machine event by event.
//   drain                finished words are packed LSB-first into an LDS bit stage and whole
//                        32-bit words are written to the unit's payload slot in HBM (coalesced).
// Written with the SPMD macros of wave.hpp (see there for the tests-only CPU build).
#pragma once
#include "icer_tables.hpp"
#include "wave.hpp"

#ifdef ICER_WAVE_EMU
extern unsigned long long g_emu_chunks[2];          // tests only: [0] fast-path chunks, [1] exact-path chunks
#define ICER_EMU_COUNT(i) (g_emu_chunks[i]++)
#else
#define ICER_EMU_COUNT(i)
#endif

// Optional per-phase cycle counters (s_memtime) for tools/phase_profile.py; compiled in only with
// -DICER_PHASE_TIMERS (a separate profiling build of the library, never the shipped one).
#if defined(ICER_PHASE_TIMERS) && !defined(ICER_WAVE_EMU)
#define ICER_NUM_TIMERS 12
#define ICER_TIMERS_DECL uint64_t tacc_[ICER_NUM_TIMERS] = {}; uint64_t tlast_ = __builtin_amdgcn_s_memtime();
#define ICER_TIMER_PARAMS , uint64_t *tacc_, uint64_t &tlast_
#define ICER_TIMER_PASS , tacc_, tlast_
#define ICER_TICK(k) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); tacc_[k] += t_ - tlast_; tlast_ = t_; }
#define ICER_TIMERS_STORE(dst) { if (dst) { _Pragma("unroll") for (int i_ = 0; i_ < ICER_NUM_TIMERS; i_++) if (lane == 0) atomicAdd((unsigned long long *)&(dst)[i_], (unsigned long long)tacc_[i_]); } }
#else
#define ICER_TIMERS_DECL
#define ICER_TIMER_PARAMS
#define ICER_TIMER_PASS
#define ICER_TICK(k)
#define ICER_TIMERS_STORE(dst)
#endif

namespace icer {

constexpr uint32_t kStageWords = 1024;      // LDS bit stage (circular, 32-bit words)
constexpr uint32_t kUnitTooBig = 0xFFFFFFFFu;

// ring word: open  -> owner bin (bit 15 clear)
//            done  -> 0x8000 | nbits << 11 | code (<= 10 bits)
constexpr uint32_t kWordDone = 0x8000u;

struct CoderShared {
    uint32_t stage[kStageWords];
    uint16_t ring[kRingWords];
    CoderTables tab;
    uint32_t crc_tab[256];
    uint8_t ev[128];            // events of the current chunk in coding order: 0x80 | bit << 5 | bin
    uint8_t evflag[128];        // bins 1..7, written by the walker lanes: bit0 word starts here, bit1 word ends here
    uint8_t evstart[128];       // position of the start event of the word that ends here (255: carried-in word)
    uint8_t binseq[8][128];     // bins 1..7: that bin's events in coding order, bit7 = input bit, bits 6..0 = position
    uint32_t ctx_zero[kNumContexts], ctx_total[kNumContexts];   // adaptive model of the unit (C5)
    uint8_t evacc[128];         // bins 1..7: completed input value of the word that ends here (code looked up by the event lane)
    uint8_t bin_open_pos[32];   // per bin after the chunk: 255 unchanged, 254 closed, else start position of its open word
    int32_t bin_slot[kNumBins]; // ring index of the bin's open word, -1 if none
    uint32_t bin_acc[kNumBins]; // Golomb: zero-run length so far; bins 1..7: partial input value
    uint32_t bin_nin[kNumBins]; // bins 1..7: input bits accumulated
    uint32_t head, used;        // ring state
    uint32_t bitpos;            // payload bits produced so far
    uint32_t flushed_words;     // payload words already written to HBM
    uint32_t resume;            // exact path: event index at which the single-lane replay paused
};

struct UnitArgs {
    const uint16_t *seg;        // first coefficient of the segment (sign-magnitude words)
    uint32_t stride;            // plane row stride in elements
    uint32_t w, h;              // segment size
    int subband, lsb;
    uint32_t *out_words;        // payload slot (4-byte aligned)
    uint32_t cap_words;         // slot capacity in 32-bit words
    uint64_t *timers;           // profiling build only (may be null)
};

// ------------------------------------------------------------------------------------------
// sequential coder steps (executed by a single lane; exact restatement of E1-E6 except that
// draining is left to wave_drain)
// ------------------------------------------------------------------------------------------
// Golomb codeword for a run of k zeros ended by a one (icer_encoding.c:73-80)
ICER_DEV uint32_t golomb_word(const CoderTables &t, int bin, uint32_t k)
{
    const uint32_t gi = t.gi[bin];
    const uint32_t code = k + (k >= gi ? gi : 0u);
    const uint32_t n = t.gl[bin] + (k >= gi ? 1u : 0u);
    return kWordDone | (n << 11) | ((brev32(code) >> (32u - n)) & 0x3FFu);
}

// first half of icer_flush_encode (icer_encoding.c:141-189): force-complete the oldest word.
// The caller drains afterwards (wave_drain).
ICER_DEV void seq_complete_head(CoderShared &s)
{
    const uint32_t w = s.ring[s.head];
    if (!(w & kWordDone)) {
        const int bin = (int)(w & 31u);
        if (bin >= 8) {
            const uint32_t k = s.bin_acc[bin];
            s.ring[s.head] = (uint16_t)((k == (uint32_t)s.tab.gm[bin] - 1u) ? (kWordDone | (1u << 11) | 1u)
                                                                            : golomb_word(s.tab, bin, k));
            s.bin_acc[bin] = 0;
            s.bin_slot[bin] = -1;
        } else if (bin >= 1) {
            const uint32_t nin = s.bin_nin[bin];
            const uint32_t pv = s.bin_acc[bin] > 8u ? 8u : s.bin_acc[bin];          // partial values are <= 8
            const uint32_t f = s.tab.v2v_flush[bin][pv][nin > 5u ? 5u : nin];
            const uint32_t pre = (s.bin_acc[bin] | ((f & 15u) << nin)) & 31u;
            const uint32_t e = s.tab.v2v[bin][pre];
            // QUIRK (kept): the completed input is not checked to be a real code word
            s.ring[s.head] = (uint16_t)(kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8));
            s.bin_acc[bin] = 0;
            s.bin_nin[bin] = 0;
            s.bin_slot[bin] = -1;
        }
    }
}

// icer_encode_bit after bin selection (icer_encoding.c:59-112), without the drain that follows every
// event in the reference: finished words are only *observable* through `used` when a new word is
// allocated with the ring apparently full, which seq_run checks for (it then pauses for a drain).
ICER_DEV void seq_put(CoderShared &s, int bin, uint32_t bit)
{
    int slot = s.bin_slot[bin];
    if (slot < 0) {
        slot = (int)((s.head + s.used) & (kRingWords - 1));
        s.used++;
        s.ring[slot] = (uint16_t)bin;
        s.bin_slot[bin] = slot;
    }
    if (bin >= 8) {
        if (bit) {
            s.ring[slot] = (uint16_t)golomb_word(s.tab, bin, s.bin_acc[bin]);
            s.bin_acc[bin] = 0;
            s.bin_slot[bin] = -1;
        } else {
            const uint32_t k = s.bin_acc[bin] + 1;
            if (k >= s.tab.gm[bin]) {
                s.ring[slot] = (uint16_t)(kWordDone | (1u << 11) | 1u);
                s.bin_acc[bin] = 0;
                s.bin_slot[bin] = -1;
            } else s.bin_acc[bin] = k;
        }
    } else if (bin >= 1) {
        const uint32_t nin = s.bin_nin[bin] + 1;
        const uint32_t pre = s.bin_acc[bin] | (bit << (nin - 1));
        const uint32_t e = s.tab.v2v[bin][pre & 31u];
        if ((e & 15u) == nin) {
            s.ring[slot] = (uint16_t)(kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8));
            s.bin_acc[bin] = 0;
            s.bin_nin[bin] = 0;
            s.bin_slot[bin] = -1;
        } else {
            s.bin_acc[bin] = pre;
            s.bin_nin[bin] = nin;
        }
    } else {
        s.ring[slot] = (uint16_t)(kWordDone | (1u << 11) | bit);
        s.bin_slot[0] = -1;
    }
}

// replay events [e, 128) of s.ev in order; stops BEFORE an event that needs a new word while the ring
// holds 2048 (possibly already finished) words, returning its index; 128 when the chunk is done
ICER_DEV uint32_t seq_run(CoderShared &s, uint32_t e)
{
    for (; e < 128u; e++) {
        const uint32_t v = s.ev[e];
        if (!(v & 0x80u)) continue;
        const int bin = (int)(v & 31u);
        if (s.bin_slot[bin] < 0 && s.used == (uint32_t)kRingWords) return e;
        seq_put(s, bin, (v >> 5) & 1u);
    }
    return 128u;
}

// ------------------------------------------------------------------------------------------
// context tables as arithmetic (icer_config.c:26-67)
// ------------------------------------------------------------------------------------------
ICER_DEV uint32_t ctx_plain(uint32_t h, uint32_t v, uint32_t d)      // LL / LH / (swapped) HL
{
    if (h == 2) return 8;
    if (h == 1) return (v == 0) ? (d < 2 ? 5 + d : 7u) : 7u;
    if (v == 0) return d > 2 ? 2u : d;
    return 2 + v;                                                    // v = 1 -> 3, v = 2 -> 4
}
ICER_DEV uint32_t ctx_hh(uint32_t hv, uint32_t d)
{
    if (d >= 3) return 8;
    const uint32_t k = hv > 2 ? 2u : hv;
    if (d == 0) return k;
    if (d == 1) return 3 + k;
    return hv == 0 ? 6u : 7u;
}

// pick the coder bin from a folded (zero >= total/2) probability estimate: the number of
// cut-offs not above zero/total (icer_compute_bin, icer_util.c:48-56; cut-offs are ascending)
ICER_DEV uint32_t pick_bin(const uint32_t *cut, uint32_t zero, uint32_t total)
{
    const uint32_t lhs = zero << 16;
    uint32_t bin = 0;
#pragma unroll
    for (int b = 0; b < 16; b++) bin += (lhs >= total * cut[b]) ? 1u : 0u;
    return bin;
}

// ------------------------------------------------------------------------------------------
// the unit coder.  Returns the payload length in bits, or kUnitTooBig when the slot is too small.
// ------------------------------------------------------------------------------------------
// Adaptive counts for every event of context C in this chunk (phase 2).  A context is rescaled
// when its total reaches 500 (-> 250); with at most 64 events per context and chunk that can
// happen at most once per chunk.  QUIRK C5: at a rescale `zero` is halved only if it exceeds the
// halved total.
#define ICER_CTX_STEP(C, PRED, ISZERO, ZOUT, TOUT)                                                    \
    {                                                                                                 \
        const uint64_t m_ = BALLOT(PRED);                                                             \
        if (m_) {                                                                                     \
            const uint64_t zm_ = BALLOT((PRED) && (ISZERO));                                          \
            const uint32_t n_ = (uint32_t)popc64(m_), nz_ = (uint32_t)popc64(zm_);                    \
            const uint32_t t0_ = s.ctx_total[C], z0_ = s.ctx_zero[C];                                 \
            if (t0_ + n_ < kRescaleCap) {                                                             \
                FOR_LANES                                                                             \
                {                                                                                     \
                    if (PRED) {                                                                       \
                        LV(TOUT) = t0_ + (uint32_t)mbcnt64(m_, lane);                                 \
                        LV(ZOUT) = z0_ + (uint32_t)mbcnt64(zm_, lane);                                \
                    }                                                                                 \
                }                                                                                     \
                FOR_LANES { if (lane == 0) { s.ctx_total[C] = t0_ + n_; s.ctx_zero[C] = z0_ + nz_; } }       \
            } else {                                                                                  \
                const uint32_t kc_ = kRescaleCap - 1 - t0_; /* rank of the event that triggers it */  \
                const int lc_ = ffs64(BALLOT((PRED) && (uint32_t)mbcnt64(m_, lane) == kc_));          \
                const uint32_t zc_ = (uint32_t)popc64(zm_ & ((2ull << lc_) - 1ull));                  \
                const uint32_t zat_ = z0_ + zc_;                                                      \
                const uint32_t zr_ = (zat_ > kRescaleCap / 2) ? (zat_ >> 1) : zat_;                   \
                FOR_LANES                                                                             \
                {                                                                                     \
                    if (PRED) {                                                                       \
                        const uint32_t rk_ = (uint32_t)mbcnt64(m_, lane), zb_ = (uint32_t)mbcnt64(zm_, lane); \
                        if (rk_ <= kc_) { LV(TOUT) = t0_ + rk_; LV(ZOUT) = z0_ + zb_; }               \
                        else { LV(TOUT) = kRescaleCap / 2 + (rk_ - kc_ - 1); LV(ZOUT) = zr_ + (zb_ - zc_); } \
                    }                                                                                 \
                }                                                                                     \
                FOR_LANES { if (lane == 0) { s.ctx_total[C] = kRescaleCap / 2 + (n_ - kc_ - 1); s.ctx_zero[C] = zr_ + (nz_ - zc_); } } \
            }                                                                                         \
        }                                                                                             \
    }

// write the complete 32-bit words of the bit stage to HBM; returns false when the slot is full
ICER_DEV bool flush_stage(CoderShared &s, const UnitArgs &a, bool final_partial)
{
    DECL_LANE;
    const uint32_t bp = s.bitpos;
    const uint32_t first = s.flushed_words;
    uint32_t last = bp >> 5;
    if (final_partial && (bp & 31u)) last++;
    const bool fits = last <= a.cap_words;
    const uint32_t stop = fits ? last : a.cap_words;
    FOR_LANES
    {
        for (uint32_t wi = first + (uint32_t)lane; wi < last; wi += 64) {
            const uint32_t v = s.stage[wi & (kStageWords - 1)];
            if (wi < stop) a.out_words[wi] = v;
            s.stage[wi & (kStageWords - 1)] = 0;
        }
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane == 0) s.flushed_words = last;
    }
    WAVE_SYNC();
    // a unit whose complete bytes reach the capacity can never fit (see P3 in DESIGN.md)
    return fits && (bp >> 3) < a.cap_words * 4u;
}

// ------------------------------------------------------------------------------------------
// phase 4, fast path: the whole chunk's events coded by 64 lanes
// ------------------------------------------------------------------------------------------
// Event coordinates: lane L carries the magnitude-bit event at position 2L and the sign event at
// position 2L+1 (coding order = position order).  A set of events is a pair of 64-bit lane masks
// (A1 for magnitude events, A2 for sign events).
ICER_DEV uint64_t below64(uint32_t x) { return x >= 64u ? ~0ull : ((1ull << x) - 1ull); }
// events of the set strictly before position pos
ICER_DEV uint32_t cnt_lt(uint64_t A1, uint64_t A2, uint32_t pos)
{
    return (uint32_t)(popc64(A1 & below64((pos + 1u) >> 1)) + popc64(A2 & below64(pos >> 1)));
}
// latest position <= pos in the set, -1 if none
ICER_DEV int last_le(uint64_t A1, uint64_t A2, uint32_t pos)
{
    const uint64_t c1 = A1 & below64((pos >> 1) + 1u), c2 = A2 & below64((pos + 1u) >> 1);
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
ICER_DEV int last_lt(uint64_t A1, uint64_t A2, uint32_t pos) { return pos == 0u ? -1 : last_le(A1, A2, pos - 1u); }

// one Golomb-bin event at position POS with bit BIT (lane-local); Z*/O* = zero/one events of the bin
#define ICER_GOLOMB_EVENT(POS, BIT, FL, WD)                                                           \
    {                                                                                                 \
        const uint32_t zb_ = cnt_lt(Z1, Z2, (POS));                                                   \
        const int lo_ = last_lt(O1, O2, (POS));                                                       \
        const uint32_t z_ = lo_ >= 0 ? zb_ - cnt_lt(Z1, Z2, (uint32_t)lo_) : k_in + zb_;              \
        const uint32_t kb_ = z_ - ((z_ * inv) >> 20) * m;                                             \
        FL = (kb_ == 0u ? 1u : 0u) | (((BIT) || kb_ + 1u == m) ? 2u : 0u);                            \
        WD = (BIT) ? golomb_word(s.tab, b, kb_) : (kWordDone | (1u << 11) | 1u);                      \
    }

ICER_DEV void fast_chunk(CoderShared &s, LANEARG(uint32_t, ev1), LANEARG(uint32_t, ev2), LANEARG(uint32_t, term) ICER_TIMER_PARAMS)
{
    DECL_LANE;
    LANEVAR(uint32_t, fl1); LANEVAR(uint32_t, fl2);     // bit0: a word starts at this event, bit1: a word ends here
    LANEVAR(uint32_t, wd1); LANEVAR(uint32_t, wd2);     // finished ring word of an end event
    LANEVAR(uint32_t, sp1); LANEVAR(uint32_t, sp2);     // start position of the word an end event closes (255: carried in)
    LANEVAR(uint32_t, wn);                              // lanes 1..7: number of events of bin `lane` in this chunk

    FOR_LANES
    {
        LV(fl1) = 0; LV(fl2) = 0; LV(wd1) = 0; LV(wd2) = 0; LV(sp1) = 255; LV(sp2) = 255;
        if (lane < kNumBins) s.bin_open_pos[lane] = 255;
        // bin 0 (uncoded): every event is a complete one-bit word (E3)
        if ((LV(ev1) & 0x9Fu) == 0x80u) { LV(fl1) = 3; LV(wd1) = kWordDone | (1u << 11) | ((LV(ev1) >> 5) & 1u); LV(sp1) = 2u * (uint32_t)lane; }
        if ((LV(ev2) & 0x9Fu) == 0x80u) { LV(fl2) = 3; LV(wd2) = kWordDone | (1u << 11) | ((LV(ev2) >> 5) & 1u); LV(sp2) = 2u * (uint32_t)lane + 1u; }
    }
    WAVE_SYNC();

    // ---- Golomb bins 8..16: run length since the bin's previous one-event, modulo m ------------
    // one step per Golomb bin PRESENT in the chunk
    for (uint64_t rem1 = BALLOT((LV(ev1) & 0x98u) >= 0x88u), rem2 = BALLOT((LV(ev2) & 0x98u) >= 0x88u); rem1 | rem2;) {
        const int b = (int)(rem1 ? READLANE(ev1, ffs64(rem1)) & 31u : READLANE(ev2, ffs64(rem2)) & 31u);
        const uint64_t M1 = BALLOT((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b));
        const uint64_t M2 = BALLOT((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b));
        rem1 &= ~M1;
        rem2 &= ~M2;
        const uint64_t O1 = BALLOT((LV(ev1) & 0xBFu) == (0xA0u | (uint32_t)b));
        const uint64_t O2 = BALLOT((LV(ev2) & 0xBFu) == (0xA0u | (uint32_t)b));
        const uint64_t Z1 = M1 & ~O1, Z2 = M2 & ~O2;
        const uint32_t m = s.tab.gm[b], inv = s.tab.ginv[b], k_in = s.bin_acc[b];
        FOR_LANES
        {
            if ((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b)) ICER_GOLOMB_EVENT(2u * (uint32_t)lane, (LV(ev1) >> 5) & 1u, LV(fl1), LV(wd1))
            if ((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b)) ICER_GOLOMB_EVENT(2u * (uint32_t)lane + 1u, (LV(ev2) >> 5) & 1u, LV(fl2), LV(wd2))
        }
        const uint64_t SB1 = BALLOT((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b) && (LV(fl1) & 1u));
        const uint64_t SB2 = BALLOT((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b) && (LV(fl2) & 1u));
        FOR_LANES
        {
            if ((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b) && (LV(fl1) & 2u)) {
                const int sp = last_le(SB1, SB2, 2u * (uint32_t)lane);
                LV(sp1) = sp < 0 ? 255u : (uint32_t)sp;
            }
            if ((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b) && (LV(fl2) & 2u)) {
                const int sp = last_le(SB1, SB2, 2u * (uint32_t)lane + 1u);
                LV(sp2) = sp < 0 ? 255u : (uint32_t)sp;
            }
        }
        // bin state after the chunk (wave-uniform)
        const int lastone = last_le(O1, O2, 127u);
        const uint32_t ztot = (uint32_t)(popc64(Z1) + popc64(Z2));
        const uint32_t zafter = lastone >= 0 ? ztot - cnt_lt(Z1, Z2, (uint32_t)lastone) : k_in + ztot;
        const uint32_t k_out = zafter - ((zafter * inv) >> 20) * m;
        const int laststart = last_le(SB1, SB2, 127u);
        FOR_LANES
        {
            if (lane == 0) {
                s.bin_acc[b] = k_out;
                s.bin_open_pos[b] = (uint8_t)(k_out ? (laststart >= 0 ? laststart : 255) : 254);
            }
        }
    }

    ICER_TICK(4)
    // ---- bins 1..7 (variable-to-variable codes) ---------------------------------------------------
    // Every event of a present bin writes (position, bit) at its rank into that bin's sequence in LDS;
    // lane b then walks the dense sequence of bin b (all <= 7 walkers run in lockstep).
    bool any_v2v = false;
    FOR_LANES { LV(wn) = 0; }
    for (uint64_t rem1 = BALLOT((LV(ev1) & 0x98u) == 0x80u && (LV(ev1) & 7u)), rem2 = BALLOT((LV(ev2) & 0x98u) == 0x80u && (LV(ev2) & 7u)); rem1 | rem2;) {
        const int b = (int)(rem1 ? READLANE(ev1, ffs64(rem1)) & 31u : READLANE(ev2, ffs64(rem2)) & 31u);
        const uint64_t M1 = BALLOT((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b));
        const uint64_t M2 = BALLOT((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b));
        rem1 &= ~M1;
        rem2 &= ~M2;
        any_v2v = true;
        const uint32_t n = (uint32_t)(popc64(M1) + popc64(M2));
        FOR_LANES
        {
            if ((LV(ev1) & 0x9Fu) == (0x80u | (uint32_t)b))
                s.binseq[b][cnt_lt(M1, M2, 2u * (uint32_t)lane)] = (uint8_t)(2u * (uint32_t)lane | ((LV(ev1) << 2) & 0x80u));
            if ((LV(ev2) & 0x9Fu) == (0x80u | (uint32_t)b))
                s.binseq[b][cnt_lt(M1, M2, 2u * (uint32_t)lane + 1u)] = (uint8_t)((2u * (uint32_t)lane + 1u) | ((LV(ev2) << 2) & 0x80u));
            if (lane == b) LV(wn) = n;
        }
    }
    if (any_v2v) {
        WAVE_SYNC();
        FOR_LANES
        {
            if (lane >= 1 && lane <= 7 && LV(wn)) {
                const int b = lane;
                const uint32_t n = LV(wn), tmask = LV(term);
                uint32_t acc = s.bin_acc[b], top = 1u << s.bin_nin[b];   // partial input and the weight of its next bit
                uint32_t cur_start = 255;                                // an unfinished word carried in from earlier chunks
                uint32_t x = s.binseq[b][0];
                for (uint32_t r = 0; r < n; r++) {
                    const uint32_t pos = x & 127u, bit = x >> 7;
                    if (r + 1 < n) x = s.binseq[b][r + 1];              // next event's record is fetched early
                    const uint32_t starts = top == 1u ? 1u : 0u;
                    cur_start = starts ? pos : cur_start;
                    acc |= bit ? top : 0u;
                    top <<= 1;
                    // (acc | top) numbers the node of the code tree; all 5-bit inputs are code words
                    const uint32_t ends = (top == 32u || ((tmask >> (acc | top)) & 1u)) ? 1u : 0u;
                    s.evflag[pos] = (uint8_t)(starts | (ends << 1));
                    s.evacc[pos] = (uint8_t)acc;
                    s.evstart[pos] = (uint8_t)cur_start;
                    acc = ends ? 0u : acc;
                    top = ends ? 1u : top;
                }
                s.bin_acc[b] = acc;
                s.bin_nin[b] = 31u - (uint32_t)clz32(top);
                s.bin_open_pos[b] = (uint8_t)(top != 1u ? cur_start : 254u);
            }
        }
        WAVE_SYNC();
        FOR_LANES
        {
            const uint32_t b1 = LV(ev1) & 0x9Fu, b2 = LV(ev2) & 0x9Fu;
            if (b1 >= 0x81u && b1 <= 0x87u) {
                LV(fl1) = s.evflag[2 * lane];
                if (LV(fl1) & 2u) {
                    const uint32_t e = s.tab.v2v[b1 & 31u][s.evacc[2 * lane] & 31u];
                    LV(wd1) = kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8);
                    LV(sp1) = s.evstart[2 * lane];
                }
            }
            if (b2 >= 0x81u && b2 <= 0x87u) {
                LV(fl2) = s.evflag[2 * lane + 1];
                if (LV(fl2) & 2u) {
                    const uint32_t e = s.tab.v2v[b2 & 31u][s.evacc[2 * lane + 1] & 31u];
                    LV(wd2) = kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8);
                    LV(sp2) = s.evstart[2 * lane + 1];
                }
            }
        }
    }

    ICER_TICK(5)
    // ---- ring slots in allocation order = order of the words' first events (E2) ------------------
    const uint64_t S1 = BALLOT(LV(fl1) & 1u), S2 = BALLOT(LV(fl2) & 1u);
    const uint32_t used = s.used, tail = s.head + used;
    FOR_LANES
    {
        if (LV(fl1) & 1u) s.ring[(tail + cnt_lt(S1, S2, 2u * (uint32_t)lane)) & (kRingWords - 1)] = (uint16_t)(LV(ev1) & 31u);
        if (LV(fl2) & 1u) s.ring[(tail + cnt_lt(S1, S2, 2u * (uint32_t)lane + 1u)) & (kRingWords - 1)] = (uint16_t)(LV(ev2) & 31u);
    }
    FOR_LANES
    {
        if (LV(fl1) & 2u) {
            const uint32_t slot = LV(sp1) == 255u ? (uint32_t)s.bin_slot[LV(ev1) & 31u] : (tail + cnt_lt(S1, S2, LV(sp1)));
            s.ring[slot & (kRingWords - 1)] = (uint16_t)LV(wd1);
        }
        if (LV(fl2) & 2u) {
            const uint32_t slot = LV(sp2) == 255u ? (uint32_t)s.bin_slot[LV(ev2) & 31u] : (tail + cnt_lt(S1, S2, LV(sp2)));
            s.ring[slot & (kRingWords - 1)] = (uint16_t)LV(wd2);
        }
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane < kNumBins) {
            const uint32_t op = s.bin_open_pos[lane];
            if (op == 254u) s.bin_slot[lane] = -1;
            else if (op < 128u) s.bin_slot[lane] = (int32_t)((tail + cnt_lt(S1, S2, op)) & (kRingWords - 1));
        }
        if (lane == 0) s.used = used + (uint32_t)(popc64(S1) + popc64(S2));
    }
    WAVE_SYNC();
    ICER_TICK(6)
}

// drain finished words from the head of the ring, 64 per round: lengths -> prefix sum -> bit offsets,
// code bits OR-ed into the LDS bit stage (icer_popbuf_while_avail, icer_encoding.c:114-139)
ICER_DEV void wave_drain(CoderShared &s)
{
    DECL_LANE;
    uint32_t head = s.head, used = s.used, bitpos = s.bitpos;
    for (;;) {
        LANEVAR(uint32_t, w); LANEVAR(uint32_t, len); LANEVAR(uint32_t, off);
        FOR_LANES
        {
            LV(w) = (uint32_t)lane < used ? s.ring[(head + (uint32_t)lane) & (kRingWords - 1)] : 0u;
        }
        const uint64_t done = BALLOT((LV(w) & kWordDone) != 0u);
        const uint32_t n = (uint32_t)ffs64(~done);               // leading finished words
        if (n == 0) break;
        FOR_LANES
        {
            LV(len) = (uint32_t)lane < n ? ((LV(w) >> 11) & 15u) : 0u;
        }
        uint32_t total;
        WAVE_EXCL_SCAN(uint32_t, off, len, total);
        FOR_LANES
        {
            if (LV(len)) {
                const uint32_t p = bitpos + LV(off), wi = (p >> 5) & (kStageWords - 1), sh = p & 31u;
                const uint32_t code = LV(w) & 0x3FFu;
                LDS_OR(s.stage[wi], code << sh);
                if (sh + LV(len) > 32u) LDS_OR(s.stage[(wi + 1) & (kStageWords - 1)], code >> (32u - sh));
            }
        }
        bitpos += total;
        head = (head + n) & (kRingWords - 1);
        used -= n;
        if (n < 64u) break;
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane == 0) { s.head = head; s.used = used; s.bitpos = bitpos; }
    }
    WAVE_SYNC();
}

ICER_DEV uint32_t code_unit_wave(CoderShared &s, const UnitArgs &a)
{
    DECL_LANE;
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)lane; i < kStageWords; i += 64) s.stage[i] = 0;
        if (lane < kNumBins) { s.bin_slot[lane] = -1; s.bin_acc[lane] = 0; s.bin_nin[lane] = 0; }
        if (lane < kNumContexts) { s.ctx_zero[lane] = 2; s.ctx_total[lane] = 4; }      // icer_context_modeller.c:607-613
        if (lane == 0) { s.head = 0; s.used = 0; s.bitpos = 0; s.flushed_words = 0; }
    }
    WAVE_SYNC();


    ICER_TIMERS_DECL
    // lanes 1..7 walk bins 1..7: that bin's code-tree termination mask stays in a register.
    // Node numbering: (partial input | 1 << bits so far), < 32 for inputs of up to 4 bits.
    LANEVAR(uint32_t, term);
    FOR_LANES
    {
        const int wb = lane & 7;
        LV(term) = 0;
        for (uint32_t n = 1; n <= 4; n++)
            for (uint32_t v = 0; v < (1u << n); v++)
                if ((s.tab.v2v_term[wb][n] >> v) & 1u) LV(term) |= 1u << (v | (1u << n));
    }
    const uint32_t npix = a.w * a.h;
    const uint32_t lsb = (uint32_t)a.lsb;
    const bool is_hl = a.subband == kHL, is_hh = a.subband == kHH;
    bool ok = true;

    // coefficient window of the NEXT chunk is fetched while the current one is coded (the loads stay in
    // flight across the whole chunk: WAVE_SYNC does not drain vmcnt)
    LANEVAR(uint32_t, nC); LANEVAR(uint32_t, nW); LANEVAR(uint32_t, nE); LANEVAR(uint32_t, nN); LANEVAR(uint32_t, nS);
    LANEVAR(uint32_t, nNW); LANEVAR(uint32_t, nNE); LANEVAR(uint32_t, nSW); LANEVAR(uint32_t, nSE);
#define ICER_FETCH_WINDOW(BASE)                                                                        \
    FOR_LANES                                                                                          \
    {                                                                                                  \
        const uint32_t p_ = (BASE) + (uint32_t)lane;                                                   \
        const uint32_t pp_ = p_ < npix ? p_ : 0u;                                                      \
        const uint32_t r_ = pp_ / a.w, c_ = pp_ - r_ * a.w;                                            \
        const uint16_t *q_ = a.seg + (size_t)r_ * a.stride + c_;                                       \
        const bool hasW_ = c_ > 0, hasE_ = c_ + 1 < a.w, hasN_ = r_ > 0, hasS_ = r_ + 1 < a.h;          \
        LV(nC) = q_[0];                                                                                \
        LV(nW) = hasW_ ? q_[-1] : 0u;                                                                  \
        LV(nE) = hasE_ ? q_[1] : 0u;                                                                   \
        LV(nN) = hasN_ ? *(q_ - a.stride) : 0u;                                                        \
        LV(nS) = hasS_ ? *(q_ + a.stride) : 0u;                                                        \
        LV(nNW) = (hasN_ && hasW_) ? *(q_ - a.stride - 1) : 0u;                                        \
        LV(nNE) = (hasN_ && hasE_) ? *(q_ - a.stride + 1) : 0u;                                        \
        LV(nSW) = (hasS_ && hasW_) ? *(q_ + a.stride - 1) : 0u;                                        \
        LV(nSE) = (hasS_ && hasE_) ? *(q_ + a.stride + 1) : 0u;                                        \
    }
    if (npix) ICER_FETCH_WINDOW(0u)

    for (uint32_t base = 0; base < npix && ok; base += 64) {
        LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
        LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
        LANEVAR(uint32_t, z1); LANEVAR(uint32_t, t1); LANEVAR(uint32_t, z2); LANEVAR(uint32_t, t2);
        LANEVAR(uint32_t, cC); LANEVAR(uint32_t, cW); LANEVAR(uint32_t, cE); LANEVAR(uint32_t, cN); LANEVAR(uint32_t, cS);
        LANEVAR(uint32_t, cNW); LANEVAR(uint32_t, cNE); LANEVAR(uint32_t, cSW); LANEVAR(uint32_t, cSE);
        FOR_LANES
        {
            LV(cC) = LV(nC); LV(cW) = LV(nW); LV(cE) = LV(nE); LV(cN) = LV(nN); LV(cS) = LV(nS);
            LV(cNW) = LV(nNW); LV(cNE) = LV(nNE); LV(cSW) = LV(nSW); LV(cSE) = LV(nSE);
        }
        if (base + 64u < npix) ICER_FETCH_WINDOW(base + 64u)

        // ---- phase 1: context formation (C1-C6) -------------------------------------------
        FOR_LANES
        {
            const bool valid = base + (uint32_t)lane < npix;
            const uint32_t x = LV(cC), xW = LV(cW), xE = LV(cE), xN = LV(cN), xS = LV(cS);
            const uint32_t xNW = LV(cNW), xNE = LV(cNE), xSW = LV(cSW), xSE = LV(cSE);

            const uint32_t mag = x & 0x7FFFu;
            const int msb = 31 - clz32(mag | 1u);
            int cat = msb - (int)lsb;
            cat = cat < 0 ? 0 : (cat > 3 ? 3 : cat);
            const uint32_t bit = (mag >> lsb) & 1u;
            // already-visited neighbours are judged at this plane, the others one plane up
#define ICER_SIG(v, l) ((((v)&0x7FFFu) >> (l)) != 0u ? 1u : 0u)
            const uint32_t sW = ICER_SIG(xW, lsb), sE = ICER_SIG(xE, lsb + 1);
            const uint32_t sN = ICER_SIG(xN, lsb), sS = ICER_SIG(xS, lsb + 1);
            uint32_t hh = sW + sE, vv = sN + sS;
            const uint32_t dd = ICER_SIG(xNW, lsb) + ICER_SIG(xNE, lsb) + ICER_SIG(xSW, lsb + 1) + ICER_SIG(xSE, lsb + 1);
#undef ICER_SIG
            uint32_t ctx;
            if (cat == 3) ctx = 31;                               // uncoded: no model context
            else if (cat == 2) ctx = 11;
            else if (cat == 1) ctx = (hh + vv == 0) ? 9u : 10u;
            else {
                if (is_hl) { const uint32_t t = hh; hh = vv; vv = t; }
                ctx = is_hh ? ctx_hh(hh + vv, dd) : ctx_plain(hh, vv, dd);
            }
            LV(valid1) = valid ? 1u : 0u;
            LV(ctx1) = ctx;
            LV(bit1) = bit;
            LV(z1) = 1; LV(t1) = 2;                               // what an uncoded event presents (C2)

            // sign event (C6): only negative significant neighbours count
            const bool sgn = valid && cat == 0 && bit;
            uint32_t sh = 2 - ((xW >> 15) & sW) - ((xE >> 15) & sE);
            uint32_t sv = 2 - ((xN >> 15) & sN) - ((xS >> 15) & sS);
            if (is_hl) { const uint32_t t = sh; sh = sv; sv = t; }
            // icer_sign_context_table / icer_sign_prediction_table restricted to sh, sv in {0,1,2}
            const uint32_t sctx = (sh == 2) ? (sv == 2 ? 12u : 13u) : (sv == 2 ? 15u : 14u);
            const uint32_t pred = (sh == 2) ? 0u : 1u;
            LV(valid2) = sgn ? 1u : 0u;
            LV(ctx2) = sctx;
            LV(bit2) = (pred ^ (x >> 15)) & 1u;
            LV(z2) = 0; LV(t2) = 0;
        }

        ICER_TICK(0)
        // ---- phase 2: adaptive counts per event (C5) --------------------------------------
        // one step per context PRESENT in the chunk (typically 3-8 of the 17)
        for (uint64_t rem = BALLOT(LV(valid1) && LV(ctx1) != 31u); rem;) {
            const uint32_t c = READLANE(ctx1, ffs64(rem));
            ICER_CTX_STEP(c, LV(valid1) && LV(ctx1) == c, LV(bit1) == 0u, z1, t1)
            rem &= ~BALLOT(LV(valid1) && LV(ctx1) == c);
        }
        for (uint64_t rem = BALLOT(LV(valid2) != 0u); rem;) {
            const uint32_t c = READLANE(ctx2, ffs64(rem));
            ICER_CTX_STEP(c, LV(valid2) && LV(ctx2) == c, LV(bit2) == 0u, z2, t2)
            rem &= ~BALLOT(LV(valid2) && LV(ctx2) == c);
        }
        WAVE_SYNC();

        ICER_TICK(1)
        // ---- phase 3: fold + bin (E1) -----------------------------------------------------
        LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);           // 0x80 | bit << 5 | bin, 0 = no event
        FOR_LANES
        {
            uint32_t e1 = 0, e2 = 0;
            if (LV(valid1)) {
                uint32_t z = LV(z1), t = LV(t1), b = LV(bit1);
                if (z < (t >> 1)) { z = t - z; b ^= 1u; }
                e1 = 0x80u | (b << 5) | pick_bin(s.tab.cut, z, t);
            }
            if (LV(valid2)) {
                uint32_t z = LV(z2), t = LV(t2), b = LV(bit2);
                if (z < (t >> 1)) { z = t - z; b ^= 1u; }
                e2 = 0x80u | (b << 5) | pick_bin(s.tab.cut, z, t);
            }
            LV(ev1) = e1;
            LV(ev2) = e2;
        }

        ICER_TICK(2)
        // ---- phase 4: interleaved entropy coder ------------------------------------------
        const uint32_t nev = (uint32_t)(popc64(BALLOT(LV(ev1) != 0u)) + popc64(BALLOT(LV(ev2) != 0u)));
        if (s.used + nev <= (uint32_t)kRingWords) {
            // every event could open at most one word, so the ring cannot fill up in this chunk:
            // no forced flush (E5) is possible and word boundaries depend on each bin alone
            fast_chunk(s, ev1, ev2, term ICER_TIMER_PASS);
            wave_drain(s);
            ICER_TICK(7)
            ICER_EMU_COUNT(0);
        } else {
            ICER_EMU_COUNT(1);
            // ring nearly full: replay the reference state machine exactly, event by event
            FOR_LANES
            {
                s.ev[2 * lane] = (uint8_t)LV(ev1);
                s.ev[2 * lane + 1] = (uint8_t)LV(ev2);
            }
            WAVE_SYNC();
            uint32_t e = 0;
            for (;;) {
                FOR_LANES
                {
                    if (lane == 0) s.resume = seq_run(s, e);
                }
                WAVE_SYNC();
                e = s.resume;
                if (e >= 128u) break;
                // a new word is needed and the ring holds 2048 words: pop what is finished (64 lanes),
                // and if the oldest word is still open force-complete it (E5, icer_encoding.c:59-64)
                wave_drain(s);
                if (s.used == (uint32_t)kRingWords) {
                    FOR_LANES
                    {
                        if (lane == 0) seq_complete_head(s);
                    }
                    WAVE_SYNC();
                    wave_drain(s);
                }
            }
            wave_drain(s);
            ICER_TICK(8)
        }
        ok = flush_stage(s, a, false);
        ICER_TICK(9)
    }

    if (!ok) return kUnitTooBig;
    ICER_TICK(10)
    // end of unit: force-complete whatever is still open (C8, icer_context_modeller.c:452-455)
    while (s.used > 0) {
        FOR_LANES
        {
            if (lane == 0) seq_complete_head(s);
        }
        WAVE_SYNC();
        wave_drain(s);
    }
    ok = flush_stage(s, a, true);
    ICER_TICK(11)
    ICER_TIMERS_STORE(a.timers)
    return ok ? s.bitpos : kUnitTooBig;
}

}  // namespace icer
