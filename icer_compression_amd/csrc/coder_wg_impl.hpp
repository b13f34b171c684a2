// coder_wg_impl.hpp -- the body of the workgroup-window coder (see coder_wg.hpp for what it does), compiled once per
// INSTANCE: the including header sets ICER_WG_NS (the namespace inside icer::) and ICER_WG_WAVES (wavefronts per
// workgroup = chunks per window) and includes this file; coder_wg.hpp makes icer::wg with 16 waves, coder_wg_small.hpp
// icer::wgs with 2.  No include guard on purpose.
namespace icer {
namespace ICER_WG_NS {
#ifdef ICER_WAVE_EMU
// tests only: set by a failed WG_ASSERT (the run goes on, so that a test can report it instead of aborting)
static int g_wg_assert_line = 0;
static unsigned g_wg_order = 0, g_wg_order_state = 1;
static unsigned long long g_wg_stats[8] = {0};   // windows, detailed flush tests, exact chunks, forced flushes
#undef WG_STAT
#define WG_STAT(i) (icer::ICER_WG_NS::g_wg_stats[i]++)
static inline uint32_t wg_wave_order(uint32_t i)
{
    const uint32_t n = (uint32_t)ICER_WG_WAVES;
    if (g_wg_order == 0) return i;
    if (g_wg_order == 1) return n - 1u - i;
    if (i == 0) g_wg_order_state = g_wg_order_state * 1664525u + 1013904223u;     // a new permutation per region
    const uint32_t mul = ((g_wg_order_state >> 8) | 1u) % n, add = (g_wg_order_state >> 16) % n;   // odd multiplier: a bijection mod 2^k
    return (i * (mul | 1u) + add) % n;
}
static inline void wg_assert_fail(const char *what, int line) { if (!g_wg_assert_line) { g_wg_assert_line = line; if (getenv("ICER_WG_TRACE")) fprintf(stderr, "WG_ASSERT line %d: %s\n", line, what); } }
#endif

constexpr uint32_t kWgWaves = ICER_WG_WAVES;
static_assert(kWgWaves * 128u <= (uint32_t)kRingWords, "a window must not be able to open more words than the ring holds (E5 test)");
constexpr uint32_t kPhysRing = 2u * kRingWords;     // physical ring entries: ring occupancy + one window of new words
constexpr uint32_t kStageWords = 2048;              // LDS bit stage (circular, 32-bit words): one full drain of the physical ring
constexpr uint32_t kBlankRunMin = 2, kBlankRunMax = 1024;   // blank chunks coded in closed form at a time (blank_run)
constexpr uint32_t kBlankLook = 16;                          // chunk-table entries a lane looks at when a run is measured
constexpr uint32_t kUnitTooBig = 0xFFFFFFFFu;
constexpr uint32_t kUnitStopped = 0xFFFFFFFDu;      // progressive mode: the quota cut lies before this unit

// ring word: 0 while its code word is unfinished (a slot is cleared when it is popped; which bin an open word
//            belongs to follows from the bins' open slots), 0x8000 | nbits << 11 | code (<= 10 bits) once it is finished.
//            Only END events store to the ring, so no two waves ever store to the same slot.
constexpr uint32_t kWordDone = 0x8000u;

struct UnitArgs {
    const uint16_t *seg;        // first coefficient of the segment (sign-magnitude words)
    uint32_t stride;            // plane row stride in elements
    uint32_t w, h;              // segment size
    int subband, lsb;
    uint32_t *out_words;        // payload slot (4-byte aligned)
    uint32_t cap_words;         // slot capacity in 32-bit words
    // progressive mode (small byte quota): the frame's per-unit results so far in priority order, this unit's place in
    // that order and the quota; null / 0 otherwise.  See quota_already_spent.
    const uint32_t *done_bytes;
    uint32_t prio_index;
    uint64_t early_quota;
    uint64_t *timers;           // profiling build: per-phase cycle counters (null otherwise)
    const uint8_t *sig;         // chunk table of the unit's family: chunk j is blank at every plane >= sig[j] (chunk_blank_plane); null: none
};

struct WaveLds {                // per-wave LDS: the chunk's summaries and scratch
    uint16_t cnt[20];           // per context: events | zeros << 8
    uint8_t zpre[17][64];       // [context][rank]: zeros among the context's events of rank <= this one
    // bins 1..7, compacted per bin in coding order (rank = number of earlier events of the same bin):
    uint8_t binseq[8][128];     // rank -> position of the event
    uint32_t binbits[8][6];     // rank -> input bit, stored with an offset of 8 bits
    uint32_t binstart[8][6];    // rank -> a code word starts here (same offset)
    uint8_t binn[8], bincarry[8], post_nin[8];
    uint32_t sumC[17];          // bins 1..7: end node of the chunk's walk for every start node (3 bits each, compact numbers);
                                // bins 8..16: 0x8000 | zeros after the last one-event, or the zeros if there is none
    uint32_t gk[17];            // bins 8..16: zero-run length carried into the chunk
    uint32_t binst[17];         // per bin after the chunk: bits 0..7 open_pos -- 255 untouched, 254 closed, else the first event
                                // of its open word --, bits 8..23 Golomb run length / partial input value, bits 24..31 input bits
    int32_t bslot[17];          // ring slot (allocation count) of each bin's open word at chunk start, -1 if none
    uint16_t sumE[17];          // per bin: 255 untouched, 254 closed, else the chunk-relative slot of the word it leaves open
    uint8_t fe[20];             // per bin: position of its first end event in the chunk (255: none)
    // bins 8..16, by bin - 8: the bin's events at even / odd positions (bit = lane), its one-events likewise; event count
    uint64_t gmask[9][4];
    uint8_t gn[12];
    uint8_t onez[128];          // position of a Golomb bin's one-event -> zeros of that bin before it
    uint8_t srank[128];         // position of a word start -> number of word starts before it in the chunk
    uint32_t nst;               // words the chunk opens
    uint32_t segtot;            // drain: code bits of this wave's 64 ring words
};

struct Shared {
    uint32_t stage[kStageWords];
    uint16_t ring[kPhysRing];
    CoderTables tab;
    uint32_t crc_tab[256];
    WaveLds wl[kWgWaves];
    // coder state as of the first chunk that is not committed yet
    uint32_t ctot[2][20], czer[2][20];  // adaptive counts (icer_context_model_typedef, icer.h:195-199) as of the window start; [window parity]
    uint32_t bin_state[20];         // as WaveLds::binst (bits 0..7 unused)
    int32_t bin_slot[20];           // ring slot (allocation count) of the bin's open word, -1 if none
    uint8_t ctx_tab[48];            // context table of the unit's subband (phase A)
    // ring occupancy = alloc - popped (both count words since the start of the unit; physical index = count mod 4096).
    // These four live in registers (identical in every wave) while the waves work together and here while ONE wave
    // works alone (exact_chunk, end of unit).
    uint32_t alloc, popped, bitpos, flushed_words;
    uint32_t stop;                  // progressive mode: the unit was abandoned
    uint32_t run_tail[2];           // blank_run: allocation count after the run, [counts copy the run started from]
};

// Progressive mode.  The stream keeps units in priority order until the first one that does not fit the byte quota
// (icer_partition.c:321-336, `break` in the packet loop); everything after it is dropped.  done_bytes[j] is 0 while
// unit j is unfinished, its size (header + payload bytes) once it is coded, ~0 if it can not fit whatever comes
// before it.  If the finished units of higher priority ALONE already exceed the quota, the cut lies before this unit
// and it can stop: its result can not be part of the stream.  (A lower bound of the prefix sum, so never a false
// positive; units before the cut always run to completion.)  One wavefront; wave-uniform result.
#ifndef ICER_WAVE_EMU
ICER_DEV bool quota_already_spent(const UnitArgs &a)
{
    if (!a.early_quota) return false;
    DECL_LANE;
    unsigned long long sum = 0;
    for (uint32_t j = (uint32_t)lane; j < a.prio_index; j += 64) {
        const uint32_t d = __hip_atomic_load(&a.done_bytes[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += d == ~0u ? a.early_quota + 1ull : (unsigned long long)d;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    return sum > a.early_quota;
}
#else
// tests only: the stop is a flag the test raises (done_bytes[0] != 0)
ICER_DEV bool quota_already_spent(const UnitArgs &a) { return a.early_quota && a.done_bytes && a.done_bytes[0] != 0u; }
#endif

#define WRING_LD(i) ((uint32_t)s.ring[(i) & (kPhysRing - 1)])
#define WRING_ST(i, v) (s.ring[(i) & (kPhysRing - 1)] = (uint16_t)(v))

// ------------------------------------------------------------------------------------------
// exact coder steps (restatement of E1-E6)
// ------------------------------------------------------------------------------------------
// Golomb codeword for a run of k zeros ended by a one (icer_encoding.c:73-80)
ICER_DEV uint32_t wg_golomb_word(const CoderTables &t, int bin, uint32_t k)
{
    const uint32_t gi = t.gi[bin];
    const uint32_t code = k + (k >= gi ? gi : 0u);
    const uint32_t n = t.gl[bin] + (k >= gi ? 1u : 0u);
    return kWordDone | (n << 11) | ((brev32(code) >> (32u - n)) & 0x3FFu);
}

// packed per-bin coder state (WaveLds::binst, Shared::bin_state)
ICER_DEV uint32_t st_acc(uint32_t st) { return (st >> 8) & 0xFFFFu; }
ICER_DEV uint32_t st_nin(uint32_t st) { return st >> 24; }
ICER_DEV uint32_t st_pack(uint32_t op, uint32_t acc, uint32_t nin) { return op | (acc << 8) | (nin << 24); }

// first half of icer_flush_encode (icer_encoding.c:141-189): force-complete the oldest word.
// The caller drains afterwards (wave_drain).  One lane.
ICER_DEV void seq_complete_head(Shared &s)
{
    const uint32_t head = s.popped;
    // the oldest word is open (everything finished has been popped): it is the open word of the bin whose slot it is
    int bin = 0;
    for (int b = 1; b < kNumBins; b++) if (s.bin_slot[b] == (int32_t)head) bin = b;
    if (bin >= 8) {
        const uint32_t k = st_acc(s.bin_state[bin]);
        WRING_ST(head, (k == (uint32_t)s.tab.gm[bin] - 1u) ? (kWordDone | (1u << 11) | 1u) : wg_golomb_word(s.tab, bin, k));
        s.bin_state[bin] = 0;
        s.bin_slot[bin] = -1;
    } else if (bin >= 1) {
        const uint32_t nin = st_nin(s.bin_state[bin]), acc = st_acc(s.bin_state[bin]);
        const uint32_t pv = acc > 8u ? 8u : acc;                                // partial values are <= 8
        const uint32_t f = s.tab.v2v_flush[bin][pv][nin > 5u ? 5u : nin];
        const uint32_t pre = (acc | ((f & 15u) << nin)) & 31u;
        const uint32_t e = s.tab.v2v[bin][pre];
        // QUIRK (kept): the completed input is not checked to be a real code word
        WRING_ST(head, (kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8)));
        s.bin_state[bin] = 0;
        s.bin_slot[bin] = -1;
    }
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
// cut-offs not above zero/total (icer_compute_bin, icer_util.c:48-56; cut-offs are ascending).
// zero * 65536 >= total * cut  <=>  floor(zero * 65536 / total) >= cut, so one exact division (total <= 500: a
// float reciprocal estimate is within 1 of the quotient, then corrected) and one table look-up replace the 16 compares.
ICER_DEV uint32_t pick_bin(const uint32_t *binlut, uint32_t zero, uint32_t total)
{
    const uint32_t a = zero << 16;
#ifdef ICER_WAVE_EMU
    uint32_t q = (uint32_t)((float)a * (1.0f / (float)total));
    int32_t rem = (int32_t)a - (int32_t)(q * total);
#else
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)total));
    int32_t rem = (int32_t)a - (int32_t)__umul24(q, total);                  // q <= 2^16 + 1, total <= 500
#endif
    if (rem < 0) { q--; rem += (int32_t)total; }
    if (rem >= (int32_t)total) q++;
    const uint32_t e = binlut[q >> 8];
    return (e & 255u) + (q >= (e >> 8) ? 1u : 0u);
}

// Adaptive counts for every event of context C in this chunk when the context is rescaled inside the chunk.  A context
// is rescaled when its total reaches 500 (-> 250); with at most 64 events per context and chunk that can happen at
// most once per chunk.  QUIRK C5: at a rescale `zero` is halved only if it exceeds the halved total.
#undef ICER_CTX_STEP
#define ICER_CTX_STEP(C, PRED, ISZERO, ZOUT, TOUT)                                                    \
    {                                                                                                 \
        const uint64_t m_ = BALLOT(PRED);                                                             \
        const uint64_t zm_ = BALLOT((PRED) && (ISZERO));                                              \
        const uint32_t n_ = (uint32_t)popc64(m_), nz_ = (uint32_t)popc64(zm_);                        \
        const uint32_t t0_ = READLANE(ctot, C), z0_ = READLANE(czer, C);                              \
        const uint32_t kc_ = kRescaleCap - 1 - t0_; /* rank of the event that triggers it */          \
        const int lc_ = ffs64(BALLOT((PRED) && (uint32_t)mbcnt64(m_, lane) == kc_));                  \
        const uint32_t zc_ = (uint32_t)popc64(zm_ & ((2ull << lc_) - 1ull));                          \
        const uint32_t zat_ = z0_ + zc_;                                                              \
        const uint32_t zr_ = (zat_ > kRescaleCap / 2) ? (zat_ >> 1) : zat_;                           \
        FOR_LANES                                                                                     \
        {                                                                                             \
            if (PRED) {                                                                               \
                const uint32_t rk_ = (uint32_t)mbcnt64(m_, lane), zb_ = (uint32_t)mbcnt64(zm_, lane); \
                if (rk_ <= kc_) { LV(TOUT) = t0_ + rk_; LV(ZOUT) = z0_ + zb_; }                       \
                else { LV(TOUT) = kRescaleCap / 2 + (rk_ - kc_ - 1); LV(ZOUT) = zr_ + (zb_ - zc_); }  \
            }                                                                                         \
        }                                                                                             \
        FOR_LANES { if ((uint32_t)lane == (C)) { LV(ctot) = kRescaleCap / 2 + (n_ - kc_ - 1); LV(czer) = zr_ + (nz_ - zc_); } } \
    }

ICER_DEV uint64_t below64(uint32_t x) { return x >= 64u ? ~0ull : ((1ull << x) - 1ull); }
// this lane's bit of a mask that differs from lane to lane (no 64-bit shift by a variable)
ICER_DEV uint32_t own_bit(uint64_t A, int lane) { return ((lane < 32 ? (uint32_t)A : (uint32_t)(A >> 32)) >> (lane & 31)) & 1u; }
// events of the set strictly before position pos (A1: even positions 2 * lane, A2: odd positions 2 * lane + 1)
ICER_DEV uint32_t cnt_lt(uint64_t A1, uint64_t A2, uint32_t pos)
{
    return (uint32_t)(popc64(A1 & below64((pos + 1u) >> 1)) + popc64(A2 & below64(pos >> 1)));
}
// the same for this lane's own events (position 2 * lane + slot): two v_mbcnt pairs instead of 64-bit shifts
ICER_DEV uint32_t cnt_lt_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    return (uint32_t)(mbcnt64(A1, lane) + mbcnt64(A2, lane)) + (slot ? own_bit(A1, lane) : 0u);
}
// latest position <= pos in the set, -1 if none
ICER_DEV int last_le(uint64_t A1, uint64_t A2, uint32_t pos)
{
    const uint64_t c1 = A1 & below64((pos >> 1) + 1u), c2 = A2 & below64((pos + 1u) >> 1);
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
ICER_DEV int last_lt(uint64_t A1, uint64_t A2, uint32_t pos) { return pos == 0u ? -1 : last_le(A1, A2, pos - 1u); }
// earliest position in the set, 255 if none
ICER_DEV uint32_t first_pos(uint64_t A1, uint64_t A2)
{
    const uint32_t k1 = A1 ? 2u * (uint32_t)ffs64(A1) : 255u, k2 = A2 ? 2u * (uint32_t)ffs64(A2) + 1u : 255u;
    return k1 < k2 ? k1 : k2;
}

// The same questions about THIS lane's own events (position 2 * lane + slot) of masks that differ from lane to lane
// (vector registers): lane-relative masks from 32-bit operations, no 64-bit shifts by a variable.
ICER_DEV uint64_t lanes_below(int lane)         // bits of the lanes below this one
{
    const uint32_t lo = lane < 32 ? (1u << (lane & 31)) - 1u : ~0u, hi = lane < 32 ? 0u : (1u << (lane & 31)) - 1u;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// latest position < (2 * lane + slot) in the set, -1 if none
ICER_DEV int last_lt_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    const uint64_t lt = lanes_below(lane);
    const uint64_t c1 = A1 & (slot ? (lt << 1) | 1ull : lt), c2 = A2 & lt;
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
// latest position <= (2 * lane + slot) in the set, -1 if none
ICER_DEV int last_le_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    const uint64_t lt = lanes_below(lane), le = (lt << 1) | 1ull;
    const uint64_t c1 = A1 & le, c2 = A2 & (slot ? le : lt);
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}

// bits [start, start + 6) of a bit string stored with an offset of 8 (rank r lives at bit r + 8, so that
// reads a few ranks before rank 0 see zeros); `start` is a rank - 4 >= -4
ICER_DEV uint32_t window6(const uint32_t *words, int start)
{
    const uint32_t pos = (uint32_t)(start + 8);
    const uint64_t two = (uint64_t)words[pos >> 5] | ((uint64_t)words[(pos >> 5) + 1] << 32);
    return (uint32_t)(two >> (pos & 31u)) & 63u;
}

// ------------------------------------------------------------------------------------------
// single-wave drain (exact_chunk, end of unit): finished words popped from the head of the ring, 64 per round:
// lengths -> prefix sum -> bit offsets, code bits OR-ed into the LDS bit stage (icer_popbuf_while_avail,
// icer_encoding.c:114-139).  `limit` = allocation count up to which ring slots are valid.
// ------------------------------------------------------------------------------------------
ICER_DEV uint32_t wave_drain(Shared &s, uint32_t limit)
{
    DECL_LANE;
    const uint32_t popped0 = s.popped;
    uint32_t head = popped0, used = limit - popped0, bitpos = s.bitpos;
    uint32_t npop = 0;
    for (;;) {
        LANEVAR(uint32_t, w); LANEVAR(uint32_t, len); LANEVAR(uint32_t, off);
        FOR_LANES
        {
            LV(w) = (uint32_t)lane < used ? WRING_LD(head + (uint32_t)lane) : 0u;
        }
        const uint64_t done = BALLOT((LV(w) & kWordDone) != 0u);
        const uint32_t n = (uint32_t)ffs64(~done);               // leading finished words
        if (n == 0) break;
        FOR_LANES
        {
            LV(len) = (uint32_t)lane < n ? ((LV(w) >> 11) & 15u) : 0u;
        }
        // exclusive prefix sum of the lengths (< 16): one ballot per bit of the length, lane-masked popcounts
        const uint64_t L0 = BALLOT(LV(len) & 1u), L1 = BALLOT(LV(len) & 2u), L2 = BALLOT(LV(len) & 4u), L3 = BALLOT(LV(len) & 8u);
        const uint32_t total = (uint32_t)(popc64(L0) + 2 * popc64(L1) + 4 * popc64(L2) + 8 * popc64(L3));
        FOR_LANES
        {
            LV(off) = (uint32_t)(mbcnt64(L0, lane) + 2 * mbcnt64(L1, lane) + 4 * mbcnt64(L2, lane) + 8 * mbcnt64(L3, lane));
        }
        FOR_LANES
        {
            if (LV(len)) {
                const uint32_t p = bitpos + LV(off), wi = (p >> 5) & (kStageWords - 1), sh = p & 31u;
                const uint32_t code = LV(w) & 0x3FFu;
                LDS_OR(s.stage[wi], code << sh);
                if (sh + LV(len) > 32u) LDS_OR(s.stage[(wi + 1) & (kStageWords - 1)], code >> (32u - sh));
            }
        }
        FOR_LANES
        {
            if ((uint32_t)lane < n) WRING_ST(head + (uint32_t)lane, 0u);          // popped: the slot is free again
        }
        bitpos += total;
        head += n;
        used -= n;
        npop += n;
        if (n < 64u) break;
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane == 0) { s.bitpos = bitpos; s.popped = popped0 + npop; }
    }
    WAVE_SYNC();
    return npop;
}

// single-wave: write the complete 32-bit words of the bit stage to HBM; returns false when the slot is full
ICER_DEV bool flush_stage(Shared &s, const UnitArgs &a, bool final_partial)
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
    // a unit whose complete bytes reach the capacity can never fit (P3 of SURVEY.md 2.3; HISTORY.md 3)
    return fits && (bp >> 3) < a.cap_words * 4u;
}

// ==========================================================================================
// chunk tables
// ==========================================================================================
// A chunk is BLANK at a bit plane when its 64 pixels are and stay insignificant there and have no significant
// neighbour: 64 zero events of context 0, no sign event (icer_compress_bitplane_uint16, icer_context_modeller.c:312-457:
// category 0, bit 0, h = v = d = 0).  With pixel magnitude m that is  m >> lsb == 0  for the pixel and its W, N, NW, NE
// neighbours (judged at this plane) and  m >> (lsb + 1) == 0  for E, S, SW, SE (judged one plane up), so a chunk is blank
// at every plane >= T and at none below, T = max(bitlen(pixels, W, N, NW, NE), bitlen(E, S, SW, SE) - 1); T = 255 for a
// chunk with fewer than 64 pixels (never blank).  One wavefront, chunk j of the segment; wave-uniform result.  It
// depends on the coefficients only, so it is computed once per (channel, level, subband, segment) for all bit planes.
ICER_DEV uint32_t chunk_blank_plane(const uint16_t *seg, uint32_t stride, uint32_t sw, uint32_t sh, uint32_t j)
{
    DECL_LANE;
    const uint32_t npix = sw * sh;
    LANEVAR(uint32_t, t);
    FOR_LANES
    {
        const uint32_t np = j * 64u + (uint32_t)lane;
        const bool in_ = np < npix;
        const uint32_t r_ = in_ ? np / sw : 0u, c_ = in_ ? np - r_ * sw : 0u;
        const bool hasW_ = c_ > 0, hasE_ = c_ + 1 < sw, hasN_ = r_ > 0, hasS_ = r_ + 1 < sh;
        const uint32_t cW_ = hasW_ ? c_ - 1u : c_, cE_ = hasE_ ? c_ + 1u : c_;
        const uint16_t *pC_ = seg + (size_t)r_ * stride;
        const uint16_t *pN_ = hasN_ ? pC_ - stride : pC_, *pS_ = hasS_ ? pC_ + stride : pC_;
        // (clamped positions repeat a pixel that is in the right group already, or, for E / S, one of the other group:
        // those are masked)
        uint32_t a_ = pC_[c_] & 0x7FFFu;
        a_ |= pC_[cW_] & 0x7FFFu;                                   // OR keeps the bit length of the maximum
        a_ |= pN_[c_] & 0x7FFFu; a_ |= pN_[cW_] & 0x7FFFu;
        a_ |= (hasN_ && hasE_) ? (pN_[cE_] & 0x7FFFu) : 0u;
        uint32_t b_ = hasE_ ? (pC_[cE_] & 0x7FFFu) : 0u;
        b_ |= hasS_ ? (pS_[c_] & 0x7FFFu) : 0u;
        b_ |= (hasS_ && hasW_) ? (pS_[cW_] & 0x7FFFu) : 0u;
        b_ |= (hasS_ && hasE_) ? (pS_[cE_] & 0x7FFFu) : 0u;
        const uint32_t la_ = 32u - (uint32_t)clz32(a_), lb_ = 32u - (uint32_t)clz32(b_);
        LV(t) = in_ ? (la_ > lb_ - (lb_ ? 1u : 0u) ? la_ : lb_ - (lb_ ? 1u : 0u)) : 255u;
    }
    uint32_t tmax;
    WAVE_MAX(tmax, t)
    return tmax;
}

// ==========================================================================================
// per-wave register state
// ==========================================================================================
struct MergeChunk {             // one chunk's events with their code-word roles, one pixel per lane
    LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);     // raw event bytes: 0x80 | bit << 5 | bin, 0 = none
    LANEVAR(uint32_t, fl1); LANEVAR(uint32_t, fl2);     // bit0: a word starts at this event, bit1: a word ends here
    LANEVAR(uint32_t, wd1); LANEVAR(uint32_t, wd2);     // finished ring word of an end event
    LANEVAR(uint32_t, sp1); LANEVAR(uint32_t, sp2);     // start position of the word an end event closes (255: carried in)
    LANEVAR(uint32_t, st);                              // lane b: WaveLds::binst of bin b
    uint64_t S1, S2;                                    // word-start flags of the chunk
};

struct Wave {
    // the 3x3 coefficient windows of this wave's chunk in the NEXT window, one pixel per lane (prefetch)
    LANEVAR(uint32_t, nC); LANEVAR(uint32_t, nW); LANEVAR(uint32_t, nE); LANEVAR(uint32_t, nN); LANEVAR(uint32_t, nS);
    LANEVAR(uint32_t, nNW); LANEVAR(uint32_t, nNE); LANEVAR(uint32_t, nSW); LANEVAR(uint32_t, nSE);
    LANEVAR(uint32_t, has);                             // which neighbours exist: bit 0 W, 1 E, 2 N, 3 S
    uint32_t pf;                                        // the chunk these belong to (~0: none)
    // phase A -> B: per pixel, word 1 the magnitude-bit event, word 2 the sign event:
    //   bits 0..7   0x80 | bit << 5 | context (31 = uncoded), 0 = none; the sign event's bit is the agreement bit
    //   bits 8..15  rank of the event among the chunk's events of the same context (coding order)
    //   bits 16..23 how many of those earlier ones were zeros
    LANEVAR(uint32_t, w1); LANEVAR(uint32_t, w2);
    LANEVAR(uint32_t, cn);                              // lane c: events | zeros << 8 of context c in the chunk
    LANEVAR(uint32_t, czer); LANEVAR(uint32_t, ctot);   // lane c: adaptive counts of context c after the chunk
    LANEVAR(uint32_t, rb1); LANEVAR(uint32_t, rb2);     // bins 1..7: rank of this lane's events inside their bin
    // phase C -> D, lane L = one node of one bin's code tree (CoderTables::cand_*): the walk of the bin's bit string
    // entered at that node: word-start flags by rank (128 bits) and the node it ends in
    LANEVAR(uint32_t, stl0); LANEVAR(uint32_t, stl1); LANEVAR(uint32_t, stl2); LANEVAR(uint32_t, stl3); LANEVAR(uint32_t, wnode);
    LANEVAR(uint32_t, cb); LANEVAR(uint32_t, ce);       // candidate lanes: their bin and the (compact) node they assume
    LANEVAR(uint32_t, slot);                            // lane b: ring slot of bin b's open word at chunk start / after it
    LANEVAR(uint32_t, slot0);                           // lane b: ring slot of bin b's open word before the pending chunks, ~0 if none
    LANEVAR(uint32_t, dw); LANEVAR(uint32_t, doff);     // drain: this lane's two ring words and their bit offsets between the two halves of a round
    MergeChunk c;
    uint32_t blank, has_v2v;
    // uniform (the same value in every wave)
    uint32_t tail, popped, bitpos, flushed_words;       // see Shared
    uint32_t tailw;                                     // allocation count before this wave's chunk
    uint32_t wL;                                        // first chunk of the pending ones that needs exact_chunk (or nwin)
    uint32_t nflush;                                    // drain: words to pop
    uint32_t too_big;
#if defined(ICER_PHASE_TIMERS) && !defined(ICER_WAVE_EMU)
    uint32_t tacc[24];
    uint64_t tlast;
#endif
};

// ==========================================================================================
// phase A: pixels -> events
// ==========================================================================================
// fetch the 3x3 windows of the pixels of chunk j (one pixel per lane) into the prefetch registers; R.pf = j
ICER_DEV void fetch_window(const UnitArgs &a, Wave &R, uint32_t j)
{
    DECL_LANE;
    const uint32_t npix = a.w * a.h;
    FOR_LANES
    {
        const uint32_t np = j * 64u + (uint32_t)lane;
        const bool in_ = np < npix;
        const uint32_t r_ = in_ ? np / a.w : 0u, c_ = in_ ? np - r_ * a.w : 0u;
        const bool hasW_ = c_ > 0, hasE_ = c_ + 1 < a.w, hasN_ = r_ > 0, hasS_ = r_ + 1 < a.h;
        /* nine unconditional loads from clamped (always valid) positions, then selects: no divergent branches */
        const uint32_t cW_ = hasW_ ? c_ - 1u : c_, cE_ = hasE_ ? c_ + 1u : c_;
        const uint16_t *pC_ = a.seg + (size_t)r_ * a.stride;
        const uint16_t *pN_ = hasN_ ? pC_ - a.stride : pC_, *pS_ = hasS_ ? pC_ + a.stride : pC_;
        LV(R.nC) = pC_[c_]; LV(R.nW) = pC_[cW_]; LV(R.nE) = pC_[cE_];
        LV(R.nN) = pN_[c_]; LV(R.nNW) = pN_[cW_]; LV(R.nNE) = pN_[cE_];
        LV(R.nS) = pS_[c_]; LV(R.nSW) = pS_[cW_]; LV(R.nSE) = pS_[cE_];
        /* the loaded values are not touched before the chunk is processed (the loads stay in flight meanwhile):
         * which neighbours exist is kept as a mask and applied then */
        LV(R.has) = (hasW_ ? 1u : 0u) | (hasE_ ? 2u : 0u) | (hasN_ ? 4u : 0u) | (hasS_ ? 8u : 0u);
    }
    R.pf = j;
}

// is chunk j blank at this unit's bit plane, going by the family's chunk table (false without a table)
ICER_DEV bool table_says_blank(const UnitArgs &a, uint32_t j)
{
    return a.sig != nullptr && (uint32_t)a.lsb >= (uint32_t)a.sig[j];
}

ICER_DEV void wave_init(Shared &s, const UnitArgs &a, Wave &R, uint32_t w)
{
    DECL_LANE;
    FOR_LANES
    {
        LV(R.cb) = (lane >= 8) ? (uint32_t)s.tab.cand_bin[lane] : 0u;
        LV(R.ce) = LV(R.cb) ? (uint32_t)s.tab.node_c[LV(R.cb) & 7u][s.tab.cand_node[lane]] & 7u : 0u;
    }
    FOR_LANES
    {
        for (uint32_t i = w * 64u + (uint32_t)lane; i < kPhysRing / 2u; i += 64u * kWgWaves) reinterpret_cast<uint32_t *>(s.ring)[i] = 0u;
    }
    R.tail = 0; R.popped = 0; R.bitpos = 0; R.flushed_words = 0; R.too_big = 0;
#if defined(ICER_PHASE_TIMERS) && !defined(ICER_WAVE_EMU)
    for (int i_ = 0; i_ < 24; i_++) R.tacc[i_] = 0;
    R.tlast = __builtin_amdgcn_s_memtime();
#endif
    R.pf = ~0u;
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    if (w < nchunks && !table_says_blank(a, w)) fetch_window(a, R, w);
}

// state every wave relies on; run by ONE wave before the others start (a workgroup barrier follows)
ICER_DEV void unit_state_init(Shared &s, const UnitArgs &a)
{
    DECL_LANE;
    const bool is_hh = a.subband == kHH;
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)lane; i < kStageWords; i += 64) s.stage[i] = 0;
        if (lane < 20) { s.bin_slot[lane] = -1; s.bin_state[lane] = 0; s.czer[0][lane] = 2u; s.ctot[0][lane] = 4u; }   // icer_context_modeller.c:607-613
        // context of a not-yet-significant pixel by neighbour counts (icer_config.c:26-67): HH indexed
        // (h + v) * 5 + d, the other subbands (h * 3 + v) * 5 + d with h, v <= 2, d <= 4
        if (lane < 45) s.ctx_tab[lane] = (uint8_t)(is_hh ? ctx_hh((uint32_t)lane / 5u, (uint32_t)lane % 5u)
                                                        : ctx_plain((uint32_t)lane / 15u, ((uint32_t)lane / 5u) % 3u, (uint32_t)lane % 5u));
        if (lane == 0) { s.alloc = 0; s.popped = 0; s.bitpos = 0; s.flushed_words = 0; s.stop = 0; }
    }
    WAVE_SYNC();
}

// chunk j (pixels j * 64 ...) of the unit: events, context groups, per-context summary
ICER_DEV void phase_a(Shared &s, const UnitArgs &a, Wave &R, uint32_t w, uint32_t j)
{
    DECL_LANE;
    WaveLds &l = s.wl[w];
    const uint32_t npix = a.w * a.h;
    const uint32_t lsb = (uint32_t)a.lsb;
    const bool is_hl = a.subband == kHL, is_hh = a.subband == kHH;
    const uint32_t base = j * 64u;
    LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
    LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
    LANEVAR(uint32_t, cC); LANEVAR(uint32_t, cW); LANEVAR(uint32_t, cE); LANEVAR(uint32_t, cN); LANEVAR(uint32_t, cS);
    LANEVAR(uint32_t, cNW); LANEVAR(uint32_t, cNE); LANEVAR(uint32_t, cSW); LANEVAR(uint32_t, cSE);
    // A chunk that the family's chunk table says is blank needs no pixels at all.
    const bool tblank = base + 64u <= npix && table_says_blank(a, j);
#ifdef ICER_WAVE_EMU
    const bool tskip = false;                         // (tests: the pixels are always read and the chunk table is checked against them)
#else
    const bool tskip = tblank;
#endif
    if (!tskip && R.pf != j) fetch_window(a, R, j);                    // (not prefetched: the windows did not follow each other)
    FOR_LANES
    {
        const uint32_t hm = tskip ? 0u : LV(R.has);
        const bool hW = hm & 1u, hE = hm & 2u, hN = hm & 4u, hS = hm & 8u;
        LV(cC) = tskip ? 0u : LV(R.nC); LV(cW) = hW ? LV(R.nW) : 0u; LV(cE) = hE ? LV(R.nE) : 0u;
        LV(cN) = hN ? LV(R.nN) : 0u; LV(cS) = hS ? LV(R.nS) : 0u;
        LV(cNW) = (hN && hW) ? LV(R.nNW) : 0u; LV(cNE) = (hN && hE) ? LV(R.nNE) : 0u;
        LV(cSW) = (hS && hW) ? LV(R.nSW) : 0u; LV(cSE) = (hS && hE) ? LV(R.nSE) : 0u;
    }
    // ---- context formation (C1-C6) ------------------------------------------------------------
    if (!tskip) FOR_LANES
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
        // category 0: the subband's context table (built once per unit, ctx_tab); 1: 9 / 10; 2: 11; 3: uncoded
        if (is_hl) { const uint32_t t = hh; hh = vv; vv = t; }
        const uint32_t c0 = s.ctx_tab[is_hh ? (hh + vv) * 5u + dd : (hh * 3u + vv) * 5u + dd];
        const uint32_t c1 = (hh + vv == 0) ? 9u : 10u;
        const uint32_t ctx = cat == 0 ? c0 : cat == 1 ? c1 : cat == 2 ? 11u : 31u;
        LV(valid1) = valid ? 1u : 0u;
        LV(ctx1) = ctx;
        LV(bit1) = bit;

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
    }
    // ---- context groups --------------------------------------------------------------------------
    // What the adaptive counts (C5) need from the chunk: an event sees the counts at chunk start + its rank among the
    // chunk's events of the same context (+ the zeros among those).  Lanes with the same context are found from
    // per-bit ballots of the context number (no loop over contexts); lane c applies the same match to "context c".
#define ICER_MATCH(KEY, V, B0, B1, B2, B3) \
    ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
    // A blank chunk -- 64 pixels that are and stay insignificant with no significant neighbour, i.e. 64 zero events
    // of context 0 and no sign event; more than half of all chunks, the high planes mostly -- needs no matching:
    // the rank of an event is its lane number and so is the number of zeros before it.
#ifdef ICER_WAVE_EMU
    const bool blank = BALLOT(!LV(valid1) || LV(ctx1) != 0u || LV(bit1) != 0u || LV(valid2)) == 0ull;
    WG_ASSERT(a.sig == nullptr || blank == tblank);
#else
    const bool blank = tblank || BALLOT(!LV(valid1) || LV(ctx1) != 0u || LV(bit1) != 0u || LV(valid2)) == 0ull;
#endif
    R.blank = blank ? 1u : 0u;
    if (blank) {
        FOR_LANES
        {
            LV(R.w1) = 0x80u | ((uint32_t)lane << 8) | ((uint32_t)lane << 16);
            LV(R.w2) = 0;
            LV(R.cn) = lane == 0 ? (64u | (64u << 8)) : 0u;
            l.zpre[0][lane] = (uint8_t)(lane + 1);
        }
    } else {
        // magnitude-bit events: contexts 0..11; sign events: contexts 12..16, keyed by context - 12
        const uint64_t V = BALLOT(LV(valid1) && LV(ctx1) != 31u);
        const uint64_t B0 = BALLOT(LV(ctx1) & 1u), B1 = BALLOT(LV(ctx1) & 2u), B2 = BALLOT(LV(ctx1) & 4u), B3 = BALLOT(LV(ctx1) & 8u);
        const uint64_t ZM = BALLOT(LV(bit1) == 0u);
        const uint64_t U = BALLOT(LV(valid2) != 0u);
        const uint64_t C0 = BALLOT((LV(ctx2) - 12u) & 1u), C1 = BALLOT((LV(ctx2) - 12u) & 2u), C2 = BALLOT((LV(ctx2) - 12u) & 4u);
        const uint64_t ZN = BALLOT(LV(bit2) == 0u);
        FOR_LANES
        {
            LV(R.w1) = 0; LV(R.w2) = 0; LV(R.cn) = 0;
            if (LV(valid1)) {
                LV(R.w1) = 0x80u | (LV(bit1) << 5) | LV(ctx1);
                if (LV(ctx1) != 31u) {
                    const uint64_t m = ICER_MATCH(LV(ctx1), V, B0, B1, B2, B3);
                    const uint32_t rk = (uint32_t)mbcnt64(m, lane), zb = (uint32_t)mbcnt64(m & ZM, lane);
                    LV(R.w1) |= (rk << 8) | (zb << 16);
                    l.zpre[LV(ctx1)][rk] = (uint8_t)(zb + (LV(bit1) ? 0u : 1u));
                }
            }
            if (LV(valid2)) {
                const uint64_t m = ICER_MATCH(LV(ctx2) - 12u, U, C0, C1, C2, 0ull);
                const uint32_t rk = (uint32_t)mbcnt64(m, lane), zb = (uint32_t)mbcnt64(m & ZN, lane);
                LV(R.w2) = 0x80u | (LV(bit2) << 5) | LV(ctx2) | (rk << 8) | (zb << 16);
                l.zpre[LV(ctx2)][rk] = (uint8_t)(zb + (LV(bit2) ? 0u : 1u));
            }
            if (lane < 12) {
                const uint64_t m = ICER_MATCH((uint32_t)lane, V, B0, B1, B2, B3);
                LV(R.cn) = (uint32_t)popc64(m) | ((uint32_t)popc64(m & ZM) << 8);
            } else if (lane <= 16) {
                const uint64_t m = ICER_MATCH((uint32_t)lane - 12u, U, C0, C1, C2, 0ull);
                LV(R.cn) = (uint32_t)popc64(m) | ((uint32_t)popc64(m & ZN) << 8);
            }
        }
    }
#undef ICER_MATCH
    FOR_LANES
    {
        if (lane < 17) l.cnt[lane] = (uint16_t)LV(R.cn);
    }
}

// ==========================================================================================
// phase B: adaptive counts (C5), probability fold + bin (E1)
// ==========================================================================================
ICER_DEV void phase_b(Shared &s, Wave &R, uint32_t w, uint32_t par)
{
    DECL_LANE;
    LANEVAR(uint32_t, czer); LANEVAR(uint32_t, ctot);
    // counts at the start of this wave's chunk = counts at the window start advanced over the chunks before it
    FOR_LANES
    {
        LV(ctot) = lane < 17 ? s.ctot[par][lane] : 0u;
        LV(czer) = lane < 17 ? s.czer[par][lane] : 0u;
    }
    for (uint32_t v = 0; v < w; v++) {
        const WaveLds &p = s.wl[v];
        FOR_LANES
        {
            if (lane < 17) {
                const uint32_t cw = p.cnt[lane], n = cw & 255u, nz = cw >> 8, t = LV(ctot);
                if (t + n >= kRescaleCap) {
                    // the context is rescaled by its event of rank kc in that chunk (ICER_CTX_STEP, QUIRK C5)
                    const uint32_t kc = kRescaleCap - 1u - t, zc = p.zpre[lane][kc];
                    const uint32_t zat = LV(czer) + zc, zr = (zat > kRescaleCap / 2) ? (zat >> 1) : zat;
                    LV(czer) = zr + (nz - zc);
                    LV(ctot) = t + n - kRescaleCap / 2;
                } else {
                    LV(ctot) = t + n;
                    LV(czer) += nz;
                }
            }
        }
    }
    LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
    LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
    LANEVAR(uint32_t, z1); LANEVAR(uint32_t, t1); LANEVAR(uint32_t, z2); LANEVAR(uint32_t, t2);
    LANEVAR(uint32_t, rk1); LANEVAR(uint32_t, zb1); LANEVAR(uint32_t, rk2); LANEVAR(uint32_t, zb2);
    LANEVAR(uint32_t, cn); LANEVAR(uint32_t, cnz);
    FOR_LANES
    {
        const uint32_t c1 = LV(R.w1), c2 = LV(R.w2);
        LV(valid1) = (c1 >> 7) & 1u; LV(ctx1) = c1 & 31u; LV(bit1) = (c1 >> 5) & 1u; LV(rk1) = (c1 >> 8) & 255u; LV(zb1) = (c1 >> 16) & 255u;
        LV(valid2) = (c2 >> 7) & 1u; LV(ctx2) = c2 & 31u; LV(bit2) = (c2 >> 5) & 1u; LV(rk2) = (c2 >> 8) & 255u; LV(zb2) = (c2 >> 16) & 255u;
        const uint32_t cw = lane < 17 ? LV(R.cn) : 0u;
        LV(cn) = cw & 255u; LV(cnz) = cw >> 8;
        LV(z1) = 1; LV(t1) = 2;                               // what an uncoded event presents (C2)
        LV(z2) = 0; LV(t2) = 0;
    }
    // counts an event sees = its context's counts at chunk start + its rank; lane c owns context c and advances its
    // counters by the chunk's totals.  A context that reaches the rescale point inside this chunk (total 500, at most
    // once per chunk) is redone by ICER_CTX_STEP.
    {
        LANEVAR(uint32_t, t0); LANEVAR(uint32_t, zz0); LANEVAR(uint32_t, idx);
        FOR_LANES { LV(idx) = LV(ctx1) & 15u; }
        WAVE_GATHER(t0, ctot, idx)
        WAVE_GATHER(zz0, czer, idx)
        FOR_LANES
        {
            if (LV(valid1) && LV(ctx1) != 31u) { LV(t1) = LV(t0) + LV(rk1); LV(z1) = LV(zz0) + LV(zb1); }
            LV(idx) = LV(ctx2);
        }
        if (BALLOT(LV(valid2) != 0u)) {                       // (no sign events in most chunks of the high planes)
            WAVE_GATHER(t0, ctot, idx)
            WAVE_GATHER(zz0, czer, idx)
        }
        LANEVAR(uint32_t, cross);
        FOR_LANES
        {
            if (LV(valid2)) { LV(t2) = LV(t0) + LV(rk2); LV(z2) = LV(zz0) + LV(zb2); }
            LV(cross) = 0;
            if (lane < 17) {
                if (LV(ctot) + LV(cn) < kRescaleCap) { LV(ctot) += LV(cn); LV(czer) += LV(cnz); }
                else LV(cross) = 1;
            }
        }
        for (uint64_t rem = BALLOT(LV(cross) != 0u); rem; rem &= rem - 1ull) {     // rare: contexts that rescale in this chunk
            const uint32_t c = (uint32_t)ffs64(rem);
            if (c < 12u) ICER_CTX_STEP(c, LV(valid1) && LV(ctx1) == c, LV(bit1) == 0u, z1, t1)
            else ICER_CTX_STEP(c, LV(valid2) && LV(ctx2) == c, LV(bit2) == 0u, z2, t2)
        }
    }
    FOR_LANES
    {
        uint32_t e1 = 0, e2 = 0;
        if (LV(valid1)) {
            uint32_t z = LV(z1), t = LV(t1), b = LV(bit1);
            if (z < (t >> 1)) { z = t - z; b ^= 1u; }
            e1 = 0x80u | (b << 5) | pick_bin(s.tab.binlut, z, t);
        }
        if (LV(valid2)) {
            uint32_t z = LV(z2), t = LV(t2), b = LV(bit2);
            if (z < (t >> 1)) { z = t - z; b ^= 1u; }
            e2 = 0x80u | (b << 5) | pick_bin(s.tab.binlut, z, t);
        }
        LV(R.c.ev1) = e1;
        LV(R.c.ev2) = e2;
        LV(R.czer) = LV(czer); LV(R.ctot) = LV(ctot);
    }
}

// ==========================================================================================
// phase C: the chunk's per-bin summaries (they do not depend on the coder state)
// ==========================================================================================
#define ICER_MATCH4(KEY, V, B0, B1, B2, B3) \
    ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
#define ICER_MATCH3(KEY, V, B0, B1, B2) ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)))

ICER_DEV void phase_c(Shared &s, Wave &R, uint32_t w)
{
    DECL_LANE;
    WaveLds &q = s.wl[w];
    const uint64_t V1 = BALLOT((LV(R.c.ev1) & 0x98u) == 0x80u && (LV(R.c.ev1) & 7u)), V2 = BALLOT((LV(R.c.ev2) & 0x98u) == 0x80u && (LV(R.c.ev2) & 7u));
    R.has_v2v = (V1 | V2) ? 1u : 0u;
    uint32_t maxn = 0;
    FOR_LANES
    {
        if (lane < 48) (&q.binbits[0][0])[lane] = 0;
        if (lane < 17) q.sumC[lane] = 0;
        LV(R.rb1) = 0; LV(R.rb2) = 0;
    }
    WAVE_SYNC();
    if (V1 | V2) {
        // bins 1..7: every event puts its input bit at its rank into the bin's bit string and its position into the
        // bin's position list.  Rank = number of earlier events of the same bin; the lanes of a bin are found from
        // per-bit ballots of the bin number (no loop over bins).
        const uint64_t P0 = BALLOT(LV(R.c.ev1) & 1u), P1 = BALLOT(LV(R.c.ev1) & 2u), P2 = BALLOT(LV(R.c.ev1) & 4u);
        const uint64_t Q0 = BALLOT(LV(R.c.ev2) & 1u), Q1 = BALLOT(LV(R.c.ev2) & 2u), Q2 = BALLOT(LV(R.c.ev2) & 4u);
        LANEVAR(uint32_t, nb);
        FOR_LANES
        {
            LV(nb) = 0;
            if ((V1 >> lane) & 1ull) {
                const uint32_t b = LV(R.c.ev1) & 7u;
                const uint32_t r = (uint32_t)(mbcnt64(ICER_MATCH3(b, V1, P0, P1, P2), lane) + mbcnt64(ICER_MATCH3(b, V2, Q0, Q1, Q2), lane));
                LV(R.rb1) = r;
                q.binseq[b][r] = (uint8_t)(2u * (uint32_t)lane);
                if (LV(R.c.ev1) & 0x20u) LDS_OR(q.binbits[b][(r + 8u) >> 5], 1u << ((r + 8u) & 31u));
            }
            if ((V2 >> lane) & 1ull) {
                const uint32_t b = LV(R.c.ev2) & 7u;
                const uint64_t m1 = ICER_MATCH3(b, V1, P0, P1, P2);          // this lane's own magnitude event comes first
                const uint32_t r = (uint32_t)(mbcnt64(m1, lane) + (int)((m1 >> lane) & 1ull) + mbcnt64(ICER_MATCH3(b, V2, Q0, Q1, Q2), lane));
                LV(R.rb2) = r;
                q.binseq[b][r] = (uint8_t)(2u * (uint32_t)lane + 1u);
                if (LV(R.c.ev2) & 0x20u) LDS_OR(q.binbits[b][(r + 8u) >> 5], 1u << ((r + 8u) & 31u));
            }
            if (lane < 8) {
                LV(nb) = (uint32_t)(popc64(ICER_MATCH3((uint32_t)lane, V1, P0, P1, P2)) + popc64(ICER_MATCH3((uint32_t)lane, V2, Q0, Q1, Q2)));
                if (lane == 0) LV(nb) = 0;
                q.binn[lane] = (uint8_t)LV(nb);
            }
        }
        for (uint32_t b = 1; b < 8; b++) { const uint32_t n = READLANE(nb, b); maxn = n > maxn ? n : maxn; }
        WAVE_SYNC();
    } else {
        FOR_LANES
        {
            if (lane < 8) q.binn[lane] = 0;
        }
        WAVE_SYNC();
    }
    // The walk of a bin's bit string through its code tree, six input bits per table look-up, from EVERY node of the
    // tree (one lane per node, 46 in all): which of them is the real one is known once the states are scanned.
    {
        LANEVAR(uint64_t, lo); LANEVAR(uint64_t, hi); LANEVAR(uint64_t, st_lo); LANEVAR(uint64_t, st_hi);
        LANEVAR(uint32_t, node); LANEVAR(uint32_t, n);
        FOR_LANES
        {
            const uint32_t b = LV(R.cb);
            LV(n) = b ? (uint32_t)q.binn[b] : 0u;
            LV(node) = LV(R.ce);
            LV(st_lo) = 0; LV(st_hi) = 0; LV(lo) = 0; LV(hi) = 0;
            if (LV(n)) {
                uint64_t l0 = ((uint64_t)q.binbits[b][0] | ((uint64_t)q.binbits[b][1] << 32)) >> 8;      // ranks 0..55
                uint64_t h0 = (uint64_t)q.binbits[b][2] | ((uint64_t)q.binbits[b][3] << 32);             // ranks 56..119
                const uint32_t top8 = q.binbits[b][4];                                                     // ranks 120..127
                l0 |= h0 << 56;
                h0 = (h0 >> 8) | ((uint64_t)top8 << 56);                                                   // ranks 64..127
                LV(lo) = l0; LV(hi) = h0;
            }
        }
        for (uint32_t r = 0; r + 6u <= maxn; r += 6) {
            FOR_LANES
            {
                if (r + 6u <= LV(n)) {
                    const uint32_t e = s.tab.v2v_step6[LV(R.cb) & 7u][LV(node)][(uint32_t)LV(lo) & 63u];
                    const uint64_t f = (uint64_t)(e >> 4);
                    if (r < 64u) { LV(st_lo) |= f << r; if (r > 58u) LV(st_hi) |= f >> (64u - r); }
                    else LV(st_hi) |= f << (r - 64u);
                    LV(node) = e & 7u;
                    LV(lo) = (LV(lo) >> 6) | (LV(hi) << 58);
                    LV(hi) >>= 6;
                }
            }
        }
        FOR_LANES
        {
            const uint32_t b = LV(R.cb) & 7u;
            const uint32_t full = (LV(n) / 6u) * 6u, k = LV(n) - full;
            if (k) {                                        // last 1..5 bits: flags from the zero-padded step, node from the tail table
                const uint32_t bits = (uint32_t)LV(lo) & ((1u << k) - 1u);
                const uint32_t e = s.tab.v2v_step6[b][LV(node)][bits];
                const uint64_t f = (uint64_t)((e >> 4) & ((1u << k) - 1u));
                if (full < 64u) { LV(st_lo) |= f << full; if (full > 58u) LV(st_hi) |= f >> (64u - full); }
                else LV(st_hi) |= f << (full - 64u);
                LV(node) = s.tab.v2v_tail[b][LV(node)][(1u << k) | bits];
            }
            LV(R.stl0) = (uint32_t)LV(st_lo); LV(R.stl1) = (uint32_t)(LV(st_lo) >> 32);
            LV(R.stl2) = (uint32_t)LV(st_hi); LV(R.stl3) = (uint32_t)(LV(st_hi) >> 32);
            LV(R.wnode) = LV(node) & 7u;
            // the bin's transfer function: end node for every start node
            if (LV(R.cb)) LDS_OR(q.sumC[b], (LV(node) & 7u) << (3u * LV(R.ce)));
        }
    }
    // bins 8..16: zeros after the bin's last one-event, or all its (zero) events if there is none
    {
        const uint64_t G1 = BALLOT((LV(R.c.ev1) & 0x98u) >= 0x88u), G2 = BALLOT((LV(R.c.ev2) & 0x98u) >= 0x88u);
        if (G1 | G2) {
            const uint64_t K0 = BALLOT(LV(R.c.ev1) & 1u), K1 = BALLOT(LV(R.c.ev1) & 2u), K2 = BALLOT(LV(R.c.ev1) & 4u), K3 = BALLOT((LV(R.c.ev1) & 31u) == 16u);
            const uint64_t J0 = BALLOT(LV(R.c.ev2) & 1u), J1 = BALLOT(LV(R.c.ev2) & 2u), J2 = BALLOT(LV(R.c.ev2) & 4u), J3 = BALLOT((LV(R.c.ev2) & 31u) == 16u);
            const uint64_t O1 = G1 & BALLOT(LV(R.c.ev1) & 0x20u), O2 = G2 & BALLOT(LV(R.c.ev2) & 0x20u);       // one-events
            FOR_LANES
            {
                if (lane >= 8 && lane <= 16) {
                    const uint32_t key = ((uint32_t)lane & 7u) | (lane == 16 ? 8u : 0u);
                    const uint64_t m1 = ICER_MATCH4(key, G1, K0, K1, K2, K3), m2 = ICER_MATCH4(key, G2, J0, J1, J2, J3);
                    const uint32_t all = (uint32_t)(popc64(m1) + popc64(m2));
                    const int lo1 = last_le(m1 & O1, m2 & O2, 127u);
                    q.sumC[lane] = lo1 < 0 ? all : (0x8000u | (all - cnt_lt(m1, m2, (uint32_t)lo1 + 1u)));
                    // (kept for the records, phase E)
                    q.gmask[lane - 8][0] = m1; q.gmask[lane - 8][1] = m2; q.gmask[lane - 8][2] = m1 & O1; q.gmask[lane - 8][3] = m2 & O2;
                    q.gn[lane - 8] = (uint8_t)all;
                }
            }
        }
    }
}

// ==========================================================================================
// phases D + E: coder state at the start of this wave's chunk; per-event code-word roles
// ==========================================================================================
// `wbase` = first chunk of the window that is not committed yet; Shared::bin_state is the state before it
ICER_DEV void phase_de(Shared &s, Wave &R, uint32_t w, uint32_t wbase)
{
    DECL_LANE;
    WaveLds &q = s.wl[w];
    MergeChunk &c = R.c;
    // ---- D: scan ---------------------------------------------------------------------------------
    LANEVAR(uint32_t, stv);         // lane b = 1..7: compact node; lane b = 8..16: zero-run length
    FOR_LANES
    {
        const uint32_t st = lane < 17 ? s.bin_state[lane] : 0u;
        LV(stv) = st_acc(st);
        if (lane >= 1 && lane <= 7) LV(stv) = (uint32_t)s.tab.node_c[lane][st_acc(st) | (1u << st_nin(st))] & 7u;
    }
    for (uint32_t v = wbase; v < w; v++) {
        const WaveLds &p = s.wl[v];
        FOR_LANES
        {
            if (lane >= 1 && lane <= 16) {
                const uint32_t sc = p.sumC[lane];
                if (lane <= 7) LV(stv) = (sc >> (3u * LV(stv))) & 7u;
                else {
                    uint32_t z = (sc & 0x8000u) ? (sc & 0x7FFFu) : LV(stv) + sc;
                    const uint32_t m = s.tab.gm[lane];
                    z -= ((z * s.tab.ginv[lane]) >> 20) * m;             // (z < 2048: exact)
                    LV(stv) = z;
                }
            }
        }
    }
    // ---- D: bins 1..7 take the walk that was entered at their node -------------------------------------------
    LANEVAR(uint32_t, src); LANEVAR(uint32_t, g0); LANEVAR(uint32_t, g1); LANEVAR(uint32_t, g2); LANEVAR(uint32_t, g3); LANEVAR(uint32_t, gnode);
    FOR_LANES
    {
        LV(src) = (uint32_t)lane;
        if (lane >= 1 && lane <= 7) LV(src) = s.tab.cand_lane[lane][s.tab.node_full[lane][LV(stv)]];
        if (lane >= 8 && lane <= 16) q.gk[lane] = LV(stv);
        if (lane == 0) q.gk[0] = 0;
        // a bin without events in this chunk keeps its run length and its open word (open_pos 255)
        if (lane == 0 || (lane >= 8 && lane <= 16)) q.binst[lane] = st_pack(255u, lane ? LV(stv) : 0u, 0u);
    }
    WAVE_GATHER(g0, R.stl0, src)
    WAVE_GATHER(g1, R.stl1, src)
    WAVE_GATHER(g2, R.stl2, src)
    WAVE_GATHER(g3, R.stl3, src)
    WAVE_GATHER(gnode, R.wnode, src)
    FOR_LANES
    {
        if (lane >= 1 && lane <= 7) {
            const int b = lane;
            const uint32_t n = q.binn[b];
            const uint64_t st_lo = (uint64_t)LV(g0) | ((uint64_t)LV(g1) << 32), st_hi = (uint64_t)LV(g2) | ((uint64_t)LV(g3) << 32);
            const uint32_t node_in = s.tab.node_full[b][LV(stv)];
            const uint32_t node = s.tab.node_full[b][LV(gnode) & 7u];     // back to the tree's own numbering (acc | 1 << bits)
            q.bincarry[b] = (uint8_t)node_in;
            // start flags with the same offset of 8 as the bit string
            q.binstart[b][0] = (uint32_t)(st_lo << 8);
            q.binstart[b][1] = (uint32_t)(st_lo >> 24);
            q.binstart[b][2] = (uint32_t)(st_lo >> 56) | (uint32_t)(st_hi << 8);
            q.binstart[b][3] = (uint32_t)(st_hi >> 24);
            q.binstart[b][4] = (uint32_t)(st_hi >> 56);
            q.binstart[b][5] = 0;
            const uint32_t nin = 31u - (uint32_t)clz32(node);
            q.post_nin[b] = (uint8_t)nin;
            uint32_t op = 255;
            if (n) {
                // open word after the chunk: the last start, unless everything after it completed
                const int last = st_hi ? 64 + 63 - clz64(st_hi) : (st_lo ? 63 - clz64(st_lo) : -1);
                op = node == 1u ? 254u : (last >= 0 ? (uint32_t)q.binseq[b][last] : 255u);
            }
            q.binst[b] = st_pack(op, node ^ (1u << nin), nin);
        }
    }
    WAVE_SYNC();

    // ---- E: records ------------------------------------------------------------------------------------
    FOR_LANES
    {
        LV(c.fl1) = 0; LV(c.fl2) = 0; LV(c.wd1) = 0; LV(c.wd2) = 0; LV(c.sp1) = 255; LV(c.sp2) = 255;
        // bin 0 (uncoded): every event is a complete one-bit word (E3)
        if ((LV(c.ev1) & 0x9Fu) == 0x80u) { LV(c.fl1) = 3; LV(c.wd1) = kWordDone | (1u << 11) | ((LV(c.ev1) >> 5) & 1u); LV(c.sp1) = 2u * (uint32_t)lane; }
        if ((LV(c.ev2) & 0x9Fu) == 0x80u) { LV(c.fl2) = 3; LV(c.wd2) = kWordDone | (1u << 11) | ((LV(c.ev2) >> 5) & 1u); LV(c.sp2) = 2u * (uint32_t)lane + 1u; }
    }
    if (R.has_v2v) {
        // bins 1..7: from the bin's start flags every event lane derives whether a code word starts / ends at its
        // event and, for an end, the finished ring word and the position of the word's first event
#define ICER_V2V_RECORD(EV, RK, FL, SP, WD)                                                            \
        if (((EV)&0x98u) == 0x80u && ((EV)&7u)) {                                                      \
            const uint32_t b_ = (EV)&7u, r_ = (RK), n_ = q.binn[b_];                                    \
            const uint32_t sw_ = window6(q.binstart[b_], (int)r_ - 4);  /* starts at ranks r-4 .. r+1 */ \
            const uint32_t bw_ = window6(q.binbits[b_], (int)r_ - 4);   /* input bits, same ranks */    \
            const uint32_t starts_ = (sw_ >> 4) & 1u;                                                   \
            const uint32_t carry_ = q.bincarry[b_];                                                     \
            const uint32_t ends_ = (r_ + 1u < n_) ? ((sw_ >> 5) & 1u) : (q.post_nin[b_] == 0u ? 1u : 0u); \
            uint32_t wd_ = 0, sp_ = 255;                                                                \
            if (ends_) {                                                                                \
                const uint32_t back_ = sw_ & 31u;                        /* starts at r-4 .. r */       \
                uint32_t acc_;                                                                          \
                if (back_) {                                                                            \
                    const uint32_t k_ = 31u - (uint32_t)clz32(back_);    /* start at rank r-4+k */      \
                    acc_ = (bw_ & 31u) >> k_;                                                           \
                    sp_ = q.binseq[b_][r_ - 4u + k_];                                                   \
                } else {                                                 /* the carried-in word */      \
                    const uint32_t cn_ = 31u - (uint32_t)clz32(carry_);                                 \
                    acc_ = (carry_ ^ (1u << cn_)) | (((bw_ & 31u) >> (4u - r_)) << cn_);                \
                }                                                                                       \
                const uint32_t e_ = s.tab.v2v[b_][acc_ & 31u];                                          \
                wd_ = kWordDone | (((e_ >> 4) & 15u) << 11) | (e_ >> 8);                                \
            }                                                                                           \
            FL = starts_ | (ends_ << 1); SP = sp_; WD = wd_;                                            \
        }
        FOR_LANES
        {
            ICER_V2V_RECORD(LV(c.ev1), LV(R.rb1), LV(c.fl1), LV(c.sp1), LV(c.wd1))
            ICER_V2V_RECORD(LV(c.ev2), LV(R.rb2), LV(c.fl2), LV(c.sp2), LV(c.wd2))
        }
#undef ICER_V2V_RECORD
    }
    // Golomb bins 8..16, no loop over bins: the lanes of one bin are found from per-bit ballots of (bin - 8); an
    // event's run length = zeros of its bin since the bin's previous one-event (or since the chunk start, plus the
    // run carried in), modulo m; a word starts where that is 0 and ends at a one or when the run reaches m - 1.
    // The lane holding a bin's last event of the chunk also leaves the bin's state (run length, open word).
    {
        LANEVAR(uint32_t, ka1); LANEVAR(uint32_t, ka2);     // run length after the event if it is the bin's last one, else ~0
        // 64 zero events of Golomb bins in at most two runs of lanes (first bin b0, then bin b1) and nothing else -- what
        // a blank chunk turns into once its context's estimate has settled; the bin changes where the estimate crosses
        // a cut-off or is rescaled.  The runs simply continue: the r-th event of a run sees run length (k + r) mod m.
        uint32_t cc = 64, e1 = 0;                                            // first lane and event of the second run
        bool runs = R.blank != 0u;
        if (runs) {
            const uint32_t e0 = READLANE(c.ev1, 0);
            const uint64_t D = BALLOT(LV(c.ev1) != e0);
            cc = D ? (uint32_t)ffs64(D) : 64u;
            e1 = READLANE(c.ev1, cc & 63u);
            runs = BALLOT((LV(c.ev1) & 0xB8u) < 0x88u || (LV(c.ev1) & 0x20u) != 0u || LV(c.ev2) != 0u || ((uint32_t)lane >= cc && LV(c.ev1) != e1)) == 0ull;
        }
        if (runs) {
            FOR_LANES
            {
                const uint32_t b = LV(c.ev1) & 31u, m = s.tab.gm[b], inv = s.tab.ginv[b];
                const uint32_t r = (uint32_t)lane >= cc ? (uint32_t)lane - cc : (uint32_t)lane;      // rank inside the run
                const uint32_t z = q.gk[b] + r;
                const uint32_t kb = z - ((z * inv) >> 20) * m;
                const uint32_t ends = kb + 1u == m ? 1u : 0u;
                const uint32_t first = kb <= r ? 2u * ((uint32_t)lane - kb) : 255u;               // first event of the word this event is in
                LV(c.fl1) = (kb == 0u ? 1u : 0u) | (ends << 1);
                LV(c.wd1) = kWordDone | (1u << 11) | 1u;
                if (ends) LV(c.sp1) = first;
                LV(ka1) = ((uint32_t)lane == 63u || (uint32_t)lane + 1u == cc) ? (ends ? 0u : kb + 1u) : ~0u;   // last event of its bin
                LV(ka2) = first;
            }
            WAVE_SYNC();
            FOR_LANES
            {
                if (LV(ka1) != ~0u) {
                    const uint32_t b = LV(c.ev1) & 31u;
                    q.binst[b] = st_pack(LV(ka1) ? LV(ka2) : 254u, LV(ka1), 0u);
                }
            }
        } else {
            const uint64_t G1 = BALLOT((LV(c.ev1) & 0x98u) >= 0x88u), G2 = BALLOT((LV(c.ev2) & 0x98u) >= 0x88u);
            if (G1 | G2) {
                // The events of a lane's bin (and the one-events among them) come from the masks phase C left in LDS;
                // everything an event needs to know about ITS OWN position is a lane-masked count of such a mask.
                // Zeros since the bin's previous one-event: that event's lane publishes the zeros before IT (onez).
                LANEVAR(uint32_t, zb1); LANEVAR(uint32_t, zb2);
                FOR_LANES
                {
                    LV(ka1) = ~0u; LV(ka2) = ~0u; LV(zb1) = 0; LV(zb2) = 0;
                    if ((LV(c.ev1) & 0x98u) >= 0x88u) {
                        const uint64_t *g = q.gmask[(LV(c.ev1) & 31u) - 8u];
                        LV(zb1) = cnt_lt_own(g[0] & ~g[2], g[1] & ~g[3], lane, 0u);
                        if (LV(c.ev1) & 0x20u) q.onez[2 * lane] = (uint8_t)LV(zb1);
                    }
                    if ((LV(c.ev2) & 0x98u) >= 0x88u) {
                        const uint64_t *g = q.gmask[(LV(c.ev2) & 31u) - 8u];
                        LV(zb2) = cnt_lt_own(g[0] & ~g[2], g[1] & ~g[3], lane, 1u);
                        if (LV(c.ev2) & 0x20u) q.onez[2 * lane + 1] = (uint8_t)LV(zb2);
                    }
                }
                WAVE_SYNC();
#define ICER_GOLOMB_LANE(EV, SLOT, ZB, KA, FL, WD)                                                              \
                if (((EV) & 0x98u) >= 0x88u) {                                                                \
                    const uint32_t b_ = (EV) & 31u;                                                            \
                    const uint64_t *g_ = q.gmask[b_ - 8u];                                                     \
                    const int lo_ = last_lt_own(g_[2], g_[3], lane, (SLOT));                                   \
                    const uint32_t z_ = lo_ >= 0 ? (ZB) - (uint32_t)q.onez[lo_] : q.gk[b_] + (ZB);              \
                    const uint32_t m = s.tab.gm[b_], inv = s.tab.ginv[b_];                                      \
                    const uint32_t kb_ = z_ - ((z_ * inv) >> 20) * m;                                           \
                    const uint32_t bit_ = ((EV) >> 5) & 1u;                                                     \
                    FL = (kb_ == 0u ? 1u : 0u) | ((bit_ || kb_ + 1u == m) ? 2u : 0u);                           \
                    WD = bit_ ? wg_golomb_word(s.tab, (int)b_, kb_) : (kWordDone | (1u << 11) | 1u);            \
                    const uint32_t before_ = cnt_lt_own(g_[0], g_[1], lane, (SLOT));                            \
                    if (before_ + 1u == (uint32_t)q.gn[b_ - 8u]) KA = (FL & 2u) ? 0u : kb_ + 1u;                \
                }
                FOR_LANES
                {
                    ICER_GOLOMB_LANE(LV(c.ev1), 0u, LV(zb1), LV(ka1), LV(c.fl1), LV(c.wd1))
                    ICER_GOLOMB_LANE(LV(c.ev2), 1u, LV(zb2), LV(ka2), LV(c.fl2), LV(c.wd2))
                }
#undef ICER_GOLOMB_LANE
                // word starts of the Golomb bins; an end event's word began at its bin's latest start, and so did the
                // word a bin's last event leaves open
                WAVE_SYNC();
                const uint64_t SB1 = G1 & BALLOT(LV(c.fl1) & 1u), SB2 = G2 & BALLOT(LV(c.fl2) & 1u);
                FOR_LANES
                {
                    if ((LV(c.ev1) & 0x98u) >= 0x88u && ((LV(c.fl1) & 2u) || LV(ka1) != ~0u)) {
                        const uint64_t *g = q.gmask[(LV(c.ev1) & 31u) - 8u];
                        const int sp = last_le_own(SB1 & g[0], SB2 & g[1], lane, 0u);
                        if (LV(c.fl1) & 2u) LV(c.sp1) = sp < 0 ? 255u : (uint32_t)sp;
                        if (LV(ka1) != ~0u) q.binst[LV(c.ev1) & 31u] = st_pack(LV(ka1) ? (sp < 0 ? 255u : (uint32_t)sp) : 254u, LV(ka1), 0u);
                    }
                    if ((LV(c.ev2) & 0x98u) >= 0x88u && ((LV(c.fl2) & 2u) || LV(ka2) != ~0u)) {
                        const uint64_t *g = q.gmask[(LV(c.ev2) & 31u) - 8u];
                        const int sp = last_le_own(SB1 & g[0], SB2 & g[1], lane, 1u);
                        if (LV(c.fl2) & 2u) LV(c.sp2) = sp < 0 ? 255u : (uint32_t)sp;
                        if (LV(ka2) != ~0u) q.binst[LV(c.ev2) & 31u] = st_pack(LV(ka2) ? (sp < 0 ? 255u : (uint32_t)sp) : 254u, LV(ka2), 0u);
                    }
                }
            }
        }
    }
    WAVE_SYNC();
    c.S1 = BALLOT(LV(c.fl1) & 1u);
    c.S2 = BALLOT(LV(c.fl2) & 1u);
    // a word's ring slot = allocation count before the chunk + the number of word starts before its first event
    // (E2): every start event publishes that rank for the end events and the bins' open words
    FOR_LANES
    {
        if (LV(c.fl1) & 1u) q.srank[2 * lane] = (uint8_t)cnt_lt_own(c.S1, c.S2, lane, 0u);
        if (LV(c.fl2) & 1u) q.srank[2 * lane + 1] = (uint8_t)cnt_lt_own(c.S1, c.S2, lane, 1u);
    }
    WAVE_SYNC();
    // the chunk's ring summary: words opened; per bin closed / the chunk-relative slot of the word left open
    FOR_LANES
    {
        LV(c.st) = lane < 17 ? q.binst[lane] : 255u;
        if (lane < 17) {
            const uint32_t op = LV(c.st) & 255u;
            q.sumE[lane] = (uint16_t)(op < 128u ? (uint32_t)q.srank[op] : op);
        }
        if (lane == 0) q.nst = (uint32_t)(popc64(c.S1) + popc64(c.S2));
    }
}

// position of the first end event of every bin in this wave's chunk (needed by the forced-flush test only)
ICER_DEV void first_ends(Shared &s, Wave &R, uint32_t w)
{
    DECL_LANE;
    WaveLds &q = s.wl[w];
    const MergeChunk &c = R.c;
    const uint64_t F1 = BALLOT(LV(c.fl1) & 2u), F2 = BALLOT(LV(c.fl2) & 2u);
    const uint64_t A0 = BALLOT(LV(c.ev1) & 1u), A1 = BALLOT(LV(c.ev1) & 2u), A2 = BALLOT(LV(c.ev1) & 4u), A3 = BALLOT(LV(c.ev1) & 8u), A4 = BALLOT(LV(c.ev1) & 16u);
    const uint64_t D0 = BALLOT(LV(c.ev2) & 1u), D1 = BALLOT(LV(c.ev2) & 2u), D2 = BALLOT(LV(c.ev2) & 4u), D3 = BALLOT(LV(c.ev2) & 8u), D4 = BALLOT(LV(c.ev2) & 16u);
    FOR_LANES
    {
        if (lane < 17) {
            const uint32_t k = (uint32_t)lane;
            const uint64_t m1 = ICER_MATCH4(k, F1, A0, A1, A2, A3) & ((k & 16u) ? A4 : ~A4);
            const uint64_t m2 = ICER_MATCH4(k, F2, D0, D1, D2, D3) & ((k & 16u) ? D4 : ~D4);
            q.fe[lane] = (uint8_t)first_pos(m1, m2);
        }
    }
}

// ==========================================================================================
// phase F: ring slots, forced-flush test, commit
// ==========================================================================================
// ring slot of every bin's open word at the start of this wave's chunk, allocation count before it; returns the
// window's total
ICER_DEV void phase_f_scan(Shared &s, Wave &R, uint32_t w, uint32_t wbase, uint32_t nwin)
{
    DECL_LANE;
    FOR_LANES { LV(R.slot) = LV(R.slot0); }
    uint32_t run = R.tail;
    for (uint32_t v = wbase; v < w && v < nwin; v++) {
        const WaveLds &p = s.wl[v];
        FOR_LANES
        {
            if (lane < 17) {
                const uint32_t op = p.sumE[lane];
                if (op == 254u) LV(R.slot) = ~0u;
                else if (op <= 128u) LV(R.slot) = run + op;
            }
        }
        run += p.nst;
    }
    R.tailw = run;
    FOR_LANES { if (lane < 17) s.wl[w].bslot[lane] = (int32_t)LV(R.slot); }
}

// Forced-flush test (see the header).  hmin = slot of the oldest word that is open before chunk wbase.  Returns
// true when the window's pending chunks [wbase, nwin) could reach slot hmin + 2048, i.e. the detailed test is needed.
ICER_DEV bool flush_possible(Shared &s, Wave &R, uint32_t wbase, uint32_t nwin, uint32_t *total_out)
{
    DECL_LANE;
    uint32_t total = 0;
    for (uint32_t v = wbase; v < nwin; v++) total += s.wl[v].nst;
    *total_out = total;
    const uint64_t open = BALLOT(lane < 17 && (int32_t)LV(R.slot0) >= 0 && R.tail + total - LV(R.slot0) > (uint32_t)kRingWords);
    return open != 0ull;
}

// first pending chunk in which a word start may find the ring full (nwin if none).  Every wave computes the same value.
ICER_DEV uint32_t flush_chunk(Shared &s, Wave &R, uint32_t wbase, uint32_t nwin, uint32_t total)
{
    DECL_LANE;
    LANEVAR(uint32_t, cand);
    FOR_LANES
    {
        LV(cand) = nwin;
        if (lane < 17 && (int32_t)LV(R.slot0) >= 0) {
            // the word start that would allocate slot s_b + 2048 is the trig-th start of the pending chunks
            const uint32_t trig = LV(R.slot0) + (uint32_t)kRingWords - R.tail;
            if (trig < total) {
                uint32_t cum = 0, wT = nwin, wE = nwin;
                for (uint32_t v = wbase; v < nwin; v++) {
                    const uint32_t n = s.wl[v].nst;
                    if (wT == nwin && trig < cum + n) wT = v;
                    cum += n;
                    if (wE == nwin && s.wl[v].fe[lane] != 255u) wE = v;
                }
                // the bin's word ends in an earlier chunk: the slot is free by then.  Same chunk: left to exact_chunk.
                if (wT <= wE) LV(cand) = wT;
            }
        }
    }
    uint32_t wL = nwin;
    for (uint32_t b = 1; b < 17; b++) { const uint32_t v = READLANE(cand, b); wL = v < wL ? v : wL; }
    return wL;
}

// the bins' open words and coder state after the chunk (slots are allocation counts)
ICER_DEV void commit_bins(Shared &s, MergeChunk &c, uint32_t tail)
{
    DECL_LANE;
    const uint64_t S1 = c.S1, S2 = c.S2;
    FOR_LANES
    {
        if (lane >= 1 && lane < kNumBins) {
            const uint32_t op = LV(c.st) & 255u;
            if (op == 254u) s.bin_slot[lane] = -1;
            else if (op < 128u) s.bin_slot[lane] = (int32_t)(tail + cnt_lt(S1, S2, op));
            s.bin_state[lane] = LV(c.st);
        }
    }
}

// ring stores for the chunk's events at positions [lo, hi): the finished words of the code words that end there (slot of a word = tail0 + number of word starts before its first event, E2).
// `bslot` = the bins' open slots at chunk start.
ICER_DEV void commit_range(Shared &s, MergeChunk &c, const int32_t *bslot, uint32_t tail0, uint32_t lo, uint32_t hi, const uint8_t *srank)
{
    DECL_LANE;
    const uint64_t S1 = c.S1, S2 = c.S2;
    // `srank`: the start ranks phase E published (null: the start flags have changed since, count them)
    FOR_LANES
    {
        const uint32_t p1 = 2u * (uint32_t)lane, p2 = p1 + 1u;
        if ((LV(c.fl1) & 2u) && p1 >= lo && p1 < hi) {
            const uint32_t slot = LV(c.sp1) == 255u ? (uint32_t)bslot[LV(c.ev1) & 31u] : tail0 + (srank ? (uint32_t)srank[LV(c.sp1) & 127u] : cnt_lt(S1, S2, LV(c.sp1)));
            WRING_ST(slot, LV(c.wd1));
        }
        if ((LV(c.fl2) & 2u) && p2 >= lo && p2 < hi) {
            const uint32_t slot = LV(c.sp2) == 255u ? (uint32_t)bslot[LV(c.ev2) & 31u] : tail0 + (srank ? (uint32_t)srank[LV(c.sp2) & 127u] : cnt_lt(S1, S2, LV(c.sp2)));
            WRING_ST(slot, LV(c.wd2));
        }
    }
    WAVE_SYNC();
}

// next event of a bin: lowest position in (Q1: even positions 2 * lane, Q2: odd positions 2 * lane + 1); removes it
#define ICER_NEXT_EVENT(Q1, Q2, POS, V)                                                                    \
    {                                                                                                      \
        const uint32_t q1_ = (Q1) ? 2u * (uint32_t)ffs64(Q1) : 999u, q2_ = (Q2) ? 2u * (uint32_t)ffs64(Q2) + 1u : 999u; \
        if (q1_ < q2_) { POS = q1_; V = READLANE(c.ev1, q1_ >> 1); (Q1) &= (Q1) - 1ull; }                  \
        else { POS = q2_; V = READLANE(c.ev2, q2_ >> 1); (Q2) &= (Q2) - 1ull; }                             \
    }

// A chunk inside which the ring may fill up, coded by ITS wave alone while the others wait at the barrier; Shared
// holds the state before the chunk and the ring is drained (popped = the oldest open word).  The records (c) are right
// up to the first word start that finds the ring full -- position P, known from the ring occupancy alone -- so events
// before P are committed as computed.  Then everything finished is popped (the reference pops after every event;
// popping is only observable through `used` when a word is allocated) and, if the ring is still full, the oldest word
// is force-completed (E5, icer_flush_encode icer_encoding.c:141-189: it belongs to bin hb, whose state at P follows
// from hb's events since the word's start).  A forced flush changes the word boundaries of bin hb only: hb's
// events from P on are replayed one by one (icer_encode_bit, icer_encoding.c:37-112) with the bin starting afresh,
// all other results stay valid, and the scheme repeats from P.
ICER_DEV void exact_chunk(Shared &s, MergeChunk &c, uint32_t tail0)
{
    DECL_LANE;
    uint32_t base = 0;
    for (;;) {
        const uint64_t S1 = c.S1, S2 = c.S2;
        const uint32_t t = (uint32_t)kRingWords - (tail0 - s.popped);           // rank of the first word start that finds the ring full
        if (t >= (uint32_t)(popc64(S1) + popc64(S2))) break;
        const uint64_t h1 = BALLOT((LV(c.fl1) & 1u) && cnt_lt_own(S1, S2, lane, 0u) == t);
        const uint64_t h2 = BALLOT((LV(c.fl2) & 1u) && cnt_lt_own(S1, S2, lane, 1u) == t);
        const uint32_t P = h1 ? 2u * (uint32_t)ffs64(h1) : 2u * (uint32_t)ffs64(h2) + 1u;
        commit_range(s, c, s.bin_slot, tail0, base, P, nullptr);
        const uint32_t alloc = tail0 + t;
        wave_drain(s, alloc);
        if (alloc - s.popped == (uint32_t)kRingWords) {
            // still full: the head word is open.  Its bin, and that bin's state just before P:
            const uint32_t head = s.popped;
            // (the ring holds kRingWords words and a chunk opens at most 128: the head was open before this chunk)
            const uint32_t hb = (uint32_t)ffs64(BALLOT(lane >= 1 && lane < kNumBins && s.bin_slot[lane] == (int32_t)head));
            WG_ASSERT(hb >= 1u && hb < (uint32_t)kNumBins);
            WG_STAT(3);
            const uint64_t E1 = BALLOT((LV(c.ev1) & 0x9Fu) == (0x80u | hb)), E2 = BALLOT((LV(c.ev2) & 0x9Fu) == (0x80u | hb));
            const int x = last_lt(S1 & E1, S2 & E2, P);                           // first event of the open word, -1: carried in
            const uint32_t lo = x < 0 ? 0u : (uint32_t)x;
            uint64_t R1 = E1 & below64((P + 1u) >> 1) & ~below64((lo + 1u) >> 1);   // hb's events in [lo, P)
            uint64_t R2 = E2 & below64(P >> 1) & ~below64(lo >> 1);
            uint32_t hacc = x < 0 ? st_acc(s.bin_state[hb]) : 0u, hnin = x < 0 ? st_nin(s.bin_state[hb]) : 0u;
            uint32_t word;
            if (hb >= 8u) {
                hacc += (uint32_t)(popc64(R1) + popc64(R2));                       // all zeros, or the word would have ended
                word = (hacc == (uint32_t)s.tab.gm[hb] - 1u) ? (kWordDone | (1u << 11) | 1u) : wg_golomb_word(s.tab, (int)hb, hacc);
            } else {
                while (R1 | R2) {
                    uint32_t q, v;
                    ICER_NEXT_EVENT(R1, R2, q, v)
                    (void)q;
                    hacc |= ((v >> 5) & 1u) << hnin;
                    hnin++;
                }
                const uint32_t f = s.tab.v2v_flush[hb][hacc > 8u ? 8u : hacc][hnin > 5u ? 5u : hnin];
                const uint32_t en = s.tab.v2v[hb][(hacc | ((f & 15u) << hnin)) & 31u];
                word = kWordDone | (((en >> 4) & 15u) << 11) | (en >> 8);       // QUIRK (kept): not checked to be a code word
            }
            FOR_LANES
            {
                if (lane == 0) { WRING_ST(head, word); s.bin_slot[hb] = -1; }
            }
            WAVE_SYNC();
            // bin hb starts afresh at P: replay its remaining events of the chunk
            uint64_t Q1 = E1 & ~below64((P + 1u) >> 1), Q2 = E2 & ~below64(P >> 1);
            bool open = false;
            uint32_t acc = 0, nin = 0, spos = 255;
            while (Q1 | Q2) {
                uint32_t q, v;
                ICER_NEXT_EVENT(Q1, Q2, q, v)
                const uint32_t bit = (v >> 5) & 1u;
                const uint32_t starts = open ? 0u : 1u;
                if (!open) { open = true; spos = q; }
                uint32_t wd = 0;
                bool close = false;
                if (hb >= 8u) {
                    if (bit) { wd = wg_golomb_word(s.tab, (int)hb, acc); close = true; }
                    else if (acc + 1u >= s.tab.gm[hb]) { wd = kWordDone | (1u << 11) | 1u; close = true; }
                    else acc++;
                } else if (hb >= 1u) {
                    acc |= bit << nin;
                    nin++;
                    const uint32_t en = s.tab.v2v[hb][acc & 31u];
                    if ((en & 15u) == nin) { wd = kWordDone | (((en >> 4) & 15u) << 11) | (en >> 8); close = true; }
                } else {
                    wd = kWordDone | (1u << 11) | bit;
                    close = true;
                }
                const uint32_t fl = starts | (close ? 2u : 0u), sp = close ? spos : 255u;
                FOR_LANES
                {
                    if ((uint32_t)lane == (q >> 1)) {
                        if (q & 1u) { LV(c.fl2) = fl; LV(c.sp2) = sp; LV(c.wd2) = wd; }
                        else { LV(c.fl1) = fl; LV(c.sp1) = sp; LV(c.wd1) = wd; }
                    }
                }
                if (close) { open = false; acc = 0; nin = 0; }
            }
            FOR_LANES
            {
                // (no event of hb after P: the bin is simply closed; 254 also keeps commit_bins from re-opening it)
                if ((uint32_t)lane == hb) LV(c.st) = st_pack(open ? spos : 254u, acc, nin);
            }
            c.S1 = BALLOT(LV(c.fl1) & 1u);
            c.S2 = BALLOT(LV(c.fl2) & 1u);
            wave_drain(s, alloc);
        }
        base = P;
    }
    commit_range(s, c, s.bin_slot, tail0, base, 128u, nullptr);
    commit_bins(s, c, tail0);
    WAVE_SYNC();
}
#undef ICER_NEXT_EVENT

// ==========================================================================================
// the unit
// ==========================================================================================
// Returns the payload length in bits, kUnitTooBig (payload slot too small) or kUnitStopped (progressive mode).
// GPU: called by all kWgWaves * 64 threads of the workgroup, `regs` in registers; Shared initialised (unit_state_init,
// tables) and a barrier passed.
//
// Barriers per window (no forced flush in sight): A | B C | D E | F | drain 1 | drain 2 -- the store of the drained
// payload words (drain 3) is done by the next region that comes along.

// phase F of one wave: the first pending chunk that may hit a full ring (detailed test only), ring slots, ring stores
// of the chunks before it, the coder state they leave, the allocation count
ICER_DEV void phase_f_commit(Shared &s, Wave &R, uint32_t w, uint32_t wbase, uint32_t nwin, bool detailed)
{
    DECL_LANE;
    if (detailed) { R.wL = flush_chunk(s, R, wbase, nwin, R.nflush); if (w == 0) WG_STAT(1); }
    const uint32_t wL = R.wL;
    phase_f_scan(s, R, w, wbase, wL);
    if (w >= wbase && w < wL) {
        commit_range(s, R.c, s.wl[w].bslot, R.tailw, 0u, 128u, s.wl[w].srank);
        if (w + 1u == wL) {
            // the last committed chunk leaves the coder state (the next chunk's start state)
            const MergeChunk &c = R.c;
            FOR_LANES
            {
                if (lane >= 1 && lane < kNumBins) {
                    const uint32_t op = LV(c.st) & 255u;
                    s.bin_slot[lane] = op == 254u ? -1 : op < 128u ? (int32_t)(R.tailw + (uint32_t)s.wl[w].sumE[lane]) : (int32_t)LV(R.slot);
                    s.bin_state[lane] = LV(c.st);
                }
            }
        }
    }
    // (uniform) allocation count after the committed chunks
    { uint32_t t = R.tail; for (uint32_t v = wbase; v < wL; v++) t += s.wl[v].nst; R.tail = t; }
}

// ---- runs of blank chunks in closed form ------------------------------------------------------------------
// nev = 64 * (number of blank chunks) zero events of context 0 and nothing else.  Their bins follow from the context's
// counts alone -- the estimate zero / total only rises from event to event, except for a slight drop where the counts are
// halved (at most one bin down) --, so the events fall into a few SEGMENTS of one Golomb bin each, ending where the
// estimate crosses the bin's upper cut-off or the counts are rescaled; and n zero events of a Golomb bin with parameter
// m just fill up words of m zeros (code word "1", icer_encoding.c:68-72).  blank_run_chunks: how many of the nb chunks the closed form may take (0: it does not apply) -- the
// estimate is not folded, every bin on the way is a Golomb bin, and no word start can find the ring full whatever the
// segments turn out to be (an upper bound of the words they open; also keeps the physical ring from overflowing, since
// nothing is popped here).  Every wave evaluates it (wave-uniform, same answer).
ICER_DEV uint32_t blank_run_chunks(Shared &s, const Wave &R, uint32_t par, uint32_t nb)
{
    const uint32_t z = s.czer[par][0], t = s.ctot[par][0];
    if (z < (t >> 1)) return 0u;
    const uint32_t b0 = pick_bin(s.tab.binlut, z, t);
    if (b0 < 9u) return 0u;
    // words the run may open: no bin on the way lies more than one below the first one (see above), a word holds m zeros,
    // and every bin may have one more word begun than finished; as many chunks as that leaves room for
    const uint32_t used = R.tail - R.popped + (uint32_t)kNumBins;
    if (used >= (uint32_t)kRingWords) return 0u;
    const uint32_t fit = ((uint32_t)kRingWords - used) * (uint32_t)s.tab.gm[b0 - 1u] / 64u;
    return nb < fit ? nb : fit;
}

// ceil(num / den) for 0 < num < 2^25, 144 <= den < 2^16 (one float reciprocal, corrected: no integer division)
ICER_DEV uint32_t ceil_div_small(uint32_t num, uint32_t den)
{
#ifdef ICER_WAVE_EMU
    uint32_t q = (uint32_t)((float)num * (1.0f / (float)den));
#else
    uint32_t q = (uint32_t)((float)num * __builtin_amdgcn_rcpf((float)den));
#endif
    int32_t rem = (int32_t)num - (int32_t)(q * den);            // the estimate is within 2 of the quotient
    if (rem < 0) { q--; rem += (int32_t)den; }
    if (rem < 0) { q--; rem += (int32_t)den; }
    if (rem >= (int32_t)den) { q++; rem -= (int32_t)den; }
    if (rem >= (int32_t)den) { q++; rem -= (int32_t)den; }
    return q + (rem ? 1u : 0u);
}

// floor(num / den) for num < 2^24, 1 <= den < 2^16 (the same scheme)
ICER_DEV uint32_t floor_div_small(uint32_t num, uint32_t den)
{
#ifdef ICER_WAVE_EMU
    uint32_t q = (uint32_t)((float)num * (1.0f / (float)den));
#else
    uint32_t q = (uint32_t)((float)num * __builtin_amdgcn_rcpf((float)den));
#endif
    int32_t rem = (int32_t)num - (int32_t)(q * den);
    if (rem < 0) { q--; rem += (int32_t)den; }
    if (rem < 0) { q--; rem += (int32_t)den; }
    if (rem >= (int32_t)den) { q++; rem -= (int32_t)den; }
    if (rem >= (int32_t)den) q++;
    return q;
}

// One wave; the others wait at the barrier that follows.  Leaves the counts in the OTHER copy (the caller flips `par`)
// and the allocation count in Shared::run_tail[par].  Lane b keeps bin b's run length, open slot and Golomb parameters in
// registers for the whole run (read by readlane with the wave-uniform bin number): the loop over the segments touches LDS
// only to store finished words.
ICER_DEV void blank_run(Shared &s, const Wave &R, uint32_t par, uint32_t nev)
{
    DECL_LANE;
    uint32_t z = s.czer[par][0], t = s.ctot[par][0], tail = R.tail;
    const uint32_t one_word = kWordDone | (1u << 11) | 1u;
    LANEVAR(uint32_t, run); LANEVAR(uint32_t, slot); LANEVAR(uint32_t, gm); LANEVAR(uint32_t, ginv); LANEVAR(uint32_t, cutv);
    FOR_LANES
    {
        const bool gb = lane >= 8 && lane < kNumBins;
        LV(run) = gb ? st_acc(s.bin_state[lane]) : 0u;
        LV(slot) = gb ? (uint32_t)s.bin_slot[lane] : ~0u;
        LV(gm) = gb ? (uint32_t)s.tab.gm[lane] : 1u;
        LV(ginv) = gb ? s.tab.ginv[lane] : 0u;
        LV(cutv) = lane < 16 ? s.tab.cut[lane] : 0u;            // the cut-off above bin `lane`
    }
    uint32_t touched = 0;                                       // bins whose state changed
    const uint32_t half = kRescaleCap / 2;
    while (nev) {
        // ---- steady state.  With only zero events the counts of context 0 run in cycles of `half` events from one
        // rescale to the next (total half -> kRescaleCap -> half, zero -> (zero + half) >> 1), and a cycle that leaves the
        // zero count where it found it repeats for as long as the chunks stay blank: the same one or two segments (bin,
        // length) every time.  All whole cycles that are left are then taken in one step -- one lane per code word that
        // gets started, its ring slot from counting the words the bins start before it.
        if (t == half && nev >= half && ((z + half) >> 1) == z) {
            uint32_t pb[2] = {0u, 0u}, pn[2] = {0u, 0u}, nseg = 0, zz = z, tt = half, rest = half;
            while (rest && nseg < 2u) {                        // the segments of one cycle
                const uint32_t bb = pick_bin(s.tab.binlut, zz, tt);
                uint32_t nn = rest;
                if (bb < 16u) {
                    const uint32_t cut = READLANE(cutv, bb);
                    const uint32_t nb = ceil_div_small(tt * cut - (zz << 16), 65536u - cut);
                    nn = nb < nn ? nb : nn;
                }
                pb[nseg] = bb; pn[nseg] = nn; nseg++;
                zz += nn; tt += nn; rest -= nn;
            }
            const bool usable = rest == 0u && pb[0] >= 8u && (nseg == 1u || (pb[1] >= 8u && pb[1] != pb[0]));
            if (usable) {
                uint32_t m_[2], k_[2], sl_[2];
                for (uint32_t q = 0; q < 2u; q++) {
                    const bool on = q < nseg;
                    m_[q] = on ? READLANE(gm, pb[q]) : 1u;
                    sl_[q] = on ? READLANE(slot, pb[q]) : ~0u;
                    k_[q] = (on && (int32_t)sl_[q] >= 0) ? READLANE(run, pb[q]) : 0u;
                }
                const uint32_t cycles = nev / half;
                uint32_t w_[2];                                // words the bins finish in these cycles
                w_[0] = floor_div_small(k_[0] + cycles * pn[0], m_[0]);
                w_[1] = nseg > 1u ? floor_div_small(k_[1] + cycles * pn[1], m_[1]) : 0u;
                {
                    // ring slot of the j-th word of pattern bin q (counting the one that may be open at the start, which has
                    // its slot).  It starts at the (j * m - k)-th zero the bin gets here, in cycle c; before that event lie
                    // the bin's own earlier starts and the other bin's starts through cycle c (its segment comes first) or
                    // before cycle c (it comes second).
                    const auto slot_of = [&](uint32_t q, uint32_t j) -> uint32_t {
                        if (j == 0u && k_[q] > 0u) return sl_[q];
                        const uint32_t o = q ^ 1u, c = floor_div_small(j * m_[q] - k_[q], pn[q]);
                        uint32_t before = j - (k_[q] > 0u ? 1u : 0u);
                        if (nseg > 1u) {
                            const uint32_t fed = k_[o] + (o < q ? c + 1u : c) * pn[o];
                            before += floor_div_small(fed + m_[o] - 1u, m_[o]) - (k_[o] > 0u ? 1u : 0u);
                        }
                        return tail + before;
                    };
                    const uint32_t nwords = w_[0] + w_[1];     // every finished word here is a full run; a lane per word
                    FOR_LANES
                    {
                        for (uint32_t i = (uint32_t)lane; i < nwords; i += 64u) {
                            const uint32_t q = i < w_[0] ? 0u : 1u;
                            WRING_ST(slot_of(q, q ? i - w_[0] : i), one_word);
                        }
                    }
                    uint32_t started = 0;
                    for (uint32_t q = 0; q < nseg; q++) {
                        const uint32_t fed = k_[q] + cycles * pn[q], kept = fed - w_[q] * m_[q];
                        const uint32_t open_at = kept ? slot_of(q, w_[q]) : ~0u;         // (the word in progress)
                        const uint32_t bq = pb[q];
                        FOR_LANES
                        {
                            if ((uint32_t)lane == bq) { LV(run) = kept; LV(slot) = open_at; }
                        }
                        touched |= 1u << bq;
                        started += floor_div_small(fed + m_[q] - 1u, m_[q]) - (k_[q] > 0u ? 1u : 0u);
                    }
                    tail += started;
                    nev -= cycles * half;
                    WG_STAT(5);
                    continue;
                }
            }
        }
        const uint32_t b = pick_bin(s.tab.binlut, z, t);
        WG_ASSERT(b >= 8u && z >= (t >> 1));
        uint32_t n = kRescaleCap - t;                 // events up to and including the one that triggers the rescale
        if (b < 16u) {
            // first event that sees the next bin: (z + i) * 65536 >= (t + i) * cut
            const uint32_t cut = READLANE(cutv, b);
            const uint32_t nb = ceil_div_small(t * cut - (z << 16), 65536u - cut);
            n = nb < n ? nb : n;
        }
        n = nev < n ? nev : n;
        const uint32_t m = READLANE(gm, b), inv = READLANE(ginv, b);
        uint32_t k = READLANE(run, b);
        uint32_t sl = READLANE(slot, b);
        uint32_t left = n;
        if ((int32_t)sl < 0) { sl = tail++; k = 0; }                             // a word starts at the segment's first event
        if (left >= m - k) {
            left -= m - k;                                                   // the open word is filled up ...
            const uint32_t q = (left * inv) >> 20;                           // ... then q whole words, each started by the next event
            left -= q * m;
            FOR_LANES
            {
                if (lane == 0) WRING_ST(sl, one_word);
                for (uint32_t i = (uint32_t)lane; i < q; i += 64u) WRING_ST(tail + i, one_word);
            }
            tail += q;
            if (left) { sl = tail++; k = left; }
            else { sl = ~0u; k = 0; }
        } else k += left;
        FOR_LANES
        {
            if ((uint32_t)lane == b) { LV(run) = k; LV(slot) = sl; }
        }
        touched |= 1u << b;
        z += n; t += n;
        if (t >= kRescaleCap) { t = kRescaleCap / 2; if (z > kRescaleCap / 2) z >>= 1; }     // QUIRK C5
        nev -= n;
    }
    // results go where nobody is reading: the other copy of the counts (a slower wave may still be evaluating
    // blank_run_chunks on this one) and this run's copy of the allocation count
    FOR_LANES
    {
        if (lane < 32 && ((touched >> lane) & 1u)) { s.bin_state[lane] = st_pack(0u, LV(run), 0u); s.bin_slot[lane] = (int32_t)LV(slot); }
        if (lane < 17) { s.czer[par ^ 1u][lane] = lane ? s.czer[par][lane] : z; s.ctot[par ^ 1u][lane] = lane ? s.ctot[par][lane] : t; }
        if (lane == 0) s.run_tail[par] = tail;
    }
    WAVE_SYNC();
}

// drain 3: complete 32-bit words of the bit stage -> HBM (every word is stored exactly once, by the whole workgroup)
ICER_DEV void store_stage_words(Shared &s, const UnitArgs &a, Wave &R, uint32_t w)
{
    DECL_LANE;
    const uint32_t first = R.flushed_words, last = R.bitpos >> 5;
    const bool fits = last <= a.cap_words;
    const uint32_t stop = fits ? last : a.cap_words;
    FOR_LANES
    {
        for (uint32_t wi = first + w * 64u + (uint32_t)lane; wi < last; wi += 64u * kWgWaves) {
            const uint32_t v = s.stage[wi & (kStageWords - 1)];
            if (wi < stop) a.out_words[wi] = v;
            s.stage[wi & (kStageWords - 1)] = 0;
        }
    }
    R.flushed_words = last;
    // a unit whose complete bytes reach the capacity can never fit (P3 of SURVEY.md 2.3; HISTORY.md 3)
    if (!(fits && (R.bitpos >> 3) < a.cap_words * 4u)) R.too_big = 1u;
}

ICER_DEV uint32_t code_unit_wg(Shared &s, const UnitArgs &a, WG_REGS_PARAM(Wave, regs))
{
    DECL_LANE;
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    WG_EACH_WAVE
        wave_init(s, a, R, w);
    WG_BARRIER
    uint32_t windows = 0, par = 0;
    bool store_pending = false;          // (uniform) drained payload words are waiting in the bit stage
    const uint32_t nfull = (a.w * a.h) / 64u;                    // (a last chunk with fewer than 64 pixels is never blank)
    uint32_t look_base = ~0u;                                    // (uniform) the 64 chunks whose blank mask is in look_mask
    uint64_t look_mask = 0;
    for (uint32_t j0 = 0; j0 < nchunks; windows++) {
        // progressive mode: has the byte quota been used up by units of higher priority in the meantime?
        const bool check_stop = a.early_quota && (windows & 15u) == 15u;
        // ---- a run of blank chunks (chunk table) is coded in closed form by one wave
        if (a.sig) {
            uint32_t nb;
            // The 64 chunks around j0 as a mask (one byte of the chunk table per lane, one ballot), kept from iteration to iteration: in a
            // mid-sparse unit -- a run of a few dozen blank chunks, a chunk with content, the next run -- most iterations start inside the
            // block the last one looked at and cost no load at all.  Only a run that reaches the block's end takes the wide look below
            // (round 6: the wide look at EVERY iteration, sixteen dependent-latency loads per lane, was half of such a unit's chain).
            // A run that reaches the block's end is followed block by block (one load each; the last block looked at is the one the next
            // iteration starts in), up to four blocks; only a longer one takes the wide look at 1 024 chunks.
            nb = 0;
            bool ended = false;
            for (uint32_t step = 0; step < 4u && !ended; step++) {
                const uint32_t pos = j0 + nb;
                if ((pos & ~63u) != look_base) {
                    look_base = pos & ~63u;
                    look_mask = BALLOT(look_base + (uint32_t)lane < nfull && (uint32_t)a.lsb >= (uint32_t)a.sig[look_base + (uint32_t)lane < nfull ? look_base + (uint32_t)lane : 0u]);
                }
                const uint32_t in_block = 64u - (pos & 63u);
                const uint64_t not_blank = ~(look_mask >> (pos & 63u));                  // (bits at and above in_block: ones)
                const uint32_t run_here = (uint32_t)ffs64(not_blank) < in_block ? (uint32_t)ffs64(not_blank) : in_block;
                nb += run_here;
                ended = run_here < in_block;
            }
            if (!ended) {
                // (every lane looks at kBlankLook entries of the chunk table: 64 * kBlankLook chunks per look)
                const uint32_t from = j0 + nb;
                LANEVAR(uint32_t, lead);
                FOR_LANES
                {
                    uint32_t c = 0, open = 1u;
                    for (uint32_t i = 0; i < kBlankLook; i++) {
                        const uint32_t j = from + (uint32_t)lane * kBlankLook + i;
                        open &= (j < nfull && (uint32_t)a.lsb >= (uint32_t)a.sig[j < nfull ? j : 0u]) ? 1u : 0u;
                        c += open;
                    }
                    LV(lead) = c;
                }
                const uint32_t first = (uint32_t)ffs64(~BALLOT(LV(lead) == kBlankLook));
                nb += first * kBlankLook + (first < 64u ? READLANE(lead, first) : 0u);
            }
            if (nb > kBlankRunMax) nb = kBlankRunMax;
            bool ok = false;
            WG_EACH_WAVE
                (void)w;
                if (nb >= kBlankRunMin) nb = blank_run_chunks(s, R, par, nb);
                ok = nb >= kBlankRunMin;
            WG_BARRIER_NONE
            if (ok) {
                WG_EACH_WAVE
                    WG_TICK(0)
                    if (w == 0) {
                        if (check_stop && quota_already_spent(a)) { FOR_LANES { if (lane == 0) s.stop = 1u; } }
                        blank_run(s, R, par, nb * 64u);
                        WG_STAT(4);
                    }
                    WG_TICK(12)
                    WG_COUNT(13)
                WG_BARRIER
                if (s.stop) return kUnitStopped;
                WG_EACH_WAVE
                    (void)w;
                    R.tail = s.run_tail[par];
                WG_BARRIER_NONE
                par ^= 1u;
                j0 += nb;
                continue;
            }
        }
        const uint32_t nwin = nchunks - j0 < kWgWaves ? nchunks - j0 : kWgWaves;
        WG_STAT(0);
        WG_EACH_WAVE
            WG_TICK(0)
            if (store_pending) store_stage_words(s, a, R, w);
            WG_TICK(9)
            if (check_stop && w == 0 && quota_already_spent(a)) { FOR_LANES { if (lane == 0) s.stop = 1u; } }
            phase_a(s, a, R, w, j0 + w);
            WG_TICK(1)
        WG_BARRIER
        store_pending = false;
        if (s.stop) return kUnitStopped;
        if (WG_UNIFORM(too_big)) return kUnitTooBig;
        WG_EACH_WAVE
            WG_TICK(0)
            phase_b(s, R, w, par);
            // the adaptive counts after the window: where phase B of the next window starts (the other copy: waves are
            // still reading this window's)
            if (w + 1u == nwin) { FOR_LANES { if (lane < 17) { s.ctot[par ^ 1u][lane] = LV(R.ctot); s.czer[par ^ 1u][lane] = LV(R.czer); } } }
            WG_TICK(2)
            phase_c(s, R, w);
            WG_TICK(3)
        WG_BARRIER
        for (uint32_t wbase = 0; wbase < nwin;) {
            WG_EACH_WAVE
                WG_TICK(0)
                // (the bins' open slots as of chunk wbase are taken into registers here: the commit two regions further on
                // overwrites them while other waves may still be at the forced-flush test)
                FOR_LANES { LV(R.slot0) = lane < 17 ? (uint32_t)s.bin_slot[lane] : ~0u; }
                if (w >= wbase) phase_de(s, R, w, wbase);
                WG_TICK(4)
            WG_BARRIER
            // forced-flush test: quick (uniform, every wave computes it) -- only if it can not exclude a flush do the
            // waves publish their first end events and meet once more
            bool detailed = false;
            WG_EACH_WAVE
                WG_TICK(0)
                uint32_t total;
                const bool fp = flush_possible(s, R, wbase, nwin, &total);
                R.nflush = total;
                R.wL = nwin;
                detailed = fp;
                if (fp && w >= wbase && w < nwin) first_ends(s, R, w);
                WG_TICK(5)
                if (!fp) phase_f_commit(s, R, w, wbase, nwin, false);
                WG_TICK(6)
            WG_BARRIER
            if (detailed) {
                WG_EACH_WAVE
                    WG_TICK(0)
                    phase_f_commit(s, R, w, wbase, nwin, true);
                    WG_TICK(6)
                WG_BARRIER
            }
            const uint32_t wL = WG_UNIFORM(wL);
            // ---- drain: everything before the oldest open word is finished ------------------------------------
            // (two ring words per lane: 128 * kWgWaves words per round, so that one round is almost always enough)
            for (;;) {
                WG_EACH_WAVE
                    WG_TICK(0)
                    if (store_pending) store_stage_words(s, a, R, w);          // (a second round: the stage must be free again)
                    LANEVAR(uint32_t, bs);
                    FOR_LANES { LV(bs) = (lane >= 1 && lane < 17) ? (uint32_t)s.bin_slot[lane] : ~0u; }
                    uint32_t head = R.tail;
                    for (uint32_t b = 1; b < 17; b++) { const uint32_t v = READLANE(bs, b); if (v != ~0u && (int32_t)(v - head) < 0) head = v; }
                    uint32_t n = head - R.popped;
                    if (n > 128u * kWgWaves) n = 128u * kWgWaves;
                    R.nflush = n;
                    // this wave's 128 ring words: lengths, their sum
                    LANEVAR(uint32_t, wa); LANEVAR(uint32_t, wb);
                    FOR_LANES
                    {
                        const uint32_t ia = w * 128u + (uint32_t)lane, ib = ia + 64u;
                        LV(wa) = ia < n ? WRING_LD(R.popped + ia) : 0u;
                        LV(wb) = ib < n ? WRING_LD(R.popped + ib) : 0u;
                        WG_ASSERT(ia >= n || (LV(wa) & kWordDone));
                        WG_ASSERT(ib >= n || (LV(wb) & kWordDone));
                    }
                    const uint64_t L0 = BALLOT(LV(wa) & (1u << 11)), L1 = BALLOT(LV(wa) & (2u << 11)), L2 = BALLOT(LV(wa) & (4u << 11)), L3 = BALLOT(LV(wa) & (8u << 11));
                    const uint64_t M0 = BALLOT(LV(wb) & (1u << 11)), M1 = BALLOT(LV(wb) & (2u << 11)), M2 = BALLOT(LV(wb) & (4u << 11)), M3 = BALLOT(LV(wb) & (8u << 11));
                    const uint32_t suma = (uint32_t)(popc64(L0) + 2 * popc64(L1) + 4 * popc64(L2) + 8 * popc64(L3));
                    const uint32_t sumb = (uint32_t)(popc64(M0) + 2 * popc64(M1) + 4 * popc64(M2) + 8 * popc64(M3));
                    FOR_LANES { if (lane == 0) s.wl[w].segtot = suma + sumb; }
                    // (kept for the second half: the words, and each one's bit offset inside this wave's 128)
                    FOR_LANES
                    {
                        const uint32_t offa = (uint32_t)(mbcnt64(L0, lane) + 2 * mbcnt64(L1, lane) + 4 * mbcnt64(L2, lane) + 8 * mbcnt64(L3, lane));
                        const uint32_t offb = suma + (uint32_t)(mbcnt64(M0, lane) + 2 * mbcnt64(M1, lane) + 4 * mbcnt64(M2, lane) + 8 * mbcnt64(M3, lane));
                        LV(R.dw) = LV(wa) | (LV(wb) << 16);
                        LV(R.doff) = offa | (offb << 16);
                    }
                    // this wave's chunk of the next window is fetched now: the loads are in flight during the drain, and
                    // their registers are free while the records are worked out (phases D to F)
                    if (wL == nwin) {
                        const uint32_t jn = j0 + kWgWaves + w;
                        if (jn < nchunks && R.pf != jn && !table_says_blank(a, jn)) fetch_window(a, R, jn);
                    }
                    WG_TICK(7)
                WG_BARRIER
                store_pending = false;
                if (WG_UNIFORM(too_big)) return kUnitTooBig;
                const uint32_t n = WG_UNIFORM(nflush);
                WG_EACH_WAVE
                    WG_TICK(0)
                    uint32_t before = 0, total = 0;
                    for (uint32_t v = 0; v < kWgWaves; v++) { const uint32_t tsum = s.wl[v].segtot; if (v < w) before += tsum; total += tsum; }
                    FOR_LANES
                    {
                        const uint32_t ia = w * 128u + (uint32_t)lane, ib = ia + 64u;
                        if (ia < n) WRING_ST(R.popped + ia, 0u);               // popped: the slots are free again
                        if (ib < n) WRING_ST(R.popped + ib, 0u);
#define ICER_PACK_WORD(WD, OFF)                                                                            \
                        {                                                                                  \
                            const uint32_t wd_ = (WD), len_ = (wd_ >> 11) & 15u;                           \
                            if (len_) {                                                                    \
                                const uint32_t p_ = R.bitpos + before + (OFF), wi_ = (p_ >> 5) & (kStageWords - 1), sh_ = p_ & 31u; \
                                const uint32_t code_ = wd_ & 0x3FFu;                                       \
                                LDS_OR(s.stage[wi_], code_ << sh_);                                        \
                                if (sh_ + len_ > 32u) LDS_OR(s.stage[(wi_ + 1) & (kStageWords - 1)], code_ >> (32u - sh_)); \
                            }                                                                              \
                        }
                        ICER_PACK_WORD(LV(R.dw) & 0xFFFFu, LV(R.doff) & 0xFFFFu)
                        ICER_PACK_WORD(LV(R.dw) >> 16, LV(R.doff) >> 16)
#undef ICER_PACK_WORD
                    }
                    R.bitpos += total;
                    R.popped += n;
                    WG_TICK(8)
                WG_BARRIER
                store_pending = true;
                if (n < 128u * kWgWaves) break;
            }
            if (wL < nwin) {
                // ---- the chunk in which a word start may find the ring full: its wave alone -------------------
                WG_EACH_WAVE
                    WG_TICK(0)
                    store_stage_words(s, a, R, w);
                    WG_TICK(9)
                WG_BARRIER
                store_pending = false;
                if (WG_UNIFORM(too_big)) return kUnitTooBig;
                WG_EACH_WAVE
                    WG_TICK(0)
                    if (w == wL) {
                        FOR_LANES { if (lane == 0) { s.alloc = R.tail; s.popped = R.popped; s.bitpos = R.bitpos; s.flushed_words = R.flushed_words; } }
                        WAVE_SYNC();
                        WG_STAT(2);
                        exact_chunk(s, R.c, R.tail);
                        const uint32_t nst = (uint32_t)(popc64(R.c.S1) + popc64(R.c.S2));
                        const bool ok = flush_stage(s, a, false);
                        FOR_LANES { if (lane == 0) { s.alloc = R.tail + nst; if (!ok) s.stop = 2u; } }
                    }
                    WG_TICK(10)
                WG_BARRIER
                if (s.stop == 2u) return kUnitTooBig;
                WG_EACH_WAVE
                    (void)w;
                    R.tail = s.alloc; R.popped = s.popped; R.bitpos = s.bitpos; R.flushed_words = s.flushed_words;
                WG_BARRIER
            }
            wbase = wL + 1u;
        }
        par ^= 1u;
        j0 += kWgWaves;
    }
    // end of unit: force-complete whatever is still open (C8, icer_context_modeller.c:452-455)
    uint32_t bits = kUnitTooBig;
    WG_EACH_WAVE
        if (store_pending) store_stage_words(s, a, R, w);
        WG_GLOBAL_RELEASE();
#if defined(ICER_PHASE_TIMERS) && !defined(ICER_WAVE_EMU)
        R.tacc[11] = windows;
        if (a.timers && lane == 0) for (int i_ = 0; i_ < 14; i_++) atomicAdd((unsigned long long *)&a.timers[i_], (unsigned long long)R.tacc[i_]);
#endif
    WG_BARRIER
    if (WG_UNIFORM(too_big)) return kUnitTooBig;
    WG_EACH_WAVE
        if (w == 0) {
            FOR_LANES { if (lane == 0) { s.alloc = R.tail; s.popped = R.popped; s.bitpos = R.bitpos; s.flushed_words = R.flushed_words; } }
            WAVE_SYNC();
            wave_drain(s, s.alloc);
            while (s.alloc != s.popped) {
                FOR_LANES
                {
                    if (lane == 0) seq_complete_head(s);
                }
                WAVE_SYNC();
                wave_drain(s, s.alloc);
            }
            const bool ok = flush_stage(s, a, true);
            FOR_LANES { if (lane == 0) s.alloc = ok ? s.bitpos : kUnitTooBig; }
        }
    WG_BARRIER
    bits = s.alloc;
    return bits;
}

}  // namespace wg
}  // namespace icer
