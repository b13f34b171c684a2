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
//   phase 4              interleaved entropy coder.  The 2048-word ring, the allocation-order
//                        output rule and the forced flush of the oldest open word (E5) make this
//                        a sequential state machine; here it is executed exactly, in event order.
//   drain                finished words are packed LSB-first into an LDS bit stage and whole
//                        32-bit words are written to the unit's payload slot in HBM (coalesced).
// Written with the SPMD macros of wave.hpp (see there for the tests-only CPU build).
#pragma once
#include "icer_tables.hpp"
#include "wave.hpp"

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
    int32_t bin_slot[kNumBins]; // ring index of the bin's open word, -1 if none
    uint32_t bin_acc[kNumBins]; // Golomb: zero-run length so far; bins 1..7: partial input value
    uint32_t bin_nin[kNumBins]; // bins 1..7: input bits accumulated
    uint32_t head, used;        // ring state
    uint32_t bitpos;            // payload bits produced so far
    uint32_t flushed_words;     // payload words already written to HBM
};

struct UnitArgs {
    const uint16_t *seg;        // first coefficient of the segment (sign-magnitude words)
    uint32_t stride;            // plane row stride in elements
    uint32_t w, h;              // segment size
    int subband, lsb;
    uint32_t *out_words;        // payload slot (4-byte aligned)
    uint32_t cap_words;         // slot capacity in 32-bit words
};

// ------------------------------------------------------------------------------------------
// sequential coder steps (executed by a single lane; exact restatement of E1-E6)
// ------------------------------------------------------------------------------------------
ICER_DEV void seq_emit(CoderShared &s, uint32_t code, uint32_t n)
{
    const uint32_t bp = s.bitpos, wi = (bp >> 5) & (kStageWords - 1), sh = bp & 31;
    s.stage[wi] |= code << sh;
    if (sh + n > 32) s.stage[(wi + 1) & (kStageWords - 1)] |= code >> (32 - sh);
    s.bitpos = bp + n;
}

// icer_popbuf_while_avail, icer_encoding.c:114-139
ICER_DEV void seq_drain(CoderShared &s)
{
    uint32_t head = s.head, used = s.used;
    while (used > 0) {
        const uint32_t w = s.ring[head];
        if (!(w & kWordDone)) break;
        seq_emit(s, w & 0x3FFu, (w >> 11) & 15u);
        head = (head + 1) & (kRingWords - 1);
        used--;
    }
    s.head = head;
    s.used = used;
}

ICER_DEV uint32_t reverse_low_bits(uint32_t v, uint32_t n)   // icer.h:601-610
{
    uint32_t r = 0;
    for (uint32_t k = 0; k < n; k++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// Golomb codeword for a run of k zeros ended by a one (icer_encoding.c:73-80)
ICER_DEV uint32_t golomb_word(const CoderTables &t, int bin, uint32_t k)
{
    const uint32_t gi = t.gi[bin];
    const uint32_t code = k + (k >= gi ? gi : 0u);
    const uint32_t n = t.gl[bin] + (k >= gi ? 1u : 0u);
    return kWordDone | (n << 11) | (reverse_low_bits(code, n) & 0x3FFu);
}

// icer_flush_encode, icer_encoding.c:141-189: force-complete the oldest word, then drain
ICER_DEV void seq_flush_head(CoderShared &s)
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
    seq_drain(s);
}

// icer_encode_bit after bin selection, icer_encoding.c:59-112
ICER_DEV void seq_put(CoderShared &s, int bin, uint32_t bit)
{
    int slot = s.bin_slot[bin];
    if (slot < 0) {
        if (s.used == (uint32_t)kRingWords) seq_flush_head(s);      // E5: ring full
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
    seq_drain(s);
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
            const uint32_t t0_ = total[C], z0_ = zero[C];                                             \
            if (t0_ + n_ < kRescaleCap) {                                                             \
                FOR_LANES                                                                             \
                {                                                                                     \
                    if (PRED) {                                                                       \
                        LV(TOUT) = t0_ + (uint32_t)mbcnt64(m_, lane);                                 \
                        LV(ZOUT) = z0_ + (uint32_t)mbcnt64(zm_, lane);                                \
                    }                                                                                 \
                }                                                                                     \
                total[C] = t0_ + n_;                                                                  \
                zero[C] = z0_ + nz_;                                                                  \
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
                total[C] = kRescaleCap / 2 + (n_ - kc_ - 1);                                          \
                zero[C] = zr_ + (nz_ - zc_);                                                          \
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

ICER_DEV uint32_t code_unit_wave(CoderShared &s, const UnitArgs &a)
{
    DECL_LANE;
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)lane; i < kStageWords; i += 64) s.stage[i] = 0;
        if (lane < kNumBins) { s.bin_slot[lane] = -1; s.bin_acc[lane] = 0; s.bin_nin[lane] = 0; }
        if (lane == 0) { s.head = 0; s.used = 0; s.bitpos = 0; s.flushed_words = 0; }
    }
    WAVE_SYNC();

    uint32_t zero[kNumContexts], total[kNumContexts];         // wave-uniform (SGPR) model state
#pragma unroll
    for (int c = 0; c < kNumContexts; c++) { zero[c] = 2; total[c] = 4; }   // icer_context_modeller.c:607-613

    const uint32_t npix = a.w * a.h;
    const uint32_t lsb = (uint32_t)a.lsb;
    const bool is_hl = a.subband == kHL, is_hh = a.subband == kHH;
    bool ok = true;

    for (uint32_t base = 0; base < npix && ok; base += 64) {
        LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
        LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
        LANEVAR(uint32_t, z1); LANEVAR(uint32_t, t1); LANEVAR(uint32_t, z2); LANEVAR(uint32_t, t2);

        // ---- phase 1: context formation (C1-C6) -------------------------------------------
        FOR_LANES
        {
            const uint32_t p = base + (uint32_t)lane;
            const bool valid = p < npix;
            const uint32_t pp = valid ? p : 0u;
            const uint32_t r = pp / a.w, c = pp - r * a.w;
            const uint16_t *q = a.seg + (size_t)r * a.stride + c;
            const bool hasW = c > 0, hasE = c + 1 < a.w, hasN = r > 0, hasS = r + 1 < a.h;
            const uint32_t x = q[0];
            const uint32_t xW = hasW ? q[-1] : 0u, xE = hasE ? q[1] : 0u;
            const uint32_t xN = hasN ? *(q - a.stride) : 0u, xS = hasS ? *(q + a.stride) : 0u;
            const uint32_t xNW = (hasN && hasW) ? *(q - a.stride - 1) : 0u, xNE = (hasN && hasE) ? *(q - a.stride + 1) : 0u;
            const uint32_t xSW = (hasS && hasW) ? *(q + a.stride - 1) : 0u, xSE = (hasS && hasE) ? *(q + a.stride + 1) : 0u;

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

        // ---- phase 2: adaptive counts per event (C5) --------------------------------------
#pragma unroll
        for (int c = 0; c <= 11; c++) ICER_CTX_STEP(c, LV(valid1) && LV(ctx1) == (uint32_t)c, LV(bit1) == 0u, z1, t1)
#pragma unroll
        for (int c = 12; c <= 16; c++) ICER_CTX_STEP(c, LV(valid2) && LV(ctx2) == (uint32_t)c, LV(bit2) == 0u, z2, t2)

        // ---- phase 3: fold + bin (E1) -----------------------------------------------------
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
            s.ev[2 * lane] = (uint8_t)e1;
            s.ev[2 * lane + 1] = (uint8_t)e2;
        }
        WAVE_SYNC();

        // ---- phase 4: interleaved entropy coder, exact event order -------------------------
        FOR_LANES
        {
            if (lane == 0) {
                for (int e = 0; e < 128; e++) {
                    const uint32_t v = s.ev[e];
                    if (v & 0x80u) seq_put(s, (int)(v & 31u), (v >> 5) & 1u);
                }
            }
        }
        WAVE_SYNC();
        ok = flush_stage(s, a, false);
    }

    if (!ok) return kUnitTooBig;
    // end of unit: force-complete whatever is still open (C8, icer_context_modeller.c:452-455)
    FOR_LANES
    {
        if (lane == 0)
            while (s.used > 0) seq_flush_head(s);
    }
    WAVE_SYNC();
    ok = flush_stage(s, a, true);
    return ok ? s.bitpos : kUnitTooBig;
}

}  // namespace icer
