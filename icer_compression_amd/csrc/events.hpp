// events.hpp -- the stateless half of the context modeller, ONCE PER FAMILY instead of once per bit plane.
//
// What icer_compress_bitplane_uint16 (lib_icer/src/icer_context_modeller.c:312-457) derives from a pixel's 3x3 window at
// bit plane p -- category (:340-352), magnitude bit, the 8-neighbour context (:361-396, tables icer_config.c:26-67), the
// sign context and prediction (:417-440) -- is a pure function of the nine coefficient words and p.  The nine coding units of
// a family (channel, level, subband, segment) read the same words, and everything but the final compares depends on the
// words' BIT LENGTHS alone ("is the neighbour significant at plane p" = "its bit length exceeds p"; the not-yet-visited
// neighbours E, S, SW, SE are judged one plane up, :631-642).  So one pass over the family -- nine gathers per pixel, nine
// bit lengths -- leaves one EVENT BYTE per pixel and bit plane, stored in the coding order of the plane's unit (64-pixel
// chunks, raster order inside the segment), and the pipeline coder's pixel wave reads 64 consecutive bytes per chunk instead
// of gathering and classifying nine words again for every plane.  The same pass leaves the chunk table (the bit plane from
// which a chunk is blank: the maximum of the adjusted bit lengths) that chunk_sig_kernel used to make on its own.
//
// event byte:  bits 0..3  context of the magnitude-bit event: 0..8 category 0 (icer_config.c:26-67), 9 / 10 category 1,
//                         11 category 2, 15 category 3 (uncoded, C2), 14 = no pixel (beyond the end of the segment)
//              bit 4      the magnitude bit
//              bits 5..6  sign context - 12 (meaningful when the pixel becomes significant here: context <= 8 and bit 4 set)
//              bit 7      the sign event's bit: prediction XOR sign (agreement bit)
// A chunk that is blank at plane p -- 64 x (context 0, bit 0) -- is not stored: the chunk table says so.
//
// Written with the SPMD macros of wave.hpp; the tests-only CPU build (tests/emu) runs the same source.
#pragma once
#include "icer_tables.hpp"
#include "wave.hpp"

namespace icer {

constexpr uint32_t kEvNone = 14u, kEvUncoded = 15u;

// context of a not-yet-significant pixel by neighbour counts (icer_config.c:26-67) as arithmetic
ICER_HD uint32_t ev_ctx_plain(uint32_t h, uint32_t v, uint32_t d)      // LL / LH / (swapped) HL
{
    if (h == 2) return 8;
    if (h == 1) return (v == 0) ? (d < 2 ? 5 + d : 7u) : 7u;
    if (v == 0) return d > 2 ? 2u : d;
    return 2 + v;                                                      // v = 1 -> 3, v = 2 -> 4
}
ICER_HD uint32_t ev_ctx_hh(uint32_t hv, uint32_t d)
{
    if (d >= 3) return 8;
    const uint32_t k = hv > 2 ? 2u : hv;
    if (d == 0) return k;
    if (d == 1) return 3 + k;
    return hv == 0 ? 6u : 7u;
}
// the subband's table: HH indexed (h + v) * 5 + d, the others (h * 3 + v) * 5 + d with h, v <= 2, d <= 4 (45 entries)
ICER_HD uint32_t ev_ctx_entry(bool is_hh, uint32_t i)
{
    return is_hh ? ev_ctx_hh(i / 5u, i % 5u) : ev_ctx_plain(i / 15u, (i / 5u) % 3u, i % 5u);
}

// what a pixel's window contributes to every plane: bit lengths (adjusted: the neighbours judged one plane up lose one),
// the centre's magnitude and the signs -- one pixel
struct PixelLens {
    uint32_t mag;               // |C|
    uint32_t msb;               // max(bit length of |C|, 1) - 1
    uint32_t aW, aE, aN, aS, aNW, aNE, aSW, aSE;    // neighbour k is significant at plane p  <=>  a_k > p
    uint32_t neg;               // bit 0 W, 1 E, 2 N, 3 S negative; bit 4: C negative
    uint32_t t;                 // the pixel is blank from this plane on
};

ICER_HD uint32_t ev_len(uint32_t v) { const uint32_t m = v & 0x7FFFu; return m ? 32u - (uint32_t)__builtin_clz(m) : 0u; }

// words of neighbours that do not exist must be passed as 0
ICER_HD PixelLens pixel_lens(uint32_t vC, uint32_t vW, uint32_t vE, uint32_t vN, uint32_t vS, uint32_t vNW, uint32_t vNE, uint32_t vSW, uint32_t vSE)
{
    PixelLens L;
    const uint32_t lc = ev_len(vC);
    L.mag = vC & 0x7FFFu;
    L.msb = lc ? lc - 1u : 0u;
    L.aW = ev_len(vW); L.aN = ev_len(vN); L.aNW = ev_len(vNW); L.aNE = ev_len(vNE);
    const uint32_t lE = ev_len(vE), lS = ev_len(vS), lSW = ev_len(vSW), lSE = ev_len(vSE);
    L.aE = lE ? lE - 1u : 0u; L.aS = lS ? lS - 1u : 0u; L.aSW = lSW ? lSW - 1u : 0u; L.aSE = lSE ? lSE - 1u : 0u;
    L.neg = ((vW >> 15) & 1u) | (((vE >> 15) & 1u) << 1) | (((vN >> 15) & 1u) << 2) | (((vS >> 15) & 1u) << 3) | (((vC >> 15) & 1u) << 4);
    uint32_t t = lc;
    t = L.aW > t ? L.aW : t; t = L.aN > t ? L.aN : t; t = L.aNW > t ? L.aNW : t; t = L.aNE > t ? L.aNE : t;
    t = L.aE > t ? L.aE : t; t = L.aS > t ? L.aS : t; t = L.aSW > t ? L.aSW : t; t = L.aSE > t ? L.aSE : t;
    L.t = t;
    return L;
}

// the event byte of one pixel at bit plane p (C1-C6 of SURVEY 2.3; icer_context_modeller.c:340-440)
ICER_HD uint32_t event_byte(const PixelLens &L, uint32_t p, bool is_hl, bool is_hh, const uint8_t *ctx_tab)
{
    const uint32_t sW = L.aW > p, sE = L.aE > p, sN = L.aN > p, sS = L.aS > p;
    uint32_t hh = sW + sE, vv = sN + sS;
    const uint32_t dd = (uint32_t)(L.aNW > p) + (uint32_t)(L.aNE > p) + (uint32_t)(L.aSW > p) + (uint32_t)(L.aSE > p);
    const uint32_t up = L.msb > p ? L.msb - p : 0u, cat = up > 3u ? 3u : up;
    const uint32_t bit = (L.mag >> p) & 1u;
    // sign (C6): only negative significant neighbours count
    uint32_t A = (L.neg & sW) | ((L.neg >> 1) & sE), B = ((L.neg >> 2) & sN) | ((L.neg >> 3) & sS);
    if (is_hl) { uint32_t x = hh; hh = vv; vv = x; x = A; A = B; B = x; }
    const uint32_t c0 = ctx_tab[is_hh ? (hh + vv) * 5u + dd : (hh * 3u + vv) * 5u + dd];
    const uint32_t ctx = cat == 0u ? c0 : cat == 1u ? (hh + vv == 0u ? 9u : 10u) : cat == 2u ? 11u : kEvUncoded;
    // icer_sign_context_table / icer_sign_prediction_table restricted to "no / some negative significant neighbour":
    // context - 12 = A ? (B ? 2 : 3) : (B ? 1 : 0), prediction = A
    const uint32_t s2 = ((A & 1u) << 1) | ((A ^ B) & 1u);
    const uint32_t bit2 = (A ^ (L.neg >> 4)) & 1u;
    return ctx | (bit << 4) | (s2 << 5) | (bit2 << 7);
}

// Where a family's events live: plane p of the family's chunk j starts at  ev + ((size_t)p * chunks_per_frame + chunk_off + j) * 64
// (chunks_per_frame = Plan::sig_bytes, chunk_off = UnitDesc::sig_off: the chunk table's own indexing)
ICER_HD size_t ev_offset(uint32_t p, size_t chunks_per_frame, uint32_t chunk_off, uint32_t j) { return ((size_t)p * chunks_per_frame + chunk_off + j) * 64u; }

}  // namespace icer
