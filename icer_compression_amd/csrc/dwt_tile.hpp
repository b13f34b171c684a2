// dwt_tile.hpp -- one stage of the 2-D lifting transform as a fused, LDS-staged tile pass.
//
// Replaces icer_wavelet_transform_2d_uint16 (lib_icer/src/icer_wavelet.c:155-171): the reference runs the
// 1-D lifting over every row of the current LL region and then over every column (stride W, cache hostile).
// Here a workgroup owns a tile of kTileKX x kTileKY output *pairs*:
//   load    the (2*KY+8) x (2*KX+8) input window (tile + the 4/3-sample halo the filters reach) into LDS with
//           coalesced row reads;
//   rows    every window row is lifted: low[k], high[k] for the tile's KX pair columns (dwt_pair on the LDS row);
//   columns the KX low columns and KX high columns are lifted along the window rows for the tile's KY pair rows,
//           giving LL/LH (from the low columns) and HL/HH (from the high columns);
//   store   HL, LH, HH go straight to their final place in the coefficient plane -- as the sign-magnitude words the
//           coder reads (icer_to_sign_magnitude_int16 fused into the store) --, LL to the buffer the next stage
//           reads (so no stage reads what another workgroup of the same stage writes).
// HBM traffic per stage: the region once in (+ halo), once out; the row-pass intermediate never leaves LDS.
// The phase bodies are plain per-thread functions so the tests-only CPU build (tests/emu) can run them in a loop.
#pragma once
#include "dwt_core.hpp"
#include "icer_tables.hpp"

namespace icer {

constexpr int kTileKX = 64, kTileKY = 16;                    // output pairs per tile
constexpr int kWinW = 2 * kTileKX + 8, kWinH = 2 * kTileKY + 8;
constexpr int kTileThreads = 256;

struct DwtTileShared {
    int16_t win[kWinH][kWinW];          // input window
    int16_t lo[kWinH][kTileKX];         // row pass: lows / highs of the tile's pair columns, per window row
    int16_t hi[kWinH][kTileKX];
};

struct DwtStageArgs {
    const int16_t *src;                 // current LL region (stage 0: the frame), plane base
    uint32_t src_stride;
    int cw, ch;                         // region size
    int16_t *coef;                      // coefficient plane (stride W): HL, LH, HH of this stage go here
    uint32_t coef_stride;
    int16_t *ll;                        // where this stage's LL goes (next stage's source, or the plane itself)
    uint32_t ll_stride;
    FilterTaps f;
    int32_t lim;                        // largest storable sample: 32767, or 127 for the uint8 twins
    int sm;                             // detail bands are stored as sign-magnitude words (what the coder reads): 16 = int16
                                        // (icer_to_sign_magnitude_int16, icer_wavelet.c:871-877), 8 = the int8 word of the uint8
                                        // twins widened to s|0..0|mmmmmmm (icer_wavelet.c:852-858), 0 = plain two's complement
};

// two's complement -> the coder's sign-magnitude word (fused into the store of HL / LH / HH: these bands are final
// when a stage writes them; LL still loses its mean first, finalize_ll_kernel)
DWT_HD int16_t to_coder_word(int16_t v, int sm)
{
    if (sm == 16) {
        const uint16_t mask = (uint16_t)(v >> 15);
        return (int16_t)((((uint16_t)v + mask) ^ mask) | ((uint16_t)v & 0x8000u));
    }
    if (sm == 8) {
        const int8_t v8 = (int8_t)v;
        const uint8_t m8 = (uint8_t)(v8 >> 7);
        const uint8_t w8 = (uint8_t)((((uint8_t)v8 + m8) ^ m8) | ((uint8_t)v8 & 0x80u));
        return (int16_t)(uint16_t)(((w8 & 0x80u) << 8) | (w8 & 0x7Fu));
    }
    return v;
}

// window origin of the tile (tx, ty) in region coordinates (may be negative: clamped on load)
DWT_HD int tile_x0(int tx) { return 2 * tx * kTileKX - 4; }
DWT_HD int tile_y0(int ty) { return 2 * ty * kTileKY - 4; }

// phase 1, thread t: load window elements t, t + T, ... (out-of-region elements are never read later)
DWT_HD void dwt_tile_load(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int x0 = tile_x0(tx), y0 = tile_y0(ty);
    for (int i = t; i < kWinH * kWinW; i += kTileThreads) {
        const int r = i / kWinW, c = i - r * kWinW;
        const int gx = x0 + c, gy = y0 + r;
        sh.win[r][c] = (gx >= 0 && gx < a.cw && gy >= 0 && gy < a.ch) ? a.src[(size_t)gy * a.src_stride + gx] : (int16_t)0;
    }
}

// phase 2, thread t: row lifting of window rows; returns true on int16 overflow
DWT_HD bool dwt_tile_rows(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int x0 = tile_x0(tx), y0 = tile_y0(ty);
    const int nl = (a.cw + 1) >> 1;
    bool ovf = false;
    for (int i = t; i < kWinH * kTileKX; i += kTileThreads) {
        const int r = i / kTileKX, kk = i - r * kTileKX;
        const int gy = y0 + r, k = tx * kTileKX + kk;
        if (gy < 0 || gy >= a.ch || k >= nl) continue;
        const int16_t *row = sh.win[r];
        const DwtPair p = dwt_pair([row, x0](int x) { return row[x - x0]; }, a.cw, k, a.f.am1, a.f.a0, a.f.a1, a.f.be, a.lim);
        sh.lo[r][kk] = p.low;
        sh.hi[r][kk] = p.has_high ? p.high : (int16_t)0;
        ovf |= p.overflow;
    }
    return ovf;
}

// phase 3, thread t: column lifting + stores; returns true on int16 overflow
DWT_HD bool dwt_tile_cols(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int y0 = tile_y0(ty);
    const int nlw = (a.cw + 1) >> 1, nhw = a.cw >> 1, nlh = (a.ch + 1) >> 1;
    bool ovf = false;
    for (int i = t; i < 2 * kTileKX * kTileKY; i += kTileThreads) {
        // consecutive threads -> consecutive pair columns (coalesced stores); low columns first, then high columns
        const int kk = i % kTileKX, which = (i / kTileKX) & 1, jj = i / (2 * kTileKX);
        const int kx = tx * kTileKX + kk, ky = ty * kTileKY + jj;
        if (ky >= nlh || kx >= (which ? nhw : nlw)) continue;
        const int16_t(*col)[kTileKX] = which ? sh.hi : sh.lo;
        const DwtPair p = dwt_pair([col, kk, y0](int y) { return col[y - y0][kk]; }, a.ch, ky, a.f.am1, a.f.a0, a.f.a1, a.f.be, a.lim);
        ovf |= p.overflow;
        if (!which) {                                           // low column: LL (top) and LH (below)
            a.ll[(size_t)ky * a.ll_stride + kx] = p.low;
            if (p.has_high) a.coef[(size_t)(nlh + ky) * a.coef_stride + kx] = to_coder_word(p.high, a.sm);
        } else {                                                // high column: HL (right) and HH (diagonal)
            a.coef[(size_t)ky * a.coef_stride + nlw + kx] = to_coder_word(p.low, a.sm);
            if (p.has_high) a.coef[(size_t)(nlh + ky) * a.coef_stride + nlw + kx] = to_coder_word(p.high, a.sm);
        }
    }
    return ovf;
}

}  // namespace icer
