// dwt_tile.hpp -- one stage of the 2-D lifting transform as a fused, LDS-staged tile pass.
//
// Replaces icer_wavelet_transform_2d_uint16 (lib_icer/src/icer_wavelet.c:155-171): the reference runs the
// 1-D lifting over every row of the current LL region and then over every column (stride W, cache hostile).
// Here a workgroup owns a tile of kTileKX x kTileKY output *pairs*:
//   load    the (2*KY+8) x (2*KX+8) input window (tile + the 4/3-sample halo the filters reach) into LDS with
//           coalesced row reads, two samples (one 32-bit word) per lane where the rows are word-aligned;
//   rows    every window row is lifted in two steps: first the pair averages (lows) and pair differences of the whole
//           row -- each computed ONCE and left in LDS --, then the highs of the tile's KX pair columns from them
//           (a high needs the four lows around it and two differences, with the lows shared
//           instead of recomputed by each of the four pairs that need them);
//   columns the same two steps down the KX low columns and KX high columns for the tile's KY pair rows,
//           giving LL/LH (from the low columns) and HL/HH (from the high columns);
//   store   HL, LH, HH go straight to their final place in the coefficient plane -- as the sign-magnitude words the
//           coder reads (icer_to_sign_magnitude_int16 fused into the store) --, LL to the buffer the next stage
//           reads (so no stage reads what another workgroup of the same stage writes).
// HBM traffic per stage: the region once in (+ halo), once out; the row-pass intermediate never leaves LDS.
// The phase bodies are plain per-thread functions so the tests-only CPU build (tests/emu) can run them in a loop;
// tests/test_emu_pipeline.py::test_dwt_core holds them to the oracle for every filter, odd sizes and int16 overflow.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DWT_HD __host__ __device__ __forceinline__
#else
#define DWT_HD static inline
#endif
#include "icer_tables.hpp"

#if defined(__HIPCC__)
#define DWT_UNROLL _Pragma("unroll")
#else
#define DWT_UNROLL
#endif

namespace icer {

constexpr int kTileKX = 64, kTileKY = 16;                    // output pairs per tile
constexpr int kWinW = 2 * kTileKX + 8, kWinH = 2 * kTileKY + 8;
constexpr int kWinPX = kWinW / 2, kWinPY = kWinH / 2;        // pairs per window row / column (the window starts 2 pairs before the tile)
constexpr int kTileThreads = 256;

// (two blocks of LDS, each used twice: 21.8 KB per workgroup, seven workgroups per compute unit; round 4: 32 KB, five)
struct DwtTileShared {
    union {
        int16_t win[kWinH][kWinW];          // input window (dead once the row lows / differences exist) ...
        struct {                            // ... then the row pass result: lows / highs of the tile's pair columns, per window row
            int16_t lo[kWinH][kTileKX];
            int16_t hi[kWinH][kTileKX];
        };
    };
    union {
        struct { int16_t rlo[kWinH][kWinPX], rdi[kWinH][kWinPX]; };   // row step 1: lows and differences of every pair of every window row (dead after row step 2) ...
        struct { int16_t lo[2][kWinPY][kTileKX], di[2][kWinPY][kTileKX]; } col;   // ... then column step 1: [low / high column set]
    };
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

// phase 1, thread t: load window sample pairs t, t + T, ... (out-of-region elements read as 0 and are never used)
DWT_HD void dwt_tile_load(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int x0 = tile_x0(tx), y0 = tile_y0(ty);
    // a pair of samples is one aligned 32-bit word when the rows start word-aligned (x0 is even)
    const bool words = ((a.src_stride & 1u) == 0u) && ((reinterpret_cast<uintptr_t>(a.src) & 3u) == 0u);
    for (int i = t; i < kWinH * kWinPX; i += kTileThreads) {
        const int r = i / kWinPX, p = i - r * kWinPX;
        const int gx = x0 + 2 * p, gy = y0 + r;
        int16_t v0 = 0, v1 = 0;
        if (gy >= 0 && gy < a.ch && gx + 1 >= 0 && gx < a.cw) {
            const int16_t *q = a.src + (size_t)gy * a.src_stride + gx;
            if (words && gx >= 0 && gx + 1 < a.cw) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(q);
                v0 = (int16_t)(w & 0xFFFFu); v1 = (int16_t)(w >> 16);
            } else {
                if (gx >= 0) v0 = q[0];
                if (gx + 1 < a.cw) v1 = q[1];
            }
        }
        sh.win[r][2 * p] = v0; sh.win[r][2 * p + 1] = v1;
    }
}

// One line of `n` samples, step 1 for pair index k (0 <= k < ceil(n/2)): its low (pair average, or the odd tail sample)
// and, for k < floor(n/2), its difference; exactly the values the reference forms for that pair (icer_wavelet.c:410-428).  Returns the overflow flag.
DWT_HD bool dwt_pair_step1(int32_t a, int32_t b, int n, int k, int32_t lim, int16_t *low, int16_t *dif)
{
    const int nh = n >> 1;
    bool ovf = false;
    if (k < nh) {
        const int32_t v = (a + b) >> 1, d = a - b;
        ovf = v < -lim - 1 || v > lim || d < -lim - 1 || d > lim;
        *low = (int16_t)v; *dif = (int16_t)d;
    } else { *low = (int16_t)a; *dif = 0; }                 // (odd n, k == nh: the tail sample is a low)
    return ovf;
}

// Step 2: the high of pair k from the line's lows and differences.  Output pair k of a line of n samples needs
// x[2k-4 .. 2k+3] only, so the reference's in-place pair lift -> un-shuffle (icer_deinterleave_uint16, icer_wavelet.c:765-820)
// -> predict collapses to this local rule.  Bit-exactness: lows / highs are stored truncated to int16 with a sticky
// overflow flag on the untruncated value; r[j] = (int16)(low[j-1] - low[j]) wraps silently (get_r_int16 :206-208); boundary
// rules k == 0, k == nh-1 (even n), missing d[k+1] (odd n) as :430-462; filter C (alpha_-1 != 0) at k == 1 uses d[1] where
// the ICER paper has d[2] (reference quirk W3, offset = low_N at :437-440), and 0 when n == 5.
// LO(j) / DI(j): low / difference of pair j, called with 0 <= j < ceil(n/2) only.
template <class Lo, class Di>
DWT_HD bool dwt_pair_step2(const Lo &LO, const Di &DI, int n, int k, int am1, int a0, int a1, int be, int32_t lim, int16_t *high)
{
    const int nl = (n + 1) >> 1, nh = n >> 1;
    const bool odd = (n & 1) != 0;
    const auto clampi = [nl](int j) { return j < 0 ? 0 : (j > nl - 1 ? nl - 1 : j); };
    const int32_t l0 = LO(clampi(k - 2)), l1 = LO(clampi(k - 1)), l2 = LO(k), l3 = LO(clampi(k + 1));
    const int32_t r_km1 = (int16_t)(l0 - l1), r_k = (int16_t)(l1 - l2), r_kp1 = (int16_t)(l2 - l3);
    const int32_t dk = DI(k);
    const int32_t dn = (k + 1 < nh) ? (int32_t)DI(k + 1) : 0;
    int32_t sub;
    if (k == 0) sub = r_kp1 >> 2;
    else if (k == 1 && am1 != 0) {
        const int32_t x = (odd && nl == 3) ? 0 : dk;            // QUIRK W3
        sub = (2 * r_k + 3 * r_kp1 - 2 * x + 4) >> 3;
    } else if (!odd && k == nh - 1) sub = r_k >> 2;
    else {
        const int32_t rm = (k >= 2) ? r_km1 : 1;                // r[0] reads as 1 (times alpha_-1 == 0)
        sub = (am1 * rm + a0 * r_k + a1 * r_kp1 - be * dn + 8) >> 4;
    }
    const int32_t h = dk - sub;
    *high = (int16_t)h;
    return h < -lim - 1 || h > lim;
}

// phase 2a, thread t: lows and differences of every pair of every window row; returns true on int16 overflow
DWT_HD bool dwt_tile_rows_step1(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int y0 = tile_y0(ty);
    const int nl = (a.cw + 1) >> 1;
    bool ovf = false;
    for (int i = t; i < kWinH * kWinPX; i += kTileThreads) {
        const int r = i / kWinPX, p = i - r * kWinPX;
        const int gy = y0 + r, k = tx * kTileKX - 2 + p;
        if (gy < 0 || gy >= a.ch || k < 0 || k >= nl) continue;
        ovf |= dwt_pair_step1(sh.win[r][2 * p], sh.win[r][2 * p + 1], a.cw, k, a.lim, &sh.rlo[r][p], &sh.rdi[r][p]);
    }
    return ovf;
}

// phase 2b, thread t: row highs of the tile's pair columns (and a copy of their lows)
DWT_HD bool dwt_tile_rows_step2(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int y0 = tile_y0(ty);
    const int nl = (a.cw + 1) >> 1, nh = a.cw >> 1;
    const int kbase = tx * kTileKX - 2;
    bool ovf = false;
    for (int i = t; i < kWinH * kTileKX; i += kTileThreads) {
        const int r = i / kTileKX, kk = i - r * kTileKX;
        const int gy = y0 + r, k = tx * kTileKX + kk;
        if (gy < 0 || gy >= a.ch || k >= nl) continue;
        const int16_t *rl = sh.rlo[r], *rd = sh.rdi[r];
        sh.lo[r][kk] = rl[k - kbase];
        int16_t h = 0;
        if (k < nh)
            ovf |= dwt_pair_step2([rl, kbase](int j) { return (int32_t)rl[j - kbase]; }, [rd, kbase](int j) { return (int32_t)rd[j - kbase]; },
                                  a.cw, k, a.f.am1, a.f.a0, a.f.a1, a.f.be, a.lim, &h);
        sh.hi[r][kk] = h;
    }
    return ovf;
}

// phase 3a, thread t: lows and differences down the tile's low columns (which = 0) and high columns (1) -- into the
// LDS block the input window lived in
DWT_HD bool dwt_tile_cols_step1(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int nlw = (a.cw + 1) >> 1, nhw = a.cw >> 1, nlh = (a.ch + 1) >> 1;
    bool ovf = false;
    for (int i = t; i < 2 * kWinPY * kTileKX; i += kTileThreads) {
        const int kk = i % kTileKX, q = (i / kTileKX) % kWinPY, which = i / (kTileKX * kWinPY);
        const int kx = tx * kTileKX + kk, ky = ty * kTileKY - 2 + q;
        if (ky < 0 || ky >= nlh || kx >= (which ? nhw : nlw)) continue;
        const int16_t(*col)[kTileKX] = which ? sh.hi : sh.lo;
        ovf |= dwt_pair_step1(col[2 * q][kk], col[2 * q + 1][kk], a.ch, ky, a.lim, &sh.col.lo[which][q][kk], &sh.col.di[which][q][kk]);
    }
    return ovf;
}

// phase 3b, thread t: column highs + stores; returns true on int16 overflow
DWT_HD bool dwt_tile_cols_step2(DwtTileShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int nlw = (a.cw + 1) >> 1, nhw = a.cw >> 1, nlh = (a.ch + 1) >> 1, nhh = a.ch >> 1;
    const int qbase = ty * kTileKY - 2;
    bool ovf = false;
    for (int i = t; i < 2 * kTileKX * kTileKY; i += kTileThreads) {
        // consecutive threads -> consecutive pair columns (coalesced stores); low columns first, then high columns
        const int kk = i % kTileKX, which = (i / kTileKX) & 1, jj = i / (2 * kTileKX);
        const int kx = tx * kTileKX + kk, ky = ty * kTileKY + jj;
        if (ky >= nlh || kx >= (which ? nhw : nlw)) continue;
        const int16_t(*cl)[kTileKX] = sh.col.lo[which];
        const int16_t(*cd)[kTileKX] = sh.col.di[which];
        const int16_t low = cl[ky - qbase][kk];
        int16_t high = 0;
        const bool has_high = ky < nhh;
        if (has_high)
            ovf |= dwt_pair_step2([cl, kk, qbase](int j) { return (int32_t)cl[j - qbase][kk]; }, [cd, kk, qbase](int j) { return (int32_t)cd[j - qbase][kk]; },
                                  a.ch, ky, a.f.am1, a.f.a0, a.f.a1, a.f.be, a.lim, &high);
        if (!which) {                                           // low column: LL (top) and LH (below)
            a.ll[(size_t)ky * a.ll_stride + kx] = low;
            if (has_high) a.coef[(size_t)(nlh + ky) * a.coef_stride + kx] = to_coder_word(high, a.sm);
        } else {                                                // high column: HL (right) and HH (diagonal)
            a.coef[(size_t)ky * a.coef_stride + nlw + kx] = to_coder_word(low, a.sm);
            if (has_high) a.coef[(size_t)(nlh + ky) * a.coef_stride + nlw + kx] = to_coder_word(high, a.sm);
        }
    }
    return ovf;
}

// ------------------------------------------------------------------------------------------ interior tiles
// A tile whose whole input window lies inside the region and none of whose pairs sits at a line end -- nine tenths of the
// tiles of a large stage -- needs none of the generic phases' bounds tests, clamps and boundary rules, and its phases map
// threads to (row, pair of pairs) by shifts: 256 threads = 8 rows x 32 x 2 pairs per step.  Lows and differences travel as ONE 32-bit
// LDS word per pair (low | difference << 16), so row step 2 reads four words where the generic phase reads six halves, and
// row step 1 is done on the loaded words themselves: the input window never goes to LDS.  Same arithmetic, same results
// (tests/test_emu_pipeline.py runs both paths against the oracle).
struct alignas(8) DwtW2 { uint32_t x, y; };           // two adjacent pairs' words: one 64-bit LDS / global access
struct DwtFastShared {
    union {
        alignas(8) uint32_t rw[kWinH][kWinPX];        // row step 1: low | dif << 16 of every pair of every window row (dead after row step 2) ...
        struct {                                      // ... then column step 1, low | dif << 16:
            alignas(8) uint32_t cl[kWinPY][kTileKX];  //     down the low columns
            alignas(8) uint32_t chh[kWinPY][kTileKX]; //     down the high columns
        };
    };
    alignas(8) uint32_t rr[kWinH][kTileKX];           // row pass: low | high << 16 of the tile's pair columns, per window row
};
static_assert(sizeof(DwtFastShared) <= sizeof(DwtTileShared), "the kernel's LDS block is the generic path's");

// Round 5: a thread owns TWO adjacent pairs in every phase (64-bit loads from HBM and LDS, 32-bit stores of two coefficients:
// half the memory instructions per byte), the loads of a phase are issued together before anything waits for one of them
// (the loops are unrolled by hand: left as loops the compiler kept one load in flight per thread, ten round trips to HBM in a
// row per tile), and the column pass keeps a thread on two adjacent output rows (five LDS rows read for two, not eight).
DWT_HD bool dwt_tile_is_interior(const DwtStageArgs &a, int tx, int ty)
{
    const int x0 = tile_x0(tx), y0 = tile_y0(ty);
    const int nlw = (a.cw + 1) >> 1;
    // (window inside the region; then every pair k of the tile has 2 <= k, k + 1 < n / 2 - 1 + 1, i.e. only the general rule applies;
    // rows of the source 8-byte aligned for the 64-bit loads, every destination 4-byte aligned at even columns for the 32-bit stores)
    return x0 >= 0 && y0 >= 0 && x0 + kWinW <= a.cw && y0 + kWinH <= a.ch && (a.src_stride & 3u) == 0u &&
           (reinterpret_cast<uintptr_t>(a.src) & 7u) == 0u && (a.coef_stride & 1u) == 0u && (reinterpret_cast<uintptr_t>(a.coef) & 3u) == 0u &&
           (a.ll_stride & 1u) == 0u && (reinterpret_cast<uintptr_t>(a.ll) & 3u) == 0u && (nlw & 1) == 0;
}
DWT_HD uint32_t dwt_pack(int32_t lo, int32_t hi) { return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16); }
DWT_HD int32_t dwt_lo16(uint32_t w) { return (int32_t)(int16_t)(w & 0xFFFFu); }
DWT_HD int32_t dwt_hi16(uint32_t w) { return (int32_t)(int16_t)(w >> 16); }
// pair (a, b) -> low | dif << 16, overflow flag as dwt_pair_step1
DWT_HD uint32_t dwt_fast_step1(int32_t a, int32_t b, int32_t lim, bool *ovf)
{
    const int32_t v = (a + b) >> 1, d = a - b;
    *ovf |= v < -lim - 1 || v > lim || d < -lim - 1 || d > lim;
    return dwt_pack(v, d);
}
// the general rule of dwt_pair_step2 from the four words around pair k: w0..w3 = pairs k - 2 .. k + 1
DWT_HD int32_t dwt_fast_step2(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, const FilterTaps &f, int32_t lim, bool *ovf)
{
    const int32_t l0 = dwt_lo16(w0), l1 = dwt_lo16(w1), l2 = dwt_lo16(w2), l3 = dwt_lo16(w3);
    const int32_t r_km1 = (int16_t)(l0 - l1), r_k = (int16_t)(l1 - l2), r_kp1 = (int16_t)(l2 - l3);
    const int32_t sub = (f.am1 * r_km1 + f.a0 * r_k + f.a1 * r_kp1 - f.be * dwt_hi16(w3) + 8) >> 4;
    const int32_t h = dwt_hi16(w2) - sub;
    *ovf |= h < -lim - 1 || h > lim;
    return h;
}
DWT_HD DwtW2 dwt_ld2(const uint32_t *p) { return *reinterpret_cast<const DwtW2 *>(p); }
DWT_HD void dwt_st2(uint32_t *p, uint32_t x, uint32_t y) { DwtW2 v; v.x = x; v.y = y; *reinterpret_cast<DwtW2 *>(p) = v; }

constexpr int kFastRowGroups = kTileThreads / 32;                 // 8: a thread = (row group, pair of pairs)
constexpr int kFastRowIters = kWinH / kFastRowGroups;             // 5 window rows per thread in the row phases
static_assert(kWinH % kFastRowGroups == 0 && kTileKY == 2 * kFastRowGroups && kWinPX == kTileKX + 4, "thread map of the interior path");

// phase F1, thread t: load the window as 64-bit words (two pairs) and leave low | dif of every pair in LDS
DWT_HD bool dwt_fast_rows_step1(DwtFastShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int x0 = tile_x0(tx), y0 = tile_y0(ty);
    const DwtW2 *base = reinterpret_cast<const DwtW2 *>(a.src + (size_t)y0 * a.src_stride + x0);
    const size_t row_w2 = a.src_stride >> 2;
    const int pp = t & 31, rg = t >> 5;
    bool ovf = false;
    DwtW2 v[kFastRowIters];
    DWT_UNROLL
    for (int i = 0; i < kFastRowIters; i++) v[i] = base[(size_t)(rg + kFastRowGroups * i) * row_w2 + pp];       // (all in flight together)
    const bool extra = t < 2 * kWinH;                         // the four pairs beyond the 64th of every row: two lanes per row
    const int er = t >> 1, ep = 32 + (t & 1);
    DwtW2 ve; ve.x = 0; ve.y = 0;
    if (extra) ve = base[(size_t)er * row_w2 + ep];
    DWT_UNROLL
    for (int i = 0; i < kFastRowIters; i++) {
        const int r = rg + kFastRowGroups * i;
        dwt_st2(&sh.rw[r][2 * pp], dwt_fast_step1((int16_t)(v[i].x & 0xFFFFu), (int16_t)(v[i].x >> 16), a.lim, &ovf),
                dwt_fast_step1((int16_t)(v[i].y & 0xFFFFu), (int16_t)(v[i].y >> 16), a.lim, &ovf));
    }
    if (extra)
        dwt_st2(&sh.rw[er][2 * ep], dwt_fast_step1((int16_t)(ve.x & 0xFFFFu), (int16_t)(ve.x >> 16), a.lim, &ovf),
                dwt_fast_step1((int16_t)(ve.y & 0xFFFFu), (int16_t)(ve.y >> 16), a.lim, &ovf));
    return ovf;
}
// phase F2: row highs of the tile's 64 pair columns for every window row (pair kk of the tile = window pair kk + 2)
DWT_HD bool dwt_fast_rows_step2(DwtFastShared &sh, const DwtStageArgs &a, int t)
{
    const int kk = 2 * (t & 31), rg = t >> 5;
    bool ovf = false;
    DwtW2 q0[kFastRowIters], q1[kFastRowIters], q2[kFastRowIters];
    DWT_UNROLL
    for (int i = 0; i < kFastRowIters; i++) {
        const uint32_t *q = &sh.rw[rg + kFastRowGroups * i][kk];
        q0[i] = dwt_ld2(q); q1[i] = dwt_ld2(q + 2); q2[i] = dwt_ld2(q + 4);
    }
    DWT_UNROLL
    for (int i = 0; i < kFastRowIters; i++) {
        const int32_t h0 = dwt_fast_step2(q0[i].x, q0[i].y, q1[i].x, q1[i].y, a.f, a.lim, &ovf);
        const int32_t h1 = dwt_fast_step2(q0[i].y, q1[i].x, q1[i].y, q2[i].x, a.f, a.lim, &ovf);
        dwt_st2(&sh.rr[rg + kFastRowGroups * i][kk], dwt_pack(dwt_lo16(q1[i].x), h0), dwt_pack(dwt_lo16(q1[i].y), h1));
    }
    return ovf;
}
// phase F3: step 1 down the low and the high columns (window pair rows q = 0 .. 19) -- into the LDS block the row words lived in
DWT_HD bool dwt_fast_cols_step1(DwtFastShared &sh, const DwtStageArgs &a, int t)
{
    const int kk = 2 * (t & 31), rg = t >> 5;
    bool ovf = false;
    constexpr int iters = (kWinPY + kFastRowGroups - 1) / kFastRowGroups;     // 3 (8 + 8 + 4 rows)
    DwtW2 u[iters], v[iters];
    DWT_UNROLL
    for (int i = 0; i < iters; i++) {
        const int q = rg + kFastRowGroups * i;
        if (q < kWinPY) { u[i] = dwt_ld2(&sh.rr[2 * q][kk]); v[i] = dwt_ld2(&sh.rr[2 * q + 1][kk]); }
        else { u[i].x = u[i].y = v[i].x = v[i].y = 0; }
    }
    DWT_UNROLL
    for (int i = 0; i < iters; i++) {
        const int q = rg + kFastRowGroups * i;
        if (q >= kWinPY) continue;
        dwt_st2(&sh.cl[q][kk], dwt_fast_step1(dwt_lo16(u[i].x), dwt_lo16(v[i].x), a.lim, &ovf), dwt_fast_step1(dwt_lo16(u[i].y), dwt_lo16(v[i].y), a.lim, &ovf));
        dwt_st2(&sh.chh[q][kk], dwt_fast_step1(dwt_hi16(u[i].x), dwt_hi16(v[i].x), a.lim, &ovf), dwt_fast_step1(dwt_hi16(u[i].y), dwt_hi16(v[i].y), a.lim, &ovf));
    }
    return ovf;
}
// phase F4: column highs and the four stores, two adjacent coefficients per store (tile pair row jj = window pair row jj + 2);
// a thread takes the two adjacent pair rows 2 * rg, 2 * rg + 1
DWT_HD bool dwt_fast_cols_step2(DwtFastShared &sh, const DwtStageArgs &a, int tx, int ty, int t)
{
    const int nlw = (a.cw + 1) >> 1, nlh = (a.ch + 1) >> 1;
    const int kk = 2 * (t & 31), jj0 = 2 * (t >> 5), kx = tx * kTileKX + kk;
    bool ovf = false;
    DwtW2 c[5], h[5];
    DWT_UNROLL
    for (int i = 0; i < 5; i++) { c[i] = dwt_ld2(&sh.cl[jj0 + i][kk]); h[i] = dwt_ld2(&sh.chh[jj0 + i][kk]); }
    DWT_UNROLL
    for (int d = 0; d < 2; d++) {
        const int ky = ty * kTileKY + jj0 + d;
        const int32_t lh0 = dwt_fast_step2(c[d].x, c[d + 1].x, c[d + 2].x, c[d + 3].x, a.f, a.lim, &ovf);
        const int32_t lh1 = dwt_fast_step2(c[d].y, c[d + 1].y, c[d + 2].y, c[d + 3].y, a.f, a.lim, &ovf);
        const int32_t hh0 = dwt_fast_step2(h[d].x, h[d + 1].x, h[d + 2].x, h[d + 3].x, a.f, a.lim, &ovf);
        const int32_t hh1 = dwt_fast_step2(h[d].y, h[d + 1].y, h[d + 2].y, h[d + 3].y, a.f, a.lim, &ovf);
        *reinterpret_cast<uint32_t *>(a.ll + (size_t)ky * a.ll_stride + kx) = dwt_pack(dwt_lo16(c[d + 2].x), dwt_lo16(c[d + 2].y));
        *reinterpret_cast<uint32_t *>(a.coef + (size_t)(nlh + ky) * a.coef_stride + kx) =
            dwt_pack(to_coder_word((int16_t)lh0, a.sm), to_coder_word((int16_t)lh1, a.sm));
        *reinterpret_cast<uint32_t *>(a.coef + (size_t)ky * a.coef_stride + nlw + kx) =
            dwt_pack(to_coder_word((int16_t)dwt_lo16(h[d + 2].x), a.sm), to_coder_word((int16_t)dwt_lo16(h[d + 2].y), a.sm));
        *reinterpret_cast<uint32_t *>(a.coef + (size_t)(nlh + ky) * a.coef_stride + nlw + kx) =
            dwt_pack(to_coder_word((int16_t)hh0, a.sm), to_coder_word((int16_t)hh1, a.sm));
    }
    return ovf;
}

}  // namespace icer
