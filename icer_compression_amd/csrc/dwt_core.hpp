// dwt_core.hpp -- one output pair of the ICER integer lifting transform, as a pure function of
// the eight input samples it depends on.
//
// Replaces icer_wavelet_transform_1d_uint16 (lib_icer/src/icer_wavelet.c:385-465) together with
// the in-place un-shuffle it relies on (icer_deinterleave_uint16 :765-820 and helpers): the
// reference lifts pairs in place, permutes to [lows | highs] and then predicts in place; the
// result is a local function -- output pair k of a line of n samples needs x[2k-4 .. 2k+3] --
// so a GPU thread computes (low_k, high_k) directly and writes the final layout.
//
// Bit-exactness notes (see DESIGN.md "DWT"):
//  * lows/highs are stored truncated to int16 with a sticky overflow flag on the untruncated
//    value; r[j] = (int16)(low[j-1]-low[j]) wraps silently (get_r_int16 :206-208).
//  * boundary rules k==0, k==nh-1 (even n), missing d[k+1] (odd n) exactly as :430-462.
//  * filter C (alpha_-1 != 0) at k==1 uses d[1] where the ICER paper has d[2] (reference quirk,
//    offset=low_N at :437-440), and 0 when n == 5.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DWT_HD __host__ __device__ __forceinline__
#else
#define DWT_HD static inline
#endif

namespace icer {

struct DwtPair {
    int16_t low, high;
    bool has_high;
    bool overflow;
};

// `X(i)` returns sample i (0 <= i < n) of the line as int16.  k in [0, ceil(n/2)).
// `lim` = largest storable value: 32767, or 127 for the int8 twin (icer_wavelet_transform_1d_uint8,
// icer_wavelet.c:215-296) whose samples live sign-extended in int16 -- an out-of-range store only matters through
// the overflow flag there (the frame is refused), so the values themselves are formed exactly as for int16.
template <class Load>
DWT_HD DwtPair dwt_pair(const Load &X, int n, int k, int am1, int a0, int a1, int be, int32_t lim = 32767)
{
    const auto fits_i16 = [lim](int32_t v) { return v >= -lim - 1 && v <= lim; };
    const int nl = (n + 1) >> 1, nh = n >> 1;
    const bool odd = (n & 1) != 0;
    DwtPair out;
    out.overflow = false;
    out.has_high = k < nh;

    // lows k-2 .. k+1 (clamped to the valid range; clamped copies are never used un-multiplied)
    int32_t lo[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int idx = k - 2 + j;
        idx = idx < 0 ? 0 : (idx > nl - 1 ? nl - 1 : idx);
        int32_t v;
        if (idx < nh) {
            const int32_t a = X(2 * idx), b = X(2 * idx + 1);
            v = (a + b) >> 1;                                   // floor((a+b)/2)
            if (j == 2 && !fits_i16(v)) out.overflow = true;    // own pair only: neighbours flag theirs
        } else v = X(n - 1);                                    // odd tail sample is a low
        lo[j] = (int16_t)v;
    }
    out.low = (int16_t)lo[2];
    if (!out.has_high) { out.high = 0; return out; }

    // highs k, k+1 before prediction
    const int32_t dk_full = (int32_t)X(2 * k) - (int32_t)X(2 * k + 1);
    if (!fits_i16(dk_full)) out.overflow = true;
    const int32_t dk = (int16_t)dk_full;
    int32_t dn = 0;
    if (k + 1 < nh) dn = (int16_t)((int32_t)X(2 * k + 2) - (int32_t)X(2 * k + 3));

#define R_(j) ((int32_t)(int16_t)(lo[(j)-1] - lo[(j)]))          /* r[k-2+j] in terms of lo[] slots */
    // slots: lo[0]=low[k-2], lo[1]=low[k-1], lo[2]=low[k], lo[3]=low[k+1]
    const int32_t r_km1 = R_(1), r_k = R_(2), r_kp1 = R_(3);
#undef R_
    int32_t sub;
    if (k == 0) {
        sub = r_kp1 >> 2;                                       // floor(r[1]/4)
    } else if (k == 1 && am1 != 0) {
        const int32_t x = (odd && nl == 3) ? 0 : dk;            // QUIRK W3
        sub = (2 * r_k + 3 * r_kp1 - 2 * x + 4) >> 3;
    } else if (!odd && k == nh - 1) {
        sub = r_k >> 2;                                         // floor(r[nh-1]/4)
    } else {
        const int32_t rm = (k >= 2) ? r_km1 : 1;                // r[0] reads as 1 (times alpha_-1 == 0)
        sub = (am1 * rm + a0 * r_k + a1 * r_kp1 - be * dn + 8) >> 4;
    }
    const int32_t h = dk - sub;
    if (!fits_i16(h)) out.overflow = true;
    out.high = (int16_t)h;
    return out;
}

}  // namespace icer
