// dwt_core.hpp -- one output pair of the ICER integer lifting transform, as a pure function of
// the eight input samples it depends on.
//
// Replaces icer_wavelet_transform_1d_uint16 (lib_icer/src/icer_wavelet.c:385-465) together with
// the in-place un-shuffle it relies on (icer_deinterleave_uint16 :765-820 and helpers): the
// reference lifts pairs in place, permutes to [lows | highs] and then predicts in place; the
// result is a local function -- output pair k of a line of n samples needs x[2k-4 .. 2k+3] --
// so a GPU thread computes (low_k, high_k) directly and writes the final layout (dwt_tile.hpp: the lows and
// pair differences of a line are formed once in LDS, a high then reads the four lows and two differences it needs).
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
// (the two steps of a pair -- dwt_pair_step1: low + difference, dwt_pair_step2: the high from the line's lows and
// differences -- live in dwt_tile.hpp next to the tile pass that shares the lows between the pairs that need them)
}  // namespace icer
