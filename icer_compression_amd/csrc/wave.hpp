// wave.hpp -- wavefront-level SPMD vocabulary for the gfx950 kernels.
//
// The coding-unit kernel is written "one wavefront = one coding unit": 64 lanes cooperate
// through ballots, lane-masked popcounts (v_mbcnt) and LDS.  Kernel bodies are written with the
// macros below.  In the product build (hipcc, gfx950) they are the plain SIMT constructs:
//   LANEVAR(T, x)   ->  T x            (a VGPR)
//   FOR_LANES {..}  ->  {..}           (executed by every lane)
//   BALLOT(e)       ->  __ballot(e)    (64-bit SGPR mask)
// The SAME source also compiles with g++ under -DICER_WAVE_EMU, where a lane variable becomes an
// array of 64 and FOR_LANES a loop.  That build exists only for tests/ (tests/emu): it lets the
// wave-parallel algorithm be checked against the oracle in the authoring container, which has
// no GPU.  It is not part of libicer_hip.so and no product entry point can reach it.
#pragma once
#include <stdint.h>

#ifdef ICER_WAVE_EMU
// ------------------------------------------------------------------ CPU lane-loop (tests only)
#define ICER_DEV
#define ICER_HD
#define LANEVAR(T, name) T name[64]
#define FOR_LANES for (int lane = 0; lane < 64; ++lane)
#define LV(x) x[lane]
#define DECL_LANE
#define WAVE_SYNC()
template <class F> static inline uint64_t emu_ballot(F f)
{
    uint64_t m = 0;
    for (int lane = 0; lane < 64; ++lane)
        if (f(lane)) m |= 1ull << lane;
    return m;
}
#define BALLOT(expr) emu_ballot([&](int lane) { (void)lane; return (bool)(expr); })
static inline int mbcnt64(uint64_t m, int lane) { return __builtin_popcountll(m & ((1ull << lane) - 1ull)); }
static inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
static inline int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }      // index of lowest set bit
static inline int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline int clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }
static inline uint32_t brev32(uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
#define LDS_OR(REF, VAL) ((REF) |= (VAL))
#define READLANE(X, L) (X[L])
// DST[lane] = SRC[IDX[lane]] (statement; not inside FOR_LANES)
#define WAVE_GATHER(DST, SRC, IDX) { uint32_t t_[64]; for (int l_ = 0; l_ < 64; ++l_) t_[l_] = SRC[l_]; for (int l_ = 0; l_ < 64; ++l_) DST[l_] = t_[IDX[l_] & 63u]; }
// cross-lane reductions / scans over a lane variable
#define WAVE_XOR(DST, X) { DST = 0; for (int l_ = 0; l_ < 64; ++l_) DST ^= X[l_]; }
#define WAVE_MAX(DST, X) { DST = 0; for (int l_ = 0; l_ < 64; ++l_) DST = X[l_] > DST ? X[l_] : DST; }
#define WAVE_EXCL_SCAN(T, OUT, IN, TOTAL) { T run_ = 0; for (int l_ = 0; l_ < 64; ++l_) { const T v_ = IN[l_]; OUT[l_] = run_; run_ += v_; } TOTAL = run_; }
#else
// ------------------------------------------------------------------ gfx950 (product)
#include <hip/hip_runtime.h>
#define ICER_DEV __device__ __forceinline__
#define ICER_HD __host__ __device__ __forceinline__
#define LANEVAR(T, name) T name
#define FOR_LANES
#define LV(x) x
#define DECL_LANE const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))
// The coding-unit kernel runs ONE wavefront per workgroup: LDS traffic of a single wave is executed in
// program order by the LDS pipeline, so cross-lane hand-offs through LDS only need the compiler not to
// reorder or cache across the point (no s_barrier, no vmcnt drain -> global prefetches stay in flight).
#define WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
#define BALLOT(expr) ((uint64_t)__ballot((int)(expr)))
// number of set bits of m strictly below this lane (two v_mbcnt instructions)
static __device__ __forceinline__ int mbcnt64(uint64_t m, int /*lane*/)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
static __device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }
static __device__ __forceinline__ int ffs64(uint64_t m) { return m ? (int)__builtin_ctzll(m) : 64; }
static __device__ __forceinline__ int clz32(uint32_t v) { return v ? (int)__builtin_clz(v) : 32; }
static __device__ __forceinline__ int clz64(uint64_t v) { return v ? (int)__builtin_clzll(v) : 64; }
static __device__ __forceinline__ uint32_t brev32(uint32_t v) { return __brev(v); }
#define LDS_OR(REF, VAL) atomicOr(&(REF), (VAL))
#define READLANE(X, L) ((uint32_t)__builtin_amdgcn_readlane((int)(X), (int)(L)))
#define WAVE_GATHER(DST, SRC, IDX) { DST = (uint32_t)__shfl((int)(SRC), (int)(IDX)); }
#define WAVE_XOR(DST, X) { uint32_t t_ = (X); for (int o_ = 32; o_ > 0; o_ >>= 1) t_ ^= (uint32_t)__shfl_xor((int)t_, o_); DST = t_; }
#define WAVE_MAX(DST, X) { uint32_t t_ = (X); for (int o_ = 32; o_ > 0; o_ >>= 1) { const uint32_t u_ = (uint32_t)__shfl_xor((int)t_, o_); t_ = u_ > t_ ? u_ : t_; } DST = t_; }
#define WAVE_EXCL_SCAN(T, OUT, IN, TOTAL) { const T v_ = (IN); T s_ = v_; \
    for (int o_ = 1; o_ < 64; o_ <<= 1) { const T u_ = (T)__shfl_up(s_, o_); if (lane >= o_) s_ += u_; } \
    OUT = s_ - v_; TOTAL = (T)__shfl(s_, 63); }
#endif
