// valu_rate_ubench.hip -- what does one wave64 instruction of the classes the coder uses cost on gfx950?
//
// bench.py priced a VALU wave-instruction at 4 SIMD cycles; the guide measures wave64 v_fma_f32 at 2.
// For every instruction class: a kernel whose waves run ITERS x 64 copies of the instruction
//   * on 8 independent registers  -> throughput: SIMD cycles per wave-instruction once enough waves are resident
//   * on 1 register (dependent)   -> what a lone dependent chain pays per instruction (the coder's waves are such chains)
// at 1, 2, 4 and 8 waves per SIMD (256 * W workgroups of 256 threads: W workgroups per compute unit, one wave of each per SIMD).
// Reported per class and occupancy: cycles per wave-instruction per SIMD from the kernel's duration
// (duration x 2.4 GHz x 1024 SIMDs / wave-instructions) and the per-wave view (s_memtime around the loop / instructions).
// Plus two protocol costs of the pipeline: an LDS poll round trip (load counter, compare, branch) and s_sleep 1.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate_ubench.hip -o /tmp/valu_rate_ubench ; prints one JSON object
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 256;       // outer loop
constexpr int kUnroll = 64;       // instructions per iteration

// ---- instruction classes: OP(acc, other) must read and write acc ----------------------------------------------------
#define OP_AND(a, b)     asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(b))
#define OP_ADD(a, b)     asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(b))
#define OP_LSHL(a, b)    asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a))
#define OP_BFE(a, b)     asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a))
#define OP_CNDMASK(a, b) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : )
#define OP_MBCNT(a, b)   asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(a))
#define OP_MBCNTHI(a, b) asm volatile("v_mbcnt_hi_u32_b32 %0, -1, %0" : "+v"(a))
#define OP_FMA(a, b)     asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b))
#define OP_MULLO(a, b)   asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_MUL24(a, b)   asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_CMP(a, b)     asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc")   /* 2 instructions */
#define OP_BCNT(a, b)    asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_ALIGNBIT(a, b) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a) : "v"(b))
#define OP_LSHL64(a, b)  asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a))          /* a is 64-bit */
#define OP_BPERM(a, b)   asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b))
#define OP_SWIZ(a, b)    asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n s_waitcnt lgkmcnt(0)" : "+v"(a))
#define OP_DPP(a, b)     asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a))
#define OP_READLANE(a, b) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(a)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(s_)); }   /* 2 instructions */
#define OP_CNDMASK_SGPR(a, b) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(b) : )
#define OP_CMPSEL(a, b)  asm volatile("v_cmp_lt_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc")   /* 2 instructions */
#define OP_CMPSEL64(a, b) asm volatile("v_cmp_lt_u32_e64 s[10:11], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a) : "v"(b) : "s10", "s11")   /* 2 instructions */
#define OP_CMP64(a, b)   asm volatile("v_cmp_lt_u32_e64 s[10:11], %1, %0" : : "v"(a), "v"(b) : "s10", "s11")
#define OP_OR(a, b)      asm volatile("v_or_b32 %0, %1, %0" : "+v"(a) : "v"(b))
#define OP_XOR(a, b)     asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(b))
#define OP_SUB(a, b)     asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_MIN(a, b)     asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_MOV(a, b)     asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b))
#define OP_LSHLADD(a, b) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b))
#define OP_ANDOR(a, b)   asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b))
#define OP_ADD3(a, b)    asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b))
#define OP_LSHLOR(a, b)  asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a) : "v"(b))
#define OP_LSHR(a, b)    asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a))
#define OP_FFBH(a, b)    asm volatile("v_ffbh_u32 %0, %0" : "+v"(a))
#define OP_CVT(a, b)     asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a))
#define OP_RCP(a, b)     asm volatile("v_rcp_f32 %0, %0" : "+v"(a))
#define OP_MULF(a, b)    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b))
#define OP_PERM(a, b)    asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b))
#define OP_READFIRST(a, b) { uint32_t s_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s_) : "v"(a)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "s"(s_)); }
#define OP_DSREAD(a, b)  asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a))
#define OP_DSREADU8(a, b) asm volatile("ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a))
#define OP_SADD(a, b)    asm volatile("s_add_u32 %0, %0, 3" : "+s"(a))
#define OP_SBCNT(a, b)   asm volatile("s_bcnt1_i32_b64 %0, %1" : "+s"(a) : "s"(b) : "scc")

template <int ILP, class F>
__device__ __forceinline__ void run_loop(F body)
{
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll / ILP; u++) body();
    }
}

#define VALU_KERNEL(NAME, OP, TYPE)                                                                          \
    template <int ILP> __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint64_t *cyc)            \
    {                                                                                                        \
        TYPE a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = blockIdx.x | 1u;                                                                        \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < kIters; it++) {                                                                \
            _Pragma("unroll") for (int u = 0; u < kUnroll / ILP; u++) {                                      \
                OP(a0, b);                                                                                   \
                if (ILP > 1) { OP(a1, b); }                                                                  \
                if (ILP > 2) { OP(a2, b); OP(a3, b); }                                                       \
                if (ILP > 4) { OP(a4, b); OP(a5, b); OP(a6, b); OP(a7, b); }                                 \
            }                                                                                                \
        }                                                                                                    \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                    \
    }

VALU_KERNEL(k_and, OP_AND, uint32_t)
VALU_KERNEL(k_add, OP_ADD, uint32_t)
VALU_KERNEL(k_lshl, OP_LSHL, uint32_t)
VALU_KERNEL(k_bfe, OP_BFE, uint32_t)
VALU_KERNEL(k_cndmask, OP_CNDMASK, uint32_t)
VALU_KERNEL(k_mbcnt_lo, OP_MBCNT, uint32_t)
VALU_KERNEL(k_mbcnt_hi, OP_MBCNTHI, uint32_t)
VALU_KERNEL(k_fma, OP_FMA, float)
VALU_KERNEL(k_mullo, OP_MULLO, uint32_t)
VALU_KERNEL(k_mul24, OP_MUL24, uint32_t)
VALU_KERNEL(k_cmp_addc, OP_CMP, uint32_t)
VALU_KERNEL(k_bcnt, OP_BCNT, uint32_t)
VALU_KERNEL(k_alignbit, OP_ALIGNBIT, uint32_t)
VALU_KERNEL(k_lshl64, OP_LSHL64, uint64_t)
VALU_KERNEL(k_bpermute, OP_BPERM, uint32_t)
VALU_KERNEL(k_swizzle, OP_SWIZ, uint32_t)
VALU_KERNEL(k_dpp, OP_DPP, uint32_t)
VALU_KERNEL(k_readlane_add, OP_READLANE, uint32_t)
VALU_KERNEL(k_cndmask_sgpr, OP_CNDMASK_SGPR, uint32_t)
VALU_KERNEL(k_cmp_sel_vcc, OP_CMPSEL, uint32_t)
VALU_KERNEL(k_cmp_sel_sgpr, OP_CMPSEL64, uint32_t)
VALU_KERNEL(k_cmp_e64, OP_CMP64, uint32_t)
VALU_KERNEL(k_or, OP_OR, uint32_t)
VALU_KERNEL(k_xor, OP_XOR, uint32_t)
VALU_KERNEL(k_sub, OP_SUB, uint32_t)
VALU_KERNEL(k_min, OP_MIN, uint32_t)
VALU_KERNEL(k_mov, OP_MOV, uint32_t)
VALU_KERNEL(k_lshl_add, OP_LSHLADD, uint32_t)
VALU_KERNEL(k_and_or, OP_ANDOR, uint32_t)
VALU_KERNEL(k_add3, OP_ADD3, uint32_t)
VALU_KERNEL(k_lshl_or, OP_LSHLOR, uint32_t)
VALU_KERNEL(k_lshr, OP_LSHR, uint32_t)
VALU_KERNEL(k_ffbh, OP_FFBH, uint32_t)
VALU_KERNEL(k_cvt, OP_CVT, uint32_t)
VALU_KERNEL(k_rcp, OP_RCP, float)
VALU_KERNEL(k_mulf, OP_MULF, float)
VALU_KERNEL(k_perm, OP_PERM, uint32_t)
VALU_KERNEL(k_readfirst_add, OP_READFIRST, uint32_t)

// compiler-made selects: what `c ? x : y` on lane values becomes
template <int ILP> __global__ void __launch_bounds__(256) k_select_cpp(uint32_t *out, uint64_t *cyc)
{
    uint32_t a = threadIdx.x, b = blockIdx.x | 1u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) { a = (a & 4u) ? a + b : a ^ b; asm volatile("" : "+v"(a)); }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// scalar: s_add chain, s_bcnt1_i32_b64
template <int ILP> __global__ void __launch_bounds__(256) k_sadd(uint32_t *out, uint64_t *cyc)
{
    uint32_t a0 = blockIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll / ILP; u++) {
            OP_SADD(a0, 0);
            if (ILP > 1) { OP_SADD(a1, 0); }
            if (ILP > 2) { OP_SADD(a2, 0); OP_SADD(a3, 0); }
            if (ILP > 4) { OP_SADD(a0, 0); OP_SADD(a1, 0); OP_SADD(a2, 0); OP_SADD(a3, 0); }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// ballot as the coder does it: v_cmp into an SGPR pair, then a scalar popcount of it, added back (vector) -- 3 instructions
template <int ILP> __global__ void __launch_bounds__(256) k_ballot_popc(uint32_t *out, uint64_t *cyc)
{
    uint32_t a = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            const uint64_t m = __ballot((a & 1u) != 0u);
            a += (uint32_t)__popcll(m);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// dependent LDS round trip: every load's address comes from the previous load (pointer chase inside the wave's own region)
template <int ILP> __global__ void __launch_bounds__(256) k_lds_chase(uint32_t *out, uint64_t *cyc)
{
    __shared__ uint32_t tab[256 * 4];
    for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = (uint32_t)((i * 7 + 64) & 1023);
    __syncthreads();
    uint32_t p = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int u = 0; u < kUnroll; u++) p = tab[p];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = p;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the pipeline's poll: atomic (relaxed, workgroup) load of two LDS words, compare, branch; the counter never moves, the
// loop runs a fixed number of times.  With SLEEP: s_sleep 1 between polls, as ICER_WAIT_CNT does.
template <int SLEEP> __global__ void __launch_bounds__(256) k_lds_poll(uint32_t *out, uint64_t *cyc)
{
    __shared__ uint32_t ctl[8];
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
    __syncthreads();
    uint32_t seen = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters * kUnroll / 4; it++) {
        const uint32_t v = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t ab = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((v > 12345u) || ab) break;
        seen += v + 1;
        if (SLEEP) __builtin_amdgcn_s_sleep(1);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = seen;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// publish: release fence (LDS only) + lane-0 store + the matching acquire on a value that is already there
template <int ILP> __global__ void __launch_bounds__(256) k_publish(uint32_t *out, uint64_t *cyc)
{
    __shared__ uint32_t ctl[8];
    __shared__ uint32_t data[256];
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
    __syncthreads();
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters * kUnroll / 8; it++) {
        data[threadIdx.x] = acc + it;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&ctl[threadIdx.x >> 6], (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t v = __hip_atomic_load(&ctl[threadIdx.x >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        acc += v + data[threadIdx.x ^ 1];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

struct Case {
    const char *name;
    void (*fn)(uint32_t *, uint64_t *);
    double insts_per_wave;          // wave-instructions of the class per wave
    const char *what;
};

int main()
{
    int dev_count = 0;
    HIP_OK(hipGetDeviceCount(&dev_count));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = 2.4;
    uint32_t *d_out; uint64_t *d_cyc;
    const int max_blocks = cus * 8;
    HIP_OK(hipMalloc(&d_out, (size_t)max_blocks * 256 * 4));
    HIP_OK(hipMalloc(&d_cyc, (size_t)max_blocks * 4 * 8));
    const double N = (double)kIters * kUnroll;
#define C2(NAME, MULT, WHAT) {#NAME " x8 independent", NAME<8>, N * (MULT), WHAT}, {#NAME " dependent", NAME<1>, N * (MULT), WHAT}
    const Case cases[] = {
        C2(k_and, 1, "v_and_b32"), C2(k_add, 1, "v_add_u32"), C2(k_lshl, 1, "v_lshlrev_b32"), C2(k_bfe, 1, "v_bfe_u32"),
        C2(k_cndmask, 1, "v_cndmask_b32"), C2(k_mbcnt_lo, 1, "v_mbcnt_lo_u32_b32"), C2(k_mbcnt_hi, 1, "v_mbcnt_hi_u32_b32"),
        C2(k_bcnt, 1, "v_bcnt_u32_b32"), C2(k_alignbit, 1, "v_alignbit_b32"), C2(k_mul24, 1, "v_mul_u32_u24"), C2(k_mullo, 1, "v_mul_lo_u32"),
        C2(k_lshl64, 1, "v_lshlrev_b64"), C2(k_fma, 1, "v_fma_f32 (the guide's reference point)"),
        C2(k_cmp_addc, 2, "v_cmp_ne_u32 vcc + v_addc_co_u32 (2 instructions)"), C2(k_dpp, 1, "v_mov_b32_dpp row_shr:1"),
        C2(k_readlane_add, 2, "v_readlane_b32 + v_add_u32 with the SGPR (2 instructions)"),
        C2(k_bpermute, 1, "ds_bpermute_b32 + s_waitcnt"), C2(k_swizzle, 1, "ds_swizzle_b32 + s_waitcnt"),
        C2(k_cndmask_sgpr, 1, "v_cndmask_b32_e64 with an SGPR pair"), C2(k_cmp_sel_vcc, 2, "v_cmp_lt_u32 vcc + v_cndmask_b32 vcc (2 instructions)"),
        C2(k_cmp_sel_sgpr, 2, "v_cmp_lt_u32_e64 s[10:11] + v_cndmask_b32_e64 (2 instructions)"), C2(k_cmp_e64, 1, "v_cmp_lt_u32_e64 into an SGPR pair"),
        C2(k_or, 1, "v_or_b32"), C2(k_xor, 1, "v_xor_b32"), C2(k_sub, 1, "v_sub_u32"), C2(k_min, 1, "v_min_u32"), C2(k_mov, 1, "v_mov_b32"),
        C2(k_lshl_add, 1, "v_lshl_add_u32"), C2(k_and_or, 1, "v_and_or_b32"), C2(k_add3, 1, "v_add3_u32"), C2(k_lshl_or, 1, "v_lshl_or_b32"),
        C2(k_lshr, 1, "v_lshrrev_b32"), C2(k_ffbh, 1, "v_ffbh_u32"), C2(k_cvt, 1, "v_cvt_f32_u32"), C2(k_rcp, 1, "v_rcp_f32"), C2(k_mulf, 1, "v_mul_f32"),
        C2(k_perm, 1, "v_perm_b32"), C2(k_readfirst_add, 2, "v_readfirstlane_b32 + v_add_u32 (2 instructions)"),
        {"k_select_cpp dependent", k_select_cpp<1>, N * 4, "a = (a & 4) ? a + b : a ^ b as hipcc compiles it (v_and, v_cmp, v_add, v_xor, v_cndmask: ~4-5 instructions)"},
        C2(k_sadd, 1, "s_add_u32"),
        {"k_ballot_popc dependent", k_ballot_popc<1>, N * 4, "v_and -> v_cmp -> s_bcnt1_i32_b64 -> v_add (4 instructions, the coder's ballot + popcount)"},
        {"k_lds_chase dependent", k_lds_chase<1>, N, "ds_read_b32 whose address is the previous load (round trip)"},
        {"k_lds_poll nosleep", k_lds_poll<0>, N / 4, "poll: 2 x ds_read_b32 (relaxed atomic), compare, branch"},
        {"k_lds_poll sleep1", k_lds_poll<1>, N / 4, "the same + s_sleep 1 (ICER_WAIT_CNT)"},
        {"k_publish dependent", k_publish<1>, N / 8, "ds_write, release fence, lane-0 store, load, acquire fence, ds_read (one hand-off, both sides)"},
    };
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_ghz_assumed\": %.2f, \"iters\": %d, \"unroll\": %d, \"cases\": [\n", prop.gcnArchName, cus, ghz, kIters, kUnroll);
    bool first = true;
    for (const Case &c : cases) {
        for (int W : {1, 2, 4, 8}) {
            const int blocks = cus * W;
            hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);      // warm-up
            HIP_OK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                HIP_OK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
                HIP_OK(hipEventRecord(e1, 0));
                HIP_OK(hipEventSynchronize(e1));
                float ms;
                HIP_OK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<uint64_t> cyc((size_t)blocks * 4);
            HIP_OK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            uint64_t mx = 0;
            for (uint64_t v : cyc) { sum += (double)v; if (v > mx) mx = v; }
            const double per_wave = sum / (double)cyc.size() / c.insts_per_wave;
            const double total_insts = c.insts_per_wave * blocks * 4.0;
            const double simd_cycles = best * 1e-3 * ghz * 1e9 * cus * 4.0;
            printf("%s {\"case\": \"%s\", \"what\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"simd_cycles_per_wave_inst\": %.3f, "
                   "\"wave_cycles_per_inst_s_memtime\": %.3f, \"slowest_wave_cycles_per_inst\": %.3f}",
                   first ? " " : ",\n ", c.name, c.what, W, best, simd_cycles / total_insts, per_wave, (double)mx / c.insts_per_wave);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
