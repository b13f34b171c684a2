// salu_rate_ubench.hip -- how many scalar instructions does a gfx950 compute unit issue per cycle, and what do the scalar
// idioms of the wave-per-plane decoder (decoder_planes.hpp) cost?  (round 4: two long decoder chains on one compute unit
// run at 0.6 x the speed of one, with 0.6-0.75 scalar instructions per cycle and compute unit on the counters.)
//
// Every kernel runs ITERS x one asm block of 64 (or 32 x 2 ...) instructions per wave; the grid puts 1, 4, 8, 16 or 32 waves
// on every compute unit.  Reported: instructions per cycle per compute unit at 2.4 GHz, and cycles per instruction of ONE wave.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/salu_rate_ubench.hip -o /tmp/salu_rate_ubench ; prints a markdown table
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

constexpr int kIters = 2048;

// mode 0: one dependent chain of s_add_u32
// mode 1: four independent chains
// mode 2: s_cmp + taken s_cbranch (forward, over one s_nop) -- 16 x (cmp, branch) = 32 instructions issued (+ the skipped nop is not)
// mode 3: v_readlane -> s_add on its result (VALU -> SALU hand-over), 32 x 2
// mode 4: s_bfe / s_lshl / s_and / s_cselect mix, dependent, 64
// mode 5: v_cmp -> s_bcnt1 -> s_add (ballot + popcount), 16 x 3 (+ 16 v_mov to feed back) = 64
// mode 6: not-taken s_cbranch after s_cmp, 32 x 2
// mode 7: taken backward-free long forward branch over 16 dwords (instruction fetch restart), 16 x (cmp, branch)
template <int MODE>
__global__ void __launch_bounds__(256) salu_kernel(uint32_t *out, uint64_t *cyc)
{
    uint32_t a = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 1u, b = a + 1u, c = a + 2u, d = a + 3u;
    uint32_t v = threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
        if (MODE == 0) asm volatile(R64("s_add_u32 %0, %0, 3\n") : "+s"(a) : : "scc");
        if (MODE == 1) asm volatile(R16("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 3\n s_add_u32 %3, %3, 3\n") : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
        if (MODE == 2) asm volatile(R16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n") : "+s"(a) : : "scc");
        if (MODE == 3) asm volatile(R16("v_readlane_b32 %1, %2, 3\n s_add_u32 %0, %0, %1\n v_readlane_b32 %1, %2, 5\n s_add_u32 %0, %0, %1\n") : "+s"(a), "+s"(b) : "v"(v) : "scc");
        if (MODE == 4) asm volatile(R16("s_bfe_u32 %1, %0, 0x100004\n s_lshl_b32 %1, %1, 1\n s_and_b32 %1, %1, 0xffff\n s_add_u32 %0, %0, %1\n") : "+s"(a), "+s"(b) : : "scc");
        if (MODE == 5) asm volatile(R16("v_cmp_lt_u32 vcc, %1, %2\n s_bcnt1_i32_b64 %1, vcc\n s_add_u32 %0, %0, %1\n v_add_u32 %2, %1, %2\n") : "+s"(a), "+s"(b), "+v"(v) : : "scc", "vcc");
        if (MODE == 6) asm volatile(R16("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 3\n s_add_u32 %0, %0, 5\n 1:\n") : "+s"(a) : : "scc");
        if (MODE == 7) asm volatile(R16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n" R16("s_nop 0\n") "1:\n") : "+s"(a) : : "scc");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63u) == 0u) {
        out[blockIdx.x * 4u + (threadIdx.x >> 6)] = a + b + c + d + v;
        if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    }
}

struct Case { const char *name; int issued; void (*fn)(uint32_t *, uint64_t *); };

int main()
{
    uint32_t *out; uint64_t *cyc;
    HIP_OK(hipMalloc(&out, 4u * 256u * 64u * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    const Case cases[] = {
        {"s_add_u32, one dependent chain", 64, salu_kernel<0>},
        {"s_add_u32, four independent chains", 64, salu_kernel<1>},
        {"s_cmp + taken s_cbranch over one instruction", 32, salu_kernel<2>},
        {"v_readlane -> s_add on the result", 64, salu_kernel<3>},
        {"s_bfe, s_lshl, s_and, s_add (dependent)", 64, salu_kernel<4>},
        {"v_cmp -> s_bcnt1 -> s_add -> v_add (ballot + popcount)", 64, salu_kernel<5>},
        {"s_cmp + s_cbranch NOT taken + 2 s_add", 64, salu_kernel<6>},
        {"s_cmp + taken s_cbranch over 16 dwords", 32, salu_kernel<7>},
    };
    printf("| case | instr per block | instr/cycle/CU @1 wave per CU | @4 | @8 | @16 | @32 | one wave: cycles per instr @1 | @32 |\n|---|---|---|---|---|---|---|---|---|\n");
    for (const Case &c : cases) {
        double rate[5] = {0, 0, 0, 0, 0}, wave_cyc[5] = {0, 0, 0, 0, 0};
        const int waves_per_cu[5] = {1, 4, 8, 16, 32};
        for (int k = 0; k < 5; k++) {
            const int wpc = waves_per_cu[k];
            const dim3 block(wpc == 1 ? 64 : 256), grid(wpc == 1 ? 256 : 256 * (wpc / 4));
            for (int rep = 0; rep < 2; rep++) {
                HIP_OK(hipEventRecord(e0));
                hipLaunchKernelGGL(c.fn, grid, block, 0, 0, out, cyc);
                HIP_OK(hipEventRecord(e1));
                HIP_OK(hipEventSynchronize(e1));
            }
            float ms = 0;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t cy = 0;
            HIP_OK(hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost));
            const double instr = (double)c.issued * kIters * wpc * 256.0;
            rate[k] = instr / (ms * 1e-3 * 2.4e9 * 256.0);
            wave_cyc[k] = (double)cy / ((double)c.issued * kIters);
        }
        printf("| %s | %d | %.3f | %.3f | %.3f | %.3f | %.3f | %.1f | %.1f |\n", c.name, c.issued, rate[0], rate[1], rate[2], rate[3], rate[4], wave_cyc[0], wave_cyc[4]);
    }
    return 0;
}
