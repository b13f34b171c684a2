// dwt_bw_ubench.hip -- why does dwt_tile_kernel move only ~0.8 TB/s?  (measurement aid, not part of the library)
// On a 4096 x 4096 int16 plane:
//   copy_rows        every thread copies 32-bit words, rows contiguous                       -> what plain streaming reaches here
//   copy_bands       the DWT's access pattern without any arithmetic: read a 128 x 32 tile (+ nothing), write the four
//                    bands where the DWT writes them (16-bit stores, 64 lanes = 128 contiguous bytes)
//   copy_bands32     the same with 32-bit stores (two adjacent band samples per lane)
//   dwt              the product kernel (stage 0 geometry)
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/dwt_bw_ubench.hip -o /tmp/dwt_bw_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "../../icer_compression_amd/csrc/assemble_core.hpp"
#include "../../icer_compression_amd/csrc/dwt_tile.hpp"
using namespace icer;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) copy_rows(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t nwords)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// tile = 64 x 16 pairs like the product: read 128 x 32 samples as 32-bit words, write LL / HL / LH / HH as 16-bit
__global__ void __launch_bounds__(256) copy_bands(const int16_t *__restrict__ src, int16_t *__restrict__ coef, int16_t *__restrict__ ll, int W, int H)
{
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    const int nlw = W / 2, nlh = H / 2;
    for (int i = t; i < 64 * 32; i += 256) {
        const int r = i / 64, p = i % 64;                       // window row, pair
        const uint32_t w = *reinterpret_cast<const uint32_t *>(src + (size_t)(ty * 32 + r) * W + tx * 128 + 2 * p);
        const int ky = ty * 16 + (r >> 1), kx = tx * 64 + p;
        const int16_t a = (int16_t)(w & 0xFFFF), b = (int16_t)(w >> 16);
        if (r & 1) { coef[(size_t)(nlh + ky) * W + kx] = a; coef[(size_t)(nlh + ky) * W + nlw + kx] = b; }
        else { ll[(size_t)ky * nlw + kx] = a; coef[(size_t)ky * W + nlw + kx] = b; }
    }
}

__global__ void __launch_bounds__(256) copy_bands32(const int16_t *__restrict__ src, int16_t *__restrict__ coef, int16_t *__restrict__ ll, int W, int H)
{
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    const int nlw = W / 2, nlh = H / 2;
    for (int i = t; i < 32 * 32; i += 256) {
        const int r = i / 32, p2 = i % 32;                      // window row, pair of pairs
        const uint2 w = *reinterpret_cast<const uint2 *>(src + (size_t)(ty * 32 + r) * W + tx * 128 + 4 * p2);
        const int ky = ty * 16 + (r >> 1), kx = tx * 64 + 2 * p2;
        const uint32_t lo = (w.x & 0xFFFF) | (w.y << 16), hi = (w.x >> 16) | (w.y & 0xFFFF0000u);
        if (r & 1) { *reinterpret_cast<uint32_t *>(coef + (size_t)(nlh + ky) * W + kx) = lo; *reinterpret_cast<uint32_t *>(coef + (size_t)(nlh + ky) * W + nlw + kx) = hi; }
        else { *reinterpret_cast<uint32_t *>(ll + (size_t)ky * nlw + kx) = lo; *reinterpret_cast<uint32_t *>(coef + (size_t)ky * W + nlw + kx) = hi; }
    }
}

__global__ void __launch_bounds__(kTileThreads) dwt_kernel(DwtStageArgs a, int *ovf)
{
    __shared__ DwtTileShared sh;
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    dwt_tile_load(sh, a, tx, ty, t);
    __syncthreads();
    bool o = dwt_tile_rows_step1(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_rows_step2(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_cols_step1(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_cols_step2(sh, a, tx, ty, t);
    if (o) atomicOr(ovf, 1);
}

// variants of the product kernel that stop early: where does the time go?
template <int STOP> __global__ void __launch_bounds__(kTileThreads) dwt_partial(DwtStageArgs a, int *ovf)
{
    __shared__ DwtTileShared sh;
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    dwt_tile_load(sh, a, tx, ty, t);
    __syncthreads();
    bool o = false;
    if (STOP >= 1) { o = dwt_tile_rows_step1(sh, a, tx, ty, t); __syncthreads(); }
    if (STOP >= 2) { o |= dwt_tile_rows_step2(sh, a, tx, ty, t); __syncthreads(); }
    if (STOP >= 3) { o |= dwt_tile_cols_step1(sh, a, tx, ty, t); __syncthreads(); }
    if (o || sh.lo[t & 31][t & 63] == 12345) atomicOr(ovf, 1);
}

int main()
{
    const int W = 4096, H = 4096;
    const size_t n = (size_t)W * H;
    int16_t *src, *coef, *ll; int *ovf;
    HIP_OK(hipMalloc(&src, n * 2)); HIP_OK(hipMalloc(&coef, n * 2)); HIP_OK(hipMalloc(&ll, n * 2)); HIP_OK(hipMalloc(&ovf, 4));
    HIP_OK(hipMemset(src, 1, n * 2)); HIP_OK(hipMemset(ovf, 0, 4));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    DwtStageArgs a;
    a.f = filter_taps(0); a.lim = 32767; a.sm = 16; a.coef = coef; a.coef_stride = W; a.src = src; a.src_stride = W; a.cw = W; a.ch = H; a.ll = ll; a.ll_stride = W / 2;
    const dim3 grid((W / 2 + kTileKX - 1) / kTileKX, (H / 2 + kTileKY - 1) / kTileKY);
    auto time = [&](const char *name, auto launch, double bytes) -> int {
        launch();
        HIP_OK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            HIP_OK(hipEventRecord(e0, 0));
            launch();
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
            float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-28s %8.1f us  %7.2f TB/s (of %.0f MB moved)\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
        return 0;
    };
    time("copy_rows (4096 blocks)", [&] { hipLaunchKernelGGL(copy_rows, dim3(4096), dim3(256), 0, 0, (const uint32_t *)src, (uint32_t *)coef, n / 2); }, 4.0 * n);
    time("copy_rows (16384 blocks)", [&] { hipLaunchKernelGGL(copy_rows, dim3(16384), dim3(256), 0, 0, (const uint32_t *)src, (uint32_t *)coef, n / 2); }, 4.0 * n);
    time("copy_bands (16-bit stores)", [&] { hipLaunchKernelGGL(copy_bands, dim3(W / 128, H / 32), dim3(256), 0, 0, src, coef, ll, W, H); }, 4.0 * n);
    time("copy_bands32 (32-bit stores)", [&] { hipLaunchKernelGGL(copy_bands32, dim3(W / 128, H / 32), dim3(256), 0, 0, src, coef, ll, W, H); }, 4.0 * n);
    time("dwt load only", [&] { hipLaunchKernelGGL(dwt_partial<0>, grid, dim3(kTileThreads), 0, 0, a, ovf); }, 2.0 * n);
    time("dwt load + rows1", [&] { hipLaunchKernelGGL(dwt_partial<1>, grid, dim3(kTileThreads), 0, 0, a, ovf); }, 2.0 * n);
    time("dwt load + rows1,2", [&] { hipLaunchKernelGGL(dwt_partial<2>, grid, dim3(kTileThreads), 0, 0, a, ovf); }, 2.0 * n);
    time("dwt load + rows + cols1", [&] { hipLaunchKernelGGL(dwt_partial<3>, grid, dim3(kTileThreads), 0, 0, a, ovf); }, 2.0 * n);
    time("dwt (product, stage 0)", [&] { hipLaunchKernelGGL(dwt_kernel, grid, dim3(kTileThreads), 0, 0, a, ovf); }, 4.0 * n);
    return 0;
}
