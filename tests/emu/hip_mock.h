// hip_mock.h -- TEST ONLY.  Just enough of the HIP runtime, on the CPU, to run the HOST pipeline of decoder.hip in a
// container without a GPU: device memory is host memory (filled with 0xCD so that reading what was never written shows),
// a kernel launch is a loop over the grid that calls the kernel once per thread (ICER_LAUNCH) or once per wavefront for
// kernels written with the lane-loop macros of wave.hpp (ICER_LAUNCH_WAVE).  Built only by tests/test_emu_decoder.py
// (g++ -x c++ -DICER_HOST_MOCK -DICER_WAVE_EMU -include tests/emu/hip_mock.h ...); no product library is built this way.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __launch_bounds__(n)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;
static thread_local void *g_mock_lds = nullptr;

typedef int hipError_t;
static const hipError_t hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost };
static inline const char *hipGetErrorString(hipError_t) { return "mock error"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = getenv("ICER_MOCK_NO_DEVICE") ? 0 : 1; return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n)
{
    *p = (T *)malloc(n ? n : 1);
    if (!*p) return 2;
    memset(*p, 0xCD, n);
    return hipSuccess;
}
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
typedef void *hipStream_t;                       // (launches of the mock are plain calls: a stream is a name only)
static inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
static inline uint32_t atomicXor(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o ^ v; return o; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }

#define ICER_MOCK_GRID(grid, block, shmem, per_block)                                          \
    do {                                                                                       \
        const dim3 g_(grid), b_(block);                                                        \
        const size_t lds_ = (size_t)(shmem);                                                   \
        g_mock_lds = malloc(lds_ ? lds_ : 1);                                                  \
        gridDim = g_; blockDim = b_;                                                           \
        for (unsigned bz_ = 0; bz_ < g_.z; bz_++)                                              \
            for (unsigned by_ = 0; by_ < g_.y; by_++)                                          \
                for (unsigned bx_ = 0; bx_ < g_.x; bx_++) {                                    \
                    blockIdx = dim3(bx_, by_, bz_);                                            \
                    memset(g_mock_lds, 0xCD, lds_);                                            \
                    per_block                                                                  \
                }                                                                              \
        free(g_mock_lds); g_mock_lds = nullptr;                                                \
    } while (0)
#define ICER_LAUNCH(kernel, grid, block, shmem, ...) \
    ICER_MOCK_GRID(grid, block, shmem, for (unsigned tx_ = 0; tx_ < b_.x; tx_++) { threadIdx = dim3(tx_); kernel(__VA_ARGS__); })
#define ICER_LAUNCH_WAVE(kernel, grid, shmem, ...) ICER_MOCK_GRID(grid, 64, shmem, threadIdx = dim3(0); kernel(__VA_ARGS__);)
#define ICER_DYNAMIC_LDS(T, name) T *name = (T *)g_mock_lds
#define ICER_LDS_TABLES(name, src) const DecoderTables &name = *(src)
