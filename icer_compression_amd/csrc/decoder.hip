// decoder.hip -- C ABI of libicer_hip_dec.so (include/icer_hip_dec.h) and the host-side pipeline of the decoder:
//   stream -> device | header candidates (one thread per byte offset) | payload CRCs (one thread per candidate)
//   | host: packet walk, table, chains (decoder_plan.hpp) | decode kernel (one thread per chain = segment, planes
//   top-down) | sign-magnitude removal + LL mean | inverse DWT, one level at a time (one thread per line)
//   | clamp, narrow, copy back.
// First version: correctness before speed; every loop is bounded by the stream / image size and no
// kernel waits on another thread.
#ifndef ICER_HOST_MOCK
#include <hip/hip_runtime.h>
// kernel launches go through these two macros so that tests/emu/hip_mock.h (CPU, tests only) can stand in for them
#define ICER_LAUNCH(kernel, grid, block, shmem, ...) kernel<<<(grid), (block), (shmem)>>>(__VA_ARGS__)
#define ICER_LAUNCH_WAVE(kernel, grid, shmem, ...) kernel<<<(grid), 64, (shmem)>>>(__VA_ARGS__)
#define ICER_DYNAMIC_LDS(T, name) extern __shared__ T name[]
#endif

#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/icer_hip_dec.h"
#include "decoder_wave.hpp"
#include "decoder_core.hpp"
#include "decoder_plan.hpp"

using namespace icer;

namespace {

thread_local std::string g_error;

int fail(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    fprintf(stderr, "libicer_hip_dec: %s\n", buf);
    return ICER_FATAL_ERROR;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        const hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) { rc = fail("%s: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

// ------------------------------------------------------------------------------------------ kernels
__global__ void __launch_bounds__(256)
count_headers_kernel(const uint8_t *__restrict__ stream, uint32_t len, const uint32_t *__restrict__ crc_tab,
                     PacketCandidate *__restrict__ out, uint32_t cap, uint32_t *__restrict__ count)
{
    const uint32_t off = blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= len) return;
    PacketCandidate c;
    if (!header_candidate(crc_tab, stream, len, off, &c)) return;
    const uint32_t at = atomicAdd(count, 1u);
    if (at < cap) out[at] = c;
}

__global__ void __launch_bounds__(64)
check_payloads_kernel(const uint8_t *__restrict__ stream, const uint32_t *__restrict__ crc_tab,
                      PacketCandidate *__restrict__ cands, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) check_payload(crc_tab, stream, &cands[i]);
}

__global__ void __launch_bounds__(64)
decode_chains_kernel(uint16_t *__restrict__ planes, size_t plane_samples, uint32_t image_w,
                     const ChainDesc *__restrict__ chains, const uint8_t *__restrict__ subbands, uint32_t n,
                     const uint8_t *__restrict__ stream, uint32_t len, const DecoderTables *__restrict__ tables,
                     int nplanes, int sign_bit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ChainDesc c = chains[i];
    decode_chain(planes + (size_t)c.chan * plane_samples, image_w, c, subbands[i], stream, len, *tables, nplanes, sign_bit);
}

// the same with one wavefront per chain and one lane per packet (decoder_wave.hpp); dynamic LDS = the row ring
__global__ void __launch_bounds__(64)
decode_chains_wave_kernel(uint16_t *__restrict__ planes, size_t plane_samples, uint32_t image_w,
                          const ChainDesc *__restrict__ chains, const uint8_t *__restrict__ subbands,
                          const uint8_t *__restrict__ stream, uint32_t len, const DecoderTables *__restrict__ tables,
                          int nplanes, int sign_bit, uint32_t pitch)
{
    ICER_DYNAMIC_LDS(uint16_t, ring);
    const ChainDesc c = chains[blockIdx.x];
    decode_chain_wave(ring, pitch, planes + (size_t)c.chan * plane_samples, image_w, c, subbands[blockIdx.x], stream, len,
                      *tables, nplanes, sign_bit, nullptr);
}

// sign-magnitude words -> int16, LL mean back in (grid.y = channel)
__global__ void __launch_bounds__(256)
unsign_kernel(uint16_t *__restrict__ planes, size_t plane_samples, uint32_t image_w, uint32_t ll_w, uint32_t ll_h,
              uint16_t mean0, uint16_t mean1, uint16_t mean2, int sign_bit, int bits)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane_samples) return;
    const uint32_t ch = blockIdx.y;
    uint16_t *p = planes + (size_t)ch * plane_samples;
    int16_t s = from_sign_magnitude(p[i], sign_bit);
    const uint32_t x = (uint32_t)(i % image_w), y = (uint32_t)(i / image_w);
    if (x < ll_w && y < ll_h) s = add_ll_mean(s, ch == 0 ? mean0 : ch == 1 ? mean1 : mean2, bits);
    p[i] = (uint16_t)s;
}

// one thread per column (rows = false) or per row (rows = true) of a level's region; grid.y = channel
__global__ void __launch_bounds__(64)
idwt_lines_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, size_t plane_samples, uint32_t image_w,
                  uint32_t cw, uint32_t ch_rows, FilterTaps taps, int bits, const uint32_t *__restrict__ pos_of, bool rows)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = (size_t)blockIdx.y * plane_samples;
    if (rows) {
        if (i < ch_rows) idwt_line(src + base + (size_t)i * image_w, dst + base + (size_t)i * image_w, cw, 1, taps, bits, pos_of);
    } else {
        if (i < cw) idwt_line(src + base + i, dst + base + i, ch_rows, image_w, taps, bits, pos_of);
    }
}

// icer_remove_negative_* (icer_util.c:70-91); `out8` != null: also narrow to bytes
__global__ void __launch_bounds__(256)
clamp_kernel(uint16_t *__restrict__ planes, size_t total, uint8_t *__restrict__ out8)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int16_t s = (int16_t)planes[i];
    if (s < 0) s = 0;
    planes[i] = (uint16_t)s;
    if (out8) out8[i] = (uint8_t)s;
}

__global__ void __launch_bounds__(256)
narrow_kernel(const uint16_t *__restrict__ planes, size_t total, uint8_t *__restrict__ out8)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out8[i] = (uint8_t)planes[i];
}

// ------------------------------------------------------------------------------------------ host pipeline
int decompress_planes(void *const planes[], int channels, size_t *image_w, size_t *image_h, size_t bufsize,
                      const uint8_t *data, size_t data_length, int stages, int filt, unsigned segments, int bits)
{
    g_error.clear();
    if (!image_w || !image_h || (!data && data_length) || (filt < 0 || filt > 6)) return ICER_INVALID_INPUT;
    for (int c = 0; c < channels; c++)
        if (!planes[c]) return ICER_INVALID_INPUT;
    if (data_length >= 0xFFFFFFFFull - 64u) return fail("stream of %zu bytes: 32-bit offsets only", data_length);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no usable HIP device");

    int rc = ICER_RESULT_OK;
    const uint32_t len = (uint32_t)data_length;
    uint8_t *d_stream = nullptr, *d_sub = nullptr, *d_out8 = nullptr;
    uint32_t *d_crc = nullptr, *d_count = nullptr, *d_pos = nullptr;
    PacketCandidate *d_cands = nullptr;
    ChainDesc *d_chains = nullptr;
    DecoderTables *d_tables = nullptr;
    uint16_t *d_planes = nullptr, *d_tmp = nullptr;
    std::vector<PacketCandidate> cands;
    DecodePlan pl;
    uint32_t crc_tab[256];
    build_crc32_table(crc_tab);

    // 1. packets
    if (len) {
        HIP_TRY(hipMalloc(&d_stream, len));
        HIP_TRY(hipMemcpy(d_stream, data, len, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&d_crc, sizeof crc_tab));
        HIP_TRY(hipMemcpy(d_crc, crc_tab, sizeof crc_tab, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&d_count, sizeof(uint32_t)));
        uint32_t cap = len / 64u + 1024u, count = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (d_cands) { HIP_TRY(hipFree(d_cands)); d_cands = nullptr; }
            HIP_TRY(hipMalloc(&d_cands, sizeof(PacketCandidate) * cap));
            HIP_TRY(hipMemset(d_count, 0, sizeof(uint32_t)));
            ICER_LAUNCH(count_headers_kernel, (len + 255u) / 256u, 256, 0, d_stream, len, d_crc, d_cands, cap, d_count);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(&count, d_count, sizeof count, hipMemcpyDeviceToHost));
            if (count <= cap) break;
            cap = count;                                    // (more header look-alikes than expected: once more, all of them)
        }
        if (count) {
            ICER_LAUNCH(check_payloads_kernel, (count + 63u) / 64u, 64, 0, d_stream, d_crc, d_cands, count);
            HIP_TRY(hipGetLastError());
            cands.resize(count);
            HIP_TRY(hipMemcpy(cands.data(), d_cands, sizeof(PacketCandidate) * count, hipMemcpyDeviceToHost));
            std::sort(cands.begin(), cands.end(), [](const PacketCandidate &a, const PacketCandidate &b) { return a.off < b.off; });
        }
    }

    // 2. plan
    plan_decode(&pl, data, cands, channels, stages, segments, bits, *image_w, *image_h, bufsize);
    *image_w = pl.w;
    *image_h = pl.h;
    rc = pl.rc;
    if (rc == kInvalidInput || rc == kTooManyStages || rc == kByteQuotaExceeded) goto done;
    {
        const size_t W = pl.w, H = pl.h, samples = W * H, total = samples * (size_t)channels;
        const int nplanes = bits == 8 ? kPlanes8 : kPlanes, sign_bit = bits == 8 ? 7 : 15;
        if (total == 0) goto done;
        if (W > 0xFFFFFFFFull || samples > 0xFFFFFFFFull) { rc = fail("image of %zu x %zu samples: 32-bit indices only", W, H); goto done; }
        HIP_TRY(hipMalloc(&d_planes, sizeof(uint16_t) * total));
        HIP_TRY(hipMemset(d_planes, 0, sizeof(uint16_t) * total));

        // 3. bit planes
        if (!pl.chains.empty()) {
            CoderTables ct;
            DecoderTables dt;
            build_coder_tables(&ct);
            build_decoder_tables(&dt, ct);
            const uint32_t n = (uint32_t)pl.chains.size();
            HIP_TRY(hipMalloc(&d_tables, sizeof dt));
            HIP_TRY(hipMemcpy(d_tables, &dt, sizeof dt, hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&d_chains, sizeof(ChainDesc) * n));
            HIP_TRY(hipMemcpy(d_chains, pl.chains.data(), sizeof(ChainDesc) * n, hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&d_sub, n));
            HIP_TRY(hipMemcpy(d_sub, pl.chain_subband.data(), n, hipMemcpyHostToDevice));
            // ICER_DEC_WAVE=1: the planes of a segment side by side (one wavefront per chain), if its rows fit the LDS ring
            uint32_t pitch = 2;
            for (const ChainDesc &c : pl.chains) pitch = std::max<uint32_t>(pitch, (c.w + 1u) & ~1u);
            const size_t ring_bytes = (size_t)kRingRows * pitch * sizeof(uint16_t);
            const char *mode = getenv("ICER_DEC_WAVE");
            if (mode && mode[0] == '1' && ring_bytes <= 65536u) {
                ICER_LAUNCH_WAVE(decode_chains_wave_kernel, n, ring_bytes, d_planes, samples, (uint32_t)W, d_chains, d_sub, d_stream, len,
                                 d_tables, nplanes, sign_bit, pitch);
            } else {
                ICER_LAUNCH(decode_chains_kernel, (n + 63u) / 64u, 64, 0, d_planes, samples, (uint32_t)W, d_chains, d_sub, n, d_stream, len,
                            d_tables, nplanes, sign_bit);
            }
            HIP_TRY(hipGetLastError());
        }

        // 4. samples
        if (pl.transform) {
            const dim3 grid_all((unsigned)((samples + 255u) / 256u), (unsigned)channels);
            ICER_LAUNCH(unsign_kernel, grid_all, 256, 0, d_planes, samples, (uint32_t)W, (uint32_t)dim_low(W, stages),
                        (uint32_t)dim_low(H, stages), pl.mean[0], pl.mean[1], pl.mean[2], sign_bit, bits);
            HIP_TRY(hipGetLastError());
            if (!pl.levels.empty()) {
                const FilterTaps taps = filter_taps(filt);
                HIP_TRY(hipMalloc(&d_tmp, sizeof(uint16_t) * total));
                HIP_TRY(hipMalloc(&d_pos, sizeof(uint32_t) * (W > H ? W : H)));
                std::vector<uint32_t> pos;
                for (const DecodeLevel &lv : pl.levels) {
                    // columns: d_planes -> d_tmp, rows: d_tmp -> d_planes (only the level's region is touched)
                    pos.resize(lv.ch);
                    interleave_positions(lv.ch, bits, pos.data());
                    HIP_TRY(hipMemcpy(d_pos, pos.data(), sizeof(uint32_t) * lv.ch, hipMemcpyHostToDevice));
                    ICER_LAUNCH(idwt_lines_kernel, dim3((lv.cw + 63u) / 64u, (unsigned)channels), 64, 0,
                                (const int16_t *)d_planes, (int16_t *)d_tmp, samples, (uint32_t)W, lv.cw, lv.ch, taps, bits, d_pos, false);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());            // (d_pos is reused for the rows)
                    pos.resize(lv.cw);
                    interleave_positions(lv.cw, bits, pos.data());
                    HIP_TRY(hipMemcpy(d_pos, pos.data(), sizeof(uint32_t) * lv.cw, hipMemcpyHostToDevice));
                    ICER_LAUNCH(idwt_lines_kernel, dim3((lv.ch + 63u) / 64u, (unsigned)channels), 64, 0,
                                (const int16_t *)d_tmp, (int16_t *)d_planes, samples, (uint32_t)W, lv.cw, lv.ch, taps, bits, d_pos, true);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());
                }
            }
        }

        // 5. results
        if (bits == 8) HIP_TRY(hipMalloc(&d_out8, total));
        if (pl.transform) {
            ICER_LAUNCH(clamp_kernel, (unsigned)((total + 255u) / 256u), 256, 0, d_planes, total, d_out8);
            HIP_TRY(hipGetLastError());
        } else if (bits == 8) {
            ICER_LAUNCH(narrow_kernel, (unsigned)((total + 255u) / 256u), 256, 0, d_planes, total, d_out8);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipDeviceSynchronize());
        for (int c = 0; c < channels; c++) {
            if (bits == 8) HIP_TRY(hipMemcpy(planes[c], d_out8 + (size_t)c * samples, samples, hipMemcpyDeviceToHost));
            else HIP_TRY(hipMemcpy(planes[c], d_planes + (size_t)c * samples, sizeof(uint16_t) * samples, hipMemcpyDeviceToHost));
        }
    }
done:
    for (void *p : {(void *)d_stream, (void *)d_crc, (void *)d_count, (void *)d_cands, (void *)d_chains, (void *)d_sub,
                    (void *)d_tables, (void *)d_planes, (void *)d_tmp, (void *)d_pos, (void *)d_out8})
        if (p) (void)hipFree(p);
    return rc;
}

}  // namespace

extern "C" {

const char *icerx_decoder_last_error(void) { return g_error.c_str(); }

int icer_get_image_dimensions(const uint8_t *datastream, size_t data_length, size_t *image_w, size_t *image_h)
{
    if (!datastream || !image_w || !image_h) return ICER_INVALID_INPUT;
    if (data_length > 0xFFFFFFFFull) return ICER_INVALID_INPUT;
    uint32_t crc_tab[256];
    build_crc32_table(crc_tab);
    for (uint32_t off = 0; off < (uint32_t)data_length; off++) {
        PacketCandidate c;
        if (!header_candidate(crc_tab, datastream, (uint32_t)data_length, off, &c)) continue;
        check_payload(crc_tab, datastream, &c);
        if (!c.payload_ok) continue;
        *image_w = load_le32(datastream + off + 8);
        *image_h = load_le32(datastream + off + 12);
        return ICER_RESULT_OK;
    }
    return ICER_DECODER_OUT_OF_DATA;
}

int icer_decompress_image_uint16(uint16_t *image, size_t *image_w, size_t *image_h, size_t image_bufsize,
                                 const uint8_t *datastream, size_t data_length, uint8_t stages,
                                 enum icer_filter_types filt, uint8_t segments)
{
    void *planes[1] = {image};
    return decompress_planes(planes, 1, image_w, image_h, image_bufsize, datastream, data_length, stages, (int)filt, segments, 16);
}

int icer_decompress_image_yuv_uint16(uint16_t *y_channel, uint16_t *u_channel, uint16_t *v_channel, size_t *image_w,
                                     size_t *image_h, size_t image_bufsize, const uint8_t *datastream,
                                     size_t data_length, uint8_t stages, enum icer_filter_types filt, uint8_t segments)
{
    void *planes[3] = {y_channel, u_channel, v_channel};
    return decompress_planes(planes, 3, image_w, image_h, image_bufsize, datastream, data_length, stages, (int)filt, segments, 16);
}

int icer_decompress_image_uint8(uint8_t *image, size_t *image_w, size_t *image_h, size_t image_bufsize,
                                const uint8_t *datastream, size_t data_length, uint8_t stages,
                                enum icer_filter_types filt, uint8_t segments)
{
    void *planes[1] = {image};
    return decompress_planes(planes, 1, image_w, image_h, image_bufsize, datastream, data_length, stages, (int)filt, segments, 8);
}

int icer_decompress_image_yuv_uint8(uint8_t *y_channel, uint8_t *u_channel, uint8_t *v_channel, size_t *image_w,
                                    size_t *image_h, size_t image_bufsize, const uint8_t *datastream,
                                    size_t data_length, uint8_t stages, enum icer_filter_types filt, uint8_t segments)
{
    void *planes[3] = {y_channel, u_channel, v_channel};
    return decompress_planes(planes, 3, image_w, image_h, image_bufsize, datastream, data_length, stages, (int)filt, segments, 8);
}

}  // extern "C"
