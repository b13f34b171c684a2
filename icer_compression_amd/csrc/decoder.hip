// decoder.hip -- C ABI of libicer_hip_dec.so (include/icer_hip_dec.h) and the host-side pipeline of the decoder, for
// a BATCH of streams (the lib_icer-shaped entry points are batches of one):
//   streams -> device | header candidates (one thread per byte offset, grid.y = stream) | payload CRCs (one thread per
//   candidate) | host: per stream the packet walk, table and chains (decoder_plan.hpp) | decode kernel over the
//   chains of all streams (one wavefront per chain; one thread per chain with ICER_DEC_WAVE=0) | sign-magnitude
//   removal + LL mean | inverse DWT, one level at a time, one thread per line, all frames of a geometry per launch
//   | clamp, narrow, copy back.
// First version: correctness before speed (HISTORY.md 6b (summary: DESIGN.md 8)); every loop is bounded by the stream / image size and no
// kernel waits on another thread.
#ifdef ICER_HOST_MOCK
#define ICER_LAUNCH_PLANES(kernel, grid, shmem, ...) ICER_LAUNCH_WAVE(kernel, grid, shmem, __VA_ARGS__)
#define ICER_LAUNCH_WAVE_ON(stream, kernel, grid, shmem, ...) ICER_LAUNCH_WAVE(kernel, grid, shmem, __VA_ARGS__)
#else
#include <hip/hip_runtime.h>
// kernel launches go through these two macros so that tests/emu/hip_mock.h (CPU, tests only) can stand in for them
#define ICER_LAUNCH(kernel, grid, block, shmem, ...) kernel<<<(grid), (block), (shmem)>>>(__VA_ARGS__)
#define ICER_LAUNCH_WAVE(kernel, grid, shmem, ...) kernel<<<(grid), 64, (shmem)>>>(__VA_ARGS__)
// (a workgroup of one wavefront per bit plane; the CPU mock runs the waves of a workgroup in turns inside one call)
#define ICER_LAUNCH_PLANES(kernel, grid, shmem, ...) kernel<<<(grid), 64 * kPwWaves, (shmem)>>>(__VA_ARGS__)
#define ICER_LAUNCH_WAVE_ON(stream, kernel, grid, shmem, ...) kernel<<<(grid), 64, (shmem), (stream)>>>(__VA_ARGS__)
#define ICER_DYNAMIC_LDS(T, name) extern __shared__ T name[]
// the decoder tables of a workgroup: a copy in LDS (every decision looks them up; from global memory each look-up is a
// chain of dependent loads)
#define ICER_LDS_TABLES(name, src)                                                                                   \
    __shared__ DecoderTables name;                                                                                   \
    for (uint32_t k_ = threadIdx.x; k_ < sizeof(DecoderTables) / 4u; k_ += blockDim.x)                               \
        reinterpret_cast<uint32_t *>(&name)[k_] = reinterpret_cast<const uint32_t *>(src)[k_];                       \
    __syncthreads()
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
#include "decoder_planes.hpp"
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
// what the kernels need to know about one stream / image of the batch
struct FrameInfo {
    uint32_t stream_off, stream_len;     // its bytes in the batch buffer
    uint32_t w, h, ll_w, ll_h;           // image and deepest-LL size
    uint16_t mean[4];
    uint32_t transform;                  // the run reaches sign-magnitude removal / mean / inverse DWT / clamping
};

__global__ void __launch_bounds__(256)
count_headers_kernel(const uint8_t *__restrict__ data, const FrameInfo *__restrict__ frames,
                     const uint32_t *__restrict__ crc_tab, PacketCandidate *__restrict__ out, uint32_t cap,
                     uint32_t *__restrict__ count)
{
    const FrameInfo f = frames[blockIdx.y];
    const uint32_t off = blockIdx.x * blockDim.x + threadIdx.x;
    if (off >= f.stream_len) return;
    PacketCandidate c;
    if (!header_candidate(crc_tab, data + f.stream_off, f.stream_len, off, &c)) return;
    c.frame = blockIdx.y;
    const uint32_t at = atomicAdd(count, 1u);
    if (at < cap) out[at] = c;
}

// payload CRCs: 64 threads per candidate, each a contiguous piece (payload_piece_crc), XOR-combined in the candidate's
// crc_acc; the host compares it with the header's field.  grid = candidates, block = 64.
__global__ void __launch_bounds__(64)
check_payloads_kernel(const uint8_t *__restrict__ data, const FrameInfo *__restrict__ frames,
                      const uint32_t *__restrict__ crc_tab, PacketCandidate *__restrict__ cands, uint32_t n)
{
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const uint32_t v = payload_piece_crc(crc_tab, data + frames[cands[i].frame].stream_off, cands[i], threadIdx.x, 64u);
    if (v) atomicXor(&cands[i].crc_acc, v);
}

// plane of channel `chan` of frame `frame`: planes + (frame * channels + chan) * frame_stride, rows of the frame's width
__global__ void __launch_bounds__(64)
decode_chains_kernel(uint16_t *__restrict__ planes, size_t frame_stride, int channels,
                     const ChainDesc *__restrict__ chains, uint32_t n, const uint8_t *__restrict__ data,
                     const FrameInfo *__restrict__ frames, const DecoderTables *__restrict__ tables, int nplanes, int sign_bit)
{
    ICER_DYNAMIC_LDS(uint8_t, state);                     // plane_block_bytes(64): the threads' per-bin arrays
    ICER_LDS_TABLES(lt, tables);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ChainDesc c = chains[i];
    const FrameInfo f = frames[c.frame];
    PlaneDecoder job;
    plane_attach_columns(job, state, 64u, threadIdx.x);
    decode_chain(job, planes + ((size_t)c.frame * channels + c.chan) * frame_stride, f.w, c, (int)c.subband, data + f.stream_off,
                 f.stream_len, lt, nplanes, sign_bit);
}

// the same with one wavefront per chain and one lane per packet (decoder_wave.hpp); dynamic LDS = the row ring
__global__ void __launch_bounds__(64)
decode_chains_wave_kernel(uint16_t *__restrict__ planes, size_t frame_stride, int channels,
                          const ChainDesc *__restrict__ chains, const uint8_t *__restrict__ data,
                          const FrameInfo *__restrict__ frames, const DecoderTables *__restrict__ tables, int nplanes,
                          int sign_bit)
{
    ICER_DYNAMIC_LDS(uint8_t, lds);                       // the lanes' per-bin arrays, then the chain's row ring
    ICER_LDS_TABLES(lt, tables);
    const ChainDesc c = chains[blockIdx.x];
    const FrameInfo f = frames[c.frame];
    uint16_t *ring = reinterpret_cast<uint16_t *>(lds + kStateBytes);
    decode_chain_wave(ring, planes + ((size_t)c.frame * channels + c.chan) * frame_stride, f.w, c, (int)c.subband,
                      data + f.stream_off, f.stream_len, lt, nplanes, sign_bit, nullptr, lds);
}

// one wavefront per bit plane, wave-uniform decisions (decoder_planes.hpp): grid = chains that take the fast entropy path,
// block = 64 * kPwWaves, dynamic LDS = the widest chain's control words + row ring
__global__ void __launch_bounds__(64 * kPwWaves)
decode_chains_planes_kernel(uint16_t *__restrict__ planes, size_t frame_stride, int channels,
                            const ChainDesc *__restrict__ chains, const uint8_t *__restrict__ data,
                            const FrameInfo *__restrict__ frames, const DecoderTables *__restrict__ tables, int nplanes,
                            int sign_bit, uint32_t *__restrict__ err)
{
    ICER_DYNAMIC_LDS(uint8_t, lds);
    const ChainDesc &c = chains[blockIdx.x];                  // (read in place: see pw_run_chain)
    const FrameInfo f = frames[c.frame];
#ifndef ICER_HOST_MOCK
    {   // control words, zero row and ring start out zero
        uint32_t *w = reinterpret_cast<uint32_t *>(lds);
        const uint32_t words = (uint32_t)((pw_lds_bytes(c.w, nplanes) + 3u) / 4u);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) w[i] = 0;
        __syncthreads();
    }
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const uint32_t wave = 0;
#endif
    (void)pw_run_chain(lds, wave, c, nplanes, sign_bit, planes + ((size_t)c.frame * channels + c.chan) * frame_stride, f.w,
                       data + f.stream_off, f.stream_len, tables, err);
}

// One level of the inverse transform for filters without a recurrence along the line (beta = 0 and alpha_-1 = 0: filter A --
// every restored high then depends on the stored lows and highs alone, idwt_line's `dn` and filter-C terms drop out): one
// thread per output PAIR instead of one per line.  `rows` = false: lines are columns (x = line, y = pair), true: lines are
// rows.  grid = (ceil(cw / 64), ceil(pairs-or-lines / 4), frames * channels), block = (64, 4).
__global__ void __launch_bounds__(256)
idwt_pairs_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, size_t frame_stride, int channels,
                  const uint32_t *__restrict__ list, uint32_t image_w, uint32_t cw, uint32_t ch_rows, FilterTaps f,
                  int bits, const uint32_t *__restrict__ pos_of, bool rows)
{
    const uint32_t gx = blockIdx.x * 64u + (threadIdx.x & 63u), gy = blockIdx.y * 4u + (threadIdx.x >> 6);
    const size_t base = ((size_t)list[blockIdx.z / (unsigned)channels] * channels + blockIdx.z % (unsigned)channels) * frame_stride;
    // (x runs along a row of the image in both passes, so that a wavefront's accesses are contiguous)
    const uint32_t n = rows ? cw : ch_rows, nl = (n + 1u) / 2u, nh = n / 2u;
    const uint32_t line = rows ? gy : gx, k = rows ? gx : gy;
    if (line >= (rows ? ch_rows : cw) || k >= nl) return;
    const size_t stride = rows ? 1 : image_w;
    const int16_t *s = src + base + (rows ? (size_t)line * image_w : (size_t)line);
    int16_t *d = dst + base + (rows ? (size_t)line * image_w : (size_t)line);
    const bool odd = (n & 1u) != 0;
#define LO(i) ((int32_t)s[(size_t)(i) * stride])
#define HI(i) ((int32_t)s[(size_t)(nl + (i)) * stride])
#define RR(i) ((int32_t)(int16_t)(LO((i) - 1) - LO(i)))
#define TR(v) (bits == 8 ? (int16_t)(int8_t)(v) : (int16_t)(v))
    if (k >= nh) { d[(size_t)pos_of[nl - 1u] * stride] = (int16_t)LO(nl - 1u); return; }       // (odd length: the last low)
    int32_t add;
    if (k == 0) add = dec_floordiv(RR(1), 4);
    else if (!odd && k == nh - 1u) add = dec_floordiv(RR(nh - 1u), 4);
    else add = dec_floordiv(f.a0 * RR(k) + f.a1 * RR(k + 1u) + 8, 16);
    const int32_t hi = TR(HI(k) + add);
    const int32_t a = LO(k) + dec_floordiv(hi + 1, 2);
    d[(size_t)pos_of[k] * stride] = TR(a);
    d[(size_t)pos_of[nl + k] * stride] = TR(a - hi);
#undef LO
#undef HI
#undef RR
#undef TR
}

// sign-magnitude words -> int16, LL mean back in (grid.y = frame * channels + channel)
__global__ void __launch_bounds__(256)
unsign_kernel(uint16_t *__restrict__ planes, size_t frame_stride, int channels, const FrameInfo *__restrict__ frames,
              int sign_bit, int bits)
{
    const FrameInfo f = frames[blockIdx.y / (unsigned)channels];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!f.transform || i >= (size_t)f.w * f.h) return;
    uint16_t *p = planes + (size_t)blockIdx.y * frame_stride;
    int16_t s = from_sign_magnitude(p[i], sign_bit);
    const uint32_t x = (uint32_t)(i % f.w), y = (uint32_t)(i / f.w);
    if (x < f.ll_w && y < f.ll_h) s = add_ll_mean(s, f.mean[blockIdx.y % (unsigned)channels], bits);
    p[i] = (uint16_t)s;
}

// one thread per column (rows = false) or per row (rows = true) of a level's region; grid.y walks the channels of the
// frames in `list` (all of the same size)
__global__ void __launch_bounds__(64)
idwt_lines_kernel(const int16_t *__restrict__ src, int16_t *__restrict__ dst, size_t frame_stride, int channels,
                  const uint32_t *__restrict__ list, uint32_t image_w, uint32_t cw, uint32_t ch_rows, FilterTaps taps,
                  int bits, const uint32_t *__restrict__ pos_of, bool rows)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t base = ((size_t)list[blockIdx.y / (unsigned)channels] * channels + blockIdx.y % (unsigned)channels) * frame_stride;
    if (rows) {
        if (i < ch_rows) idwt_line(src + base + (size_t)i * image_w, dst + base + (size_t)i * image_w, cw, 1, taps, bits, pos_of);
    } else {
        if (i < cw) idwt_line(src + base + i, dst + base + i, ch_rows, image_w, taps, bits, pos_of);
    }
}

// icer_remove_negative_* (icer_util.c:70-91) for the frames that were transformed; `out8` != null: narrow to bytes
// (every frame: a frame that stopped early keeps its sign-magnitude words, narrowed like the reference's uint8 planes)
__global__ void __launch_bounds__(256)
finish_kernel(uint16_t *__restrict__ planes, size_t frame_stride, int channels, const FrameInfo *__restrict__ frames,
              uint8_t *__restrict__ out8)
{
    const FrameInfo f = frames[blockIdx.y / (unsigned)channels];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)f.w * f.h) return;
    const size_t at = (size_t)blockIdx.y * frame_stride + i;
    int16_t s = (int16_t)planes[at];
    if (f.transform && s < 0) s = 0;
    planes[at] = (uint16_t)s;
    if (out8) out8[at] = (uint8_t)s;
}

}  // namespace

// ------------------------------------------------------------------------------------------ decoder object
// The side streams carry the kernels of the ring size classes side by side with the plane kernel on the call's own stream.  The HIP runtime
// shares GPU_MAX_HW_QUEUES (4 by default) hardware queues PER PRIORITY LEVEL out over the live streams of the process, and streams on one
// queue take turns: unless the process has asked for 6 queues or more, the side streams are created at the low priority level -- a pool of
// their own, whatever else the process has alive (the encoder library does the same, api.hip want_priority_streams).  In a quiet process
// the A/B shows no difference (64 streams per call 886-915 Mpix/s with plain or low-level side streams at 8 / 4 / default queues,
// profiles/r05_logs/r05_u.log; one earlier run with plain streams on 4 queues had given 782): it is there for the crowded process.
// ICER_HIP_STREAM_PRIO=0|1 pins the choice.
#ifndef ICER_HOST_MOCK
static hipError_t create_side_stream(hipStream_t *st)
{
    bool level = true;
    if (const char *pv = getenv("ICER_HIP_STREAM_PRIO")) level = atoi(pv) != 0;
    else if (const char *q = getenv("GPU_MAX_HW_QUEUES")) level = atoi(q) < 6;
    int least = 0, greatest = 0;
    if (level && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least &&
        hipStreamCreateWithPriority(st, hipStreamNonBlocking, least) == hipSuccess)
        return hipSuccess;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
#endif

struct icerx_decoder {
    int device = 0, channels = 1, stages = 1, filt = 0, bits = 16;
    unsigned segments = 1;
    DecoderTables tables;
    uint32_t crc_tab[256];
    // device buffers, grown on demand and kept
    struct Buf { void *p = nullptr; size_t cap = 0; };
    Buf data, frames, crc, dtables, count, cands, chains, work, tmp, out8, pos, list, err;
    int n_cus = 256;                   // compute units of the device
    bool planes_lds_raised = false;    // decode_chains_planes_kernel has been granted more than 64 KiB of dynamic LDS
    int is_gfx950 = -1;                // (-1: not asked yet)
    size_t planes_lds_fallback = 0;    // ... or, refused the hardware's figure, this much (what the runtime reports)
    bool planes_lds_refused = false;   // ... or the runtime refused: chains that need more than 48 KiB go to the lane-per-plane kernel from then on
    // the lane-per-plane kernel is launched once per size class of row ring, side by side (decode_batch)
    static constexpr int kRingClasses = 4;
    hipStream_t side[kRingClasses] = {};
    bool side_ok = false;
};

namespace {

hipError_t ensure(icerx_decoder::Buf &b, size_t bytes)
{
    if (bytes <= b.cap) return hipSuccess;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    const hipError_t e = hipMalloc(&b.p, bytes);
    if (e == hipSuccess) b.cap = bytes;
    return e;
}

// n streams: stream k = bytes [offsets[k], offsets[k] + lens[k]) of `data` (host memory, or device memory when
// data_on_device).  Results: rcs[k], ws[k] / hs[k] (in: the values kept when stream k holds no valid packet).  Frame k's
// channel c goes to host_out[k * channels + c] (host pointers, >= frame_stride samples each) or, when host_out is null, to
// dev_out + (k * channels + c) * frame_stride samples (device memory; uint16 or uint8 samples by sample_bits).
int decode_batch(icerx_decoder *d, int n, const uint8_t *data, bool data_on_device, const size_t *offsets, const size_t *lens,
                 void *const *host_out, void *dev_out, size_t frame_stride, int *rcs, size_t *ws, size_t *hs)
{
    g_error.clear();
    if (!d || n < 0 || (n && (!offsets || !lens || !rcs || !ws || !hs)) || (!host_out && !dev_out && n)) return ICER_INVALID_INPUT;
    if (n == 0) return ICER_RESULT_OK;
    int rc = ICER_RESULT_OK;
    const int channels = d->channels, bits = d->bits, stages = d->stages;
    const int nplanes = bits == 8 ? kPlanes8 : kPlanes, sign_bit = bits == 8 ? 7 : 15;
    size_t total_len = 0, max_len = 0;
    for (int k = 0; k < n; k++) {
        if (lens[k] && !data) return ICER_INVALID_INPUT;
        total_len = std::max(total_len, offsets[k] + lens[k]);
        max_len = std::max(max_len, lens[k]);
    }
    if (total_len >= 0xFFFFFFFFull - 64u) return fail("batch of %zu stream bytes: 32-bit offsets only", total_len);
    const size_t planes_total = (size_t)n * channels * frame_stride;
    if (frame_stride > 0xFFFFFFFFull) return fail("frames of %zu samples: 32-bit indices only", frame_stride);

    std::vector<FrameInfo> frames((size_t)n);
    std::vector<PacketCandidate> cands;
    std::vector<DecodePlan> plans((size_t)n);
    std::vector<ChainDesc> chains;
    const uint8_t *d_data = nullptr;
    uint16_t *d_planes = nullptr;
    uint8_t *d_out8 = nullptr;
    FrameInfo *d_frames = nullptr;

    for (int k = 0; k < n; k++) {
        memset(&frames[k], 0, sizeof(FrameInfo));
        frames[k].stream_off = (uint32_t)offsets[k];
        frames[k].stream_len = (uint32_t)lens[k];
    }
    HIP_TRY(ensure(d->frames, sizeof(FrameInfo) * n));
    d_frames = (FrameInfo *)d->frames.p;
    HIP_TRY(hipMemcpy(d_frames, frames.data(), sizeof(FrameInfo) * n, hipMemcpyHostToDevice));

    // 1. packets
    if (total_len) {
        if (data_on_device) d_data = data;
        else {
            HIP_TRY(ensure(d->data, total_len));
            HIP_TRY(hipMemcpy(d->data.p, data, total_len, hipMemcpyHostToDevice));
            d_data = (const uint8_t *)d->data.p;
        }
        HIP_TRY(ensure(d->count, sizeof(uint32_t)));
        uint32_t cap = (uint32_t)(total_len / 64u) + 1024u, count = 0;
        for (int attempt = 0; attempt < 2 && max_len; attempt++) {
            HIP_TRY(ensure(d->cands, sizeof(PacketCandidate) * cap));
            HIP_TRY(hipMemset(d->count.p, 0, sizeof(uint32_t)));
            ICER_LAUNCH(count_headers_kernel, dim3((unsigned)((max_len + 255u) / 256u), (unsigned)n), 256, 0, d_data, d_frames,
                        (const uint32_t *)d->crc.p, (PacketCandidate *)d->cands.p, cap, (uint32_t *)d->count.p);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpy(&count, d->count.p, sizeof count, hipMemcpyDeviceToHost));
            if (count <= cap) break;
            cap = count;                                    // (more header look-alikes than expected: once more, all of them)
        }
        if (count) {
            ICER_LAUNCH(check_payloads_kernel, count, 64, 0, d_data, d_frames, (const uint32_t *)d->crc.p,
                        (PacketCandidate *)d->cands.p, count);
            HIP_TRY(hipGetLastError());
            cands.resize(count);
            HIP_TRY(hipMemcpy(cands.data(), d->cands.p, sizeof(PacketCandidate) * count, hipMemcpyDeviceToHost));
            for (PacketCandidate &c : cands) c.payload_ok = (c.fits && c.crc_acc == load_le32(c.hdr + 20)) ? 1u : 0u;   // (icer_compress.c:576-577)
            std::sort(cands.begin(), cands.end(), [](const PacketCandidate &a, const PacketCandidate &b) {
                return a.frame != b.frame ? a.frame < b.frame : a.off < b.off;
            });
        }
    }

    // 2. plans
    {
        size_t at = 0;
        std::vector<PacketCandidate> mine;
        for (int k = 0; k < n; k++) {
            mine.clear();
            while (at < cands.size() && cands[at].frame == (uint32_t)k) mine.push_back(cands[at++]);
            DecodePlan &pl = plans[k];
            plan_decode(&pl, mine, channels, stages, d->segments, bits, ws[k], hs[k], frame_stride);
            ws[k] = pl.w; hs[k] = pl.h; rcs[k] = pl.rc;
            const bool runs = !(pl.rc == kInvalidInput || pl.rc == kTooManyStages || pl.rc == kByteQuotaExceeded) && pl.w * pl.h > 0;
            frames[k].w = runs ? (uint32_t)pl.w : 0u;
            frames[k].h = runs ? (uint32_t)pl.h : 0u;
            frames[k].ll_w = (uint32_t)dim_low(pl.w, stages); frames[k].ll_h = (uint32_t)dim_low(pl.h, stages);
            for (int c = 0; c < 3; c++) frames[k].mean[c] = pl.mean[c];
            frames[k].transform = (runs && pl.transform) ? 1u : 0u;
            if (!runs) continue;
            for (ChainDesc c : pl.chains) { c.frame = (uint32_t)k; chains.push_back(c); }
        }
    }
    HIP_TRY(hipMemcpy(d_frames, frames.data(), sizeof(FrameInfo) * n, hipMemcpyHostToDevice));

    // 3. bit planes: into the caller's device buffer when that holds uint16 samples, else into a work buffer
    if (planes_total == 0) goto done;
    if (dev_out && !host_out && bits == 16) d_planes = (uint16_t *)dev_out;
    else { HIP_TRY(ensure(d->work, sizeof(uint16_t) * planes_total)); d_planes = (uint16_t *)d->work.p; }
    // (only the w * h samples of a frame are defined results; the rest of its slot is scratch)
    HIP_TRY(hipMemset(d_planes, 0, sizeof(uint16_t) * planes_total));
    if (!chains.empty()) {
        // Which kernel decodes a chain (ICER_DEC_WAVE: 0 = one thread per chain, 1 = one wavefront per chain with a lane per
        // bit plane, anything else / unset = one wavefront per bit plane wherever it applies):
        //   wave-per-plane   chains whose packets all take the fast entropy path (ChainDesc::fast) and whose row ring fits LDS
        //   the rest         the lane-per-plane kernel if ITS ring fits, else one thread per chain
        // Unset, the choice goes by load.  A wave per plane is the faster DECISION (303 against 634 ms for the 160 chains of
        // the 4096 x 4096 headline stream), but a compute unit's throughput with it saturates near ONE long chain (what the
        // waves of several chains share is the compute unit's scalar unit: 0.6-0.75 scalar instructions per cycle at 8-16
        // streams, profiles/r04_logs/r04_l_planes_sq_counters.log; the wait loops are not it, r04_m), while the lane-per-plane
        // kernel keeps seven long chains per compute unit going (LDS for their row rings), each at half the speed.  With the
        // chains in longest-first order (below; profiles/r04_logs/r04_k_*.json, r04_m_*.json), streams per call -> Mpix/s:
        //   wave per plane   4: 220   8: 437   16: 525-586   32: 537   64: 519
        //   lane per plane   4: 107            16: 435       32: 922   64: 885-921   128: 1036
        // -- the cross-over lies near 20 streams = 12 chains per compute unit (24 streams by this rule: 686).  ICER_DEC_WAVE=2
        // pins the wave-per-plane kernel, ICER_DEC_PLANES_PER_CU moves the cross-over.
        const char *mode = getenv("ICER_DEC_WAVE");
        size_t n_eligible = 0;
        for (const ChainDesc &c : chains) n_eligible += c.fast ? 1u : 0u;
        const bool by_load = !mode || !mode[0];
        size_t per_cu = 12u;
        if (const char *pc = getenv("ICER_DEC_PLANES_PER_CU")) { const long v = atol(pc); if (v >= 1 && v <= 1000000) per_cu = (size_t)v; }
        const bool want_planes = !(mode && (mode[0] == '0' || mode[0] == '1')) && d->tables.lut_ok != 0u &&
                                 (!by_load || n_eligible <= per_cu * (size_t)d->n_cus);
        size_t planes_lds = 0;
#ifdef ICER_HOST_MOCK
        const size_t planes_lds_limit = (size_t)1 << 20;
#else
        // what a workgroup of this device may take (gfx950: 160 KiB), less 10 KiB for the kernel's static block and the runtime
        size_t planes_lds_limit = 0;
        {
            int max_lds = 0, dev_id = d->device;
            if (dev_id < 0 && hipGetDevice(&dev_id) != hipSuccess) { (void)hipGetLastError(); dev_id = 0; }      // (-1: the caller's current device)
            if (hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_id) != hipSuccess) { (void)hipGetLastError(); max_lds = 64 * 1024; }
            // (this runtime reports 64 KiB for gfx950, whose compute units have 160 KiB and grant a workgroup what hipFuncSetAttribute asks for --
            // the encoder's window coder runs on 115 KiB --: ask for the hardware's figure first, the reported one is the fall-back)
            if (d->is_gfx950 < 0) {
                hipDeviceProp_t prop;
                d->is_gfx950 = (hipGetDeviceProperties(&prop, dev_id) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ? 1 : 0;
                (void)hipGetLastError();
            }
            const bool gfx950 = d->is_gfx950 == 1;
            if (gfx950 && max_lds < 160 * 1024 && !d->planes_lds_refused) max_lds = 160 * 1024;
            planes_lds_limit = max_lds > 10 * 1024 ? (size_t)max_lds - 10u * 1024u : 0u;
            // more than 48 KiB of dynamic LDS has to be granted; a runtime / device that refuses loses nothing but the fast kernel
            // for the chains that need it (they go to decode_chains_wave_kernel below)
            if (planes_lds_limit > 48u * 1024u && !d->planes_lds_raised && !d->planes_lds_refused) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(decode_chains_planes_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)planes_lds_limit) == hipSuccess)
                    d->planes_lds_raised = true;
                else {
                    (void)hipGetLastError(); d->planes_lds_refused = true;
                    int rep = 0;
                    if (hipDeviceGetAttribute(&rep, hipDeviceAttributeMaxSharedMemoryPerBlock, dev_id) != hipSuccess) { (void)hipGetLastError(); rep = 64 * 1024; }
                    planes_lds_limit = rep > 10 * 1024 ? (size_t)rep - 10u * 1024u : 0u;
                    if (planes_lds_limit > 48u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void *>(decode_chains_planes_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)planes_lds_limit) == hipSuccess)
                        d->planes_lds_fallback = planes_lds_limit;
                    else (void)hipGetLastError();
                }
                // (said once per decoder: on a runtime that reports or grants less than gfx950's 160 KiB the fast planes kernel quietly loses
                // the chains that need more -- correct, slower)
                if (d->planes_lds_refused || planes_lds_limit < 150u * 1024u)
                    fprintf(stderr, "libicer_hip_dec: %zu KiB of LDS per workgroup for the planes kernel (%s); chains that need more take the wave kernel\n",
                            (d->planes_lds_refused ? (d->planes_lds_fallback ? d->planes_lds_fallback : (size_t)48u * 1024u) : planes_lds_limit) / 1024u,
                            d->planes_lds_refused ? "the runtime refused the hardware's 150 KiB" : "what the device reports, less 10 KiB");
            }
            if (d->planes_lds_refused) planes_lds_limit = d->planes_lds_fallback ? d->planes_lds_fallback : std::min(planes_lds_limit, (size_t)48u * 1024u);
        }
#endif
        std::stable_partition(chains.begin(), chains.end(), [&](const ChainDesc &c) {
            return want_planes && c.fast && frames[c.frame].stream_len >= 4u && pw_lds_bytes(c.w, nplanes) <= planes_lds_limit; });
        uint32_t n_fast = 0;
        for (const ChainDesc &c : chains) {
            if (!(want_planes && c.fast && frames[c.frame].stream_len >= 4u && pw_lds_bytes(c.w, nplanes) <= planes_lds_limit)) break;
            planes_lds = std::max(planes_lds, pw_lds_bytes(c.w, nplanes));
            n_fast++;
        }
        const uint32_t nc = (uint32_t)chains.size();
        // Longest chains first within either kernel's share (a chain's time goes with its samples): the level-1 chains of
        // every stream start at once, one or two per compute unit, and the short ones fill the gaps -- in stream order the
        // long chains of the later streams of a batch started when those of the first were half-way.
        {
            const char *ord = getenv("ICER_DEC_ORDER");
            if (!(ord && ord[0] == '0')) {
                auto longer = [](const ChainDesc &a, const ChainDesc &b) { return (uint32_t)a.w * a.h > (uint32_t)b.w * b.h; };
                std::stable_sort(chains.begin(), chains.begin() + n_fast, longer);
                std::stable_sort(chains.begin() + n_fast, chains.end(), longer);
            }
        }
        // The lane-per-plane kernel's share by size class of row ring (a launch reserves the LDS of its widest chain for every
        // workgroup, and LDS is what bounds the chains a compute unit holds: 7 of the headline stream's level-1 chains, but 14
        // of level 2 and 28 of level 3): class k = rings of at most limit >> k bytes, widest class first, each class a launch
        // of its own on a stream of its own, all of them side by side.
        uint32_t class_end[icerx_decoder::kRingClasses];
        {
            size_t widest = 2;
            for (uint32_t i = n_fast; i < nc; i++) widest = std::max(widest, ring_elems_for(chains[i].w, nplanes));
            auto cls = [&](const ChainDesc &c) {
                const size_t e = ring_elems_for(c.w, nplanes);
                int k = 0;
                while (k + 1 < icerx_decoder::kRingClasses && e * 2u <= (widest >> k)) k++;
                return k;
            };
            std::stable_sort(chains.begin() + n_fast, chains.end(), [&](const ChainDesc &a, const ChainDesc &b) { return cls(a) < cls(b); });
            uint32_t at = n_fast;
            for (int k = 0; k < icerx_decoder::kRingClasses; k++) {
                while (at < nc && cls(chains[at]) == k) at++;
                class_end[k] = at;
            }
        }
        HIP_TRY(ensure(d->chains, sizeof(ChainDesc) * nc));
        HIP_TRY(hipMemcpy(d->chains.p, chains.data(), sizeof(ChainDesc) * nc, hipMemcpyHostToDevice));
        if (n_fast) {
            HIP_TRY(ensure(d->err, sizeof(uint32_t)));
            HIP_TRY(hipMemset(d->err.p, 0, sizeof(uint32_t)));
            ICER_LAUNCH_PLANES(decode_chains_planes_kernel, n_fast, planes_lds, d_planes, frame_stride, channels, (const ChainDesc *)d->chains.p,
                               d_data, d_frames, (const DecoderTables *)d->dtables.p, nplanes, sign_bit, (uint32_t *)d->err.p);
            HIP_TRY(hipGetLastError());
        }
        if (n_fast < nc) {
            const ChainDesc *rest = (const ChainDesc *)d->chains.p + n_fast;
            const uint32_t nr = nc - n_fast;
            // the planes of a segment side by side (one wavefront per chain) if the chain's row ring fits LDS
            size_t ring_elems = 2;
            for (uint32_t i = n_fast; i < nc; i++) ring_elems = std::max(ring_elems, ring_elems_for(chains[i].w, nplanes));
            const size_t ring_bytes = ring_elems * sizeof(uint16_t) + kStateBytes;
            // (the kernel's static DecoderTables block counts against the same 64 KiB a launch gets without asking)
            if (!(mode && mode[0] == '0') && ring_bytes + sizeof(DecoderTables) + 256u <= 65536u) {
#ifndef ICER_HOST_MOCK
                if (!d->side_ok) {
                    bool made = true;
                    for (hipStream_t &st : d->side) made = made && create_side_stream(&st) == hipSuccess;
                    if (!made) {                                    // (none is kept half-made: the next call tries again)
                        for (hipStream_t &st : d->side) { if (st) (void)hipStreamDestroy(st); st = nullptr; }
                        (void)hipGetLastError();
                        rc = fail("the decoder could not create its side streams");
                        goto done;
                    }
                    d->side_ok = true;
                }
#endif
                // (everything the null stream did so far is complete: the upload of `chains` above was a blocking copy)
                uint32_t from = n_fast;
                for (int k = 0; k < icerx_decoder::kRingClasses; k++) {
                    const uint32_t to = class_end[k];
                    if (to == from) continue;
                    size_t elems = 2;
                    for (uint32_t i = from; i < to; i++) elems = std::max(elems, ring_elems_for(chains[i].w, nplanes));
                    ICER_LAUNCH_WAVE_ON(d->side[k], decode_chains_wave_kernel, to - from, elems * sizeof(uint16_t) + kStateBytes, d_planes, frame_stride, channels,
                                        (const ChainDesc *)d->chains.p + from, d_data, d_frames, (const DecoderTables *)d->dtables.p, nplanes, sign_bit);
                    HIP_TRY(hipGetLastError());
                    from = to;
                }
#ifndef ICER_HOST_MOCK
                for (hipStream_t st : d->side) HIP_TRY(hipStreamSynchronize(st));
#endif
            } else {
                ICER_LAUNCH(decode_chains_kernel, (nr + 63u) / 64u, 64, plane_block_bytes(64u), d_planes, frame_stride, channels, rest,
                            nr, d_data, d_frames, (const DecoderTables *)d->dtables.p, nplanes, sign_bit);
            }
            HIP_TRY(hipGetLastError());
        }
        if (n_fast) {
            uint32_t err = 0;
            HIP_TRY(hipMemcpy(&err, d->err.p, sizeof err, hipMemcpyDeviceToHost));
            if (err) { rc = fail("a bit-plane wave of the decoder waited longer than its spin bound (internal error)"); goto done; }
        }
    }

    // 4. samples
    {
        const dim3 grid_all((unsigned)((frame_stride + 255u) / 256u), (unsigned)(n * channels));
        ICER_LAUNCH(unsign_kernel, grid_all, 256, 0, d_planes, frame_stride, channels, d_frames, sign_bit, bits);
        HIP_TRY(hipGetLastError());
        // inverse DWT: the frames of one size together
        std::vector<char> done((size_t)n, 0);
        const FilterTaps taps = filter_taps(d->filt);
        for (int k = 0; k < n; k++) {
            if (done[k] || !frames[k].transform || plans[k].levels.empty()) continue;
            std::vector<uint32_t> list, pos, tab;
            for (int j = k; j < n; j++)
                if (!done[j] && frames[j].transform && frames[j].w == frames[k].w && frames[j].h == frames[k].h) { list.push_back((uint32_t)j); done[j] = 1; }
            const uint32_t W = frames[k].w;
            std::vector<size_t> col_at, row_at;              // where each level's position tables start
            for (const DecodeLevel &lv : plans[k].levels) {
                col_at.push_back(tab.size());
                pos.resize(lv.ch); interleave_positions(lv.ch, bits, pos.data()); tab.insert(tab.end(), pos.begin(), pos.end());
                row_at.push_back(tab.size());
                pos.resize(lv.cw); interleave_positions(lv.cw, bits, pos.data()); tab.insert(tab.end(), pos.begin(), pos.end());
            }
            HIP_TRY(ensure(d->tmp, sizeof(uint16_t) * planes_total));
            HIP_TRY(ensure(d->pos, sizeof(uint32_t) * tab.size()));
            HIP_TRY(hipMemcpy(d->pos.p, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice));
            HIP_TRY(ensure(d->list, sizeof(uint32_t) * list.size()));
            HIP_TRY(hipMemcpy(d->list.p, list.data(), sizeof(uint32_t) * list.size(), hipMemcpyHostToDevice));
            const unsigned gy = (unsigned)(list.size() * (size_t)channels);
            const bool pairwise = taps.be == 0 && taps.am1 == 0;      // (filter A: no recurrence along a line, idwt_pairs_kernel)
            for (size_t li = 0; li < plans[k].levels.size(); li++) {
                const DecodeLevel &lv = plans[k].levels[li];
                if (pairwise) {
                    ICER_LAUNCH(idwt_pairs_kernel, dim3((lv.cw + 63u) / 64u, ((lv.ch + 1u) / 2u + 3u) / 4u, gy), dim3(256), 0, (const int16_t *)d_planes,
                                (int16_t *)d->tmp.p, frame_stride, channels, (const uint32_t *)d->list.p, W, lv.cw, lv.ch, taps, bits,
                                (const uint32_t *)d->pos.p + col_at[li], false);
                    HIP_TRY(hipGetLastError());
                    ICER_LAUNCH(idwt_pairs_kernel, dim3(((lv.cw + 1u) / 2u + 63u) / 64u, (lv.ch + 3u) / 4u, gy), dim3(256), 0, (const int16_t *)d->tmp.p,
                                (int16_t *)d_planes, frame_stride, channels, (const uint32_t *)d->list.p, W, lv.cw, lv.ch, taps, bits,
                                (const uint32_t *)d->pos.p + row_at[li], true);
                    HIP_TRY(hipGetLastError());
                    continue;
                }
                // columns: planes -> tmp, rows: tmp -> planes (only the level's region is touched)
                ICER_LAUNCH(idwt_lines_kernel, dim3((lv.cw + 63u) / 64u, gy), 64, 0, (const int16_t *)d_planes, (int16_t *)d->tmp.p,
                            frame_stride, channels, (const uint32_t *)d->list.p, W, lv.cw, lv.ch, taps, bits,
                            (const uint32_t *)d->pos.p + col_at[li], false);
                HIP_TRY(hipGetLastError());
                ICER_LAUNCH(idwt_lines_kernel, dim3((lv.ch + 63u) / 64u, gy), 64, 0, (const int16_t *)d->tmp.p, (int16_t *)d_planes,
                            frame_stride, channels, (const uint32_t *)d->list.p, W, lv.cw, lv.ch, taps, bits,
                            (const uint32_t *)d->pos.p + row_at[li], true);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipDeviceSynchronize());                 // (the position tables and the list are reused by the next size)
        }

        // 5. results
        if (bits == 8) {
            if (dev_out && !host_out) d_out8 = (uint8_t *)dev_out;
            else { HIP_TRY(ensure(d->out8, planes_total)); d_out8 = (uint8_t *)d->out8.p; }
        }
        ICER_LAUNCH(finish_kernel, grid_all, 256, 0, d_planes, frame_stride, channels, d_frames, d_out8);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if (host_out)
            for (int k = 0; k < n; k++) {
                const size_t samples = (size_t)frames[k].w * frames[k].h;
                for (int c = 0; c < channels && samples; c++) {
                    const size_t at = ((size_t)k * channels + c) * frame_stride;
                    if (bits == 8) HIP_TRY(hipMemcpy(host_out[k * channels + c], d_out8 + at, samples, hipMemcpyDeviceToHost));
                    else HIP_TRY(hipMemcpy(host_out[k * channels + c], d_planes + at, sizeof(uint16_t) * samples, hipMemcpyDeviceToHost));
                }
            }
    }
done:
#ifndef ICER_HOST_MOCK
    // (an error exit may leave launches of the size classes behind on the side streams: nothing of this call runs on after it)
    if (rc != ICER_RESULT_OK && d->side_ok) { for (hipStream_t st : d->side) (void)hipStreamSynchronize(st); (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
#endif
    return rc;
}

int decompress_planes(void *const planes[], int channels, size_t *image_w, size_t *image_h, size_t bufsize,
                      const uint8_t *data, size_t data_length, int stages, int filt, unsigned segments, int bits)
{
    if (!image_w || !image_h || (!data && data_length)) return ICER_INVALID_INPUT;
    for (int c = 0; c < channels; c++)
        if (!planes[c]) return ICER_INVALID_INPUT;
    icerx_decoder *d = nullptr;
    int rc = icerx_decoder_create(&d, -1, channels, stages, filt, segments, bits);
    if (rc != ICER_RESULT_OK) return rc;
    const size_t off = 0;
    int frame_rc = ICER_RESULT_OK;
    rc = decode_batch(d, 1, data, false, &off, &data_length, planes, nullptr, bufsize, &frame_rc, image_w, image_h);
    icerx_decoder_destroy(d);
    return rc != ICER_RESULT_OK ? rc : frame_rc;
}

}  // namespace

extern "C" {

const char *icerx_decoder_last_error(void) { return g_error.c_str(); }

int icerx_decoder_create(icerx_decoder **out, int device, int channels, int stages, int filt, unsigned segments, int sample_bits)
{
    g_error.clear();
    if (!out) return ICER_INVALID_INPUT;
    *out = nullptr;
    if ((channels != 1 && channels != 3) || filt < 0 || filt > 6 || (sample_bits != 8 && sample_bits != 16)) return ICER_INVALID_INPUT;
    if (stages < 1 || stages > kMaxStages) return ICER_TOO_MANY_STAGES;        // (reference: out-of-bounds table)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no usable HIP device");
#ifndef ICER_HOST_MOCK
    if (device >= 0) {
        if (device >= ndev) return fail("device %d of %d", device, ndev);
        if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice(%d) failed", device);
    }
#endif
    icerx_decoder *d = new icerx_decoder;
    d->device = device; d->channels = channels; d->stages = stages; d->filt = filt; d->bits = sample_bits; d->segments = segments;
#ifndef ICER_HOST_MOCK
    { int cur = 0; hipDeviceProp_t prop; if (hipGetDevice(&cur) == hipSuccess && hipGetDeviceProperties(&prop, cur) == hipSuccess && prop.multiProcessorCount > 0) d->n_cus = prop.multiProcessorCount; }
#endif
    CoderTables ct;
    build_coder_tables(&ct);
    build_decoder_tables(&d->tables, ct);
    build_crc32_table(d->crc_tab);
    int rc = ICER_RESULT_OK;
    HIP_TRY(ensure(d->crc, sizeof d->crc_tab));
    HIP_TRY(hipMemcpy(d->crc.p, d->crc_tab, sizeof d->crc_tab, hipMemcpyHostToDevice));
    HIP_TRY(ensure(d->dtables, sizeof d->tables));
    HIP_TRY(hipMemcpy(d->dtables.p, &d->tables, sizeof d->tables, hipMemcpyHostToDevice));
    *out = d;
    return ICER_RESULT_OK;
done:
    icerx_decoder_destroy(d);
    return rc;
}

void icerx_decoder_destroy(icerx_decoder *d)
{
    if (!d) return;
    for (icerx_decoder::Buf *b : {&d->data, &d->frames, &d->crc, &d->dtables, &d->count, &d->cands, &d->chains, &d->work, &d->tmp,
                                  &d->out8, &d->pos, &d->list, &d->err})
        if (b->p) (void)hipFree(b->p);
#ifndef ICER_HOST_MOCK
    if (d->side_ok) for (hipStream_t st : d->side) (void)hipStreamDestroy(st);
#endif
    delete d;
}

int icerx_decode_host(icerx_decoder *dec, int n, const uint8_t *data, const size_t *offsets, const size_t *lens,
                      void *const *planes_out, size_t frame_stride, int *rcs, size_t *ws, size_t *hs)
{
    if (!planes_out && n) return ICER_INVALID_INPUT;
    return decode_batch(dec, n, data, false, offsets, lens, planes_out, nullptr, frame_stride, rcs, ws, hs);
}

int icerx_decode_device(icerx_decoder *dec, int n, const void *d_data, const size_t *offsets, const size_t *lens,
                        void *d_out, size_t frame_stride, int *rcs, size_t *ws, size_t *hs)
{
    if (!d_out && n) return ICER_INVALID_INPUT;
    return decode_batch(dec, n, (const uint8_t *)d_data, true, offsets, lens, nullptr, d_out, frame_stride, rcs, ws, hs);
}

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
