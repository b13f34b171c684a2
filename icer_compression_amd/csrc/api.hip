// api.hip -- C ABI of libicer_hip.so (declared in include/icer_hip.h) and the host-side pipeline
// driver.  The reference's drivers icer_compress_image_uint16 (lib_icer/src/icer_compress.c:279-426)
// and icer_compress_image_yuv_uint16 (icer_color.c:343-530) become: upload -> DWT stages -> LL mean
// -> sign-magnitude -> one launch that codes every (frame, unit) -> quota scan -> gather.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <sched.h>
#include <time.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/icer_hip.h"
#include "kernels.hpp"

using namespace icer;

namespace {
bool want_priority_streams();      // (defined with the host-fed batch, below)
}
namespace {

thread_local std::string g_last_error;
static std::atomic<uint64_t> g_stats[3];      // unit time-outs, fallback batches, slot re-runs (icerx_process_stats)
CoderTables g_tables;
bool g_tables_ready = false;
std::recursive_mutex g_mutex;

void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    fprintf(stderr, "icer_hip: %s\n", buf);
}

// polite spin (the wait for a batch is tens of milliseconds; see encode_device_impl)
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ICER_FATAL_ERROR;                                                          \
        }                                                                                     \
    } while (0)

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;      // elements
    int ensure(size_t want)
    {
        if (want <= n) return 0;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", want * sizeof(T), hipGetErrorString(e));
            return ICER_FATAL_ERROR;
        }
        n = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// Logical devices.  ICER_HIP_VIRTUAL_DEVICES=<N> makes the library present N devices whatever the node has: logical device
// d runs on physical device d % (devices present).  A dry run of every multi-device code path -- N host threads, N sets of
// pooled encoders / streams / staging buffers, error aggregation -- on a box with a single GPU (tests, bench.py --gpus N
// on one GPU); the streams are the same as with real devices, only slower.  Unset: logical = physical.
int physical_device_count()
{
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}
int virtual_device_count()
{
    const char *v = getenv("ICER_HIP_VIRTUAL_DEVICES");
    const int n = v ? atoi(v) : 0;
    return n >= 1 && n <= 64 ? n : 0;
}
int logical_device_count()
{
    const int phys = physical_device_count(), virt = virtual_device_count();
    return phys > 0 && virt > 0 ? virt : phys;
}
int physical_of(int logical)
{
    const int phys = physical_device_count();
    return phys > 0 && virtual_device_count() > 0 ? logical % phys : logical;
}

// The host cores next to a GPU: the NUMA node of its PCI function (/sys/bus/pci/devices/<bus id>/numa_node) and that
// node's cpulist.  A per-device worker thread of a host batch pins itself there before it allocates its page-locked
// staging words, so that first touch puts them on that node and its polling does not cross sockets
// (ICER_HIP_NUMA=0: off).  Best effort: any failure leaves the thread where it was.
bool pin_thread_near_device(int physical)
{
    if (const char *v = getenv("ICER_HIP_NUMA")) if (atoi(v) == 0) return false;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, physical) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;           // (-1: the platform reports no affinity)
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[1024] = {0};
    const bool ok = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
    if (!ok) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_set = 0;
    for (const char *c = list; *c;) {                   // "0-15,64-79"
        char *end = nullptr;
        const long lo = strtol(c, &end, 10);
        if (end == c) break;
        long hi = lo;
        c = end;
        if (*c == '-') { hi = strtol(c + 1, &end, 10); c = end; }
        for (long k = lo; k <= hi && k < CPU_SETSIZE; k++) { CPU_SET((int)k, &set); n_set++; }
        while (*c == ',' || *c == '\n' || *c == ' ') c++;
    }
    if (n_set == 0) return false;
    // never widen what the caller was given (taskset, numactl, cgroup cpusets, isolcpus): the node's cores AND the thread's
    // current mask; an empty intersection leaves the thread where it is
    cpu_set_t cur;
    CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return false;
    CPU_AND(&set, &set, &cur);
    if (CPU_COUNT(&set) == 0) return false;
    return sched_setaffinity(0, sizeof set, &set) == 0;
}

}  // namespace

constexpr int kMaxParts = 4;             // parts of a batch in flight per call (enqueue)

struct icerx_encoder {
    int device = 0;                     // physical HIP device (hipSetDevice)
    int logical_device = 0;             // what the caller named (ICER_HIP_VIRTUAL_DEVICES maps several onto one)
    size_t w = 0, h = 0;
    int channels = 1, stages = 0, filt = 0, segments = 0, max_frames = 0;
    int sample_bits = 16;               // 8: the uint8 twins (int8 storage, 7 bit planes)
    Plan plan;
    size_t slot_quota = (size_t)-1;     // quota the current slot table was built for
    unsigned bits_per_pixel = 3;        // slot bound; doubled on overflow
    bool units_uploaded = false;
    // Which kernel codes the units.  0 (default): the eight-wave pipeline (code_units_kernel) for large quotas, the
    // workgroup-window coder (code_units_wg_kernel: barriers only, runs of blank chunks in closed form) in progressive
    // mode; 1 / 2: always the pipeline / always the window coder (ICER_HIP_CODER=pipe|wg, tests and measurements).
    int coder_mode = 0;
    bool wg_once = false;               // the next enqueue uses the window coder whatever the mode (after a unit time-out)
    int pipe_waves = 0;                 // 0: shape of the pipeline's workgroups chosen per launch; 8 / 11: pinned (ICER_HIP_PIPE_WAVES)
    int n_cus = 256;                    // compute units of the device
    uint64_t n_timeouts = 0, n_fallbacks = 0, n_slot_retries = 0;   // icerx_encoder_stats
    uint64_t n_routed_units = 0, n_routed_launches = 0;             // icerx_encoder_routing
    bool last_routed = false;           // the last enqueue used both coders

    DevBuf<int16_t> coef, tmp;
    DevBuf<unsigned long long> sums;
    DevBuf<uint16_t> means;
    DevBuf<int> flags;                  // [0,P) dwt overflow  [P,2P) mean overflow  [2P,2P+F) frame skip  [2P+F] bound overflow
    DevBuf<UnitDesc> units;
    DevBuf<uint32_t> work_order, final_order, unit_bits, done_bytes;
    DevBuf<uint64_t> final_off;
    DevBuf<uint8_t> slots;
    DevBuf<uint8_t> sig;                // chunk tables (family_events_kernel), max_frames * plan.sig_bytes
    DevBuf<uint32_t> sig_hist;          // per frame and family: chunks by the bit plane from which they are blank, 16 entries (family_events_kernel -> route_units_kernel)
    DevBuf<uint32_t> sig_blocks;        // Plan::sig_blocks on the device
    DevBuf<uint8_t> events;             // event bytes (events.hpp): max_frames x bit planes x plan.sig_bytes chunks x 64, allocated at the first pipeline launch
    DevBuf<uint8_t> route;              // max_frames * units: the coder of each unit when both share a launch
    DevBuf<uint32_t> route_list, route_ctl;   // the units of the workgroup coder (frame * units + unit), [length, cursor]
    // sub-range splitting (coder_core.hpp "Sub-ranges"): launches of at most split_frames planes cut their dense units into
    // pieces of about split_chunks chunks, one workgroup each (ICER_HIP_SPLIT=<chunks, 0 = off>, ICER_HIP_SPLIT_FRAMES)
    uint32_t split_chunks = 1;          // (1: chosen per geometry when the units are planned -- plan.hpp auto_split_chunks; 2 048 on the headline frame)
    int split_frames = 1;
    int split_hybrid_percent = 90;      // ... whose units with at least this share of blank chunks go to the small workgroup coder (ICER_HIP_SPLIT_HYBRID)
    bool last_split = false;
    int last_waves = 0;                 // last launch: wavefronts per pipeline workgroup (0: the workgroup coder alone)
    uint32_t last_subs = 0;             // last launch: sub-range workgroups
    bool lone_as_batch = false;         // ICER_HIP_LONE_AS_BATCH=1: single-frame launches with the batch build of the pipeline (measurements)
    int split_wgs = 0;                  // staying workgroups of the small coder in a split launch (0: one per compute unit)
    int nosplit_percent = 20;           // split launches: units with at least this share of blank chunks are not cut into sub-ranges (their words stay
                                        // open for long stretches: the pieces would not meet; ICER_HIP_NOSPLIT)
    int list_waves = 0;                 // wavefronts per workgroup of the list kernel: 0 = by launch (4 for a split launch, 1 for a batch), ICER_HIP_LIST_WAVES = 1 | 2 | 4
    DevBuf<SubDesc> subs;
    DevBuf<uint32_t> sub_order, snap_valid;
    DevBuf<Snapshot> snaps;
    DevBuf<SubRecord> sub_recs;
#ifdef ICER_EXPERIMENT_PREFIX_CACHE
    DevBuf<uint32_t> prefix_cache;      // (experiment build only: kernels.hpp SplitLaunch::prefix_cache)
#endif
    hipStream_t side_stream = nullptr;  // the list kernel runs beside the pipeline kernel
    bool side_stream_borrowed = false;  // ... on a stream another encoder owns (the pooled encoders of a host batch share one)
    hipEvent_t fork[kMaxParts] = {}, join[kMaxParts] = {};   // per part of a batch (enqueue): list kernel on the side stream
    hipStream_t half_stream = nullptr;  // the odd parts of a batch coded by a synchronous call (enqueue)
    hipEvent_t part_fork = nullptr, part_join = nullptr;
    int overlap_parts = 2;              // parts a synchronous batch call is enqueued in (ICER_HIP_OVERLAP_PARTS; 1: one stream, as the asynchronous calls)
    int last_parts = 1;
    int wg_waves = 0;                   // 0: the window coder's instance by launch (kernels.hpp code_units_wg_kernel); 4 / 16: pinned (ICER_HIP_WG_WAVES)
    uint32_t list_heavy_min = 64;       // listed units with at least this many chunks that are not blank are taken first (ICER_HIP_LIST_HEAVY; route_units_kernel)
    int test_fail_frame = -1, test_fail_unit = -1, test_fail_calls = 0;   // ICER_HIP_TEST_FAIL_UNIT (test hook, enqueue_part)
    int overlap_first = 50;             // two parts: the first part's share of the frames in percent (ICER_HIP_OVERLAP_FIRST)
    hipEvent_t coef_ready = nullptr;    // the transform of the last enqueue is complete (coef, means, frame status): recorded before the coder
    hipStream_t io_stream = nullptr, copy_stream = nullptr;   // lib_icer-shaped entry points: their encode stream, and the coefficient write-back beside the coder
    int hybrid_percent = 95;            // units with at least this share of blank chunks go to the small workgroup coder (ICER_HIP_HYBRID; 0: none)
    int hybrid_wgs = 2;                 // staying workgroups of the small coder per compute unit in a batch launch (ICER_HIP_HYBRID_WGS).  Round 6: two (C4 6 525 ->
                                        // 6 664, C5 6 526 -> 6 671 Mpix/s per call; 3: the same, 4: C5 + 0.5 %, C4 - 1.8 %, 8: C5 - 5 %; profiles/r06_logs/r06t_list_grid.log) --
                                        // a call now enqueues its frames in two parts, each with a list kernel of its own
    bool unit_major = true;             // batches: the pipeline kernel's workgroups position-major over the frames (ICER_HIP_UNIT_MAJOR=0: frame by frame)
    int list_grid = 0;                  // ... or their number outright in a batch launch (ICER_HIP_LIST_GRID; 0: per compute unit as above)
    int hybrid_frames = 2;              // ... in launches of at least this many planes (frames x channels; ICER_HIP_HYBRID_FRAMES): one gray frame alone is bound by its dense units
    DevBuf<CoderTables> tables;
    // host-API staging
    DevBuf<uint16_t> in;
    DevBuf<uint8_t> in8;
    DevBuf<uint8_t> out;
    DevBuf<unsigned long long> sizes;
    DevBuf<int32_t> rcs;
    DevBuf<uint64_t> prof;              // profiling build only (-DICER_PHASE_TIMERS): per-phase cycle sums

    int *h_flag = nullptr;              // pinned host words: slot-bound overflow flag of the last batch, units on its route list
    hipEvent_t done = nullptr;          // end of the last batch on its stream
    bool wg_available = true;           // the workgroup coder's LDS block (> 64 KiB) was granted: progressive mode, hybrid launches, time-out fall-back
    bool sleepy_wait = false;           // waits yield the core between polls (the per-device workers of a host batch) instead of spinning
    struct Pending {                    // icerx_encode_device_async .. icerx_encoder_wait
        bool active = false;
        const uint16_t *d_frames = nullptr; int n_frames = 0; size_t quota = 0; uint8_t *d_out = nullptr; size_t out_stride = 0;
        uint64_t *d_sizes = nullptr; int32_t *d_rcs = nullptr; void *stream = nullptr;
    } pend;

    bool timing = false;
    hipEvent_t ev[ICERX_NUM_STAGES + 1] = {};
    double ms[ICERX_NUM_STAGES] = {};
    uint64_t timed_calls = 0;
    bool ev_pending = false;
};

namespace {

// What a launch must find zeroed -- status flags, LL sums, histograms, list cursors, sub-range records -- in ONE small kernel instead
// of a fill per range (round 6: a lone frame had nine fills of ~ 4 us each in front of its transform, + their dispatch gaps).
struct ClearList {
    static constexpr int kMax = 10;
    uint32_t *p[kMax];
    uint32_t words[kMax];
    int n = 0;
    void add(void *ptr, size_t bytes) { if (bytes && n < kMax) { p[n] = static_cast<uint32_t *>(ptr); words[n] = (uint32_t)(bytes / 4); n++; } }
};
__global__ void __launch_bounds__(256) clear_ranges_kernel(ClearList cl)
{
    uint32_t *p = cl.p[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < cl.words[blockIdx.y]; i += gridDim.x * 256u) p[i] = 0u;
}
static void launch_clears(const ClearList &cl, hipStream_t st)
{
    if (cl.n) hipLaunchKernelGGL(clear_ranges_kernel, dim3(8, (unsigned)cl.n), dim3(256), 0, st, cl);
}

__global__ void frame_status_kernel(const int *dwt_ovf, const int *mean_ovf, int channels, int n_frames, int *skip)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    int s = 0;
    for (int c = 0; c < channels; c++) s |= dwt_ovf[f * channels + c] | mean_ovf[f * channels + c];
    skip[f] = s;
}

// uint8 twins: the samples are int8 storage (icer_wavelet.c:231 `int8_t *signed_data = (int8_t *) data`); the kernels
// work on them sign-extended to int16
__global__ void __launch_bounds__(256) widen_s8_kernel(const uint8_t *__restrict__ src, uint16_t *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = (uint16_t)(int16_t)(int8_t)src[i];
}

// uint8 twins, the way back: the coder's 16-bit sign-magnitude words as the int8 sign-magnitude bytes the reference leaves in
// the caller's image (icer_wavelet.c:852-858)
__global__ void __launch_bounds__(256) narrow_sm8_kernel(const uint16_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t v = src[i];
        dst[i] = (uint8_t)(((v >> 8) & 0x80u) | (v & 0x7Fu));
    }
}

// 8-bit gray -> uint16 (what the reference's CLI does on the host, example/src/icer_util.c:163-168)
__global__ void __launch_bounds__(256) widen_u8_kernel(const uint8_t *__restrict__ src, uint16_t *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// packed RGB888 -> Y, Cb, Cr planes, integer formulas of the reference's callers (color_util.h:8,27-29)
__global__ void __launch_bounds__(256)
rgb8_to_ycbcr_kernel(const uint8_t *__restrict__ rgb, uint16_t *__restrict__ planes, size_t npix, int n_frames)
{
    const size_t total = npix * (size_t)n_frames;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t f = i / npix, p = i - f * npix;
        const int r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
        auto clip = [](int v) { return v > 255 ? 255 : (v < 0 ? 0 : v); };
        const int y = clip((19595 * r + 38470 * g + 7471 * b) >> 16);
        const int cb = clip(((36962 * (b - y)) >> 16) + 128);
        const int cr = clip(((46727 * (r - y)) >> 16) + 128);
        uint16_t *o = planes + f * 3 * npix + p;
        o[0] = (uint16_t)y;
        o[npix] = (uint16_t)cb;
        o[2 * npix] = (uint16_t)cr;
    }
}

int upload_units(icerx_encoder *e, size_t quota, hipStream_t st)
{
    if (e->units_uploaded && e->slot_quota == quota) return 0;
    // (sub-ranges are planned for encoders of a few planes only: they are used in launches of at most split_frames planes, and
    // their private slot areas and snapshots are per frame)
    // (... and only when a split launch is possible at all: one frame of the encoder must fit split_frames planes, and the
    // routing that goes with it must be on -- a YUV encoder would otherwise carry sub-range areas in every frame's slots that
    // no launch ever uses)
    const bool plan_split = e->wg_available && e->coder_mode == 0 && e->max_frames * e->channels <= 4 && e->split_frames > 0 &&
                            e->channels <= e->split_frames && e->hybrid_percent > 0 && e->split_chunks > 0;
    if (plan_split && e->split_chunks == 1u) e->split_chunks = auto_split_chunks(e->plan.units, e->n_cus, e->sample_bits == 8 ? kPlanes8 : kPlanes);
    assign_slots(&e->plan, quota, e->bits_per_pixel, plan_split ? e->split_chunks : 0u);
    const size_t n = e->plan.units.size();
    if (!e->plan.subs.empty()) {
        if (e->subs.ensure(e->plan.subs.size()) || e->sub_order.ensure(e->plan.split_launch.size())) return ICER_FATAL_ERROR;
        HIP_TRY(hipMemcpyAsync(e->subs.p, e->plan.subs.data(), e->plan.subs.size() * sizeof(SubDesc), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->sub_order.p, e->plan.split_launch.data(), e->plan.split_launch.size() * 4, hipMemcpyHostToDevice, st));
    }
    if (e->units.ensure(n) || e->work_order.ensure(n) || e->final_order.ensure(n)) return ICER_FATAL_ERROR;
    HIP_TRY(hipMemcpyAsync(e->units.p, e->plan.units.data(), n * sizeof(UnitDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(e->work_order.p, e->plan.work_order.data(), n * 4, hipMemcpyHostToDevice, st));
    if (e->sig_blocks.ensure(e->plan.sig_blocks.size() + 1)) return ICER_FATAL_ERROR;
    HIP_TRY(hipMemcpyAsync(e->sig_blocks.p, e->plan.sig_blocks.data(), e->plan.sig_blocks.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(e->final_order.p, e->plan.final_order.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));       // the host vectors may change on the next re-plan
    e->slot_quota = quota;
    e->units_uploaded = true;
    return 0;
}

int accumulate_timing(icerx_encoder *e)
{
    if (!e->ev_pending) return 0;
    HIP_TRY(hipEventSynchronize(e->ev[ICERX_NUM_STAGES]));
    for (int i = 0; i < ICERX_NUM_STAGES; i++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e->ev[i], e->ev[i + 1]));
        e->ms[i] += t;
    }
    e->timed_calls++;
    e->ev_pending = false;
    return 0;
}

// The forward DWT of a batch: one fused tile pass per stage.  Stage 0 reads the caller's frames; every stage writes
// HL/LH/HH to their final place in `coef` (as the coder's sign-magnitude words when `sm` != 0) and its LL band to a
// compact side buffer in `tmp` that the next stage reads (the last stage writes LL into `coef`), so no workgroup reads
// what another one of the same stage writes.  *cw, *ch: in the frame size, out the LL size.
void launch_dwt(icerx_encoder *e, const uint16_t *d_frames, int n_frames, hipStream_t st, int sm, int *dwt_ovf, size_t *cw_io, size_t *ch_io,
                int16_t *coef = nullptr, int16_t *tmp = nullptr)
{
    if (!coef) coef = e->coef.p;            // (a part of a batch: its own planes of the encoder's buffers, enqueue_part)
    if (!tmp) tmp = e->tmp.p;
    const size_t W = e->w, plane = W * e->h;
    const int P = n_frames * e->channels;
    size_t cw = *cw_io, ch = *ch_io, ll_off = 0;
    DwtStageArgs da;
    da.f = filter_taps(e->filt);
    da.lim = e->sample_bits == 8 ? 127 : 32767;
    da.sm = sm;
    da.coef = coef; da.coef_stride = (uint32_t)W;
    da.src = reinterpret_cast<const int16_t *>(d_frames); da.src_stride = (uint32_t)W;
    size_t src_plane = plane;
    for (int s = 0; s < e->stages; s++) {
        const int nlw = (int)((cw + 1) / 2), nlh = (int)((ch + 1) / 2);
        da.cw = (int)cw; da.ch = (int)ch;
        size_t ll_plane;
        if (s == e->stages - 1) { da.ll = coef; da.ll_stride = (uint32_t)W; ll_plane = plane; }
        else { da.ll = tmp + ll_off; da.ll_stride = (uint32_t)nlw; ll_plane = plane; }
        hipLaunchKernelGGL(dwt_tile_kernel, dim3((nlw + kTileKX - 1) / kTileKX, (nlh + kTileKY - 1) / kTileKY, P),
                           dim3(kTileThreads), 0, st, da, src_plane, plane, ll_plane, dwt_ovf);
        da.src = da.ll; da.src_stride = da.ll_stride; src_plane = ll_plane;
        ll_off += (size_t)nlw * nlh;
        cw = nlw;
        ch = nlh;
    }
    *cw_io = cw; *ch_io = ch;
}

// The stream the list kernel runs on beside the pipeline kernel of the same launch: the two must not share a hardware queue.  With
// hardware queues to spare (GPU_MAX_HW_QUEUES >= 6) a plain stream; otherwise a HIGH-priority one -- the runtime keeps a pool of queues
// per priority level, so it can never be given the queue of the caller's (normal-priority) stream, however many streams the process
// has alive (want_priority_streams).
// a stream of the level `which` names ("high" / "low"; anything else, or a runtime without levels: a plain stream)
hipError_t create_level_stream(hipStream_t *st, const char *which)
{
    int least = 0, greatest = 0;
    if (which && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest < least) {
        const bool hi = !strcmp(which, "high"), lo = !strcmp(which, "low");
        if ((hi || lo) && hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi ? greatest : least) == hipSuccess) return hipSuccess;
    }
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
hipError_t create_side_stream(hipStream_t *st)
{
    return create_level_stream(st, want_priority_streams() ? "high" : nullptr);
}
// the per-part events of an encoder and, for encoders of several frames, the stream of a batch's odd parts (enqueue)
hipError_t create_part_events(icerx_encoder *e)
{
    for (int k = 0; k < kMaxParts; k++) {
        hipError_t r = hipEventCreateWithFlags(&e->fork[k], hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&e->join[k], hipEventDisableTiming);
        if (r != hipSuccess) return r;
    }
    if (e->max_frames >= 4 && e->overlap_parts > 1) {
        hipError_t r = hipEventCreateWithFlags(&e->part_fork, hipEventDisableTiming);
        if (r == hipSuccess) r = hipEventCreateWithFlags(&e->part_join, hipEventDisableTiming);
        if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->half_stream, hipStreamNonBlocking);
        if (r != hipSuccess) return r;
    }
    return hipSuccess;
}
// The host-fed pipeline of a device has four streams whose kernels must overlap (three encoders + the side stream they share).  When
// hardware queues are scarce they go to the LOW level's pool together: measured with the runtime's default queues on C4 / C5, quiet
// process or crowded -- low 0.93-0.95 x the device-resident rate (what 8 queues and plain streams give), high 0.89-0.90 x (high-priority
// compute gets in the way of the copy streams' transfers), plain streams 0.68-0.71 x; the copy streams stay plain
// (profiles/r05_logs/r05_r.log, r05_t.log).  Low priority also means what it says: kernels of the host program's own streams go first.
const char *pool_compute_level() { const char *v = getenv("ICER_HIP_COMPUTE_LEVEL"); return v ? v : "low"; }

#ifndef ICER_LONE_PAD_BYTES
#define ICER_LONE_PAD_BYTES 12288
#endif
constexpr int kLonePadBytes = ICER_LONE_PAD_BYTES;      // see enqueue: LDS padding of the pipeline's workgroups in a launch of one frame

// enqueue the whole pipeline for the frames [f0, f0 + n_frames) of a batch on `st` (d_frames, d_out, d_sizes, d_rcs: of frame f0); every
// per-frame buffer of the encoder is used from frame f0 on, so that parts of a batch can be in flight on different streams (enqueue).
// `part`: which set of the per-launch resources (route list cursor, fork / join events) it takes; `timed`: it records the stage events.
// Returns 0 or ICER_FATAL_ERROR.
int enqueue_part(icerx_encoder *e, int f0, int part, bool timed, const uint16_t *d_frames, int n_frames, size_t quota, uint8_t *d_out, size_t out_stride,
                 unsigned long long *d_sizes, int32_t *d_rcs, hipStream_t st, bool clear_bound)
{
    const size_t W = e->w, H = e->h, plane = W * H;
    const int C = e->channels, P = n_frames * C;
    const uint32_t n_units = (uint32_t)e->plan.units.size();
    int *dwt_ovf = e->flags.p + (size_t)f0 * C, *mean_ovf = e->flags.p + (size_t)e->max_frames * C + (size_t)f0 * C;
    int *skip = e->flags.p + 2 * (size_t)e->max_frames * C + f0, *bound_ovf = e->flags.p + 2 * (size_t)e->max_frames * C + e->max_frames;
    // this part's planes of the encoder's per-frame buffers
    int16_t *const coef = e->coef.p + (size_t)f0 * C * plane, *const tmp = e->tmp.p + (size_t)f0 * C * plane;
    unsigned long long *const sums = e->sums.p + (size_t)f0 * C;
    uint16_t *const means = e->means.p + (size_t)f0 * C;
    uint8_t *const sig = e->sig.p + (size_t)f0 * e->plan.sig_bytes;
    uint32_t *const sig_hist = e->sig_hist.p + (size_t)f0 * e->plan.n_families * 16;
    uint8_t *const route_buf = e->route.p + (size_t)f0 * n_units;
    uint32_t *const route_list = e->route_list.p + 2 * (size_t)f0 * n_units,       /* (a part's list: light entries, then heavy ones) */ *const route_ctl = e->route_ctl.p + 4 * (size_t)part;
    uint8_t *const slots = e->slots.p + (size_t)f0 * e->plan.slot_bytes;
    uint32_t *const unit_bits = e->unit_bits.p + (size_t)f0 * n_units, *const done_bytes = e->done_bytes.p + (size_t)f0 * n_units;
    uint64_t *const final_off = e->final_off.p + (size_t)f0 * n_units;
    // which coders this launch uses (decided from the call's arguments alone: what has to be cleared follows from it)
    // Progressive mode: with a byte quota far below the lossless size only the first part of the priority order can end
    // up in the stream.  The units are then launched in priority order with the quota: a unit whose finished
    // higher-priority predecessors alone already exceed it stops (at its start, or at its next check) -- see
    // quota_already_spent.  Not used for large quotas, where the launch order is largest-first instead.
    const bool progressive = quota < (size_t)e->w * e->h * C / 2;
    const bool use_wg = e->wg_available && (e->wg_once || e->coder_mode == 2 || (e->coder_mode == 0 && progressive));
    // both coders in one batch: the bit planes that are mostly runs of blank chunks go to the workgroup coder, which closes
    // such runs in closed form; the dense ones to the pipeline (route_units_kernel)
    // a launch of very few planes (a single frame) cannot fill the chip with whole coding units: its dense units are cut into
    // sub-ranges, one workgroup each, and its all-but-blank ones go to the small workgroup coder as in a batch
    const bool split = e->wg_available && !use_wg && !progressive && e->coder_mode == 0 && e->split_chunks && n_frames * C <= e->split_frames &&
                       !e->plan.subs.empty() && e->hybrid_percent > 0;
    e->last_split = split;
    const bool hybrid = split || (e->wg_available && !use_wg && !progressive && e->coder_mode == 0 && e->hybrid_percent > 0 && n_frames * C >= e->hybrid_frames);
    const size_t sub_entries = e->plan.sub_entries;
    if (split && (e->snaps.ensure((size_t)e->max_frames * sub_entries * kMaxSnaps) || e->snap_valid.ensure((size_t)e->max_frames * sub_entries * kMaxSnaps) ||
                  e->sub_recs.ensure((size_t)e->max_frames * sub_entries))) return ICER_FATAL_ERROR;
    {
        ClearList cl;
        if (f0 == 0 && n_frames == e->max_frames && clear_bound) cl.add(e->flags.p, e->flags.n * sizeof(int));       // (the whole block at once)
        else {
            cl.add(dwt_ovf, (size_t)P * sizeof(int)); cl.add(mean_ovf, (size_t)P * sizeof(int)); cl.add(skip, (size_t)n_frames * sizeof(int));
            if (clear_bound) cl.add(bound_ovf, sizeof(int));
        }
        cl.add(sums, (size_t)P * sizeof(unsigned long long));
        if (progressive) cl.add(done_bytes, (size_t)n_frames * n_units * 4);
        if (hybrid) { cl.add(sig_hist, (size_t)n_frames * e->plan.n_families * 16 * sizeof(uint32_t)); cl.add(route_ctl, 4 * sizeof(uint32_t)); }
        if (split) { cl.add(e->snap_valid.p, (size_t)n_frames * sub_entries * kMaxSnaps * sizeof(uint32_t)); cl.add(e->sub_recs.p, (size_t)n_frames * sub_entries * sizeof(SubRecord)); }
        launch_clears(cl, st);
    }
    if (timed && e->timing) HIP_TRY(hipEventRecord(e->ev[0], st));

    size_t cw = W, ch = H;
    launch_dwt(e, d_frames, n_frames, st, e->sample_bits, dwt_ovf, &cw, &ch, coef, tmp);
    if (timed && e->timing) HIP_TRY(hipEventRecord(e->ev[1], st));

    // ---- LL mean, frame status, sign-magnitude
    const uint32_t llw = (uint32_t)cw, llh = (uint32_t)ch;
    unsigned sum_blocks = (llw * llh + 255) / 256;
    if (sum_blocks > 64) sum_blocks = 64;
    hipLaunchKernelGGL(ll_sum_kernel, dim3(sum_blocks, P), dim3(256), 0, st, reinterpret_cast<const uint16_t *>(coef),
                       plane, (uint32_t)W, llw, llh, sums, e->sample_bits == 8 ? 0xFFu : 0xFFFFu);
    hipLaunchKernelGGL(ll_mean_kernel, dim3((P + 63) / 64), dim3(64), 0, st, sums, (uint32_t)P, llw * llh,
                       means, mean_ovf, e->sample_bits);
    hipLaunchKernelGGL(frame_status_kernel, dim3((n_frames + 63) / 64), dim3(64), 0, st, dwt_ovf, mean_ovf, C, n_frames, skip);
    hipLaunchKernelGGL(finalize_ll_kernel, dim3((llw + 63u) / 64u, (llh + 3u) / 4u, P), dim3(256), 0, st,
                       reinterpret_cast<uint16_t *>(coef), plane, (uint32_t)W, llw, llh, means, skip, C, e->sample_bits);
    if (timed && e->timing) HIP_TRY(hipEventRecord(e->ev[2], st));
    if (e->coef_ready && f0 == 0) HIP_TRY(hipEventRecord(e->coef_ready, st));

    // ---- coding units
    // the stateless half of the context modeller, once per family: event bytes for the pipeline coder's pixel waves, the chunk
    // table for both coders (family_events_kernel; the window coder on its own reads the coefficients itself: table only)
    const int n_planes = e->sample_bits == 8 ? kPlanes8 : kPlanes;
    const size_t ev_frame_bytes = (size_t)n_planes * e->plan.sig_bytes * 64u;
    if (!use_wg && e->events.ensure((size_t)e->max_frames * ev_frame_bytes + 64)) return ICER_FATAL_ERROR;
    {
        hipLaunchKernelGGL(family_events_kernel, dim3((unsigned)(e->plan.sig_blocks.size() / 2), n_frames), dim3(256), 0, st,
                           reinterpret_cast<const uint16_t *>(coef), plane, (uint32_t)W, C, e->units.p, e->sig_blocks.p, skip, sig,
                           e->plan.sig_bytes, hybrid ? sig_hist : nullptr, e->plan.n_families,
                           use_wg ? nullptr : e->events.p + (size_t)f0 * ev_frame_bytes, ev_frame_bytes, (uint32_t)n_planes);
    }
    const uint8_t *route = nullptr;
    e->last_routed = hybrid;
    if (hybrid) {
        hipLaunchKernelGGL(route_units_kernel, dim3((unsigned)((n_units + 255) / 256), n_frames), dim3(256), 0, st, e->units.p, n_units, sig_hist, e->plan.n_families,
                           (uint32_t)(split ? e->split_hybrid_percent : e->hybrid_percent), 16u, route_buf, route_list, route_ctl,
                           (uint32_t)e->nosplit_percent, (uint32_t)n_frames * n_units, e->list_heavy_min);
        route = route_buf;
        // the workgroup coder takes its list on a second stream, beside the pipeline kernel (it is submitted first: its
        // workgroups need most of a compute unit's LDS, which they would not find once the pipeline's have spread out)
        HIP_TRY(hipEventRecord(e->fork[part], st));
        HIP_TRY(hipStreamWaitEvent(e->side_stream, e->fork[part], 0));
        // (a split launch wants the compute units' LDS for its pipeline workgroups: fewer staying workgroups of the small coder, ICER_HIP_SPLIT_WGS)
        // Which instance: measured (profiles/r04_logs/r04_h_list_waves.log).  A batch runs the ONE-wave instance: C4 + 4.2 %,
        // C5 + 2.0 % -- its list is thousands of all-blank units (a first window, then closed-form runs: nothing for a second
        // wave to do but wait at the barriers), and one resident wave of ~ 180 registers leaves the pipeline's workgroups more
        // of the compute unit than two of 204.  The launch of a single frame wants MORE waves per listed unit (7.8 ms with one,
        // 6.4 with two, 6.07 with FOUR, 8.7 with eight -- LDS; profiles/r04_logs/r04_zh_list_kernel_width.log): its list is led by
        // fifty long mid-sparse chains, where every further wave's chunk of a window is progress.
        // ICER_HIP_LIST_WAVES=1|2|4 pins one.
        // (a lone frame: one staying workgroup per compute unit -- 128: 6.8 ms, 256: 5.9, 512: 7.25, profiles/r06_logs/r06u_lone_list_grid.log)
        unsigned list_grid = (unsigned)(split ? (e->split_wgs ? e->split_wgs : e->n_cus) : e->n_cus * e->hybrid_wgs);
        if (!split && e->list_grid) list_grid = (unsigned)e->list_grid;
        const int list_waves = e->list_waves ? e->list_waves : (split ? 4 : 1);
#define ICER_LAUNCH_LIST(I, NS)                                                                                                          \
        hipLaunchKernelGGL((code_units_list_kernel<I>), dim3(list_grid), dim3(64 * NS::kWgWaves), sizeof(NS::Shared), e->side_stream,    \
                           reinterpret_cast<const uint16_t *>(coef), plane, (uint32_t)W, (uint32_t)H, C, e->units.p, n_units,     \
                           e->tables.p, means, skip, slots, e->plan.slot_bytes, unit_bits, sig,                     \
                           e->plan.sig_bytes, route_list, route_ctl, e->prof.p ? e->prof.p + kProfWgsOffset : nullptr, (uint32_t)n_frames * n_units)
        if (list_waves == 1) ICER_LAUNCH_LIST(WgOne, wg1); else if (list_waves == 4) ICER_LAUNCH_LIST(WgFour, wg4); else ICER_LAUNCH_LIST(WgSmall, wgs);
#undef ICER_LAUNCH_LIST
        HIP_TRY(hipEventRecord(e->join[part], e->side_stream));
    }
    SplitLaunch sp;
    if (split) {
        const size_t entries = sub_entries;
        sp.subs = e->subs.p; sp.launch = e->sub_order.p; sp.n_subs = (uint32_t)e->plan.subs.size(); sp.entries = (uint32_t)entries;
        sp.snaps = e->snaps.p; sp.snap_valid = e->snap_valid.p; sp.recs = e->sub_recs.p;
#ifdef ICER_EXPERIMENT_PREFIX_CACHE
        if (!e->prefix_cache.p) {
            if (e->prefix_cache.ensure((size_t)e->max_frames * entries * 36)) return ICER_FATAL_ERROR;
            HIP_TRY(hipMemsetAsync(e->prefix_cache.p, 0, (size_t)e->max_frames * entries * 36 * sizeof(uint32_t), st));
        }
        sp.prefix_cache = getenv("ICER_EXPERIMENT_NO_CACHE") ? nullptr : e->prefix_cache.p;
#endif
    }
    // TEST HOOK (ICER_HIP_TEST_FAIL_UNIT=<frame>:<unit>[:<calls>]): the pipeline kernel of the next <calls> (default 1) calls reports a time-out for
    // that unit of that frame of the batch; nothing else changes.  tests/test_gpu_recovery.py.
    uint32_t fail_inject = ~0u;
    if (e->test_fail_calls > 0 && !use_wg && e->test_fail_frame >= f0 && e->test_fail_frame < f0 + n_frames && e->test_fail_unit < (int)n_units) {
        fail_inject = ((uint32_t)(e->test_fail_frame - f0) << 20) | (uint32_t)e->test_fail_unit;
        e->test_fail_calls--;
    }
    if (!use_wg) {
        // the shape of the pipeline's workgroups: one frame alone cannot fill the chip and is bound by the chain of its
        // largest units, which the large shape (two pixel waves, golomb state wave + two workers) shortens; a batch wants
        // the occupancy of the small one.  ICER_HIP_PIPE_WAVES=8|11 pins one (measurements).
        // (a split launch fills the chip: the small shape's occupancy, measured 6.63 against 6.78 ms on the headline frame)
        const bool large = e->pipe_waves ? e->pipe_waves == kUnitWavesLarge : (n_frames == 1 && !split);
        // A launch of one frame is bound by the chains of its largest units, not by occupancy: it runs the build without the
        // register budget, padded to the LDS footprint of the queue-depth-8 build (49 KiB; the padding is static: the
        // `dynamic LDS' launch parameter had no effect on a kernel that declares none).  Measured on the headline frame: 37 KiB
        // 7.5 ms, 45.6 KiB 6.8 ms, 49.5 KiB 6.7-6.8 ms (profiles/archive/r03_logs/r03_aa.log, r03_ab.log).
        const bool lone = n_frames * C <= e->split_frames;
        // (a batch that is not in progressive mode -- there the priority order across frames does not matter, the order within a frame does --
        // is launched position-major over its frames: code_units_kernel; ICER_HIP_UNIT_MAJOR=0: frame by frame as before)
        const bool unit_major = n_frames > 1 && !progressive && e->unit_major;
        const dim3 pipe_grid = unit_major ? dim3((unsigned)((n_units + sp.n_subs) * (unsigned)n_frames), 1) : dim3(n_units + sp.n_subs, n_frames);
#define ICER_LAUNCH_PIPE(NW, OCC, PAD)                                                                                                       \
        hipLaunchKernelGGL((code_units_kernel<NW, OCC, PAD>), pipe_grid, dim3(64 * NW), 0, st,                                               \
                           reinterpret_cast<const uint16_t *>(coef), plane, (uint32_t)W, (uint32_t)H, C, e->units.p,                 \
                           progressive ? nullptr : e->work_order.p, n_units, e->tables.p, means, skip, slots,                  \
                           e->plan.slot_bytes, unit_bits, e->prof.p, done_bytes, progressive ? (uint64_t)quota : 0ull, route, sp, \
                           unit_major ? (uint32_t)n_frames : 1u, e->events.p + (size_t)f0 * ev_frame_bytes, ev_frame_bytes, sig, e->plan.sig_bytes, fail_inject)
        e->last_waves = large ? kUnitWavesLarge : kUnitWavesSmall;
        e->last_subs = sp.n_subs;
        if (large) ICER_LAUNCH_PIPE(kUnitWavesLarge, 1, 0);
        else if (lone && !e->lone_as_batch) ICER_LAUNCH_PIPE(kUnitWavesSmall, 1, kLonePadBytes);
        else ICER_LAUNCH_PIPE(kUnitWavesSmall, 8, 0);
#undef ICER_LAUNCH_PIPE
        if (split)
            hipLaunchKernelGGL(splice_units_kernel, dim3(n_units, n_frames), dim3(64 * kSpliceWaves), 0, st, e->units.p, n_units, e->tables.p, means, skip, C,
                               (uint32_t)W, (uint32_t)H, slots, e->plan.slot_bytes, unit_bits, route, sp);
        if (hybrid) HIP_TRY(hipStreamWaitEvent(st, e->join[part], 0));
    }
    if (use_wg) e->last_waves = 0, e->last_subs = 0;
    if (use_wg) {
        // (which instance: kernels.hpp code_units_wg_kernel; ICER_HIP_WG_WAVES=4|16 pins one)
        const bool four = e->wg_waves ? e->wg_waves == 4 : (progressive || n_frames * C >= 4);
#define ICER_LAUNCH_WG(I, NS)                                                                                                       \
        hipLaunchKernelGGL((code_units_wg_kernel<I>), dim3(n_units, n_frames), dim3(64 * NS::kWgWaves), sizeof(NS::Shared), st,  \
                           reinterpret_cast<const uint16_t *>(coef), plane, (uint32_t)W, (uint32_t)H, C, e->units.p,              \
                           progressive ? nullptr : e->work_order.p, n_units, e->tables.p, means, skip, slots,                      \
                           e->plan.slot_bytes, unit_bits, e->prof.p, done_bytes, progressive ? (uint64_t)quota : 0ull,              \
                           sig, e->plan.sig_bytes)
        if (four) ICER_LAUNCH_WG(WgFour, wg4); else ICER_LAUNCH_WG(WgFull, wg);
#undef ICER_LAUNCH_WG
    }
    if (timed && e->timing) HIP_TRY(hipEventRecord(e->ev[3], st));

    // ---- quota scan + gather into final stream order
    hipLaunchKernelGGL(scan_kernel, dim3(n_frames), dim3(64), 0, st, unit_bits, e->final_order.p, n_units,
                       (uint64_t)quota, skip, final_off, d_sizes, d_rcs, e->units.p, bound_ovf);
    hipLaunchKernelGGL(gather_kernel, dim3(n_units, n_frames), dim3(256), 0, st, slots, e->plan.slot_bytes,
                       e->units.p, n_units, unit_bits, final_off, d_out, out_stride);
    if (timed && e->timing) {
        HIP_TRY(hipEventRecord(e->ev[4], st));
        e->ev_pending = true;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// One call = one batch.  A batch of several frames coded by the SYNCHRONOUS entry points is enqueued in parts on two streams -- the
// caller's and one of the encoder's own --, so that a part's transform and event pass run beside the coder kernels of the part before
// it and the tail of a coder kernel (its last long units, most of the chip idle) hides behind the next part's: what a caller gets from two
// encoders and the asynchronous calls, inside one call.  Not for the asynchronous entry points (the caller overlaps
// whole batches itself), progressive mode, or single frames.  ICER_HIP_OVERLAP_PARTS=<1..4> (1: off).
int enqueue(icerx_encoder *e, const uint16_t *d_frames, int n_frames, size_t quota, uint8_t *d_out, size_t out_stride,
            unsigned long long *d_sizes, int32_t *d_rcs, hipStream_t st, bool overlap_ok)
{
    const int C = e->channels;
    int *bound_ovf = e->flags.p + 2 * (size_t)e->max_frames * C + e->max_frames;
    const bool progressive = quota < (size_t)e->w * e->h * C / 2;
    int parts = 1;
    if (overlap_ok && e->overlap_parts > 1 && e->half_stream && !progressive && e->coder_mode == 0 && !e->wg_once && n_frames >= 2 * e->overlap_parts &&
        n_frames * C >= e->hybrid_frames)
        parts = e->overlap_parts;
    e->last_parts = parts;
    if (parts == 1) return enqueue_part(e, 0, 0, true, d_frames, n_frames, quota, d_out, out_stride, d_sizes, d_rcs, st, true);
    const size_t plane = e->w * e->h;
    HIP_TRY(hipMemsetAsync(bound_ovf, 0, sizeof(int), st));             // (shared by the parts: before the second stream forks off)
    HIP_TRY(hipEventRecord(e->part_fork, st));                          // (the second stream starts behind whatever the caller's stream holds)
    HIP_TRY(hipStreamWaitEvent(e->half_stream, e->part_fork, 0));
    // (the stages of the parts overlap: the call's span is booked on the coder stage -- bench.py's roofline divides the call's bytes by it)
    if (e->timing) { HIP_TRY(hipEventRecord(e->ev[0], st)); HIP_TRY(hipEventRecord(e->ev[1], st)); HIP_TRY(hipEventRecord(e->ev[2], st)); }
    for (int k = 0, f0 = 0; k < parts; k++) {
        // (two parts: the first one smaller -- its transform and event pass have nothing to hide behind)
        int n = n_frames / parts + (k < n_frames % parts ? 1 : 0);
        if (parts == 2) { const int n0 = std::max(1, std::min(n_frames - 1, (n_frames * e->overlap_first + 50) / 100)); n = k == 0 ? n0 : n_frames - n0; }
        hipStream_t ps = (k & 1) ? e->half_stream : st;
        if (int rc = enqueue_part(e, f0, k, false, d_frames + (size_t)f0 * C * plane, n, quota, d_out + (size_t)f0 * out_stride, out_stride,
                                  d_sizes + f0, d_rcs + f0, ps, false)) return rc;
        f0 += n;
    }
    HIP_TRY(hipEventRecord(e->part_join, e->half_stream));
    HIP_TRY(hipStreamWaitEvent(st, e->part_join, 0));
    if (e->timing) {
        HIP_TRY(hipEventRecord(e->ev[3], st)); HIP_TRY(hipEventRecord(e->ev[4], st));
        e->ev_pending = true;
    }
    return 0;
}

}  // namespace

extern "C" {

const char *icerx_last_error(void) { return g_last_error.c_str(); }

int icer_init(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mutex);
    if (!g_tables_ready) {
        build_coder_tables(&g_tables);
        g_tables_ready = true;
    }
    return ICER_RESULT_OK;
}

int icer_init_output_struct(icer_output_data_buf_typedef *out, uint8_t *data, size_t buf_len, size_t byte_quota)
{
    if (byte_quota * 2 > buf_len) return ICER_OUTPUT_BUF_TOO_SMALL;
    out->size_used = 0;
    out->data_start = data;
    out->size_allocated = byte_quota;
    out->rearrange_start = data + byte_quota;
    return ICER_RESULT_OK;
}

int icerx_encoder_create(icerx_encoder **out, int device, size_t w, size_t h, int channels, int stages, int filt,
                         int segments, int max_frames)
{
    return icerx_encoder_create_ex(out, device, w, h, channels, stages, filt, segments, max_frames, 16);
}

int icerx_encoder_create_ex(icerx_encoder **out, int device, size_t w, size_t h, int channels, int stages, int filt,
                            int segments, int max_frames, int sample_bits)
{
    *out = nullptr;
    icer_init();
    if (filt < 0 || filt > 6 || max_frames < 1 || (sample_bits != 8 && sample_bits != 16)) return ICER_INVALID_INPUT;
    icerx_encoder *e = new icerx_encoder();
    e->device = device; e->w = w; e->h = h; e->channels = channels; e->stages = stages; e->filt = filt;
    e->segments = segments; e->max_frames = max_frames; e->sample_bits = sample_bits;
    const int rc = build_plan(&e->plan, w, h, channels, stages, segments, sample_bits);
    if (rc != kOk) { delete e; return rc; }
    // tuning knob: initial per-unit slot bound in bits per pixel (doubled automatically on overflow)
    if (const char *pw = getenv("ICER_HIP_PIPE_WAVES")) { const int v = atoi(pw); if (v == kUnitWavesSmall || v == kUnitWavesLarge) e->pipe_waves = v; }
    if (const char *cd = getenv("ICER_HIP_CODER")) e->coder_mode = !strcmp(cd, "pipe") ? 1 : !strcmp(cd, "wg") ? 2 : 0;
    if (const char *hy = getenv("ICER_HIP_HYBRID")) { const int v = atoi(hy); if (v >= 0 && v <= 100) e->hybrid_percent = v; }
    if (const char *hf = getenv("ICER_HIP_HYBRID_FRAMES")) { const int v = atoi(hf); if (v >= 1) e->hybrid_frames = v; }
    if (const char *um = getenv("ICER_HIP_UNIT_MAJOR")) e->unit_major = atoi(um) != 0;
    if (const char *lg = getenv("ICER_HIP_LIST_GRID")) { const int v = atoi(lg); if (v >= 1 && v <= 65536) e->list_grid = v; }
    if (const char *hw = getenv("ICER_HIP_HYBRID_WGS")) { const int v = atoi(hw); if (v >= 1 && v <= 4) e->hybrid_wgs = v; }
    if (const char *sc = getenv("ICER_HIP_SPLIT")) { const int v = atoi(sc); if (v == 0 || v >= 128) e->split_chunks = (uint32_t)v; }
    if (const char *sh = getenv("ICER_HIP_SPLIT_HYBRID")) { const int v = atoi(sh); if (v >= 1 && v <= 101) e->split_hybrid_percent = v; }     // (101: no unit goes to the small coder)
    if (const char *sw = getenv("ICER_HIP_LONE_AS_BATCH")) e->lone_as_batch = atoi(sw) != 0;
    if (const char *sw = getenv("ICER_HIP_SPLIT_WGS")) { const int v = atoi(sw); if (v >= 1 && v <= 4096) e->split_wgs = v; }
    if (const char *sf = getenv("ICER_HIP_SPLIT_FRAMES")) { const int v = atoi(sf); if (v >= 0) e->split_frames = v; }
    if (const char *ns = getenv("ICER_HIP_NOSPLIT")) { const int v = atoi(ns); if (v >= 1 && v <= 101) e->nosplit_percent = v; }
    if (const char *op = getenv("ICER_HIP_OVERLAP_PARTS")) { const int v = atoi(op); if (v >= 1 && v <= kMaxParts) e->overlap_parts = v; }
    if (const char *tf = getenv("ICER_HIP_TEST_FAIL_UNIT")) {
        int f = -1, u = -1, c = 1;
        if (sscanf(tf, "%d:%d:%d", &f, &u, &c) >= 2 && f >= 0 && f < (1 << 11) && u >= 0 && u < (1 << 20) && c >= 1) { e->test_fail_frame = f; e->test_fail_unit = u; e->test_fail_calls = c; }
    }
    if (const char *ww = getenv("ICER_HIP_WG_WAVES")) { const int v = atoi(ww); if (v == 4 || v == 16) e->wg_waves = v; }
    if (const char *lh = getenv("ICER_HIP_LIST_HEAVY")) { const long v = atol(lh); if (v >= 1) e->list_heavy_min = (uint32_t)v; }
    if (const char *of = getenv("ICER_HIP_OVERLAP_FIRST")) { const int v = atoi(of); if (v >= 5 && v <= 95) e->overlap_first = v; }
    if (const char *lw = getenv("ICER_HIP_LIST_WAVES")) { const int v = atoi(lw); if (v == 1 || v == 2 || v == 4) e->list_waves = v; }
    if (const char *bpp = getenv("ICER_HIP_SLOT_BPP")) {
        const int v = atoi(bpp);
        if (v >= 1 && v <= 24) e->bits_per_pixel = (unsigned)v;
    }

    int count = 0;
    hipError_t he = hipGetDeviceCount(&count);
    const int logical_count = he == hipSuccess && count > 0 ? logical_device_count() : 0;
    if (he != hipSuccess || count <= 0 || device < 0 || device >= logical_count) {
        set_error("no usable HIP device (hipGetDeviceCount: %s, count=%d, requested=%d); this library has no CPU path",
                  hipGetErrorString(he), count, device);
        delete e;
        return ICER_FATAL_ERROR;
    }
    e->logical_device = device;
    device = physical_of(device);
    e->device = device;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->n_cus = prop.multiProcessorCount; }
    // (a failing HIP call below must not leak the object and what it has allocated so far)
#define CREATE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); icerx_encoder_destroy(e); return ICER_FATAL_ERROR; } } while (0)
    CREATE_TRY(hipSetDevice(device));
    const size_t P = (size_t)max_frames * channels, plane = w * h, n_units = e->plan.units.size();
    if (e->coef.ensure(P * plane) || e->tmp.ensure(P * plane) || e->sums.ensure(P) || e->means.ensure(P) ||
        e->flags.ensure(2 * P + 2 * max_frames + 1) || e->unit_bits.ensure((size_t)max_frames * n_units) ||
        e->done_bytes.ensure((size_t)max_frames * n_units) || e->route.ensure((size_t)max_frames * n_units) || e->route_list.ensure(2 * (size_t)max_frames * n_units) || e->route_ctl.ensure(4 * kMaxParts) || e->sig.ensure((size_t)max_frames * e->plan.sig_bytes + 64) || e->sig_hist.ensure((size_t)max_frames * e->plan.n_families * 16 + 16) ||
        e->final_off.ensure((size_t)max_frames * n_units) || e->tables.ensure(1) || e->sizes.ensure(max_frames) ||
        e->rcs.ensure(max_frames)) {
        icerx_encoder_destroy(e);
        return ICER_FATAL_ERROR;
    }
    // event bytes of the pipeline coder (events.hpp): one per pixel and bit plane, in chunk order
    if (e->coder_mode != 2 && e->events.ensure((size_t)max_frames * (size_t)(sample_bits == 8 ? kPlanes8 : kPlanes) * e->plan.sig_bytes * 64u + 64)) {
        icerx_encoder_destroy(e);
        return ICER_FATAL_ERROR;
    }
    CREATE_TRY(hipMemcpy(e->tables.p, &g_tables, sizeof g_tables, hipMemcpyHostToDevice));
    // the workgroup coder's LDS block is above the 64 KiB a kernel gets without asking
    // A device / runtime that refuses it loses only the paths that need that coder (progressive mode then runs on the
    // pipeline, launches are not shared, a unit time-out becomes an error) -- reported by icerx_encoder_stats.
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(code_units_wg_kernel<WgFull>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wg::Shared)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(code_units_wg_kernel<WgFour>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wg4::Shared)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(code_units_list_kernel<WgSmall>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wgs::Shared)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(code_units_list_kernel<WgOne>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wg1::Shared)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(code_units_list_kernel<WgFour>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(wg4::Shared)) != hipSuccess ||
        create_side_stream(&e->side_stream) != hipSuccess ||
        create_part_events(e) != hipSuccess) {
        (void)hipGetLastError();
        e->wg_available = false;
        if (e->coder_mode == 2) { set_error("ICER_HIP_CODER=wg, but this device does not grant the workgroup coder its LDS block"); icerx_encoder_destroy(e); return ICER_FATAL_ERROR; }
        fprintf(stderr, "libicer_hip: the workgroup coder is not available on this device; the wave pipeline codes everything\n");
    }
#ifdef ICER_PHASE_TIMERS
    if (e->prof.ensure(kProfWords)) { icerx_encoder_destroy(e); return ICER_FATAL_ERROR; }
    CREATE_TRY(hipMemset(e->prof.p, 0, kProfWords * sizeof(uint64_t)));
#endif
    for (auto &ev : e->ev) CREATE_TRY(hipEventCreate(&ev));
    CREATE_TRY(hipHostMalloc((void **)&e->h_flag, (2 + kMaxParts) * sizeof(int), hipHostMallocDefault));
    CREATE_TRY(hipEventCreateWithFlags(&e->done, hipEventDisableTiming));
#undef CREATE_TRY
    *out = e;
    return 0;
}

void icerx_encoder_destroy(icerx_encoder *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    e->coef.release(); e->tmp.release(); e->sums.release(); e->means.release(); e->flags.release();
    e->units.release(); e->work_order.release(); e->final_order.release(); e->unit_bits.release(); e->done_bytes.release(); e->sig.release(); e->events.release(); e->sig_hist.release(); e->sig_blocks.release(); e->route.release(); e->route_list.release(); e->route_ctl.release();
    e->final_off.release(); e->slots.release(); e->tables.release(); e->in.release(); e->in8.release(); e->out.release();
    e->sizes.release(); e->rcs.release(); e->prof.release();
    e->subs.release(); e->sub_order.release(); e->snap_valid.release(); e->snaps.release(); e->sub_recs.release();
    for (auto &ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->done) (void)hipEventDestroy(e->done);
    for (auto &ev : e->fork) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->join) if (ev) (void)hipEventDestroy(ev);
    if (e->part_fork) (void)hipEventDestroy(e->part_fork);
    if (e->part_join) (void)hipEventDestroy(e->part_join);
    if (e->half_stream) (void)hipStreamDestroy(e->half_stream);
    if (e->side_stream && !e->side_stream_borrowed) (void)hipStreamDestroy(e->side_stream);
    if (e->coef_ready) (void)hipEventDestroy(e->coef_ready);
    if (e->io_stream) (void)hipStreamDestroy(e->io_stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->h_flag) (void)hipHostFree(e->h_flag);
    delete e;
}

// diagnostics of a unit time-out (rare error path): which unit, which wave at which wait, and the unit's hand-off counters
// (the record code_units_kernel leaves in the failed unit's payload slot)
static void report_timeouts(icerx_encoder *e, int n_frames)
{
    const size_t n_units = e->plan.units.size();
    std::vector<uint32_t> bits((size_t)n_frames * n_units);
    if (!n_units || hipMemcpy(bits.data(), e->unit_bits.p, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return;
    int shown = 0;
    for (size_t i = 0; i < bits.size() && shown < 4; i++) {
        if (bits[i] != kUnitFailed) continue;
        const size_t frame = i / n_units, ui = i % n_units;
        const UnitDesc &u = e->plan.units[ui];
        uint32_t dbg[kFailWords] = {0};
        if (u.cap_words >= kFailWords)
            (void)hipMemcpy(dbg, e->slots.p + frame * e->plan.slot_bytes + u.slot_off + kHeaderBytes, sizeof dbg, hipMemcpyDeviceToHost);
        fprintf(stderr, "libicer_hip: time-out in frame %zu unit %zu (chan %u level %u subband %u lsb %u seg %u, %u x %u)", frame, ui,
                (unsigned)u.chan, (unsigned)u.level, (unsigned)u.subband, (unsigned)u.lsb, (unsigned)u.seg, (unsigned)u.w, (unsigned)u.h);
        if (dbg[0] == kFailMagic)
            fprintf(stderr, ": wave %u gave up at coder_core.hpp:%u; chunks %u, done pixel/count/compact/merge %u/%u/%u/%u, ring alloc/popped "
                    "%u/%u, hold seq/ack %u/%u, generation %u (last exact %u), bitpos %u, flushed words %u, drain_exit %u",
                    dbg[1] >> 16, dbg[1] & 0xFFFFu, dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11],
                    dbg[12], dbg[13], dbg[14], dbg[15]);
        fprintf(stderr, "\n");
        shown++;
    }
}

// One encode call = begin (everything enqueued on the stream, nothing waited for) + finish (wait, then the rare re-runs).
// `flag` = two pinned host words that receive the batch's verdict: [0] bit 0 a coding unit outgrew its provisioned slot,
// bit 1 a unit timed out; [1] units on the route list.
static int encode_begin(icerx_encoder *e, const uint16_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                        size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, hipStream_t st, int *flag, hipEvent_t done, bool *regrow, bool overlap_ok = false)
{
    if (upload_units(e, byte_quota, st)) return ICER_FATAL_ERROR;
    if (e->slots.ensure((size_t)e->max_frames * e->plan.slot_bytes)) return ICER_FATAL_ERROR;
    if (out_stride < byte_quota && out_stride < e->plan.slot_bytes) {
        if (regrow) { *regrow = true; return 0; }
        set_error("icerx_encode_device: out_stride %zu smaller than the byte quota %zu", out_stride, byte_quota);
        return ICER_INVALID_INPUT;
    }
    int *bound_ovf = e->flags.p + 2 * (size_t)e->max_frames * e->channels + e->max_frames;
    if (enqueue(e, d_frames, n_frames, byte_quota, d_out, out_stride, (unsigned long long *)d_sizes, d_rcs, st, overlap_ok && flag == e->h_flag))
        return ICER_FATAL_ERROR;
    HIP_TRY(hipMemcpyAsync(flag, bound_ovf, sizeof(int), hipMemcpyDeviceToHost, st));
    if (e->last_routed) HIP_TRY(hipMemcpyAsync(flag + 1, e->route_ctl.p, sizeof(int), hipMemcpyDeviceToHost, st));
    // (a batch enqueued in parts: the other parts' list lengths behind the two words every caller has -- only e->h_flag is that long)
    for (int k = 1; k < e->last_parts; k++)
        if (e->last_routed) HIP_TRY(hipMemcpyAsync(flag + 1 + k, e->route_ctl.p + 4 * k, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(done, st));
    return 0;
}

// Wait for an event.  The synchronous entry points spin on a query instead of blocking in hipStreamSynchronize: an
// encode call is milliseconds and the wake-up latency of a blocking wait (measured: up to 3 ms per call on a busy host)
// would be charged to every frame.  The per-device workers of a host batch (sleepy) give the core away between polls:
// eight of them must not burn eight cores, and their waits are hidden behind the next sub-batch anyway.
static int wait_event(hipEvent_t ev, bool sleepy)
{
    for (uint32_t polls = 0;; polls++) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { set_error("hipEventQuery failed: %s", hipGetErrorString(q)); return ICER_FATAL_ERROR; }
        if (sleepy && polls > 64u) { const struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
        else cpu_relax();
    }
}

// The verdict of a finished batch (encode_begin's flag words): 0 = done, 1 = run it again (the encoder has been adjusted:
// larger slots, or the barrier-only coder for the next run), or an error code.
static int encode_verdict(icerx_encoder *e, int n_frames, const int *flag)
{
    const int ovf = flag[0];
    if (!ovf) {
        if (e->last_routed) {
            e->n_routed_units += (uint64_t)(uint32_t)flag[1]; e->n_routed_launches++;
            if (flag == e->h_flag) for (int k = 1; k < e->last_parts; k++) e->n_routed_units += (uint64_t)(uint32_t)flag[1 + k];
        }
        return 0;
    }
    if (ovf & 2) {
        // A wave of some coding unit of the eight-wave pipeline waited longer than its spin bound (seconds) and gave
        // the unit up (seen once in ~60 000 randomised encodes in round 1, never reproduced).  The batch is coded again
        // by the workgroup-window coder, which has no wave-to-wave hand-offs to wait for (barriers only) and produces
        // the same streams: the caller gets its result, the event is counted (icerx_encoder_stats) and reported.
        report_timeouts(e, n_frames);
        e->n_timeouts++; g_stats[0]++;
        if (e->wg_once || !e->wg_available) {          // (the window coder never reports a time-out)
            e->wg_once = false;
            set_error(e->wg_available ? "a coding unit timed out in the workgroup-window coder"
                                      : "a coding unit timed out and the barrier-only coder is not available on this device");
            return ICER_FATAL_ERROR;
        }
        e->n_fallbacks++; g_stats[1]++;
        e->wg_once = true;
        fprintf(stderr, "libicer_hip: a coding unit timed out; coding the batch again with the barrier-only coder\n");
        e->ev_pending = false;
        return 1;
    }
    e->n_slot_retries++; g_stats[2]++;
    if (e->bits_per_pixel >= 24) {
        set_error("coding-unit slot overflow at the theoretical bound");
        return ICER_FATAL_ERROR;
    }
    // a unit produced more than the provisioned bits per pixel: enlarge the slots and redo the batch
    e->ev_pending = false;
    e->bits_per_pixel = e->bits_per_pixel * 2 > 24 ? 24 : e->bits_per_pixel * 2;
    e->units_uploaded = false;
    return 1;
}

// `regrow`: the host wrappers size the output by min(quota, slot area); when a slot-bound retry enlarges the slot area
// they must re-allocate, signalled by *regrow (the batch is then re-run by them).
static int encode_device_impl(icerx_encoder *e, const uint16_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                              size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream, bool *regrow, bool already_begun = false)
{
    if (!e || !d_frames || !d_out || !d_sizes || !d_rcs || n_frames < 1 || n_frames > e->max_frames) {
        set_error("icerx_encode_device: invalid arguments");
        return ICER_INVALID_INPUT;
    }
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)stream;
    if (!already_begun) {
        if (e->pend.active) { set_error("icerx_encode_device: an asynchronous encode is pending on this encoder (icerx_encoder_wait)"); return ICER_INVALID_INPUT; }
        if (accumulate_timing(e)) return ICER_FATAL_ERROR;
        e->wg_once = false;
    }
    for (bool begun = already_begun;; begun = false) {
        if (!begun) {
            bool rg = false;
            const int rc = encode_begin(e, d_frames, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, st, e->h_flag, e->done, regrow ? &rg : nullptr,
                                        /* overlap_ok = */ !already_begun || e->last_parts > 1);
            if (rc) return rc;
            if (rg) { *regrow = true; return 0; }
        }
        if (wait_event(e->done, e->sleepy_wait)) return ICER_FATAL_ERROR;
        const int v = encode_verdict(e, n_frames, e->h_flag);
        if (v < 0) return v;
        if (v == 0) break;
    }
    e->wg_once = false;
    return 0;
}

int icerx_encode_device(icerx_encoder *e, const uint16_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                        size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream)
{
    if (e && e->sample_bits != 16) {
        set_error("icerx_encode_device: this encoder was created for 8-bit samples (use icerx_encode_device_s8)");
        return ICER_INVALID_INPUT;
    }
    return encode_device_impl(e, d_frames, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, stream, nullptr);
}

// The same call in two halves, so that a caller can overlap its own copies (or another encoder's work) with the coding:
// icerx_encode_device_async returns as soon as everything is enqueued on `stream`; icerx_encoder_wait returns once the
// batch is complete there (and has re-run it in the rare cases icerx_encode_device does).  One pending call per encoder.
int icerx_encode_device_async(icerx_encoder *e, const uint16_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                              size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream)
{
    if (!e || !d_frames || !d_out || !d_sizes || !d_rcs || n_frames < 1 || n_frames > e->max_frames || e->sample_bits != 16) {
        set_error("icerx_encode_device_async: invalid arguments");
        return ICER_INVALID_INPUT;
    }
    if (e->pend.active) { set_error("icerx_encode_device_async: the previous asynchronous encode has not been waited for"); return ICER_INVALID_INPUT; }
    HIP_TRY(hipSetDevice(e->device));
    if (accumulate_timing(e)) return ICER_FATAL_ERROR;
    e->wg_once = false;
    const int rc = encode_begin(e, d_frames, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, (hipStream_t)stream, e->h_flag, e->done, nullptr);
    if (rc) return rc;
    e->pend.active = true;
    e->pend.d_frames = d_frames; e->pend.n_frames = n_frames; e->pend.quota = byte_quota; e->pend.d_out = d_out; e->pend.out_stride = out_stride;
    e->pend.d_sizes = d_sizes; e->pend.d_rcs = d_rcs; e->pend.stream = stream;
    return 0;
}

int icerx_encoder_wait(icerx_encoder *e)
{
    if (!e) return ICER_INVALID_INPUT;
    if (!e->pend.active) return 0;
    const icerx_encoder::Pending p = e->pend;
    e->pend.active = false;
    return encode_device_impl(e, p.d_frames, p.n_frames, p.quota, p.d_out, p.out_stride, p.d_sizes, p.d_rcs, p.stream, nullptr, true);
}

int icerx_encode_device_u8(icerx_encoder *e, const uint8_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                           size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream)
{
    if (!e || !d_frames || e->channels != 1 || n_frames < 1 || n_frames > e->max_frames) {
        set_error("icerx_encode_device_u8: invalid arguments (needs a 1-channel encoder)");
        return ICER_INVALID_INPUT;
    }
    if (e->pend.active) { set_error("icerx_encode_device_u8: an asynchronous encode is pending on this encoder (icerx_encoder_wait)"); return ICER_INVALID_INPUT; }
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)n_frames * e->w * e->h;
    if (e->in.ensure((size_t)e->max_frames * e->w * e->h)) return ICER_FATAL_ERROR;
    hipLaunchKernelGGL(widen_u8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       d_frames, e->in.p, n);
    return icerx_encode_device(e, e->in.p, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, stream);
}

int icerx_encode_device_s8(icerx_encoder *e, const uint8_t *d_planes, int n_frames, size_t byte_quota, uint8_t *d_out,
                           size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream)
{
    if (!e || !d_planes || e->sample_bits != 8 || n_frames < 1 || n_frames > e->max_frames) {
        set_error("icerx_encode_device_s8: invalid arguments (needs an encoder created with sample_bits = 8)");
        return ICER_INVALID_INPUT;
    }
    if (e->pend.active) { set_error("icerx_encode_device_s8: an asynchronous encode is pending on this encoder (icerx_encoder_wait)"); return ICER_INVALID_INPUT; }
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = (size_t)n_frames * e->channels * e->w * e->h;
    if (e->in.ensure((size_t)e->max_frames * e->channels * e->w * e->h)) return ICER_FATAL_ERROR;
    hipLaunchKernelGGL(widen_s8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       d_planes, e->in.p, n);
    return encode_device_impl(e, e->in.p, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, stream, nullptr);
}

int icerx_encode_device_rgb8(icerx_encoder *e, const uint8_t *d_rgb, int n_frames, size_t byte_quota, uint8_t *d_out,
                             size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream)
{
    if (!e || !d_rgb || e->channels != 3 || n_frames < 1 || n_frames > e->max_frames) {
        set_error("icerx_encode_device_rgb8: invalid arguments (needs a 3-channel encoder)");
        return ICER_INVALID_INPUT;
    }
    if (e->pend.active) { set_error("icerx_encode_device_rgb8: an asynchronous encode is pending on this encoder (icerx_encoder_wait)"); return ICER_INVALID_INPUT; }
    HIP_TRY(hipSetDevice(e->device));
    const size_t npix = e->w * e->h;
    if (e->in.ensure((size_t)e->max_frames * 3 * npix)) return ICER_FATAL_ERROR;
    hipLaunchKernelGGL(rgb8_to_ycbcr_kernel, dim3((unsigned)std::min<size_t>((npix * n_frames + 255) / 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, d_rgb, e->in.p, npix, n_frames);
    return icerx_encode_device(e, e->in.p, n_frames, byte_quota, d_out, out_stride, d_sizes, d_rcs, stream);
}

int icerx_encode_host(icerx_encoder *e, const uint16_t *frames, int n_frames, size_t byte_quota, uint8_t *out,
                      size_t out_stride, uint64_t *sizes, int32_t *rcs)
{
    if (!e || !frames || !out || !sizes || !rcs || n_frames < 1 || n_frames > e->max_frames || e->sample_bits != 16) {
        set_error("icerx_encode_host: invalid arguments (needs a 16-bit encoder, 1 <= n_frames <= max_frames)");
        return ICER_INVALID_INPUT;
    }
    if (e->pend.active) { set_error("icerx_encode_host: an asynchronous encode is pending on this encoder (icerx_encoder_wait)"); return ICER_INVALID_INPUT; }
    HIP_TRY(hipSetDevice(e->device));
    const size_t plane = e->w * e->h, P = (size_t)n_frames * e->channels;
    if (upload_units(e, byte_quota, nullptr)) return ICER_FATAL_ERROR;
    if (e->in.ensure((size_t)e->max_frames * e->channels * plane)) return ICER_FATAL_ERROR;
    HIP_TRY(hipMemcpy(e->in.p, frames, P * plane * 2, hipMemcpyHostToDevice));
    for (;;) {   // the device stride depends on the slot bound, which a retry may enlarge
        const size_t ds = byte_quota < e->plan.slot_bytes ? byte_quota : e->plan.slot_bytes;
        if (e->out.ensure((size_t)e->max_frames * (ds + 4))) return ICER_FATAL_ERROR;
        bool regrow = false;
        const int rc = encode_device_impl(e, e->in.p, n_frames, byte_quota, e->out.p, ds + 4,
                                          (uint64_t *)e->sizes.p, e->rcs.p, nullptr, &regrow);
        if (rc) return rc;
        if (!regrow) {
            HIP_TRY(hipMemcpy(sizes, e->sizes.p, (size_t)n_frames * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(rcs, e->rcs.p, (size_t)n_frames * 4, hipMemcpyDeviceToHost));
            // out_stride is the room the caller gives every frame: checked against the streams that came out (for one
            // frame as for many), before anything is copied
            for (int f = 0; f < n_frames; f++)
                if (sizes[f] > out_stride) {
                    set_error("icerx_encode_host: stream of frame %d (%llu bytes) longer than out_stride %zu", f, (unsigned long long)sizes[f], out_stride);
                    return ICER_OUTPUT_BUF_TOO_SMALL;
                }
            for (int f = 0; f < n_frames; f++)
                if (sizes[f]) HIP_TRY(hipMemcpy(out + (size_t)f * out_stride, e->out.p + (size_t)f * (ds + 4), sizes[f], hipMemcpyDeviceToHost));
            break;
        }
    }
    return 0;
}

// ---- a batch over the GPUs of the node (SURVEY 8b "our additions", 8e) ---------------------------------------------
int icerx_device_count(void) { return logical_device_count(); }

}  // extern "C"

// Frames are independent and a frame's transform needs the whole frame, so the batch is cut into contiguous blocks
// of frames, one per device (earlier devices take the larger blocks, icer_compression_amd/shard.py has the same rule);
// one host thread and one encoder per device, no communication between them.  Every frame's bytes, length and
// return code equal a per-frame call of icer_compress_image_uint16.
//
// A device's block is coded in sub-batches over kBatchSets (3) buffer sets, each with an encoder and an encoder stream of
// its own, plus one copy-in and one copy-out stream:
//     copy-in stream    H2D of the sub-batches ahead (set k % 3 is free when the kernels of k - 3 are done)
//     encoder streams   all kernels of sub-batch k on stream k % 3: the kernels of two sets share the chip, so that the last
//                       coding units of one launch do not leave it idle
//     copy-out stream   D2H of the streams of the finished sub-batches (exactly size[f] bytes per frame)
// (Streams are a scarce resource: the HIP runtime multiplexes them onto GPU_MAX_HW_QUEUES = 4 hardware queues by default and
// streams that share a queue run one after the other -- measured on C4 / C5: 0.80-0.85 x the device-resident rate with 4
// queues, 0.92-0.95 x with 8 -- see want_priority_streams below for what the library does about it.  Copies on the
// encoder streams instead of streams of their own -- fewer streams -- measured slower: 0.78 x on C4.)
// The encoders and their staging buffers stay alive between calls (per device, re-made when the geometry changes;
// icerx_batch_release frees them): a call allocates nothing on the device.
namespace {

// Hardware queues.  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default; a PROCESS-wide choice read
// once when the HIP runtime starts, which the library does not make behind the caller's back) PER PRIORITY LEVEL, and a new stream gets
// the least-used queue of its level's pool.  Round 5: unless the process has 6 or more queues per level, the four streams of a device's
// pipeline whose kernels must overlap (three encoders + their shared side stream) are created at another level than the host program's
// streams -- a pool of four queues of their own, shared with nothing else in the process (which level: pool_compute_level above); the two
// copy streams stay plain.  With queues to spare (GPU_MAX_HW_QUEUES >= 6) everything is a plain stream.  ICER_HIP_STREAM_PRIO=0|1 pins
// the choice.
bool want_priority_streams()
{
    if (const char *pv = getenv("ICER_HIP_STREAM_PRIO")) return atoi(pv) != 0;
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    return !(q && atoi(q) >= 6);
}
void warn_hw_queues_once()
{
    static std::atomic<bool> said{false};
    // (an informational notice, not a warning: only on request -- ICER_HIP_VERBOSE=1; INTEGRATION.md "Hardware queues" has the full story)
    const char *vb = getenv("ICER_HIP_VERBOSE");
    if (!want_priority_streams() || !vb || atoi(vb) == 0 || said.exchange(true)) return;
    fprintf(stderr, "libicer_hip: host-fed batch: GPU_MAX_HW_QUEUES is below 6, so the pipeline's encoder streams are low-priority streams (a hardware-queue pool "
                    "of their own; kernels of the program's own streams go first); GPU_MAX_HW_QUEUES=8 in the environment before the process initialises HIP makes them plain ones\n");
}

constexpr int kBatchSets = 3;

struct BatchDevice {
    std::mutex mu;                       // one batch call at a time per device
    int device = -1;                     // physical HIP device
    int logical = -1;                    // the device number the caller used (key of the pool)
    bool ready = false;                  // everything below exists (a rebuild that failed part-way leaves this false)
    icerx_encoder *enc[kBatchSets] = {};   // sub-batch k runs on encoder k % sets, each on a stream of its own: the kernels of
                                         // k + 1 fill the compute units that the last coding units of k leave idle
    int sub = 0;                         // most frames per sub-batch (= the encoders' max_frames)
    size_t quota = 0, dev_stride = 0;
    hipStream_t s_in = nullptr, s_enc[kBatchSets] = {}, s_out = nullptr;
    int sets = 0;                        // buffer sets / encoders in use (2..kBatchSets)
    DevBuf<uint16_t> in[kBatchSets];
    DevBuf<uint8_t> out[kBatchSets];
    DevBuf<unsigned long long> d_sizes[kBatchSets];
    DevBuf<int32_t> d_rcs[kBatchSets];
    uint64_t *h_sizes = nullptr;         // pinned: [sets][sub]
    int32_t *h_rcs = nullptr;            // pinned: [sets][sub]
    int *h_flag = nullptr;               // pinned: [sets][2]
    hipEvent_t in_ready[kBatchSets] = {}, coded[kBatchSets] = {}, out_done[kBatchSets] = {};
    void release()
    {
        ready = false;
        if (device >= 0) (void)hipSetDevice(device);
        for (int k = kBatchSets - 1; k >= 0; k--) {          // (encoder 0 owns the side stream the others borrow: last)
            if (enc[k]) { icerx_encoder_destroy(enc[k]); enc[k] = nullptr; }
            if (s_enc[k]) (void)hipStreamDestroy(s_enc[k]);
            s_enc[k] = nullptr;
            in[k].release(); out[k].release(); d_sizes[k].release(); d_rcs[k].release();
            if (in_ready[k]) (void)hipEventDestroy(in_ready[k]);
            if (coded[k]) (void)hipEventDestroy(coded[k]);
            if (out_done[k]) (void)hipEventDestroy(out_done[k]);
            in_ready[k] = coded[k] = out_done[k] = nullptr;
        }
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        s_in = s_out = nullptr;
        if (h_sizes) (void)hipHostFree(h_sizes);
        if (h_rcs) (void)hipHostFree(h_rcs);
        if (h_flag) (void)hipHostFree(h_flag);
        h_sizes = nullptr; h_rcs = nullptr; h_flag = nullptr;
        sub = 0; sets = 0; quota = 0; dev_stride = 0;
    }
    // after an error in the middle of a call: nothing of this device's pipeline may still be reading the caller's frames or
    // writing the caller's rows when the API returns, and the pooled encoders must be idle for the next call
    void quiesce()
    {
        if (device >= 0) (void)hipSetDevice(device);
        if (s_in) (void)hipStreamSynchronize(s_in);
        for (int k = 0; k < kBatchSets; k++) {
            if (s_enc[k]) (void)hipStreamSynchronize(s_enc[k]);
            if (enc[k]) {
                if (enc[k]->side_stream) (void)hipStreamSynchronize(enc[k]->side_stream);
                enc[k]->pend.active = false; enc[k]->wg_once = false; enc[k]->ev_pending = false;
            }
        }
        if (s_out) (void)hipStreamSynchronize(s_out);
        (void)hipGetLastError();
    }
};

std::mutex g_pool_mutex;
std::map<int, std::unique_ptr<BatchDevice>> g_pool;

// (keyed by the LOGICAL device: with ICER_HIP_VIRTUAL_DEVICES every logical device has a pipeline of its own)
BatchDevice *pool_device(int logical)
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    auto &slot = g_pool[logical];
    if (!slot) { slot.reset(new BatchDevice()); slot->logical = logical; slot->device = physical_of(logical); }
    return slot.get();
}

// frames per sub-batch: about eight sub-batches per block (measured on C4 / C5 with three sets in flight: 4 of 32 frames and
// 2 of 8 are best, 8 / 1 cost 3-12 %), each small enough that three input + three output sets are a modest share of HBM
// (ICER_HIP_BATCH_SUB pins it)
int sub_batch_frames(int cnt, size_t frame_bytes)
{
    if (const char *sv = getenv("ICER_HIP_BATCH_SUB")) { const int v = atoi(sv); if (v >= 1) return v < cnt ? v : cnt; }
    size_t by_mem = ((size_t)256 << 20) / (frame_bytes ? frame_bytes : 1);
    if (by_mem < 1) by_mem = 1;
    int by_overlap = (cnt + 7) / 8;
    if (by_overlap < 1) by_overlap = 1;
    int sub = (size_t)by_overlap < by_mem ? by_overlap : (int)by_mem;
    // (a launch of one large gray frame is bound by the chain of its biggest coding units: prefer two per launch)
    if (sub < 2 && cnt >= 2 && frame_bytes <= ((size_t)512 << 20)) sub = 2;
    return sub;
}

// The sub-batches of a block of `cnt` frames, at most `sub` frames each.  Nothing overlaps the upload of the first sub-batch
// or the download of the last one, so the block may start (and, with ramp = 2, end) with smaller ones: 1, 2, 4, ... frames
// up to `sub` (ICER_HIP_BATCH_RAMP=0: all of `sub` frames -- the default, see below --, 1: rising at the start, 2: and falling
// at the end).
void sub_batch_plan(int cnt, int sub, std::vector<int> *first, std::vector<int> *count)
{
    int ramp = 0;                           // (measured, profiles/r04_logs/r04_a_host_batch_ramp.log: C5 the same with 0 / 1 / 2, C4 9 % slower with a ramp)
    if (const char *rv = getenv("ICER_HIP_BATCH_RAMP")) { const int v = atoi(rv); if (v >= 0 && v <= 2) ramp = v; }
    std::vector<int> head, tail;
    int left = cnt;
    if (ramp >= 1 && sub >= 2 && cnt >= 2 * sub)
        for (int n = 1; n < sub && left > sub; n *= 2) { head.push_back(n); left -= n; }
    if (ramp >= 2 && sub >= 2 && left >= 2 * sub)
        for (int n = 1; n < sub && left > sub; n *= 2) { tail.push_back(n); left -= n; }
    std::vector<int> sizes(head);
    while (left > 0) { const int n = left < sub ? left : sub; sizes.push_back(n); left -= n; }
    for (size_t i = tail.size(); i-- > 0;) sizes.push_back(tail[i]);
    first->clear(); count->clear();
    int at = 0;
    for (int n : sizes) { first->push_back(at); count->push_back(n); at += n; }
}

int batch_rebuild(BatchDevice *b, size_t w, size_t h, int channels, int stages, int filt, int segments, int sub, int sets)
{
    b->sub = sub;
    b->sets = sets;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(create_level_stream(&b->s_in, want_priority_streams() ? getenv("ICER_HIP_COPY_LEVEL") : nullptr));
    HIP_TRY(create_level_stream(&b->s_out, want_priority_streams() ? getenv("ICER_HIP_COPY_LEVEL") : nullptr));
    for (int k = 0; k < sets; k++) {
        const int rc = icerx_encoder_create(&b->enc[k], b->logical, w, h, channels, stages, filt, segments, sub);
        if (rc) return rc;
        b->enc[k]->sleepy_wait = true;
        // (streams are scarce -- hardware queues, see above: the pipeline's encoders run their short list kernels on ONE side
        // stream, the first encoder's; fork / join events stay per encoder)
        if (k > 0 && b->enc[k]->side_stream && b->enc[0]->side_stream) {
            (void)hipStreamDestroy(b->enc[k]->side_stream);
            b->enc[k]->side_stream = b->enc[0]->side_stream;
            b->enc[k]->side_stream_borrowed = true;
        }
        // (the encoders' streams and their shared side stream at a level of their own unless the process has hardware queues to
        // spare: want_priority_streams, pool_compute_level)
        HIP_TRY(create_level_stream(&b->s_enc[k], want_priority_streams() ? pool_compute_level() : nullptr));
        if (k == 0 && want_priority_streams() && b->enc[0]->side_stream && !b->enc[0]->side_stream_borrowed) {
            hipStream_t ss = nullptr;                           // (the shared side stream at the encoders' level)
            if (create_level_stream(&ss, pool_compute_level()) == hipSuccess) { (void)hipStreamDestroy(b->enc[0]->side_stream); b->enc[0]->side_stream = ss; }
        }
        if (b->in[k].ensure((size_t)sub * channels * w * h) || b->d_sizes[k].ensure(sub) || b->d_rcs[k].ensure(sub)) return ICER_FATAL_ERROR;
        HIP_TRY(hipEventCreateWithFlags(&b->in_ready[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b->coded[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b->out_done[k], hipEventDisableTiming));
    }
    HIP_TRY(hipHostMalloc((void **)&b->h_sizes, (size_t)sets * (size_t)sub * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&b->h_rcs, (size_t)sets * (size_t)sub * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&b->h_flag, (size_t)sets * 2 * sizeof(int), hipHostMallocDefault));
    b->quota = (size_t)-1;
    return 0;
}

int batch_prepare(BatchDevice *b, size_t w, size_t h, int channels, int stages, int filt, int segments, size_t quota, int sub, int sets)
{
    const icerx_encoder *e = b->enc[0];
    if (!b->ready || !e || b->sets != sets || e->w != w || e->h != h || e->channels != channels || e->stages != stages || e->filt != filt ||
        e->segments != segments || e->sample_bits != 16 || b->sub != sub) {
        b->release();
        // (a rebuild that fails part-way -- the second encoder, a staging buffer, an event -- must not leave a half-built
        // pipeline that the next call with the same geometry would take for a complete one)
        const int rc = batch_rebuild(b, w, h, channels, stages, filt, segments, sub, sets);
        if (rc) { b->release(); return rc; }
        b->ready = true;
    }
    HIP_TRY(hipSetDevice(b->device));
    // slots for this quota, then output rows that hold the longest possible stream
    size_t ds = 0;
    for (int k = 0; k < sets; k++) {
        if (upload_units(b->enc[k], quota, b->s_enc[k])) return ICER_FATAL_ERROR;
        const size_t d = (quota < b->enc[k]->plan.slot_bytes ? quota : b->enc[k]->plan.slot_bytes) + 4;
        if (d > ds) ds = d;
    }
    if (b->quota != quota || b->dev_stride < ds) {
        for (int k = 0; k < sets; k++) if (b->out[k].ensure((size_t)sub * ds)) return ICER_FATAL_ERROR;
        b->dev_stride = ds;
        b->quota = quota;
    }
    return 0;
}

// one device's block of the batch: frames [0, cnt) at `frames`, rows of `out` (the caller holds b->mu)
int batch_run(BatchDevice *b, const uint16_t *frames, int cnt, size_t frame_elems, int sub, int S, size_t quota, uint8_t *out, size_t out_stride,
              uint64_t *sizes, int32_t *rcs)
{
    int rc = 0;
    std::vector<int> first, count;
    sub_batch_plan(cnt, sub, &first, &count);
    const int K = (int)first.size();
    // enqueue sub-batch k: its copy-in, then its kernels
    auto issue = [&](int k) -> int {
        const int s = k % S, n = count[(size_t)k];
        // in[s] is free once the kernels of k - S are done, out[s] once the copy-out of k - S is
        if (k >= S) HIP_TRY(hipStreamWaitEvent(b->s_in, b->coded[s], 0));
        HIP_TRY(hipMemcpyAsync(b->in[s].p, frames + (size_t)first[(size_t)k] * frame_elems, (size_t)n * frame_elems * 2, hipMemcpyHostToDevice, b->s_in));
        HIP_TRY(hipEventRecord(b->in_ready[s], b->s_in));
        HIP_TRY(hipStreamWaitEvent(b->s_enc[s], b->in_ready[s], 0));
        if (k >= S) HIP_TRY(hipStreamWaitEvent(b->s_enc[s], b->out_done[s], 0));
        b->enc[s]->wg_once = false;
        return encode_begin(b->enc[s], b->in[s].p, n, quota, b->out[s].p, b->dev_stride, (uint64_t *)b->d_sizes[s].p, b->d_rcs[s].p, b->s_enc[s],
                            b->h_flag + 2 * s, b->coded[s], nullptr);
    };
    for (int q = 0; q < S; q++) if (accumulate_timing(b->enc[q])) return ICER_FATAL_ERROR;
    for (int k = 0; k < K && k < S; k++) if ((rc = issue(k))) return rc;
    for (int k = 0; k < K; k++) {
        const int s = k % S, n = count[(size_t)k], f0 = first[(size_t)k];
        icerx_encoder *e = b->enc[s];
        if (wait_event(b->coded[s], true)) return ICER_FATAL_ERROR;
        int v = encode_verdict(e, n, b->h_flag + 2 * s);
        if (v < 0) return v;
        if (v == 1) {
            // rare: larger slots, or the barrier-only coder after a time-out.  Let everything in flight finish (k + 1 was
            // enqueued with the old slot table; its own verdict is read in its turn), then code k again, synchronously.
            HIP_TRY(hipDeviceSynchronize());
            for (;;) {
                if (upload_units(e, quota, b->s_enc[s])) return ICER_FATAL_ERROR;
                if (e->slots.ensure((size_t)e->max_frames * e->plan.slot_bytes)) return ICER_FATAL_ERROR;
                const size_t ds = (quota < e->plan.slot_bytes ? quota : e->plan.slot_bytes) + 4;
                if (b->dev_stride < ds) {
                    for (int q = 0; q < S; q++) if (b->out[q].ensure((size_t)sub * ds)) return ICER_FATAL_ERROR;
                    b->dev_stride = ds;
                    // (the sub-batches after k that were in flight wrote rows of the old stride into buffers that are gone: enqueue them again)
                    for (int q = k + 1; q < K && q < k + S; q++) if ((rc = issue(q))) return rc;
                }
                if ((rc = encode_begin(e, b->in[s].p, n, quota, b->out[s].p, b->dev_stride, (uint64_t *)b->d_sizes[s].p, b->d_rcs[s].p, b->s_enc[s],
                                       b->h_flag + 2 * s, b->coded[s], nullptr))) return rc;
                if (wait_event(b->coded[s], true)) return ICER_FATAL_ERROR;
                v = encode_verdict(e, n, b->h_flag + 2 * s);
                if (v < 0) return v;
                if (v == 0) break;
            }
            e->wg_once = false;
        }
        // lengths and return codes of k, then exactly the bytes of every stream
        HIP_TRY(hipStreamWaitEvent(b->s_out, b->coded[s], 0));
        HIP_TRY(hipMemcpyAsync(b->h_sizes + (size_t)s * sub, b->d_sizes[s].p, (size_t)n * 8, hipMemcpyDeviceToHost, b->s_out));
        HIP_TRY(hipMemcpyAsync(b->h_rcs + (size_t)s * sub, b->d_rcs[s].p, (size_t)n * 4, hipMemcpyDeviceToHost, b->s_out));
        HIP_TRY(hipStreamSynchronize(b->s_out));
        for (int f = 0; f < n; f++) {
            const uint64_t sz = b->h_sizes[(size_t)s * sub + f];
            sizes[(size_t)f0 + f] = sz;
            rcs[(size_t)f0 + f] = b->h_rcs[(size_t)s * sub + f];
            if (sz > out_stride) {
                set_error("icerx_compress_batch_uint16: stream of frame %d (%llu bytes) longer than out_stride %zu", f0 + f, (unsigned long long)sz, out_stride);
                return ICER_OUTPUT_BUF_TOO_SMALL;
            }
            if (sz) HIP_TRY(hipMemcpyAsync(out + ((size_t)f0 + f) * out_stride, b->out[s].p + (size_t)f * b->dev_stride, sz, hipMemcpyDeviceToHost, b->s_out));
        }
        HIP_TRY(hipEventRecord(b->out_done[s], b->s_out));
        if (k + S < K && (rc = issue(k + S))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(b->s_out));
    return 0;
}

int batch_on_device(BatchDevice *b, const uint16_t *frames, int cnt, size_t w, size_t h, int channels, int stages, int filt, int segments,
                    size_t quota, uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs)
{
    std::lock_guard<std::mutex> lk(b->mu);
    const size_t frame_elems = w * h * (size_t)channels;
    const int sub = sub_batch_frames(cnt, frame_elems * 2);
    int sets = 3;                        // (ICER_HIP_BATCH_SETS: 2 or 3)
    if (const char *sv = getenv("ICER_HIP_BATCH_SETS")) { const int v = atoi(sv); if (v >= 2 && v <= kBatchSets) sets = v; }
    int rc = batch_prepare(b, w, h, channels, stages, filt, segments, quota, sub, sets);
    if (rc) return rc;
    rc = batch_run(b, frames, cnt, frame_elems, sub, sets, quota, out, out_stride, sizes, rcs);
    // every error exit of the pipeline ends here: drain the device's streams before the caller gets its buffers back
    if (rc) b->quiesce();
    return rc;
}

}  // namespace

extern "C" {

void icerx_batch_release(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    for (auto &kv : g_pool) if (kv.second) { std::lock_guard<std::mutex> lk2(kv.second->mu); kv.second->release(); }
}

int icerx_compress_batch_uint16_devices(const uint16_t *frames, int n_frames, size_t w, size_t h, int channels, int stages, int filt,
                                        int segments, size_t byte_quota, uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs,
                                        const int *devices, int n_devices)
{
    if (!frames || !out || !sizes || !rcs || n_frames < 1 || !devices || n_devices < 1) { set_error("icerx_compress_batch_uint16: invalid arguments"); return ICER_INVALID_INPUT; }
    const int have = icerx_device_count();
    if (have <= 0) { set_error("no usable HIP device; this library has no CPU path"); return ICER_FATAL_ERROR; }
    for (int d = 0; d < n_devices; d++)
        if (devices[d] < 0 || devices[d] >= have) { set_error("icerx_compress_batch_uint16: device %d of %d present", devices[d], have); return ICER_INVALID_INPUT; }
    warn_hw_queues_once();
    const int g = n_devices > n_frames ? n_frames : n_devices;
    try {
        std::vector<int> rc((size_t)g, 0);
        std::vector<std::string> err((size_t)g);
        const size_t frame_elems = w * h * (size_t)channels;
        auto work = [&](int d) {
            const int base = n_frames / g, extra = n_frames % g;
            const int lo = d * base + (d < extra ? d : extra), cnt = base + (d < extra ? 1 : 0);
            int r;
            try {
                g_last_error.clear();                            // (thread-local: what this block's failure leaves, if anything)
                r = batch_on_device(pool_device(devices[d]), frames + (size_t)lo * frame_elems, cnt, w, h, channels, stages, filt, segments,
                                    byte_quota, out + (size_t)lo * out_stride, out_stride, sizes + lo, rcs + lo);
                if (r) err[(size_t)d] = g_last_error.empty() ? "error code " + std::to_string(r) + " (lib_icer argument check)" : g_last_error;
            } catch (const std::exception &ex) { r = ICER_FATAL_ERROR; err[(size_t)d] = ex.what(); }
            rc[(size_t)d] = r;
        };
        if (g == 1) work(0);
        else {
            // one host thread per device, on the cores next to that device (its NUMA node): the thread allocates the
            // device's page-locked staging words and polls its events
            std::vector<std::thread> th;
            for (int d = 0; d < g; d++)
                th.emplace_back([&work, devices, d] { (void)pin_thread_near_device(physical_of(devices[d])); work(d); });
            for (auto &t : th) t.join();
        }
        int first = 0;
        std::string all;
        for (int d = 0; d < g; d++)
            if (rc[(size_t)d]) { if (!first) first = rc[(size_t)d]; all += (all.empty() ? "device " : "; device ") + std::to_string(devices[d]) + ": " + err[(size_t)d]; }
        if (first) { set_error("%s", all.c_str()); return first; }
        return 0;
    } catch (const std::exception &ex) {          // (std::thread / std::vector: nothing may cross the C boundary)
        set_error("icerx_compress_batch_uint16: %s", ex.what());
        return ICER_FATAL_ERROR;
    }
}

int icerx_compress_batch_uint16(const uint16_t *frames, int n_frames, size_t w, size_t h, int channels, int stages, int filt,
                                int segments, size_t byte_quota, uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs,
                                int n_gpus)
{
    if (n_gpus < 0) { set_error("icerx_compress_batch_uint16: invalid arguments"); return ICER_INVALID_INPUT; }
    const int have = icerx_device_count();
    if (have <= 0) { set_error("no usable HIP device; this library has no CPU path"); return ICER_FATAL_ERROR; }
    const int g = n_gpus == 0 || n_gpus > have ? have : n_gpus;
    int devices[64];
    for (int d = 0; d < g && d < 64; d++) devices[d] = d;
    return icerx_compress_batch_uint16_devices(frames, n_frames, w, h, channels, stages, filt, segments, byte_quota, out, out_stride, sizes,
                                               rcs, devices, g < 64 ? g : 64);
}

// Page-lock a caller buffer (frames in, streams out) so that the host-buffer entry points move it at PCIe speed by DMA
// instead of through the runtime's staging copies (hipHostRegister / hipHostUnregister behind a C ABI: a C caller need
// not link the HIP runtime).  The caller unpins before it frees the memory.
int icerx_pin_host(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return ICER_INVALID_INPUT;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}
int icerx_unpin_host(void *ptr)
{
    if (!ptr) return ICER_INVALID_INPUT;
    HIP_TRY(hipHostUnregister(ptr));
    return 0;
}

int icerx_get_coefficients(icerx_encoder *e, int frame, int channel, uint16_t *dst)
{
    if (!e || frame < 0 || frame >= e->max_frames || channel < 0 || channel >= e->channels) return ICER_INVALID_INPUT;
    HIP_TRY(hipSetDevice(e->device));
    const size_t plane = e->w * e->h;
    HIP_TRY(hipMemcpy(dst, e->coef.p + ((size_t)frame * e->channels + channel) * plane, plane * 2, hipMemcpyDeviceToHost));
    return 0;
}

int icerx_timing_enable(icerx_encoder *e, int on)
{
    if (!e) return ICER_INVALID_INPUT;
    e->timing = on != 0;
    return 0;
}

int icerx_timing_read(icerx_encoder *e, double ms[ICERX_NUM_STAGES], uint64_t *calls, int reset)
{
    if (!e) return ICER_INVALID_INPUT;
    HIP_TRY(hipSetDevice(e->device));
    if (accumulate_timing(e)) return ICER_FATAL_ERROR;
    for (int i = 0; i < ICERX_NUM_STAGES; i++) ms[i] = e->ms[i];
    *calls = e->timed_calls;
    if (reset) { for (auto &m : e->ms) m = 0; e->timed_calls = 0; }
    return 0;
}

// the same counters summed over every encoder of the process (the lib_icer-shaped entry points use an internal one)
int icerx_process_stats(uint64_t out[4])
{
    if (!out) return ICER_INVALID_INPUT;
    out[0] = g_stats[0]; out[1] = g_stats[1]; out[2] = g_stats[2]; out[3] = 0;
    return 0;
}

int icerx_encoder_stats(icerx_encoder *e, uint64_t out[4])
{
    if (!e || !out) return ICER_INVALID_INPUT;
    out[0] = e->n_timeouts; out[1] = e->n_fallbacks; out[2] = e->n_slot_retries; out[3] = (uint64_t)e->coder_mode;
    return 0;
}

int icerx_encoder_routing(icerx_encoder *e, uint64_t out[2])
{
    if (!e || !out) return ICER_INVALID_INPUT;
    out[0] = e->n_routed_units; out[1] = e->n_routed_launches;
    return 0;
}

int icerx_encoder_launch_info(icerx_encoder *e, uint32_t out[4])
{
    if (!e || !out) return ICER_INVALID_INPUT;
    out[0] = e->last_split ? 1u : 0u; out[1] = e->last_subs; out[2] = (uint32_t)e->last_waves; out[3] = e->last_routed ? 1u : 0u;
    return 0;
}

int icerx_encoder_parts(icerx_encoder *e) { return e ? e->last_parts : ICER_INVALID_INPUT; }

int icerx_info(icerx_encoder *e, uint32_t *units_per_frame, uint32_t *slot_bits_per_pixel, uint64_t *slot_bytes_per_frame)
{
    if (!e) return ICER_INVALID_INPUT;
    *units_per_frame = (uint32_t)e->plan.units.size();
    *slot_bits_per_pixel = e->bits_per_pixel;
    *slot_bytes_per_frame = e->plan.slot_bytes;
    return 0;
}

#ifdef ICER_PHASE_TIMERS
// profiling build only: summed s_memtime cycles per coder phase over all units since the last reset
int icerx_prof_read(icerx_encoder *e, uint64_t out[9 * 32], int reset)
{
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(out, e->prof.p, 9 * 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(e->prof.p, 0, kProfWords * sizeof(uint64_t)));
    return 0;
}
// the same for the small window coder beside the pipeline (code_units_list_kernel): 9 rows of 32
int icerx_prof_read_wgs(icerx_encoder *e, uint64_t out[9 * 32], int reset)
{
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(out, e->prof.p + kProfWgsOffset, 9 * 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(e->prof.p + kProfWgsOffset, 0, 9 * 32 * sizeof(uint64_t)));
    return 0;
}
// per list entry of code_units_list_kernel (frame 0, position < kListTrace): start / end (100 MHz wall clock), workgroup | unit << 32,
// lsb | level << 8 | subband << 16 | segment << 24 | chunks << 32
int icerx_prof_list_trace(icerx_encoder *e, uint64_t *out, int n_entries, int reset)
{
    if (!e || !out || n_entries < 0) return ICER_INVALID_INPUT;
    HIP_TRY(hipSetDevice(e->device));
    if (n_entries > kListTrace) n_entries = kListTrace;
    HIP_TRY(hipMemcpy(out, e->prof.p + kProfWgsOffset + 9 * 32, (size_t)n_entries * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(e->prof.p + kProfWgsOffset + 9 * 32, 0, (size_t)kListTrace * 4 * sizeof(uint64_t)));
    return n_entries;
}
// per workgroup of frame 0 (launch position b < kTraceUnits): start / end (100 MHz wall clock), HW_ID | XCC_ID << 32, unit index
int icerx_prof_trace(icerx_encoder *e, uint64_t *out, int n_blocks)
{
    if (!e || !out || n_blocks < 0) return ICER_INVALID_INPUT;
    HIP_TRY(hipSetDevice(e->device));
    if (n_blocks > kTraceUnits) n_blocks = kTraceUnits;
    HIP_TRY(hipMemcpy(out, e->prof.p + 9 * 32, (size_t)n_blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t hw[16];
    HIP_TRY(hipMemcpy(hw, e->prof.p + 9 * 32 + 4 * kTraceUnits, sizeof hw, hipMemcpyDeviceToHost));
    fprintf(stderr, "workgroup 0, wave -> SIMD:");
    for (int w = 0; w < kUnitWavesLarge; w++) fprintf(stderr, " %d->%d", w, (int)((hw[w] >> 4) & 3));
    fprintf(stderr, "\n");
    return n_blocks;
}
#endif

// ---- lib_icer drop-in entry points -------------------------------------------------------------
static icerx_encoder *g_cached = nullptr;

// `planes` are uint16_t* (sample_bits 16) or uint8_t* (sample_bits 8) host pointers
static int compress_planes(void *const planes[], int channels, size_t w, size_t h, int stages, int filt, int segments,
                           icer_output_data_buf_typedef *od, int sample_bits)
{
    std::lock_guard<std::recursive_mutex> lk(g_mutex);
    if (!od) return ICER_INVALID_INPUT;
    icerx_encoder *e = g_cached;
    if (!e || e->w != w || e->h != h || e->channels != channels || e->stages != stages || e->filt != filt ||
        e->segments != segments || e->sample_bits != sample_bits) {
        if (e) { icerx_encoder_destroy(e); g_cached = nullptr; }
        const char *dev = getenv("ICER_HIP_DEVICE");
        const int rc = icerx_encoder_create_ex(&e, dev ? atoi(dev) : 0, w, h, channels, stages, filt, segments, 1, sample_bits);
        if (rc == ICER_PACKET_COUNT_EXCEEDED || rc == ICER_TOO_MANY_SEGMENTS) {
            // The reference finds its packet table too small, or the segment grid impossible, only after the transform
            // and the LL-mean check (icer_color.c:31-131, icer_compress.c:279-400): an integer overflow there is what it
            // reports.  Run those on the planes as gray frames with one segment and look at their return codes.
            icerx_encoder *t = nullptr;
            if (icerx_encoder_create_ex(&t, dev ? atoi(dev) : 0, w, h, 1, stages, filt, 1, channels, sample_bits) == 0) {
                const size_t plane = w * h, n = (size_t)channels * plane;
                std::vector<int32_t> rcs(channels, 0);
                int r = 0;
                if (t->in.ensure(n) || (sample_bits == 8 && t->in8.ensure(n))) r = ICER_FATAL_ERROR;
                for (int c = 0; c < channels && !r; c++) {
                    const hipError_t he = sample_bits == 8 ? hipMemcpy(t->in8.p + (size_t)c * plane, planes[c], plane, hipMemcpyHostToDevice)
                                                           : hipMemcpy(t->in.p + (size_t)c * plane, planes[c], plane * 2, hipMemcpyHostToDevice);
                    if (he != hipSuccess) r = ICER_FATAL_ERROR;
                }
                if (!r) {
                    if (sample_bits == 8)
                        hipLaunchKernelGGL(widen_s8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, nullptr, t->in8.p, t->in.p, n);
                    for (;;) {
                        if (upload_units(t, 64, nullptr) || t->out.ensure((size_t)channels * 128)) { r = ICER_FATAL_ERROR; break; }
                        bool regrow = false;
                        r = encode_device_impl(t, t->in.p, channels, 64, t->out.p, 128, (uint64_t *)t->sizes.p, t->rcs.p, nullptr, &regrow);
                        if (r || !regrow) break;
                    }
                    if (!r && hipMemcpy(rcs.data(), t->rcs.p, sizeof(int32_t) * channels, hipMemcpyDeviceToHost) != hipSuccess) r = ICER_FATAL_ERROR;
                }
                icerx_encoder_destroy(t);
                if (r) return r;
                for (int c = 0; c < channels; c++) if (rcs[c] == ICER_INTEGER_OVERFLOW) return ICER_INTEGER_OVERFLOW;
            }
        }
        if (rc) return rc;
        g_cached = e;
    }
    // The call as a reference user makes it (example/src/example_encode.c:36-77): pageable caller memory in, stream and
    // coefficient planes out.  Everything runs on a stream of the encoder's own; the coefficient planes -- final once the
    // transform is done, 0.2 ms into the call -- go back to the caller's image on a second stream WHILE the coder runs, so
    // that of the three transfers only the upload and the (short) stream download are not hidden.
    const size_t plane = w * h, quota = od->size_allocated;
    HIP_TRY(hipSetDevice(e->device));
    if (!e->io_stream) HIP_TRY(hipStreamCreateWithFlags(&e->io_stream, hipStreamNonBlocking));
    if (!e->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!e->coef_ready) HIP_TRY(hipEventCreateWithFlags(&e->coef_ready, hipEventDisableTiming));
    hipStream_t st = e->io_stream;
    if (e->in.ensure((size_t)channels * plane)) return ICER_FATAL_ERROR;
    if (sample_bits == 16) {
        for (int c = 0; c < channels; c++)
            HIP_TRY(hipMemcpyAsync(e->in.p + (size_t)c * plane, planes[c], plane * 2, hipMemcpyHostToDevice, st));
    } else {
        if (e->in8.ensure((size_t)channels * plane)) return ICER_FATAL_ERROR;
        for (int c = 0; c < channels; c++)
            HIP_TRY(hipMemcpyAsync(e->in8.p + (size_t)c * plane, planes[c], plane, hipMemcpyHostToDevice, st));
        const size_t n = (size_t)channels * plane;
        hipLaunchKernelGGL(widen_s8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, e->in8.p, e->in.p, n);
    }
    uint64_t size = 0;
    int32_t rc = 0;
    bool coef_back = false;              // the coefficient planes are in the caller's image already
    for (bool first = true;; first = false) {
        if (upload_units(e, quota, st)) return ICER_FATAL_ERROR;
        const size_t ds = (quota < e->plan.slot_bytes ? quota : e->plan.slot_bytes) + 4;
        if (e->out.ensure(ds)) return ICER_FATAL_ERROR;
        bool regrow = false;
        int r;
        if (first) {
            if (accumulate_timing(e)) return ICER_FATAL_ERROR;
            e->wg_once = false;
            r = encode_begin(e, e->in.p, 1, quota, e->out.p, ds, (uint64_t *)e->sizes.p, e->rcs.p, st, e->h_flag, e->done, nullptr);
            if (r) return r;
            // beside the coder: the frame's status (an aborted frame keeps the caller's planes, see below), then the planes
            int skip = 0;
            const int *d_skip = e->flags.p + 2 * (size_t)e->max_frames * channels;
            HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->coef_ready, 0));
            HIP_TRY(hipMemcpyAsync(&skip, d_skip, sizeof(int), hipMemcpyDeviceToHost, e->copy_stream));
            HIP_TRY(hipStreamSynchronize(e->copy_stream));
            if (!skip) {
                if (sample_bits == 16) {
                    for (int c = 0; c < channels; c++)
                        HIP_TRY(hipMemcpyAsync(planes[c], e->coef.p + (size_t)c * plane, plane * 2, hipMemcpyDeviceToHost, e->copy_stream));
                } else {
                    // what the reference leaves in the caller's image: int8 sign-magnitude bytes; narrowed on the device (in8 is
                    // free again: the widening kernel has run, coef_ready lies behind it on the encode stream)
                    const size_t n = (size_t)channels * plane;
                    hipLaunchKernelGGL(narrow_sm8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, e->copy_stream,
                                       reinterpret_cast<const uint16_t *>(e->coef.p), e->in8.p, n);
                    for (int c = 0; c < channels; c++)
                        HIP_TRY(hipMemcpyAsync(planes[c], e->in8.p + (size_t)c * plane, plane, hipMemcpyDeviceToHost, e->copy_stream));
                }
                HIP_TRY(hipStreamSynchronize(e->copy_stream));
                coef_back = true;
            }
            r = encode_device_impl(e, e->in.p, 1, quota, e->out.p, ds, (uint64_t *)e->sizes.p, e->rcs.p, st, &regrow, true);
        } else
            r = encode_device_impl(e, e->in.p, 1, quota, e->out.p, ds, (uint64_t *)e->sizes.p, e->rcs.p, st, &regrow);
        if (r) return r;
        if (!regrow) break;
    }
    HIP_TRY(hipMemcpyAsync(&size, e->sizes.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&rc, e->rcs.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (rc == ICER_INTEGER_OVERFLOW) {
        // The reference aborts before any output; it leaves transformed (not sign-magnitude) data in
        // the planes it had already processed: channels up to the first DWT overflow, or all of them
        // when only the LL-mean check failed (icer_color.c:347-381).  (uint8 twins: that data went through
        // truncating int8 stores after the overflow; we leave the caller's planes untouched instead.)
        if (sample_bits == 8) return rc;
        std::vector<int> fl(2 * (size_t)channels);
        HIP_TRY(hipMemcpy(fl.data(), e->flags.p, sizeof(int) * channels, hipMemcpyDeviceToHost));
        int last = channels - 1;
        for (int c = 0; c < channels; c++) if (fl[c]) { last = c; break; }
        // (the detail bands are normally stored as sign-magnitude words: transform once more, plain)
        {
            size_t cw2 = w, ch2 = h;
            int *scratch_flags = e->flags.p;
            launch_dwt(e, reinterpret_cast<const uint16_t *>(e->in.p), 1, st, 0, scratch_flags, &cw2, &ch2);
            HIP_TRY(hipStreamSynchronize(st));
        }
        for (int c = 0; c <= last; c++)
            HIP_TRY(hipMemcpy(planes[c], e->coef.p + (size_t)c * plane, plane * 2, hipMemcpyDeviceToHost));
        return rc;
    }
    if (size) HIP_TRY(hipMemcpyAsync(od->rearrange_start, e->out.p, size, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!coef_back) {                    // (not reached in practice: a frame that was not aborted has its planes back already)
        if (sample_bits == 16) {
            for (int c = 0; c < channels; c++)
                HIP_TRY(hipMemcpy(planes[c], e->coef.p + (size_t)c * plane, plane * 2, hipMemcpyDeviceToHost));
        } else {
            const size_t n = (size_t)channels * plane;
            hipLaunchKernelGGL(narrow_sm8_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st,
                               reinterpret_cast<const uint16_t *>(e->coef.p), e->in8.p, n);
            for (int c = 0; c < channels; c++)
                HIP_TRY(hipMemcpyAsync(planes[c], e->in8.p + (size_t)c * plane, plane, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
    }
    od->size_used = size;
    return rc;
}

int icer_compress_image_uint16(uint16_t *image, size_t image_w, size_t image_h, uint8_t stages,
                               enum icer_filter_types filt, uint8_t segments, icer_output_data_buf_typedef *output_data)
{
    void *planes[1] = {image};
    return compress_planes(planes, 1, image_w, image_h, stages, (int)filt, segments, output_data, 16);
}

int icer_compress_image_yuv_uint16(uint16_t *y_channel, uint16_t *u_channel, uint16_t *v_channel, size_t image_w,
                                   size_t image_h, uint8_t stages, enum icer_filter_types filt, uint8_t segments,
                                   icer_output_data_buf_typedef *output_data)
{
    void *planes[3] = {y_channel, u_channel, v_channel};
    return compress_planes(planes, 3, image_w, image_h, stages, (int)filt, segments, output_data, 16);
}

int icer_compress_image_uint8(uint8_t *image, size_t image_w, size_t image_h, uint8_t stages, enum icer_filter_types filt,
                              uint8_t segments, icer_output_data_buf_typedef *output_data)
{
    void *planes[1] = {image};
    return compress_planes(planes, 1, image_w, image_h, stages, (int)filt, segments, output_data, 8);
}

int icer_compress_image_yuv_uint8(uint8_t *y_channel, uint8_t *u_channel, uint8_t *v_channel, size_t image_w, size_t image_h,
                                  uint8_t stages, enum icer_filter_types filt, uint8_t segments,
                                  icer_output_data_buf_typedef *output_data)
{
    void *planes[3] = {y_channel, u_channel, v_channel};
    return compress_planes(planes, 3, image_w, image_h, stages, (int)filt, segments, output_data, 8);
}

}  // extern "C"
