// kernels.hpp -- gfx950 kernels of the ICER encoder (included once by api.hip).
//
//   dwt_tile_kernel                     one stage of the 2-D lifting transform, fused LDS tile pass (a-1..a-4)
//   ll_sum_kernel / ll_mean_kernel      LL mean (a-5)
//   finalize_ll_kernel                  LL mean removal + sign-magnitude of LL (a-5, a-6; the detail bands: at the DWT store)
//   code_units_kernel                   context modeller + entropy coder + framing (a-9..a-15)
//   scan_kernel / gather_kernel         quota cut + final stream order (a-16, a-17)
// HBM layout: planes are row-major int16/uint16 with row stride = image width; plane p of a batch
// is frame-major then channel (p = frame * channels + chan).
#pragma once
#include <hip/hip_runtime.h>

#include "assemble_core.hpp"
#include "coder_core.hpp"
#include "coder_wg.hpp"
#include "coder_wg_small.hpp"
#include "dwt_tile.hpp"
#include "events.hpp"
#include "plan.hpp"

namespace icer {

// ------------------------------------------------------------------------------------------ DWT
// One stage of the 2-D lifting transform, fused and LDS-staged (dwt_tile.hpp): a workgroup loads its input window
// (tile + filter halo) with coalesced row reads, lifts the rows, lifts the columns, and stores the four bands.
// grid = (ceil(nlw / 64), ceil(nlh / 16), planes), block = 256.
__global__ void __launch_bounds__(kTileThreads)
dwt_tile_kernel(DwtStageArgs a, size_t src_plane, size_t coef_plane, size_t ll_plane, int *__restrict__ ovf)
{
    __shared__ union { DwtTileShared gen; DwtFastShared fast; } u;
    a.src += blockIdx.z * src_plane;
    a.coef += blockIdx.z * coef_plane;
    a.ll += blockIdx.z * ll_plane;
    const int tx = blockIdx.x, ty = blockIdx.y, t = threadIdx.x;
    if (dwt_tile_is_interior(a, tx, ty)) {                  // (uniform per workgroup) nine tenths of a large stage's tiles
        bool o = dwt_fast_rows_step1(u.fast, a, tx, ty, t);
        __syncthreads();
        o |= dwt_fast_rows_step2(u.fast, a, t);
        __syncthreads();
        o |= dwt_fast_cols_step1(u.fast, a, t);
        __syncthreads();
        o |= dwt_fast_cols_step2(u.fast, a, tx, ty, t);
        if (o) atomicOr(&ovf[blockIdx.z], 1);
        return;
    }
    DwtTileShared &sh = u.gen;
    dwt_tile_load(sh, a, tx, ty, t);
    __syncthreads();
    bool o = dwt_tile_rows_step1(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_rows_step2(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_cols_step1(sh, a, tx, ty, t);
    __syncthreads();
    o |= dwt_tile_cols_step2(sh, a, tx, ty, t);
    if (o) atomicOr(&ovf[blockIdx.z], 1);
}

// ------------------------------------------------------------------------------------------ LL mean
// sum of the LL samples read as unsigned 16-bit (icer_compress.c:286-296); uint8 twins: as unsigned 8-bit
// (icer_compress.c:24-33; `mask` = 0xFF, the samples are sign-extended int8).  grid = (blocks, planes)
__global__ void __launch_bounds__(256)
ll_sum_kernel(const uint16_t *__restrict__ coef, size_t plane, uint32_t stride, uint32_t llw, uint32_t llh,
              unsigned long long *__restrict__ sums, uint32_t mask)
{
    const uint16_t *p = coef + blockIdx.y * plane;
    const uint32_t n = llw * llh;
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t r = i / llw, c = i - r * llw;
        acc += p[(size_t)r * stride + c] & mask;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&sums[blockIdx.y], acc);
}

// mean = sum / (llw*llh) through a uint16_t, must fit int16 (icer_compress.c:298-302); uint8 twins: through a
// uint8_t, must fit int8 (icer_compress.c:35-38).  One thread per plane.
__global__ void ll_mean_kernel(const unsigned long long *__restrict__ sums, uint32_t n_planes, uint32_t ll_count,
                               uint16_t *__restrict__ means, int *__restrict__ mean_ovf, int sample_bits)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_planes) return;
    const uint16_t m = sample_bits == 8 ? (uint16_t)(uint8_t)(sums[p] / ll_count) : (uint16_t)(sums[p] / ll_count);
    means[p] = m;
    mean_ovf[p] = m > (sample_bits == 8 ? 127 : 32767) ? 1 : 0;
}

// LL -= mean (int16 wrap), then two's complement -> sign-magnitude, over the LL rectangle only: the detail bands were
// stored as sign-magnitude words by the DWT (dwt_tile.hpp).  (icer_compress.c:304-313, icer_wavelet.c:871-877;
// uint8 twins icer_compress.c:40-51, icer_wavelet.c:852-858: the subtraction wraps at 8 bits.)
// grid = (ceil(llw/64), ceil(llh/4), planes), block = 256.  An aborted frame keeps its raw LL band.
__global__ void __launch_bounds__(256)
finalize_ll_kernel(uint16_t *__restrict__ coef, size_t plane, uint32_t w, uint32_t llw, uint32_t llh,
                   const uint16_t *__restrict__ means, const int *__restrict__ frame_skip, int channels, int sample_bits)
{
    const uint32_t c = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= llw || r >= llh) return;
    if (frame_skip[blockIdx.z / channels]) return;
    uint16_t *q = coef + blockIdx.z * plane + (size_t)r * w + c;
    const int16_t v = (int16_t)((int16_t)*q - (int16_t)means[blockIdx.z]);
    *q = (uint16_t)to_coder_word(v, sample_bits);
}

// Sub-range splitting of a launch (coder_core.hpp "Sub-ranges"); n_subs = 0: off
struct SplitLaunch {
    const SubDesc *subs = nullptr;          // Plan::subs
    const uint32_t *launch = nullptr;       // Plan::split_launch: n_units + n_subs entries
    uint32_t n_subs = 0;                    // extra workgroups per frame
    uint32_t entries = 0;                   // Plan::sub_entries: per-frame entries of the arrays below
    Snapshot *snaps = nullptr;              // [frames][entries][kMaxSnaps]
    uint32_t *snap_valid = nullptr;         // [frames][entries][kMaxSnaps]
    SubRecord *recs = nullptr;              // [frames][entries]
#ifdef ICER_EXPERIMENT_PREFIX_CACHE
    // EXPERIMENT, never in the product build (tools/prefix_cache_probe.sh): the counts at every sub-range start, kept from the
    // launch before -- what the launch would cost if a sub-range's counts came for free.  [frames][entries][36]: 17 zero counts,
    // 17 totals, -, valid.  Right only while the same frames are coded again, which is all the probe does.
    uint32_t *prefix_cache = nullptr;
#endif
};

// a launch shared by the two coders (route_units_kernel): which one takes a unit
constexpr uint8_t kRoutePipeline = 0, kRouteWindows = 1, kRouteNoSplit = 2;   // (2: the pipeline, but not worth splitting: 20 .. `percent` % blank chunks)

// ------------------------------------------------------------------------------------------ coder
// One workgroup = one coding unit of one frame: pixel, count, compaction, walker, golomb state + workers, merge, records and
// drain waves, a software pipeline over 64-pixel chunks (coder_core.hpp).  WAVES = 8: one pixel wave, one golomb wave
// (three workgroups per CU: batches); WAVES = 11: two pixel waves, golomb state wave + two workers (two per CU, a
// shorter chain per chunk: single frames).
// grid = (units, frames), block = 64 * WAVES.
// OCC = resident waves per SIMD the register budget is cut for.  8 (64 VGPRs; with 37 KiB of LDS: four workgroups of the
// small shape per compute unit) is what a batch wants: + 7 % on C4, + 3 % on C5 over three per CU.  The budget costs ~ 280
// more SGPR spills (v_writelane / v_readlane), and a lone frame -- bound by the chains of its largest units, not by
// occupancy -- is 14 % slower with it: it runs the OCC = 1 build (profiles/archive/r03_logs/r03_x.log, r03_y.log).
// LDS_PAD = bytes of LDS the workgroup takes on top of what it uses (single-frame launches, api.hip enqueue).
template <int WAVES, int OCC, int LDS_PAD>
__global__ void __launch_bounds__(64 * WAVES, OCC)
code_units_kernel(const uint16_t *__restrict__ coef, size_t plane, uint32_t img_w, uint32_t img_h, int channels,
                  const UnitDesc *__restrict__ units, const uint32_t *__restrict__ work_order, uint32_t n_units,
                  const CoderTables *__restrict__ tables, const uint16_t *__restrict__ means,
                  const int *__restrict__ frame_skip, uint8_t *__restrict__ slots,
                  size_t slot_frame_stride, uint32_t *__restrict__ unit_bits, uint64_t *__restrict__ timers,
                  uint32_t *__restrict__ done_bytes, uint64_t early_quota, const uint8_t *__restrict__ route, SplitLaunch sp, uint32_t n_frames_1d,
                  const uint8_t *__restrict__ ev, size_t ev_frame_stride, const uint8_t *__restrict__ sig, size_t sig_frame_stride,
                  uint32_t fail_inject)
{
    __shared__ CoderShared s;
    if constexpr (LDS_PAD > 0) {
        __shared__ uint32_t lds_pad[LDS_PAD / 4];
        if (early_quota == ~0ull) lds_pad[threadIdx.x] = 1u;          // (never: keeps the array in the kernel's LDS size)
    }
    // Which (launch position, frame) this workgroup is.  A two-dimensional grid (positions, frames) is dispatched frame by frame: the
    // LAST frame's largest units then start when the launch is nearly over and the chip idles through their chains.  A one-dimensional
    // grid (gridDim.y == 1, several frames; round 5) is walked position-major instead -- eight positions (one per XCD: position % 8 still
    // names the XCD, so a family keeps its L2) of every frame, then the next eight -- so that the largest units of ALL frames start first
    // and what is left at the end are the smallest units of all frames.
    uint32_t frame = blockIdx.y, lpos = blockIdx.x;
    if (gridDim.y == 1u && n_frames_1d > 1u) {
        position_major(blockIdx.x, n_units + sp.n_subs, n_frames_1d, &frame, &lpos);
        if (frame >= n_frames_1d) return;               // (a grid that is not per_frame x n_frames: never launched, never out of bounds)
    }
    // A split launch (sp.n_subs > 0) has extra workgroups for its split units: an entry of sp.launch with bit 31 set codes a later
    // sub-range of a unit (coder_core.hpp "Sub-ranges"), the others one unit from its first chunk as always.
    const uint32_t entry = sp.n_subs ? sp.launch[lpos] : (work_order ? work_order[lpos] : lpos);   // (null: priority order = unit order)
    const bool sub_block = sp.n_subs != 0u && (entry >> 31) != 0u;
    const SubDesc sd = sub_block ? sp.subs[entry & 0x7FFFFFFFu] : SubDesc{};
    const uint32_t ui = sub_block ? sd.unit : entry;
    // (two coders share a launch: a unit belongs to the one route_units_kernel names)
    const uint8_t rt = route ? route[(size_t)frame * n_units + ui] : kRoutePipeline;
    if (rt == kRouteWindows) return;
    if (sub_block && rt != kRoutePipeline) return;                  // only dense units are split
#ifdef ICER_PHASE_TIMERS
    uint64_t *trace = (timers && frame == 0 && lpos < (uint32_t)kTraceUnits) ? timers + 9 * 32 + 4 * lpos : nullptr;
    if (trace && threadIdx.x == 0) {
        trace[0] = wall_clock64();
        trace[2] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((uint64_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
        trace[3] = entry;
    }
    if (timers && frame == 0 && lpos == 0 && (threadIdx.x & 63) == 0)
        timers[9 * 32 + 4 * kTraceUnits + (threadIdx.x >> 6)] = (uint64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    // (as a scalar: the roles below are then uniform branches with a register allocation of their own -- seen as divergent,
    // every role's loop kept alive what the later roles need: ~ 200 SGPR spills / reloads, v_writelane / v_readlane, in the loops)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // DWT / mean overflow: the reference emits nothing.
    if (frame_skip[frame]) {
        if (threadIdx.x == 0 && !sub_block) unit_bits[(size_t)frame * n_units + ui] = 0;
        return;
    }
    const UnitDesc u = units[ui];
    // TEST HOOK (ICER_HIP_TEST_FAIL_UNIT, api.hip): this unit reports a time-out although it has not had one, so that the recovery path --
    // diagnostics, the batch coded again by the barrier-only coder -- can be exercised on hardware.  ~0u (always, outside that test): none.
    const bool injected = fail_inject == ((frame << 20) | ui);
    if (early_quota) {
        // progressive mode: units are launched in priority order; one whose higher-priority predecessors have already
        // used up the quota can not be in the stream (quota_already_spent)
        UnitArgs pa;
        pa.done_bytes = done_bytes + (size_t)frame * n_units; pa.prio_index = ui; pa.early_quota = early_quota;
        __shared__ int skip_unit;                   // (one wave decides for the workgroup: the inputs keep changing)
        if (wave == 0) {
            const bool spent = quota_already_spent(pa);
            if (threadIdx.x == 0) skip_unit = spent ? 1 : 0;
        }
        __syncthreads();
        if (skip_unit) {
            if (threadIdx.x == 0) unit_bits[(size_t)frame * n_units + ui] = 0;
            return;
        }
    }
    switch (u.prio) {                             // s_setprio takes an immediate
    case 3: __builtin_amdgcn_s_setprio(3); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: break;
    }
    {   // tables -> LDS
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&s.tab);
        for (uint32_t i = threadIdx.x; i < sizeof(CoderTables) / 4; i += 64 * WAVES) dst[i] = src[i];
    }
    // Role of each wavefront.  Waves w and w + 4 of a workgroup share a SIMD (observed placement, used for speed
    // only): the count wave, which issues the most vector instructions per chunk, gets a SIMD to itself and the
    // other issue-heavy roles are paired with latency-bound ones.
    // (wave 8: the second pixel wave -- on the SIMD of the walker, the lightest role)
    constexpr uint32_t kWalker = 0, kPixel = 1, kGolomb = 2, kMerge = 3, kCompact = 4, kCount = 5, kRecords = 6, kDrain = 7;
    // large shape only: golomb state wave, second pixel wave, second golomb worker
    constexpr uint32_t kGolombState = 8, kPixel2 = 9, kGolomb2 = 10;
    static_assert(WAVES == kUnitWavesSmall || WAVES == kUnitWavesLarge, "two shapes of a workgroup");
    constexpr bool large = WAVES == kUnitWavesLarge;
    constexpr uint32_t npw = large ? 2u : 1u, ngw = large ? 2u : 0u;   // (0: no golomb state wave)
    if (wave == 0) unit_state_init(s);
    if (threadIdx.x == 64) s.nchunks = (u.w * u.h + 63u) / 64u;
    if (wave == kMerge) build_crc_table(s);

    uint32_t *slot_words = reinterpret_cast<uint32_t *>(slots + (size_t)frame * slot_frame_stride + u.slot_off);
    UnitArgs a;
    a.seg = coef + ((size_t)frame * channels + u.chan) * plane + (size_t)u.y0 * img_w + u.x0;
    a.stride = img_w;
    // the family's events at this unit's bit plane and its chunk table (family_events_kernel)
    a.ev = ev + (size_t)frame * ev_frame_stride + ev_offset(u.lsb, sig_frame_stride, u.sig_off, 0u);
    a.sig = sig + (size_t)frame * sig_frame_stride + u.sig_off;
    a.w = u.w; a.h = u.h;
    a.subband = (int)u.subband; a.lsb = (int)u.lsb;
    a.out_words = slot_words + kHeaderBytes / 4;
    a.cap_words = u.cap_words;
    // this workgroup's place among the sub-ranges of a split unit
    __shared__ SubLayout layout;
    const bool split = sp.n_subs != 0u && u.n_sub > 1u && rt == kRoutePipeline;
    if (split) {
        if (threadIdx.x == 0) {
            layout.n_sub = u.n_sub;
            layout.index = sub_block ? sd.index : 0u;
            for (uint32_t i = 0; i <= u.n_sub; i++) layout.first[i] = sub_first_chunk((u.w * u.h + 63u) / 64u, u.n_sub, i);
            const size_t e = (size_t)frame * sp.entries + u.sub_entry;
            layout.snaps = sp.snaps + e * kMaxSnaps;
            layout.snap_valid = sp.snap_valid + e * kMaxSnaps;
            layout.rec = sp.recs + e;
        }
        a.sub = &layout;
        if (sub_block) {
            a.out_words = reinterpret_cast<uint32_t *>(slots + (size_t)frame * slot_frame_stride + sd.slot_off);
            a.cap_words = sd.cap_words;
        }
    }
    a.done_bytes = early_quota ? done_bytes + (size_t)frame * n_units : nullptr;
    a.prio_index = ui;
    a.early_quota = early_quota;
    // profiling build: per-wave cycle counters of the level-1 (largest) units, one row per bit plane
    a.timers = (timers && u.level == 1) ? timers + u.lsb * 32 : nullptr;
    const uint32_t nchunks = (u.w * u.h + 63u) / 64u;
    __syncthreads();

    // A later sub-range starts with the adaptive counts at its first chunk j0: they depend on the coefficients before it
    // alone -- the pixel stage (all waves but one) and the counts of the count stage over [0, j0), nothing else.
    uint32_t j0 = 0;
    CountWave cs;
    if (split && sub_block) {
        j0 = layout.first[layout.index];
        constexpr uint32_t npw1 = (uint32_t)(WAVES - 1) < kMaxPixelWaves ? (uint32_t)(WAVES - 1) : kMaxPixelWaves;
        const uint32_t k = wave < kCount ? wave : wave - 1u;
#ifdef ICER_EXPERIMENT_PREFIX_CACHE
        uint32_t *pc = sp.prefix_cache ? sp.prefix_cache + (((size_t)frame * sp.entries + u.sub_entry + layout.index) * 36u) : nullptr;
        const bool cached = pc && pc[35] == 1u;
        if (cached) {
            const int lane_ = (int)(threadIdx.x & 63);
            if (wave == kCount && lane_ < 17) { cs.czer = pc[lane_]; cs.ctot = pc[17 + lane_]; }
        } else {
#endif
        if (wave == kCount) count_prefix_run(s, a, cs, 0, j0, npw1);
        else if (k < npw1) {
            PixelWave pw;
            pixel_prefix_run(s, a, pw, 0, j0, k, npw1);
        }
#ifdef ICER_EXPERIMENT_PREFIX_CACHE
            if (pc && wave == kCount) {
                const int lane_ = (int)(threadIdx.x & 63);
                if (lane_ < 17) { pc[lane_] = cs.czer; pc[17 + lane_] = cs.ctot; }
                __threadfence();
                if (lane_ == 0) pc[35] = 1u;
            }
        }
#endif
        __syncthreads();
        const uint32_t ab1 = s.abort;               // (a bounded spin expired in the prefix pass: the workgroup reports failure)
        __syncthreads();
        if (ab1) {
            if (wave == kMerge) {
                SubRecord &r = layout.rec[layout.index];
                if ((threadIdx.x & 63) == 0) { r.end_chunk = j0; r.end_bits = kUnitFailed; r.match_sub = 0; r.match_snap = 0; }
                const int lane = (int)(threadIdx.x & 63);
                ICER_AGENT_PUBLISH(&r.done, 1u)
            }
            return;
        }
        if (wave == 0) unit_state_init(s, j0);
        __syncthreads();
    }

    if (wave == kPixel || wave == kPixel2) {
        PixelWave pw;
        pixel_wave_run(s, a, pw, j0, nchunks, wave == kPixel ? 0u : 1u, npw);
    } else if (wave == kCount) {
        count_wave_run(s, a, cs, j0, nchunks, npw);
    } else if (wave == kWalker) {
        WalkWave ww;
        walk_wave_init(s, ww, j0);
        walk_wave_run(s, a, ww, nchunks, ~0u);
    } else if (wave == kGolomb || wave == kGolomb2) {
        GolombWave gw;
        golomb_wave_init(gw, j0);
        golomb_wave_run(s, a, gw, nchunks, ~0u, wave == kGolomb ? 0u : 1u, ngw);
    } else if (wave == kGolombState) {
        GolombWave gw;
        golomb_wave_init(gw, j0);
        golomb_state_run(s, a, gw, nchunks, ~0u);
    } else if (wave == kRecords) {
        RecordsWave rw;
        rw.next = j0;
        records_wave_run(s, a, rw, ~0u);
    } else if (wave == kDrain) {
        drain_wave_run(s, a, ~0u);
    } else if (wave == kCompact) {
        compact_wave_run(s, a, j0, nchunks);
    } else if (split) {
        // a workgroup of a split unit leaves its record; splice_units_kernel makes the unit's payload, header and CRCs
        const uint32_t how = merge_wave_run(s, a, j0, nchunks);
        if (how != kMergeMatched) {
            uint32_t bits = how == kMergeDone ? merge_wave_finish(s, a) : kUnitTooBig;
            if (s.abort == 2u || injected) bits = kUnitFailed;
            SubRecord &r = layout.rec[layout.index];
            if ((threadIdx.x & 63) == 0) { r.end_chunk = nchunks; r.end_bits = bits; r.match_sub = 0; r.match_snap = 0; }
            const int lane = (int)(threadIdx.x & 63);
            ICER_AGENT_PUBLISH(&r.done, 1u)
        }
#ifdef ICER_PHASE_TIMERS
        if (trace && (threadIdx.x & 63) == 0) { trace[1] = wall_clock64(); trace[3] |= (uint64_t)(how == kMergeMatched ? layout.rec[layout.index].end_chunk : nchunks) << 32; }
#endif
    } else {
        uint32_t bits = merge_wave_run(s, a, 0, nchunks) ? merge_wave_finish(s, a) : kUnitTooBig;
        if (s.abort == 2u || injected) {              // a bounded spin expired: internal error, never a silent hang
            bits = kUnitFailed;
            if (injected && (threadIdx.x & 63) == 0) s.abort_site = 0xFFFFu | (0xFFu << 16);      // (diagnostics: "wave 255 at line 65535" = the test hook)
            // leave the unit's hand-off counters where its payload would have been (api.hip prints them)
            if ((threadIdx.x & 63) == 0 && u.cap_words >= kFailWords) {
                uint32_t *dbg = slot_words + kHeaderBytes / 4;
                dbg[0] = kFailMagic; dbg[1] = s.abort_site; dbg[2] = nchunks; dbg[3] = s.p_done[0]; dbg[4] = s.a_done;
                dbg[5] = s.c_done; dbg[6] = s.b_done; dbg[7] = s.alloc; dbg[8] = s.popped; dbg[9] = s.hold_seq;
                dbg[10] = s.hold_ack; dbg[11] = s.exact_seq; dbg[12] = s.last_exact; dbg[13] = s.bitpos;
                dbg[14] = s.flushed_words; dbg[15] = s.drain_exit;
            }
        }
        if (s.abort == 3u) bits = 0;                  // progressive mode: stopped, the quota cut lies before this unit
        if (bits != kUnitTooBig && bits != kUnitFailed) {
            // make this wave's payload stores visible to its own loads before the CRC pass reads them
            __threadfence();
            FinishArgs f;
            f.slot_words = slot_words;
            f.bits = bits;
            f.mean = means[(size_t)frame * channels + u.chan];
            f.level = u.level; f.subband = u.subband; f.seg = u.seg; f.lsb = u.lsb; f.chan = u.chan;
            f.image_w = img_w; f.image_h = img_h;
            finish_unit_wave(s, f);
        }
        if ((threadIdx.x & 63) == 0) {
            unit_bits[(size_t)frame * n_units + ui] = bits;
            if (early_quota && s.abort != 3u) {
                // a unit that outgrew a slot sized by the quota can not fit; one that outgrew the bits-per-pixel bound is
                // unknown (the batch is redone with larger slots if it matters)
                const uint32_t d = bits == kUnitTooBig ? (u.cap_is_bound ? 0u : ~0u) : bits == kUnitFailed ? 0u : kHeaderBytes + ((bits + 7u) >> 3);
                __hip_atomic_store(&done_bytes[(size_t)frame * n_units + ui], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef ICER_PHASE_TIMERS
        if (trace && (threadIdx.x & 63) == 0) trace[1] = wall_clock64();
#endif
    }
}

// The units of a split launch that were coded by several workgroups: their payload from the workgroups' pieces
// (splice_unit_wave), then header and CRCs like any unit.  grid = (units, frames), block = 64; units that were not split
// (or went to the other coder) return at once.  Four wavefronts share a unit's copies and its CRC (a level-1 unit of the
// headline frame: ~ 35 KB to move bit-granular and 70 KB to checksum, after the last coding unit of the frame has ended).
constexpr uint32_t kSpliceWaves = 4;
__global__ void __launch_bounds__(64 * kSpliceWaves)
splice_units_kernel(const UnitDesc *__restrict__ units, uint32_t n_units, const CoderTables *__restrict__ tables,
                    const uint16_t *__restrict__ means, const int *__restrict__ frame_skip, int channels, uint32_t img_w, uint32_t img_h,
                    uint8_t *__restrict__ slots, size_t slot_frame_stride, uint32_t *__restrict__ unit_bits,
                    const uint8_t *__restrict__ route, SplitLaunch sp)
{
    struct SpliceShared { uint32_t crc_tab[256]; struct { uint32_t x2n[32]; } tab; uint32_t parts[kSpliceWaves]; };
    __shared__ SpliceShared s;
    const uint32_t frame = blockIdx.y, ui = blockIdx.x;
    const UnitDesc u = units[ui];
    if (u.n_sub <= 1u || frame_skip[frame]) return;
    if (route && route[(size_t)frame * n_units + ui] != kRoutePipeline) return;
    build_crc_table(s);
    if (threadIdx.x < 32) s.tab.x2n[threadIdx.x] = tables->x2n[threadIdx.x];
    __syncthreads();
    uint8_t *fs = slots + (size_t)frame * slot_frame_stride;
    uint32_t *slot_words = reinterpret_cast<uint32_t *>(fs + u.slot_off);
    uint32_t *words[kMaxSubs];
    words[0] = slot_words + kHeaderBytes / 4;
    for (uint32_t i = 1; i < u.n_sub; i++) words[i] = reinterpret_cast<uint32_t *>(fs + sp.subs[u.sub_first + i - 1u].slot_off);
    const size_t e = (size_t)frame * sp.entries + u.sub_entry;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");              // the records, snapshots and payload words of other workgroups
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t bits = splice_unit_wave(u.n_sub, sp.recs + e, sp.snaps + e * kMaxSnaps, words, u.cap_words, wv, kSpliceWaves);
    if (bits != kUnitTooBig && bits != kUnitFailed) {
        __threadfence();
        FinishArgs f;
        f.slot_words = slot_words;
        f.bits = bits;
        f.mean = means[(size_t)frame * channels + u.chan];
        f.level = u.level; f.subband = u.subband; f.seg = u.seg; f.lsb = u.lsb; f.chan = u.chan;
        f.image_w = img_w; f.image_h = img_h;
        __syncthreads();                                                // (every wave's copies, before any wave's CRC piece)
        finish_unit_wave(s, f, wv, kSpliceWaves, s.parts);
    }
    if (threadIdx.x == 0) unit_bits[(size_t)frame * n_units + ui] = bits;
}

// ------------------------------------------------------------------------------------------ family events + chunk tables
// The stateless half of the context modeller once per FAMILY (events.hpp): for every 64-pixel chunk of every (channel, level,
// subband, segment) one pass over the chunk's 3x3 windows leaves
//   * the chunk table: one byte per chunk, the lowest bit plane from which the chunk is blank (wg::chunk_blank_plane), and the family's
//     histogram of those values (route_units_kernel);
//   * with `ev` != null, one event byte per pixel for every bit plane below that (events.hpp: context, bit, sign context, sign bit), in
//     the coding order of the plane's unit -- what code_units_kernel's pixel wave reads instead of gathering and classifying the nine
//     words again for each of the family's planes (icer_context_modeller.c:340-440 once per pixel, not once per pixel and plane).
// Work list (Plan::sig_blocks): one (unit, block) pair per family and block of 64 chunks; grid = (entries, frames), block = 256
// (16 chunks per wavefront).  HBM: reads the coefficient plane once (2 B / pixel, 3x3 windows through L1 / L2), writes one byte per pixel
// and NON-BLANK bit plane (8-bit content: about five of nine).
__global__ void __launch_bounds__(256)
family_events_kernel(const uint16_t *__restrict__ coef, size_t plane, uint32_t img_w, int channels,
                     const UnitDesc *__restrict__ units, const uint32_t *__restrict__ blocks, const int *__restrict__ frame_skip,
                     uint8_t *__restrict__ sig, size_t sig_frame_stride, uint32_t *__restrict__ hist, uint32_t n_families,
                     uint8_t *__restrict__ ev, size_t ev_frame_stride, uint32_t n_planes)
{
    __shared__ uint8_t ctx_tab[48];
    const UnitDesc u = units[blocks[2u * blockIdx.x]];
    const uint32_t frame = blockIdx.y, nchunks = (u.w * u.h + 63u) / 64u, first = blocks[2u * blockIdx.x + 1u] * 64u;
    if (frame_skip[frame]) return;
    const bool is_hl = u.subband == kHL, is_hh = u.subband == kHH;
    if (ev) {
        if (threadIdx.x < 45u) ctx_tab[threadIdx.x] = (uint8_t)ev_ctx_entry(is_hh, threadIdx.x);
        __syncthreads();
    }
    const uint16_t *seg = coef + ((size_t)frame * channels + u.chan) * plane + (size_t)u.y0 * img_w + u.x0;
    uint8_t *out = sig + (size_t)frame * sig_frame_stride + u.sig_off;
    uint8_t *evf = ev ? ev + (size_t)frame * ev_frame_stride : nullptr;
    // 16 consecutive chunks per wavefront: the lane's pixel coordinates advance by 64 with one wrap instead of a division per
    // chunk, and the maximum over the lanes is four ballots (the values are bit lengths, 0 .. 15) instead of six cross-lane shuffles.
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t sw = u.w, sh = u.h, npix = sw * sh;
    uint32_t j = first + wave * 16u;
    uint32_t np = j * 64u + lane;
    uint32_t r = np / sw, c = np - r * sw;
    uint32_t mine = 0;                                           // lane v: this wave's chunks whose value is v (the family's histogram)
    for (uint32_t i = 0; i < 16u && j < nchunks; i++, j++) {
        const bool in_ = np < npix;
        const uint32_t r_ = in_ ? r : 0u, c_ = in_ ? c : 0u;
        const bool hasW_ = c_ > 0, hasE_ = c_ + 1 < sw, hasN_ = r_ > 0, hasS_ = r_ + 1 < sh;
        const uint32_t cW_ = hasW_ ? c_ - 1u : c_, cE_ = hasE_ ? c_ + 1u : c_;
        const uint16_t *pC_ = seg + (size_t)r_ * img_w;
        const uint16_t *pN_ = hasN_ ? pC_ - img_w : pC_, *pS_ = hasS_ ? pC_ + img_w : pC_;
        // nine unconditional loads from clamped positions, masked afterwards (as wg::chunk_blank_plane)
        const uint32_t vC = pC_[c_], vW = pC_[cW_], vE = pC_[cE_], vN = pN_[c_], vNW = pN_[cW_], vNE = pN_[cE_];
        const uint32_t vS = pS_[c_], vSW = pS_[cW_], vSE = pS_[cE_];
        const PixelLens L = pixel_lens(vC, hasW_ ? vW : 0u, hasE_ ? vE : 0u, hasN_ ? vN : 0u, hasS_ ? vS : 0u, (hasN_ && hasW_) ? vNW : 0u,
                                       (hasN_ && hasE_) ? vNE : 0u, (hasS_ && hasW_) ? vSW : 0u, (hasS_ && hasE_) ? vSE : 0u);
        uint32_t tmax = 0;
        for (uint32_t step = 8u; step; step >>= 1) tmax += __ballot(in_ && L.t >= tmax + step) ? step : 0u;
        const bool partial = __ballot(!in_) != 0ull;                 // a chunk with fewer than 64 pixels is never blank
        if (lane == 0u) out[j] = (uint8_t)(partial ? 255u : tmax);
        mine += (!partial && lane == tmax) ? 1u : 0u;
        if (evf) {
            // the planes at which the chunk is not blank (a partial chunk: all of them)
            const uint32_t top = partial ? n_planes : (tmax < n_planes ? tmax : n_planes);
            for (uint32_t p = 0; p < top; p++)
                evf[ev_offset(p, sig_frame_stride, u.sig_off, j) + lane] = (uint8_t)(in_ ? event_byte(L, p, is_hl, is_hh, ctx_tab) : kEvNone);
        }
        np += 64u;
        c += 64u;
        if (sw >= 64u) { if (c >= sw) { c -= sw; r++; } }
        else { r = np / sw; c = np - r * sw; }
    }
    // hist[v] = chunks of the family that are blank from bit plane v on and at none below (route_units_kernel sums the
    // entries up to a unit's plane instead of reading the whole table once per plane)
    if (hist && lane < 16u && mine) atomicAdd(&hist[((size_t)frame * n_families + u.family) * 16u + lane], mine);
}

// Which coder takes a unit when both share a launch: the pipeline (code_units_kernel) is the faster one on dense bit planes,
// the workgroup coder (code_units_wg_kernel) on planes that are mostly runs of blank chunks, which it closes in closed form
// (wg::blank_run).  One thread per unit and frame takes the unit's number of blank chunks from the histogram chunk_sig_kernel left
// for the unit's family (round 5; before: a workgroup per unit counted them in the chunk table, 0.40 ms of a C4 launch):
// route = windows when at least `percent` % of the chunks are blank; those units are also appended to a list
// (list_ctl[0] = its length).  grid = (ceil(units / 256), frames), block = 256.
__global__ void __launch_bounds__(256)
route_units_kernel(const UnitDesc *__restrict__ units, uint32_t n_units, const uint32_t *__restrict__ hist, uint32_t n_families,
                   uint32_t percent, uint32_t min_chunks, uint8_t *__restrict__ route, uint32_t *__restrict__ list,
                   uint32_t *__restrict__ list_ctl, uint32_t nosplit_percent, uint32_t list_cap, uint32_t heavy_min)
{
    const uint32_t ui = blockIdx.x * 256u + threadIdx.x, frame = blockIdx.y;
    const bool valid = ui < n_units;
    const UnitDesc u = units[valid ? ui : 0u];
    const uint32_t nchunks = (u.w * u.h + 63u) / 64u;
    // blank chunks of the unit = chunks of its family that are blank from a plane <= the unit's on (family_events_kernel's histogram;
    // a last chunk with fewer than 64 pixels is in no entry: never blank)
    const uint32_t *h = hist + ((size_t)frame * n_families + u.family) * 16u;
    uint32_t total = 0;
    for (uint32_t v = 0; v < 16u; v++) total += v <= u.lsb ? h[v] : 0u;
    const bool windows = valid && nchunks >= min_chunks && total * 100u >= percent * nchunks;
    // (a unit with a fifth of its chunks blank and more: its words stay open for long stretches, sub-ranges would not
    // meet -- see coder_core.hpp "Sub-ranges")
    if (valid) route[(size_t)frame * n_units + ui] = windows ? kRouteWindows : (total * 100u >= nosplit_percent * nchunks ? kRouteNoSplit : kRoutePipeline);
    // (code_units_list_kernel) The units with real content -- `heavy_min` chunks and more that are not blank: the mid-sparse planes of the
    // large levels, milliseconds each -- go to a list of their own (the second half of the buffer) that the staying workgroups take FIRST, the
    // all-blank ones (microseconds each) follow in their order of arrival: whatever order the threads of this kernel arrive in, the long
    // chains start at once.  (Round 6: a long unit taken 0.6 ms into the kernel ended the launch 0.6 ms later.)
    // list_ctl: [0] entries, [1] the consumers' cursor, [2] heavy, [3] light.  One atomic per wavefront and list, not one per unit: tens of
    // thousands of units of a batch bumping one counter made this kernel 0.2 ms.
    const bool heavy = windows && nchunks - total >= heavy_min, light = windows && !heavy;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mh = __ballot(heavy), ml = __ballot(light);
    if (mh | ml) {
        const uint32_t nh = (uint32_t)__popcll(mh), nl = (uint32_t)__popcll(ml);
        uint32_t bh = 0, bl = 0;
        if (lane == 0u) {
            if (nh) bh = atomicAdd(&list_ctl[2], nh);
            if (nl) bl = atomicAdd(&list_ctl[3], nl);
            atomicAdd(&list_ctl[0], nh + nl);
        }
        bh = (uint32_t)__shfl((int)bh, 0); bl = (uint32_t)__shfl((int)bl, 0);
        const uint32_t rh = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
        const uint32_t rl = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(ml >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ml, 0u));
        if (heavy) list[list_cap + bh + rh] = frame * n_units + ui;
        if (light) list[bl + rl] = frame * n_units + ui;
    }
}

// ------------------------------------------------------------------------------------------ coder (workgroup windows)
// One workgroup of wg::kWgWaves wavefronts = one coding unit of one frame; the unit is coded in windows of kWgWaves
// chunks of 64 pixels, one chunk per wave, the waves meeting at workgroup barriers only (coder_wg.hpp): no wave ever
// waits on a flag, so there is no hand-off that could stall.  grid = (units, frames), block = 64 * kWgWaves, LDS =
// sizeof(wg::Shared) (dynamic: above the 64 KiB static limit).
struct WgLaunch {
    const uint16_t *coef; size_t plane; uint32_t img_w, img_h; int channels;
    const UnitDesc *units; uint32_t n_units; const uint16_t *means; const int *frame_skip; uint8_t *slots;
    size_t slot_frame_stride; uint32_t *unit_bits; uint64_t *timers; uint32_t *done_bytes; uint64_t early_quota;
    const uint8_t *sig; size_t sig_frame_stride;
};
// the two instances of the workgroup coder: icer::wg (16 wavefronts) and icer::wgs (2, coder_wg_small.hpp)
struct WgFull {
    using Shared = wg::Shared; using UnitArgs = wg::UnitArgs; using Wave = wg::Wave;
    static constexpr uint32_t kWaves = wg::kWgWaves;
    static __device__ __forceinline__ void init(Shared &s, const UnitArgs &a) { wg::unit_state_init(s, a); }
    static __device__ __forceinline__ bool spent(const UnitArgs &a) { return wg::quota_already_spent(a); }
    static __device__ __forceinline__ uint32_t code(Shared &s, const UnitArgs &a, Wave &r) { return wg::code_unit_wg(s, a, r); }
};
struct WgSmall {
    using Shared = wgs::Shared; using UnitArgs = wgs::UnitArgs; using Wave = wgs::Wave;
    static constexpr uint32_t kWaves = wgs::kWgWaves;
    static __device__ __forceinline__ void init(Shared &s, const UnitArgs &a) { wgs::unit_state_init(s, a); }
    static __device__ __forceinline__ bool spent(const UnitArgs &a) { return wgs::quota_already_spent(a); }
    static __device__ __forceinline__ uint32_t code(Shared &s, const UnitArgs &a, Wave &r) { return wgs::code_unit_wg(s, a, r); }
};

struct WgFour {
    using Shared = wg4::Shared; using UnitArgs = wg4::UnitArgs; using Wave = wg4::Wave;
    static constexpr uint32_t kWaves = wg4::kWgWaves;
    static __device__ __forceinline__ void init(Shared &s, const UnitArgs &a) { wg4::unit_state_init(s, a); }
    static __device__ __forceinline__ bool spent(const UnitArgs &a) { return wg4::quota_already_spent(a); }
    static __device__ __forceinline__ uint32_t code(Shared &s, const UnitArgs &a, Wave &r) { return wg4::code_unit_wg(s, a, r); }
};

struct WgOne {
    using Shared = wg1::Shared; using UnitArgs = wg1::UnitArgs; using Wave = wg1::Wave;
    static constexpr uint32_t kWaves = wg1::kWgWaves;
    static __device__ __forceinline__ void init(Shared &s, const UnitArgs &a) { wg1::unit_state_init(s, a); }
    static __device__ __forceinline__ bool spent(const UnitArgs &a) { return wg1::quota_already_spent(a); }
    static __device__ __forceinline__ uint32_t code(Shared &s, const UnitArgs &a, Wave &r) { return wg1::code_unit_wg(s, a, r); }
};

// one coding unit of one frame by the calling workgroup; the tables and the CRC table are in `s` already
template <class I>
__device__ __forceinline__ void wg_code_one_unit(typename I::Shared &s, const WgLaunch &L, uint32_t frame, uint32_t ui)
{
    const uint16_t *coef = L.coef; const size_t plane = L.plane; const uint32_t img_w = L.img_w, img_h = L.img_h; const int channels = L.channels;
    const UnitDesc *units = L.units; const uint32_t n_units = L.n_units; const uint16_t *means = L.means; const int *frame_skip = L.frame_skip;
    uint8_t *slots = L.slots; const size_t slot_frame_stride = L.slot_frame_stride; uint32_t *unit_bits = L.unit_bits; uint64_t *timers = L.timers;
    uint32_t *done_bytes = L.done_bytes; const uint64_t early_quota = L.early_quota; const uint8_t *sig = L.sig; const size_t sig_frame_stride = L.sig_frame_stride;
    const uint32_t wave = threadIdx.x >> 6;
    if (frame_skip[frame]) {                      // DWT / mean overflow: the reference emits nothing
        if (threadIdx.x == 0) unit_bits[(size_t)frame * n_units + ui] = 0;
        return;
    }
    const UnitDesc u = units[ui];
    uint32_t *slot_words = reinterpret_cast<uint32_t *>(slots + (size_t)frame * slot_frame_stride + u.slot_off);
    typename I::UnitArgs a;
    a.seg = coef + ((size_t)frame * channels + u.chan) * plane + (size_t)u.y0 * img_w + u.x0;
    a.stride = img_w;
    a.w = u.w; a.h = u.h;
    a.subband = (int)u.subband; a.lsb = (int)u.lsb;
    a.out_words = slot_words + kHeaderBytes / 4;
    a.cap_words = u.cap_words;
    a.done_bytes = early_quota ? done_bytes + (size_t)frame * n_units : nullptr;
    a.prio_index = ui;
    a.early_quota = early_quota;
    // profiling build: per-phase cycle counters of the level-1 (largest) units, one row per bit plane
    a.timers = (timers && u.level == 1) ? timers + u.lsb * 32 : nullptr;
    a.sig = sig ? sig + (size_t)frame * sig_frame_stride + u.sig_off : nullptr;
    if (wave == 0) {
        I::init(s, a);
        // progressive mode: a unit whose finished higher-priority predecessors have already used up the quota can not
        // be in the stream (quota_already_spent)
        if (early_quota && I::spent(a) && threadIdx.x == 0) s.stop = 1u;
    }
    __syncthreads();
    uint32_t bits;
    if (s.stop) bits = wg::kUnitStopped;
    else {
        typename I::Wave regs;
        bits = I::code(s, a, regs);
    }
    const bool stopped = bits == wg::kUnitStopped;
    if (stopped) bits = 0;                        // the quota cut lies before this unit
    if (wave == 0) {
        if (bits != kUnitTooBig) {
            // the payload words were stored by every wave of the workgroup (released before the last barrier): the CRC
            // pass below must see them
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            FinishArgs f;
            f.slot_words = slot_words;
            f.bits = bits;
            f.mean = means[(size_t)frame * channels + u.chan];
            f.level = u.level; f.subband = u.subband; f.seg = u.seg; f.lsb = u.lsb; f.chan = u.chan;
            f.image_w = img_w; f.image_h = img_h;
            finish_unit_wave(s, f);
        }
        if (threadIdx.x == 0) {
            unit_bits[(size_t)frame * n_units + ui] = bits;
            if (early_quota && !stopped) {
                // a unit that outgrew a slot sized by the quota can not fit; one that outgrew the bits-per-pixel bound is
                // unknown (the batch is redone with larger slots if it matters)
                const uint32_t d = bits == kUnitTooBig ? (u.cap_is_bound ? 0u : ~0u) : kHeaderBytes + ((bits + 7u) >> 3);
                __hip_atomic_store(&done_bytes[(size_t)frame * n_units + ui], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <class I>
__device__ __forceinline__ void wg_shared_tables(typename I::Shared &s, const CoderTables *__restrict__ tables)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(tables);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&s.tab);
    for (uint32_t i = threadIdx.x; i < sizeof(CoderTables) / 4; i += 64 * I::kWaves) dst[i] = src[i];
    if ((threadIdx.x >> 6) == (I::kWaves > 1u ? 1u : 0u)) build_crc_table(s);
}

// Two instances (round 6): WgFull -- sixteen wavefronts per unit, the shortest chain for the units of a LONE frame coded losslessly (the
// fall-back after a unit time-out: 13 ms against 18 with four waves) -- and WgFour, which everything else wants: progressive mode (a 4096^2
// YUV frame at a 70 000-byte quota 1.22 -> 0.83 ms: windows of four chunks stop sooner, three workgroups fit a compute unit where the sixteen-wave
// one -- 115 KiB of LDS, 128 VGPRs with 48 spilled -- fits once) and the fall-back of a batch (8 x 2048^2: 21.5 -> 13.9 ms).
template <class I>
__global__ void __launch_bounds__(64 * I::kWaves)
code_units_wg_kernel(const uint16_t *__restrict__ coef, size_t plane, uint32_t img_w, uint32_t img_h, int channels,
                     const UnitDesc *__restrict__ units, const uint32_t *__restrict__ work_order, uint32_t n_units,
                     const CoderTables *__restrict__ tables, const uint16_t *__restrict__ means,
                     const int *__restrict__ frame_skip, uint8_t *__restrict__ slots,
                     size_t slot_frame_stride, uint32_t *__restrict__ unit_bits, uint64_t *__restrict__ timers,
                     uint32_t *__restrict__ done_bytes, uint64_t early_quota,
                     const uint8_t *__restrict__ sig, size_t sig_frame_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    typename I::Shared &s = *reinterpret_cast<typename I::Shared *>(wg_lds);
    const uint32_t ui = work_order ? work_order[blockIdx.x] : blockIdx.x;      // (null: priority order = unit order)
    wg_shared_tables<I>(s, tables);
    const WgLaunch L{coef, plane, img_w, img_h, channels, units, n_units, means, frame_skip, slots, slot_frame_stride, unit_bits, timers,
                     done_bytes, early_quota, sig, sig_frame_stride};
    wg_code_one_unit<I>(s, L, blockIdx.y, ui);
}

// The coder's small instances (icer::wgs: two wavefronts, 40 KiB of LDS; icer::wg1: one) over a LIST of (frame, unit) pairs -- the units
// route_units_kernel found to be all but blank -- by workgroups that stay: each takes the next entry until the list is
// used up, so the tables and the CRC table are set up once per workgroup and not once per (quickly coded) unit.  The
// kernel runs beside the pipeline kernel, whose workgroups leave room for it on every compute unit.
// grid = a few workgroups per compute unit, block = 128, LDS = sizeof(wgs::Shared).
template <class I>
__global__ void __launch_bounds__(64 * I::kWaves)
code_units_list_kernel(const uint16_t *__restrict__ coef, size_t plane, uint32_t img_w, uint32_t img_h, int channels,
                       const UnitDesc *__restrict__ units, uint32_t n_units,
                       const CoderTables *__restrict__ tables, const uint16_t *__restrict__ means,
                       const int *__restrict__ frame_skip, uint8_t *__restrict__ slots,
                       size_t slot_frame_stride, uint32_t *__restrict__ unit_bits,
                       const uint8_t *__restrict__ sig, size_t sig_frame_stride,
                       const uint32_t *__restrict__ list, uint32_t *__restrict__ list_ctl, uint64_t *__restrict__ timers, uint32_t list_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_lds[];
    typename I::Shared &s = *reinterpret_cast<typename I::Shared *>(wg_lds);
    __shared__ uint32_t next_entry;
    wg_shared_tables<I>(s, tables);
    // (`timers`: profiling build only -- the per-phase cycle counters of the level-1 units, rows of their own behind the pipeline's)
    const WgLaunch L{coef, plane, img_w, img_h, channels, units, n_units, means, frame_skip, slots, slot_frame_stride, unit_bits, timers,
                     nullptr, 0ull, sig, sig_frame_stride};
    const uint32_t count = list_ctl[0], n_heavy = list_ctl[2];   // entries in the list, the heavy ones at its front (route_units_kernel is done)
    for (;;) {
        __syncthreads();                                      // (the last unit's reads of `s` and of next_entry are over)
        if (threadIdx.x == 0) next_entry = atomicAdd(&list_ctl[1], 1u);
        __syncthreads();
        const uint32_t at = next_entry;
        if (at >= count) break;
        const uint32_t e = at < n_heavy ? list[list_cap + at] : list[at - n_heavy];
#ifdef ICER_PHASE_TIMERS
        const uint64_t t_start = wall_clock64();
#endif
        wg_code_one_unit<I>(s, L, e / n_units, e % n_units);
#ifdef ICER_PHASE_TIMERS
        // (the list's timeline: when which workgroup coded which unit -- tools/list_trace.py)
        if (timers && threadIdx.x == 0 && at < (uint32_t)kListTrace && e < n_units) {
            const UnitDesc &ud = units[e];
            uint64_t *tr = timers + 9 * 32 + 4 * at;
            tr[0] = t_start; tr[1] = wall_clock64(); tr[2] = blockIdx.x | ((uint64_t)e << 32);
            tr[3] = (uint64_t)ud.lsb | ((uint64_t)ud.level << 8) | ((uint64_t)ud.subband << 16) | ((uint64_t)ud.seg << 24) | ((uint64_t)((ud.w * ud.h + 63u) / 64u) << 32);
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------ assembly
// One wavefront per frame: quota walk + final offsets.  grid = frames, block = 64.
__global__ void __launch_bounds__(64)
scan_kernel(const uint32_t *__restrict__ unit_bits, const uint32_t *__restrict__ final_order, uint32_t n_units,
            uint64_t quota, const int *__restrict__ frame_skip, uint64_t *__restrict__ final_off,
            unsigned long long *__restrict__ sizes, int32_t *__restrict__ rcs, const UnitDesc *__restrict__ units,
            int *__restrict__ bound_overflow)
{
    const uint32_t frame = blockIdx.x;
    uint64_t *foff = final_off + (size_t)frame * n_units;
    if (frame_skip[frame]) {
        for (uint32_t i = threadIdx.x; i < n_units; i += 64) foff[i] = ~0ull;
        if (threadIdx.x == 0) { sizes[frame] = 0; rcs[frame] = kIntegerOverflow; }
        return;
    }
    const uint32_t *bits = unit_bits + (size_t)frame * n_units;
    {   // any unit that reported an internal error makes the frame fail loudly
        int failed = 0;
        for (uint32_t i = threadIdx.x; i < n_units; i += 64) failed |= bits[i] == kUnitFailed;
        if (__ballot(failed)) {
            for (uint32_t i = threadIdx.x; i < n_units; i += 64) foff[i] = ~0ull;
            if (threadIdx.x == 0) { sizes[frame] = 0; rcs[frame] = kFatalError; atomicOr(bound_overflow, 2); }
            return;
        }
    }
    uint32_t kept;
    uint64_t used;
    const int rc = scan_frame_wave(bits, final_order, n_units, quota, foff, &kept, &used);
    // a unit that overflowed a slot sized by the bits-per-pixel bound (not by the quota) at or
    // before the cut means the bound was too small: the host re-runs with a larger one
    if (kept < n_units && threadIdx.x == 0 && bits[kept] == kUnitTooBig && units[kept].cap_is_bound)
        atomicOr(bound_overflow, 1);
    if (threadIdx.x == 0) { sizes[frame] = used; rcs[frame] = rc; }
}

// copy every kept unit (header + payload) to its place in the final stream.  grid = (units, frames)
__global__ void __launch_bounds__(256)
gather_kernel(const uint8_t *__restrict__ slots, size_t slot_frame_stride, const UnitDesc *__restrict__ units,
              uint32_t n_units, const uint32_t *__restrict__ unit_bits, const uint64_t *__restrict__ final_off,
              uint8_t *__restrict__ out, size_t out_stride)
{
    const uint32_t frame = blockIdx.y, ui = blockIdx.x;
    const uint64_t off = final_off[(size_t)frame * n_units + ui];
    if (off == ~0ull) return;
    const uint32_t len = kHeaderBytes + ((unit_bits[(size_t)frame * n_units + ui] + 7u) >> 3);
    const uint8_t *src = slots + (size_t)frame * slot_frame_stride + units[ui].slot_off;   // 4-byte aligned
    uint8_t *dst = out + (size_t)frame * out_stride + off;
    // destination is byte-aligned only: peel to a 4-byte boundary, then move aligned words built
    // from two source words (v_alignbyte), then the tail
    const uint32_t mis = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u);
    const uint32_t head = mis < len ? mis : len;
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t body_words = (len - head) >> 2;
    const uint32_t *sw = reinterpret_cast<const uint32_t *>(src);
    uint32_t *dw = reinterpret_cast<uint32_t *>(dst + head);
    const uint32_t shift = head * 8;              // source byte offset of the body is `head` (0..3)
    for (uint32_t i = threadIdx.x; i < body_words; i += 256) {
        const uint32_t lo = sw[i];
        uint32_t v = lo;
        if (shift) v = (lo >> shift) | (sw[i + 1] << (32 - shift));
        dw[i] = v;
    }
    const uint32_t done = head + body_words * 4;
    if (threadIdx.x < len - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

}  // namespace icer
