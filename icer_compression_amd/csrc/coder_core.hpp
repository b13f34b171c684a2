// coder_core.hpp -- one workgroup of eight wavefronts codes one ICER coding unit
// (channel, level, subband, plane, segment).
//
// Replaces the reference's per-segment chain (the uint8 twins are the same code on 7 planes)
//   icer_compress_bitplane_uint16   lib_icer/src/icer_context_modeller.c:312-457
//   icer_encode_bit / icer_compute_bin   icer_encoding.c:37-112, icer_util.c:48-56
//   icer_popbuf_while_avail / icer_flush_encode   icer_encoding.c:114-189
// and must emit the identical payload bits.
//
// The segment is coded in chunks of 64 pixels (raster order inside the segment).  The reference is a
// sequential state machine; its state splits into parts that only depend on their own history, so eight
// wavefronts work on consecutive chunks at the same time (a software pipeline), handing chunks over
// through small LDS queues (depth kQueueDepth).  A lone wavefront issues one instruction every four cycles
// whatever its kind, so a wave's time per chunk is its instruction count x 4 plus LDS round trips: the roles
// are cut so that no wave has more than ~550 instructions per chunk.
//
//   pixel wave     per chunk: pixel category, magnitude bit, 8-neighbour context, sign context (64 lanes,
//     (stateless)    pure function of the 3x3 window; the next chunk's window is prefetched); rank of every event
//                    among the chunk's events of the same context (ballot match + v_mbcnt), per-context totals.
//   count wave     the adaptive counts every event sees = counts at chunk start + the prepared rank (the
//     (17 context    at-most-one rescale per context and chunk in closed form); probability fold + bin selection
//      counters)     (exact division + look-up).  Output: 128 event bytes.
//   compaction     bins 1..7: rank of every event inside its bin, the bin's input bits as a dense bit string, the
//     wave           rank -> position list.
//   walker wave    bins 1..7 (variable-to-variable codes): lane b walks the first half of bin b's bit string
//     (partial       through the code tree six bits per table look-up, one lane per tree node walks the second half
//      inputs)       speculatively; marks where code words start.
//   records wave   bins 1..7: from the start flags, per event: does a word start / end here, the finished word,
//     (stateless)    the position of its first event.
//   golomb wave    bin 0 and the Golomb bins 8..16 in closed form on ballot masks, no loop over bins: run length
//     (run lengths)  since the bin's previous one-event, mod m, gives word starts/ends and the finished words.
//   merge wave     ring slots = prefix count of word-start flags (allocation order = order of first events,
//     (ring tail,    E2); end events write the finished word into the slot of the word's first event; chunks in
//      open slots)   which the ring overflows (hybrid_chunk).
//   drain wave     finished words popped from the head of the ring 64 at a time (ballot of done flags, prefix
//     (ring head,    sum of lengths, ds_or into the LDS bit stage); whole 32-bit words stored to HBM.  Parks on
//      bit stage)    request when the merge wave needs the exact ring occupancy (overflow chunks, end of unit).
// The walker, records and golomb waves run ahead of the merge wave's verdict (speculation, see below).
//
// Exactness: word boundaries depend on each bin alone only while the 2048-word ring cannot fill up inside
// the chunk (used + words opened <= 2048): then no forced flush of the oldest open word (E5) can fire.  Otherwise
// (~1-2 % of the chunks of dense planes) the speculative results are right up to the first word start that finds
// the ring full; the merge wave commits up to there, force-completes the oldest word like icer_flush_encode,
// replays the remaining events of that one bin, and the speculating waves discard what they produced for later
// chunks and reload their state (generation counter).
//
// Written with the SPMD macros of wave.hpp (see there for the tests-only CPU build).
#pragma once
#include "events.hpp"
#include "icer_tables.hpp"
#include "wave.hpp"

// (tools/handoff_isa_audit.py builds with -DICER_ISA_MARKERS: assembler comments around every hand-off site, nothing else)
#if defined(ICER_ISA_MARKERS) && !defined(ICER_WAVE_EMU)
#define ICER_STR2(x) #x
#define ICER_STR(x) ICER_STR2(x)
#define ICER_MARK(what) asm volatile("; ICER_MARK " what " line " ICER_STR(__LINE__));
#else
#define ICER_MARK(what)
#endif

#if defined(ICER_WAVE_EMU) && defined(ICER_WAVE_THREADS)
// tests only (tests/emu/threads_main.cpp): the lane-loop build with every wave on its own CPU thread and REAL waits --
// the hand-off protocol under true concurrency, optionally under ThreadSanitizer (fence + relaxed atomic = the LDS
// release / acquire pairs of the GPU build, so a data race it reports is a hole in the protocol).
#include <assert.h>
#include <atomic>
#include <thread>
#define ICER_EMU_COUNT(i)
// (acquire / release on the counters themselves: ThreadSanitizer does not model stand-alone fences)
#define ICER_LOAD_CNT(x) __atomic_load_n(&(x), __ATOMIC_ACQUIRE)
#define ICER_POLL_PAUSE(n) std::this_thread::yield()
#define ICER_SET_ABORT2() __atomic_store_n(&s.abort, 2u, __ATOMIC_RELAXED)
#define ICER_FENCE_ACQ() std::atomic_thread_fence(std::memory_order_acquire)
#define ICER_FENCE_REL() std::atomic_thread_fence(std::memory_order_release)
#define ICER_STORE_CNT(x, v) __atomic_store_n(&(x), (v), __ATOMIC_RELEASE)
#define ICER_LANE0
#define ICER_GLOBAL_RELEASE() std::atomic_thread_fence(std::memory_order_release)
constexpr uint32_t kPollLimit = 50u * 1000u * 1000u;
#elif defined(ICER_WAVE_EMU)
#include <assert.h>
extern unsigned long long g_emu_chunks[4];          // tests only: [0] fast-path chunks, [1] exact-path chunks, golomb wave: [2] chunks in the general form (a sign
                                                    // event in a Golomb bin), [3] in the reduced form
#define ICER_EMU_COUNT(i) (g_emu_chunks[i]++)
// cross-wave hand-off: the emulation runs the waves in an order in which every wait is already satisfied
#define ICER_LOAD_CNT(x) (x)
#define ICER_WAIT_UNTIL(cond) { assert(cond); }
#define ICER_WAIT_CNT(X, V, PRED, AB, SLEEP) uint32_t AB = s.abort; { const uint32_t V = (X); (void)V; assert((PRED) || AB); }
#define ICER_PUBLISH(x, v) { (x) = (v); }
#define ICER_PUBLISH2(x1, v1, x2, v2) { (x1) = (v1); (x2) = (v2); }
#define ICER_ACQUIRE()
#define ICER_GLOBAL_RELEASE()
#define ICER_IDLE() break;     /* the emulation never waits: hand control back to the scheduler */
#define ICER_IDLE_DECL
#define ICER_IDLE_RESET
#else
#define ICER_EMU_COUNT(i)
// counters live in LDS; data written before a PUBLISH is visible to a wave that has seen the new value.
// LDS-only fences: they wait for this wave's LDS traffic (lgkmcnt), never for its global loads/stores.
#define ICER_LOAD_CNT(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#ifndef ICER_SLEEP_SCALE
#define ICER_SLEEP_SCALE 1
#endif
#define ICER_POLL_PAUSE(n) __builtin_amdgcn_s_sleep((n) * ICER_SLEEP_SCALE)
// (the first wave to give up also leaves its wave number and the source line of the wait: code_units_kernel passes them
// on with the unit's counters, so that a time-out names the hand-off it happened in)
#define ICER_SET_ABORT2() { if (__hip_atomic_load(&s.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 2u)            \
        __hip_atomic_store(&s.abort_site, (uint32_t)__LINE__ | ((uint32_t)(threadIdx.x >> 6) << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    __hip_atomic_store(&s.abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#define ICER_FENCE_ACQ() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define ICER_FENCE_REL() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#define ICER_STORE_CNT(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define ICER_LANE0 if (lane == 0)
// all address spaces: also waits for this wave's global stores (vmcnt) -- used once per unit, see drain_wave_run
#define ICER_GLOBAL_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define kPollLimit kSpinLimit
#endif

// cross-WORKGROUP hand-off through global memory (sub-range snapshots): data stored before ICER_AGENT_PUBLISH is visible to a
// workgroup on another compute unit that has seen the flag with ICER_AGENT_ACQUIRE (agent scope: the per-CU vector L1 and the
// per-XCD L2 are not coherent by themselves, MI355X_MICROARCH.md "Correctness boundaries")
#if defined(ICER_WAVE_EMU)
#define ICER_AGENT_PUBLISH(ptr, v) { __atomic_store_n((ptr), (v), __ATOMIC_RELEASE); }
#define ICER_AGENT_ACQUIRE(ptr) __atomic_load_n((ptr), __ATOMIC_ACQUIRE)
#else
#define ICER_AGENT_PUBLISH(ptr, v) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); if (lane == 0) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define ICER_AGENT_ACQUIRE(ptr) icer_agent_acquire_load(ptr)
static __device__ __forceinline__ uint32_t icer_agent_acquire_load(const uint32_t *ptr)
{
    const uint32_t v = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return v;
}
#endif

#if !defined(ICER_WAVE_EMU) || defined(ICER_WAVE_THREADS)
// Every spin is bounded (kPollLimit polls, seconds of wall time): a wave that would wait longer declares the unit
// failed (abort = 2), which every other wait observes; the host then reports ICER_FATAL_ERROR instead of hanging.
#define ICER_SPIN(cond, SLEEP) { ICER_MARK("WAIT_BEGIN") uint32_t spins_ = 0; while (!(cond)) { ICER_POLL_PAUSE(SLEEP); \
        if (++spins_ > kPollLimit) { ICER_SET_ABORT2(); break; } } \
    ICER_FENCE_ACQ(); ICER_MARK("WAIT_END") }
#define ICER_WAIT_UNTIL(cond) ICER_SPIN(cond, 1)
// wait until PRED holds for V = the counter X, or the unit is abandoned; the counter and the abort word are read
// together (one LDS round trip per poll) and AB receives the abort word
#define ICER_WAIT_CNT(X, V, PRED, AB, SLEEP) uint32_t AB; { ICER_MARK("WAIT_BEGIN") uint32_t spins_ = 0; for (;;) {        \
        const uint32_t V = ICER_LOAD_CNT(X); AB = ICER_LOAD_CNT(s.abort);                                             \
        if ((PRED) || AB) break;                                                                                      \
        ICER_POLL_PAUSE(SLEEP);                                                                                       \
        if (++spins_ > kPollLimit) { ICER_SET_ABORT2(); AB = 2u; break; } }                                           \
    ICER_FENCE_ACQ(); ICER_MARK("WAIT_END") }
// (experiments: -DICER_USE_WAKEUP pings the workgroup's sleeping waves after every publish; -DICER_BP_SLEEP=<n> sets the sleep of the
// back-pressure waits)
#if defined(ICER_USE_WAKEUP) && !defined(ICER_WAVE_EMU)
#define ICER_WAKE() asm volatile("s_wakeup");
#else
#define ICER_WAKE()
#endif
#define ICER_PUBLISH(x, v) { const uint32_t pv_ = (v); ICER_MARK("PUBLISH_BEGIN") ICER_FENCE_REL(); ICER_LANE0 ICER_STORE_CNT(x, pv_); ICER_WAKE() ICER_MARK("PUBLISH_END") }
#define ICER_PUBLISH2(x1, v1, x2, v2) { const uint32_t pv1_ = (v1), pv2_ = (v2); ICER_MARK("PUBLISH_BEGIN") ICER_FENCE_REL(); ICER_LANE0 { ICER_STORE_CNT(x1, pv1_); ICER_STORE_CNT(x2, pv2_); } ICER_WAKE() ICER_MARK("PUBLISH_END") }
#define ICER_ACQUIRE() { ICER_MARK("ACQUIRE") ICER_FENCE_ACQ(); }
#define ICER_IDLE() { ICER_POLL_PAUSE(1); if (++idle_spins_ > kPollLimit) { ICER_SET_ABORT2(); break; } }
#define ICER_IDLE_DECL uint32_t idle_spins_ = 0;
#define ICER_IDLE_RESET idle_spins_ = 0;
#endif

// ring words are read by the drain wave while the merge wave may be finishing a later word of the same 64-word window
// (it then sees the open marker or the finished word, both fine): plain 16-bit LDS accesses on the GPU, relaxed atomics
// in the threaded test build so that ThreadSanitizer does not report this intended overlap
#ifdef ICER_WAVE_THREADS
#define RING_LD(i) ((uint32_t)__atomic_load_n(&s.ring[i], __ATOMIC_RELAXED))
#define RING_ST(i, v) __atomic_store_n(&s.ring[i], (uint16_t)(v), __ATOMIC_RELAXED)
#else
#define RING_LD(i) ((uint32_t)s.ring[i])
#define RING_ST(i, v) (s.ring[i] = (uint16_t)(v))
#endif

// Optional per-wave cycle counters (s_memtime) for tools/phase_profile.py; compiled in only with
// -DICER_PHASE_TIMERS (a separate profiling build of the library, never the shipped one).
#if defined(ICER_PHASE_TIMERS) && !defined(ICER_WAVE_EMU)
#define ICER_NUM_TIMERS 32
#define ICER_TIMERS_DECL uint64_t tacc_[ICER_NUM_TIMERS] = {}; uint64_t tlast_ = __builtin_amdgcn_s_memtime();
#define ICER_TIMER_PARAMS , uint64_t *tacc_, uint64_t &tlast_
#define ICER_TIMER_PASS , tacc_, tlast_
#define ICER_TICK(k) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); tacc_[k] += t_ - tlast_; tlast_ = t_; }
#define ICER_COUNT(k) { tacc_[k] += 1; }
#define ICER_TIMERS_STORE(dst) { if (dst) { _Pragma("unroll") for (int i_ = 0; i_ < ICER_NUM_TIMERS; i_++) if (lane == 0 && tacc_[i_]) atomicAdd((unsigned long long *)&(dst)[i_], (unsigned long long)tacc_[i_]); } }
#else
#define ICER_TIMERS_DECL
#define ICER_TIMER_PARAMS
#define ICER_TIMER_PASS
#define ICER_TICK(k)
#define ICER_COUNT(k)
#define ICER_TIMERS_STORE(dst)
#endif

namespace icer {

constexpr uint32_t kStageWords = 1024;      // LDS bit stage (circular, 32-bit words)
constexpr uint32_t kUnitTooBig = 0xFFFFFFFFu;
constexpr uint32_t kUnitFailed = 0xFFFFFFFEu;    // internal error (a bounded spin expired): reported as ICER_FATAL_ERROR
constexpr uint32_t kFailMagic = 0x1CEBAD00u;     // first word of the diagnostic record a failed unit leaves in its payload slot
constexpr uint32_t kFailWords = 16;
constexpr uint32_t kSpinLimit = 1u << 25;
#ifndef ICER_QUEUE_DEPTH
#define ICER_QUEUE_DEPTH 4
#endif
#ifndef ICER_BP_SLEEP
#define ICER_BP_SLEEP 6
#endif
constexpr uint32_t kQueueDepth = ICER_QUEUE_DEPTH;   // chunks in flight between the waves of a unit.  4: 37 KiB of LDS per workgroup, four per
                                                     // CU (8: 48 KiB, three per CU).  Round 3, profiles/archive/r03_logs/r03_x.log: the depth itself does
                                                     // not matter (4 and 8 at three per CU are equal on C2 and C4); four workgroups per CU are
                                                     // + 7 % on C4, + 3 % on C5 -- and - 14 % on a lone frame, which therefore pads its LDS (api.hip)
// The pixel stage keeps no state from chunk to chunk, so several wavefronts can share it: wave k of `npw` takes the
// chunks j with j % npw == k.  The Golomb stage (bins 0, 8..16) is split the same way: what carries over from chunk to
// chunk is only the zero-run length of every bin, and that follows from the chunk's events by a few counts
// (golomb_state_run, one wave); the per-event work is then free of state and shared by `ngw` worker waves.
// Two shapes of a workgroup are built (code_units_kernel<...>): 8 waves -- one pixel wave, one golomb wave that keeps the
// run lengths itself; three workgroups per CU, for batches (more waves per unit cost a batch a quarter of its throughput,
// gpurun_out/r02r) -- and 11 waves -- two pixel waves, golomb state wave + two workers; two workgroups per CU, a shorter
// chain per chunk, for a launch that cannot fill the chip anyway (a single frame).
constexpr uint32_t kMaxPixelWaves = 8, kMaxGolombWorkers = 2;   // (8 pixel waves: the counts-only prefix pass of a sub-range workgroup)
constexpr int kUnitWavesSmall = 8, kUnitWavesLarge = 11;
constexpr int kTraceUnits = 4096;           // profiling build: workgroups of frame 0 whose start / end times are recorded
constexpr int kProfWgsOffset = 9 * 32 + 4 * kTraceUnits + 16;   // the small window coder's rows (code_units_list_kernel), one per bit plane
constexpr int kListTrace = 4096;            // profiling build: list entries of code_units_list_kernel whose start / end times are recorded (frame 0)
constexpr int kProfWords = kProfWgsOffset + 9 * 32 + 4 * kListTrace;              // (the first part: pipeline rows, workgroup trace, HW_ID of each wave of workgroup 0)

// ring word: open  -> owner bin (bit 15 clear)
//            done  -> 0x8000 | nbits << 11 | code (<= 10 bits)
constexpr uint32_t kWordDone = 0x8000u;

struct PixelSlot {              // pixel wave -> count wave
    // per pixel `lane`, word 0 the magnitude-bit event, word 1 the sign event:
    //   bits 0..7   0x80 | bit << 5 | context (31 = uncoded), 0 = none; the sign event's bit is the agreement bit
    //   bits 8..15  rank of the event among the chunk's events of the same context (coding order)
    //   bits 16..23 how many of those earlier ones were zeros
    uint32_t e[64][2];
    uint16_t cn[32];            // per context: events in the chunk | zeros among them << 8
};
struct EventSlot {              // count wave -> walker, golomb and merge waves
    uint8_t ev1[64];            // magnitude-bit event of pixel `lane`: 0x80 | bit << 5 | bin, 0 = none
    uint8_t ev2[64];            // sign event of pixel `lane`
    uint32_t blank;             // 1: the chunk is 64 zero events of context 0 and nothing else (PixelSlot::cn[17])
    uint32_t cnt[kNumBins + 1]; // the adaptive counts after this chunk, context c: zero | total << 16 (sub-range splicing: part of the coder state)
    // bins 1..7, compacted per bin in coding order (rank = number of earlier events of the same bin):
    uint8_t rk1[64], rk2[64];   // rank of this lane's events inside their bin
    uint8_t binseq[8][128];     // rank -> position of the event
    uint32_t binbits[8][6];     // rank -> input bit, stored with an offset of 8 bits
    uint8_t binn[8];            // number of events of the bin
};
struct RecSlot {                // golomb, walker and records waves -> merge wave
    // per event position (2 * lane + slot): bit0 a code word starts at this event, bit1 one ends here, bits 2..6 the
    // event's bin, bits 8..15 the position of that word's first event (255: carried in), bits 16..31 the finished
    // ring word of an end event; 0 where there is no event.  Bins 1..7 are written by the records wave, everything
    // else by the golomb wave.
    uint32_t rec[128];
    // per bin after the chunk (bins 1..7: walker wave, the others: golomb wave): bits 0..7 open_pos -- 255 untouched,
    // 254 closed, else the first event of its open word --, bits 8..23 Golomb run length / partial input value,
    // bits 24..31 input bits accumulated
    uint32_t binst[20];
    uint32_t gtag, rtag;        // (chunk << 8 | generation) + 1 once the golomb / records wave has written its part
};
struct WalkSlot {               // walker wave -> records wave (bins 1..7)
    // word-start flags of bin b by rank (offset of 8 bits like EventSlot::binbits), node carried in
    uint32_t binstart[8][6];
    uint8_t bincarry[8];
    uint8_t post_nin[8];        // input bits of the bin's open word after the chunk
    uint32_t tag;               // as RecSlot::gtag
};

struct RunSlot {                // golomb state wave -> golomb workers: the bins' zero-run lengths at the start of the chunk
    uint32_t tag;               // chunk_tag(chunk, generation)
    uint32_t gk[kNumBins];
};

struct CoderShared {
    uint32_t stage[kStageWords];
    uint16_t ring[kRingWords];
    CoderTables tab;
    uint32_t crc_tab[256];
    PixelSlot pq[kQueueDepth];
    EventSlot eq[kQueueDepth];
    WalkSlot wq[kQueueDepth];
    RecSlot rq[kQueueDepth];
    RunSlot kq[kQueueDepth];
    int32_t bin_slot[kNumBins]; // ring index of the bin's open word, -1 if none
    uint32_t bin_state[kNumBins];   // as RecSlot::binst, as of the last retired chunk (bits 0..7 unused)
    uint8_t srank[128];             // merge wave: position of a word start -> number of word starts before it in the chunk (merge_commit)
    uint32_t gk[kNumBins];          // golomb wave: zero-run length of each Golomb bin's open word as of its last chunk (0 = none)
    // ring occupancy = alloc - popped (both count words since the start of the unit; slot = count mod 2048)
    uint32_t alloc;             // words allocated so far          (merge wave)
    uint32_t popped;            // words popped so far             (drain wave, or the merge wave while it holds the drain)
    uint32_t bitpos;            // payload bits produced so far    (same owner as popped)
    uint32_t flushed_words;     // payload words already written to HBM
    // merge -> drain wave: odd = "park, I need the drain state", even = released; the drain wave answers in hold_ack
    uint32_t hold_seq, hold_ack, drain_exit;
    uint32_t nchunks;           // chunks of the unit
    // progress counters of the three waves (chunks completed) and the per-chunk verdicts
    uint32_t p_done[kMaxPixelWaves];   // per pixel wave: 1 + the last chunk it has handed over
    uint32_t a_done, c_done, b_done, abort;
    uint32_t abort_site;        // who set abort = 2: source line | wave << 16 (GPU build; diagnostics only)
    // speculation control: the walker and golomb waves run ahead assuming the fast path; every chunk the merge
    // wave had to replay exactly bumps exact_seq, which invalidates all results produced for later chunks
    uint32_t exact_seq, last_exact;
#ifdef ICER_LDS_PAD_BYTES                  // (experiments: occupancy at a fixed queue depth)
    uint8_t lds_pad[ICER_LDS_PAD_BYTES];
#endif
};

// ------------------------------------------------------------------------------------------
// Sub-ranges: several workgroups per coding unit (launches too small to fill the chip: a single frame)
// ------------------------------------------------------------------------------------------
// A dense unit is a chain of thousands of chunks, and a frame has fewer such units than the chip has compute units.  The
// unit's chunk range is therefore cut into K sub-ranges, each coded by a workgroup of its own, all at the same time.
// Workgroup i (first chunk c_i) cannot know the coder state at c_i -- the open code word of every bin, the ring --, but:
//   * the adaptive counts at c_i depend on the coefficients before c_i alone: the workgroup first runs the pixel and count
//     stages over [0, c_i) with all its waves (count_wave_run(counts_only), a fraction of the full cost per chunk);
//   * started COLD (no open words, empty ring) with the exact counts, its bins are the exact ones from the first event on,
//     and its state converges to the true one: a Golomb bin is in step after its first one-event, a variable-to-variable
//     bin once both walks meet in a tree node, the ring once every word that was open at c_i has been closed (on dense bit
//     planes: after 66 chunks on average, 587 at most, tests/research/heal_experiment.c).
// Exactness does not rest on that expectation.  Workgroup i writes SNAPSHOTS of its complete coder state (counts, every
// bin's state and open slot, the ring from its oldest word to its tail, its bit position) every kSnapEvery chunks after c_i;
// the workgroup before it keeps coding past c_i and compares ITS state with the snapshot at the same chunk.  Equal states
// and equal input from there on give equal output, so it stops: the unit's payload is its bits up to that point followed
// by workgroup i's bits from the snapshot's bit position on (splice_unit_wave).  Without a match (sparse planes, where words
// stay open for thousands of chunks) it simply codes on -- through the whole unit if need be, which is what a unit that
// is not split costs.  Nothing ever waits for another workgroup.
#ifndef ICER_SNAP_EVERY
#define ICER_SNAP_EVERY 64
#endif
constexpr uint32_t kSnapEvery = ICER_SNAP_EVERY;   // chunks between snapshots (the tests-only CPU build also runs with 4: small units)
constexpr uint32_t kMaxSnaps = 16;              // snapshots per sub-range (up to 1024 chunks past its first chunk)
constexpr uint32_t kMaxSubs = 8;                // sub-ranges per unit

struct Snapshot {                               // global memory
    uint32_t chunk;                             // state BEFORE this chunk
    uint32_t bitpos;                            // the writer's payload bits so far (everything finished is popped)
    uint32_t nring;                             // ring words from the oldest (open) word to the tail
    uint32_t cnt[kNumBins];                     // adaptive counts (EventSlot::cnt)
    uint32_t bin_state[kNumBins];               // CoderShared::bin_state >> 8
    uint32_t open_off[kNumBins];                // ring offset of the bin's open word from the oldest word, ~0 = none
    uint32_t pad[12];
    uint16_t ring[kRingWords];
};
struct SubRecord {                              // what a workgroup of a split unit leaves behind (global memory)
    uint32_t done;                              // 1 once the fields below are valid
    uint32_t end_chunk;                         // it coded [its first chunk, end_chunk)
    uint32_t end_bits;                          // payload bits in its slot (kUnitTooBig / kUnitFailed as unit_bits)
    uint32_t match_sub, match_snap;             // stopped because its state equalled this snapshot (match_sub 0: ran to the unit's end)
    uint32_t pad[3];
};
struct SubLayout {                              // one per (frame, split unit), built by the kernel from the plan's tables
    uint32_t n_sub;                             // K
    uint32_t index;                             // this workgroup's sub-range
    uint32_t first[kMaxSubs + 1];               // first chunk of every sub-range; first[K] = chunks of the unit
    Snapshot *snaps;                            // [K][kMaxSnaps] (row 0 unused)
    uint32_t *snap_valid;                       // [K][kMaxSnaps], 1 once the snapshot is complete (agent-scope release / acquire)
    SubRecord *rec;                             // [K]
};

struct UnitArgs {
    const uint16_t *seg;        // first coefficient of the segment (sign-magnitude words)
    const uint8_t *ev;          // the unit's event bytes (events.hpp): chunk j = bytes [64 j, 64 j + 64), unwritten where the chunk is blank
    const uint8_t *sig;         // chunk table of the unit's family: chunk j is blank at every bit plane >= sig[j]
    uint32_t stride;            // plane row stride in elements
    uint32_t w, h;              // segment size
    int subband, lsb;
    uint32_t *out_words;        // payload slot (4-byte aligned)
    uint32_t cap_words;         // slot capacity in 32-bit words
    uint64_t *timers;           // profiling build only (may be null)
    // progressive mode (small byte quota): the frame's per-unit results so far in priority order, this unit's place in
    // that order and the quota; null / 0 otherwise.  See quota_already_spent.
    const uint32_t *done_bytes;
    uint32_t prio_index;
    uint64_t early_quota;
    // sub-range splicing (see "Sub-ranges" below): null / 0 when the unit is coded by one workgroup from its first chunk
    const SubLayout *sub = nullptr;
};

// Progressive mode.  The stream keeps units in priority order until the first one that does not fit the byte quota
// (icer_partition.c:321-336, `break` in the packet loop); everything after it is dropped.  done_bytes[j] is 0 while
// unit j is unfinished, its size (header + payload bytes) once it is coded, ~0 if it can not fit whatever comes
// before it.  If the finished units of higher priority ALONE already exceed the quota, the cut lies before this unit
// and it can stop: its result can not be part of the stream.  (A lower bound of the prefix sum, so never a false
// positive; units before the cut always run to completion.)  One wavefront; wave-uniform result.
#ifndef ICER_WAVE_EMU
ICER_DEV bool quota_already_spent(const UnitArgs &a)
{
    if (!a.early_quota) return false;
    DECL_LANE;
    unsigned long long sum = 0;
    for (uint32_t j = (uint32_t)lane; j < a.prio_index; j += 64) {
        const uint32_t d = __hip_atomic_load(&a.done_bytes[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += d == ~0u ? a.early_quota + 1ull : (unsigned long long)d;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    return sum > a.early_quota;
}
#elif defined(ICER_WAVE_THREADS)
// tests only (threaded build): the stop is a flag that another thread raises at an arbitrary moment
ICER_DEV bool quota_already_spent(const UnitArgs &a)
{
    return a.early_quota && __atomic_load_n(a.done_bytes, __ATOMIC_RELAXED) != 0u;
}
#endif

// ------------------------------------------------------------------------------------------
// exact coder steps (restatement of E1-E6; draining is left to wave_drain)
// ------------------------------------------------------------------------------------------
// Golomb codeword for a run of k zeros ended by a one (icer_encoding.c:73-80)
ICER_DEV uint32_t golomb_word(const CoderTables &t, int bin, uint32_t k)
{
    const uint32_t gi = t.gi[bin];
    const uint32_t code = k + (k >= gi ? gi : 0u);
    const uint32_t n = t.gl[bin] + (k >= gi ? 1u : 0u);
    return kWordDone | (n << 11) | ((brev32(code) >> (32u - n)) & 0x3FFu);
}

// packed per-bin coder state (RecSlot::binst, CoderShared::bin_state)
ICER_DEV uint32_t st_acc(uint32_t st) { return (st >> 8) & 0xFFFFu; }
ICER_DEV uint32_t st_nin(uint32_t st) { return st >> 24; }
ICER_DEV uint32_t st_pack(uint32_t op, uint32_t acc, uint32_t nin) { return op | (acc << 8) | (nin << 24); }

// first half of icer_flush_encode (icer_encoding.c:141-189): force-complete the oldest word.
// The caller drains afterwards (wave_drain).
ICER_DEV void seq_complete_head(CoderShared &s)
{
    const uint32_t head = s.popped & (kRingWords - 1);
    const uint32_t w = RING_LD(head);
    if (!(w & kWordDone)) {
        const int bin = (int)(w & 31u);
        if (bin >= 8) {
            const uint32_t k = st_acc(s.bin_state[bin]);
            RING_ST(head, (k == (uint32_t)s.tab.gm[bin] - 1u) ? (kWordDone | (1u << 11) | 1u) : golomb_word(s.tab, bin, k));
            s.bin_state[bin] = 0;
            s.bin_slot[bin] = -1;
        } else if (bin >= 1) {
            const uint32_t nin = st_nin(s.bin_state[bin]), acc = st_acc(s.bin_state[bin]);
            const uint32_t pv = acc > 8u ? 8u : acc;                                // partial values are <= 8
            const uint32_t f = s.tab.v2v_flush[bin][pv][nin > 5u ? 5u : nin];
            const uint32_t pre = (acc | ((f & 15u) << nin)) & 31u;
            const uint32_t e = s.tab.v2v[bin][pre];
            // QUIRK (kept): the completed input is not checked to be a real code word
            RING_ST(head, (kWordDone | (((e >> 4) & 15u) << 11) | (e >> 8)));
            s.bin_state[bin] = 0;
            s.bin_slot[bin] = -1;
        }
    }
}

// pick the coder bin from a folded (zero >= total/2) probability estimate: the number of
// cut-offs not above zero/total (icer_compute_bin, icer_util.c:48-56; cut-offs are ascending).
// zero * 65536 >= total * cut  <=>  floor(zero * 65536 / total) >= cut, so one exact division (total <= 500: a
// float reciprocal estimate is within 1 of the quotient, then corrected) and one table look-up replace the 16 compares.
ICER_DEV uint32_t pick_bin(const uint32_t *binlut, uint32_t zero, uint32_t total)
{
    const uint32_t a = zero << 16;
#ifdef ICER_WAVE_EMU
    uint32_t q = (uint32_t)((float)a * (1.0f / (float)total));
    int32_t rem = (int32_t)a - (int32_t)(q * total);
#else
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)total));
    int32_t rem = (int32_t)a - (int32_t)__umul24(q, total);                  // q <= 2^16 + 1, total <= 500
#endif
    if (rem < 0) { q--; rem += (int32_t)total; }
    if (rem >= (int32_t)total) q++;
    const uint32_t e = binlut[q >> 8];
    return (e & 255u) + (q >= (e >> 8) ? 1u : 0u);
}

// ------------------------------------------------------------------------------------------
// building blocks shared by the waves
// ------------------------------------------------------------------------------------------
// Adaptive counts for every event of context C in this chunk (phase 2).  A context is rescaled
// when its total reaches 500 (-> 250); with at most 64 events per context and chunk that can
// happen at most once per chunk.  QUIRK C5: at a rescale `zero` is halved only if it exceeds the
// halved total.
#define ICER_CTX_STEP(C, PRED, ISZERO, ZOUT, TOUT)                                                    \
    {                                                                                                 \
        const uint64_t m_ = BALLOT(PRED);                                                             \
        if (m_) {                                                                                     \
            const uint64_t zm_ = BALLOT((PRED) && (ISZERO));                                          \
            const uint32_t n_ = (uint32_t)popc64(m_), nz_ = (uint32_t)popc64(zm_);                    \
            const uint32_t t0_ = READLANE(ctot, C), z0_ = READLANE(czer, C);                          \
            if (t0_ + n_ < kRescaleCap) {                                                             \
                FOR_LANES                                                                             \
                {                                                                                     \
                    if (PRED) {                                                                       \
                        LV(TOUT) = t0_ + (uint32_t)mbcnt64(m_, lane);                                 \
                        LV(ZOUT) = z0_ + (uint32_t)mbcnt64(zm_, lane);                                \
                    }                                                                                 \
                }                                                                                     \
                FOR_LANES { if ((uint32_t)lane == (C)) { LV(ctot) = t0_ + n_; LV(czer) = z0_ + nz_; } }      \
            } else {                                                                                  \
                const uint32_t kc_ = kRescaleCap - 1 - t0_; /* rank of the event that triggers it */  \
                const int lc_ = ffs64(BALLOT((PRED) && (uint32_t)mbcnt64(m_, lane) == kc_));          \
                const uint32_t zc_ = (uint32_t)popc64(zm_ & ((2ull << lc_) - 1ull));                  \
                const uint32_t zat_ = z0_ + zc_;                                                      \
                const uint32_t zr_ = (zat_ > kRescaleCap / 2) ? (zat_ >> 1) : zat_;                   \
                FOR_LANES                                                                             \
                {                                                                                     \
                    if (PRED) {                                                                       \
                        const uint32_t rk_ = (uint32_t)mbcnt64(m_, lane), zb_ = (uint32_t)mbcnt64(zm_, lane); \
                        if (rk_ <= kc_) { LV(TOUT) = t0_ + rk_; LV(ZOUT) = z0_ + zb_; }               \
                        else { LV(TOUT) = kRescaleCap / 2 + (rk_ - kc_ - 1); LV(ZOUT) = zr_ + (zb_ - zc_); } \
                    }                                                                                 \
                }                                                                                     \
                FOR_LANES { if ((uint32_t)lane == (C)) { LV(ctot) = kRescaleCap / 2 + (n_ - kc_ - 1); LV(czer) = zr_ + (nz_ - zc_); } } \
            }                                                                                         \
        }                                                                                             \
    }

// write the complete 32-bit words of the bit stage to HBM; returns false when the slot is full
ICER_DEV bool flush_stage(CoderShared &s, const UnitArgs &a, bool final_partial)
{
    DECL_LANE;
    const uint32_t bp = s.bitpos;
    const uint32_t first = s.flushed_words;
    uint32_t last = bp >> 5;
    if (final_partial && (bp & 31u)) last++;
    const bool fits = last <= a.cap_words;
    const uint32_t stop = fits ? last : a.cap_words;
    FOR_LANES
    {
        for (uint32_t wi = first + (uint32_t)lane; wi < last; wi += 64) {
            const uint32_t v = s.stage[wi & (kStageWords - 1)];
            if (wi < stop) a.out_words[wi] = v;
            s.stage[wi & (kStageWords - 1)] = 0;
        }
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane == 0) s.flushed_words = last;
    }
    WAVE_SYNC();
    // a unit whose complete bytes reach the capacity can never fit (P3 of SURVEY.md 2.3; HISTORY.md 3)
    return fits && (bp >> 3) < a.cap_words * 4u;
}

ICER_DEV uint64_t below64(uint32_t x) { return x >= 64u ? ~0ull : ((1ull << x) - 1ull); }
// The same questions about THIS lane's own events (position 2 * lane + slot) of masks that differ from lane to lane
// (vector registers): lane-relative masks from 32-bit operations, no 64-bit shift by a variable.
ICER_DEV uint32_t own_bit(uint64_t A, int lane) { return ((lane < 32 ? (uint32_t)A : (uint32_t)(A >> 32)) >> (lane & 31)) & 1u; }
ICER_DEV uint64_t lanes_below(int lane)         // bits of the lanes below this one
{
    const uint32_t lo = lane < 32 ? (1u << (lane & 31)) - 1u : ~0u, hi = lane < 32 ? 0u : (1u << (lane & 31)) - 1u;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// events of the set strictly before position pos
ICER_DEV uint32_t cnt_lt(uint64_t A1, uint64_t A2, uint32_t pos)
{
    return (uint32_t)(popc64(A1 & below64((pos + 1u) >> 1)) + popc64(A2 & below64(pos >> 1)));
}
// the same for this lane's own events (position 2 * lane + slot): two v_mbcnt pairs instead of 64-bit shifts
ICER_DEV uint32_t cnt_lt_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    return (uint32_t)(mbcnt64(A1, lane) + mbcnt64(A2, lane)) + (slot ? own_bit(A1, lane) : 0u);
}
// latest position <= pos in the set, -1 if none
ICER_DEV int last_le(uint64_t A1, uint64_t A2, uint32_t pos)
{
    const uint64_t c1 = A1 & below64((pos >> 1) + 1u), c2 = A2 & below64((pos + 1u) >> 1);
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
ICER_DEV int last_lt(uint64_t A1, uint64_t A2, uint32_t pos) { return pos == 0u ? -1 : last_le(A1, A2, pos - 1u); }
// latest position < (2 * lane + slot) / <= (2 * lane + slot) in the set, -1 if none
ICER_DEV int last_lt_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    const uint64_t lt = lanes_below(lane);
    const uint64_t c1 = A1 & (slot ? (lt << 1) | 1ull : lt), c2 = A2 & lt;
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
ICER_DEV int last_le_own(uint64_t A1, uint64_t A2, int lane, uint32_t slot)
{
    const uint64_t lt = lanes_below(lane), le = (lt << 1) | 1ull;
    const uint64_t c1 = A1 & le, c2 = A2 & (slot ? le : lt);
    const int k1 = c1 ? 2 * (63 - clz64(c1)) : -1, k2 = c2 ? 2 * (63 - clz64(c2)) + 1 : -1;
    return k1 > k2 ? k1 : k2;
}
// the same with events in the first position of a lane only (A2 = 0)
ICER_DEV uint32_t cnt_lt1(uint64_t A1, uint32_t pos) { return (uint32_t)popc64(A1 & below64((pos + 1u) >> 1)); }
ICER_DEV uint32_t cnt_lt_own1(uint64_t A1, int lane, uint32_t slot) { return (uint32_t)mbcnt64(A1, lane) + (slot ? own_bit(A1, lane) : 0u); }
ICER_DEV int last_lt_own1(uint64_t A1, int lane, uint32_t slot)
{
    const uint64_t lt = lanes_below(lane);
    const uint64_t c1 = A1 & (slot ? (lt << 1) | 1ull : lt);
    return c1 ? 2 * (63 - clz64(c1)) : -1;
}
ICER_DEV int last_le_own1(uint64_t A1, int lane)
{
    const uint64_t c1 = A1 & ((lanes_below(lane) << 1) | 1ull);
    return c1 ? 2 * (63 - clz64(c1)) : -1;
}

// drain finished words from the head of the ring, 64 per round: lengths -> prefix sum -> bit offsets,
// code bits OR-ed into the LDS bit stage (icer_popbuf_while_avail, icer_encoding.c:114-139)
// `limit` = allocation count up to which ring slots are valid; at most `max_rounds` rounds of 64 words.  Returns the
// number of words popped.
ICER_DEV uint32_t wave_drain(CoderShared &s, uint32_t limit, uint32_t max_rounds)
{
    DECL_LANE;
    const uint32_t popped0 = s.popped;
    uint32_t head = popped0 & (kRingWords - 1), used = limit - popped0, bitpos = s.bitpos;
    uint32_t npop = 0;
    for (uint32_t round = 0; round < max_rounds; round++) {
        LANEVAR(uint32_t, w); LANEVAR(uint32_t, len); LANEVAR(uint32_t, off);
        FOR_LANES
        {
            LV(w) = (uint32_t)lane < used ? RING_LD((head + (uint32_t)lane) & (kRingWords - 1)) : 0u;
        }
        const uint64_t done = BALLOT((LV(w) & kWordDone) != 0u);
        const uint32_t n = (uint32_t)ffs64(~done);               // leading finished words
        if (n == 0) break;
        FOR_LANES
        {
            LV(len) = (uint32_t)lane < n ? ((LV(w) >> 11) & 15u) : 0u;
        }
        // exclusive prefix sum of the lengths (< 16): one ballot per bit of the length, lane-masked popcounts
        const uint64_t L0 = BALLOT(LV(len) & 1u), L1 = BALLOT(LV(len) & 2u), L2 = BALLOT(LV(len) & 4u), L3 = BALLOT(LV(len) & 8u);
        const uint32_t total = (uint32_t)(popc64(L0) + 2 * popc64(L1) + 4 * popc64(L2) + 8 * popc64(L3));
        FOR_LANES
        {
            LV(off) = (uint32_t)(mbcnt64(L0, lane) + 2 * mbcnt64(L1, lane) + 4 * mbcnt64(L2, lane) + 8 * mbcnt64(L3, lane));
        }
        FOR_LANES
        {
            if (LV(len)) {
                const uint32_t p = bitpos + LV(off), wi = (p >> 5) & (kStageWords - 1), sh = p & 31u;
                const uint32_t code = LV(w) & 0x3FFu;
                LDS_OR(s.stage[wi], code << sh);
                if (sh + LV(len) > 32u) LDS_OR(s.stage[(wi + 1) & (kStageWords - 1)], code >> (32u - sh));
            }
        }
        bitpos += total;
        head = (head + n) & (kRingWords - 1);
        used -= n;
        npop += n;
        if (n < 64u) break;
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (lane == 0) s.bitpos = bitpos;
    }
    ICER_PUBLISH(s.popped, popped0 + npop)
    WAVE_SYNC();
    return npop;
}


// ==========================================================================================
// pixel wave
// ==========================================================================================
struct PixelWave {                // this wave's next chunk, prefetched
    LANEVAR(uint32_t, nb);          // event byte of pixel `lane` (events.hpp); not loaded for a blank chunk
    uint32_t fetched = ~0u;         // chunk whose bytes are in flight / in the register above (~0: none yet)
    uint64_t blank = 0;             // chunks [blank_base, blank_base + 64) of the unit that are blank at its plane (chunk table)
    uint32_t blank_base = ~0u;
};

// The unit's chunk table (family_events_kernel): chunk j is blank at the unit's bit plane iff sig[j] <= lsb.  Read 64 chunks at a
// time -- one byte per lane, one ballot -- so that a chunk costs the pixel wave a scalar bit test.
#define ICER_BLANK_MASK(J)                                                                             \
    if (((J) & ~63u) != cw.blank_base) {                                                               \
        cw.blank_base = (J) & ~63u;                                                                    \
        cw.blank = BALLOT(cw.blank_base + (uint32_t)lane < nchunks && (uint32_t)a.sig[cw.blank_base + (uint32_t)lane] <= lsb); \
    }
#define ICER_IS_BLANK(J) (((cw.blank >> ((J) & 63u)) & 1ull) != 0ull)

// event byte -> the pixel's two events (events.hpp)
#define ICER_UNPACK_EVENT_BYTE(B, V1, C1, B1, V2, C2, B2)                                              \
    {                                                                                                  \
        const uint32_t b_ = (B), c4_ = b_ & 15u;                                                       \
        V1 = c4_ != kEvNone ? 1u : 0u;                                                                 \
        C1 = c4_ == kEvUncoded ? 31u : c4_;                                                            \
        B1 = (b_ >> 4) & 1u;                                                                           \
        V2 = (c4_ <= 8u && (b_ & 16u)) ? 1u : 0u;       /* the pixel becomes significant here: its sign follows (C6) */ \
        C2 = 12u + ((b_ >> 5) & 3u);                                                                   \
        B2 = (b_ >> 7) & 1u;                                                                           \
    }

// pixel wave k of npw: its chunks (j % npw == k) among [j0, j1); consecutive calls continue the prefetch (cw.fetched)
// The events themselves -- category, bit, 8-neighbour context, sign context and prediction (C1-C6) -- were made once for all bit
// planes of the unit's family (events.hpp, family_events_kernel): one byte per pixel, 64 consecutive bytes per chunk.
ICER_DEV void pixel_wave_run(CoderShared &s, const UnitArgs &a, PixelWave &cw, uint32_t j0, uint32_t j1, uint32_t k, uint32_t npw)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    const uint32_t lsb = (uint32_t)a.lsb;
    const uint32_t jfirst = j0 + (k + npw - j0 % npw) % npw;                        // first chunk >= j0 of this wave
    if (cw.fetched != jfirst && jfirst < nchunks) {                                  // (first call, or a new start: nothing in flight)
        ICER_BLANK_MASK(jfirst)
        if (!ICER_IS_BLANK(jfirst)) { FOR_LANES { LV(cw.nb) = a.ev[(size_t)jfirst * 64u + (uint32_t)lane]; } }
        cw.fetched = jfirst;
    }

    for (uint32_t j = jfirst; j < j1; j += npw) {
        LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
        LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
        ICER_BLANK_MASK(j)
        const bool blank = ICER_IS_BLANK(j);
        LANEVAR(uint32_t, eb);
        FOR_LANES { LV(eb) = blank ? 0u : LV(cw.nb); }
        // this wave's next chunk is fetched while this one is processed (the LDS-only fences never drain vmcnt)
        if (j + npw < nchunks) {
            ICER_BLANK_MASK(j + npw)
            if (!ICER_IS_BLANK(j + npw)) { FOR_LANES { LV(cw.nb) = a.ev[(size_t)(j + npw) * 64u + (uint32_t)lane]; } }
        }
        cw.fetched = j + npw;
        FOR_LANES { ICER_UNPACK_EVENT_BYTE(LV(eb), LV(valid1), LV(ctx1), LV(bit1), LV(valid2), LV(ctx2), LV(bit2)) }
        // ---- context groups --------------------------------------------------------------------------
        // What the adaptive counts (C5) need from the chunk depends on the coefficients alone and is prepared
        // here, off the serial path: an event sees the counts at chunk start + its rank among the chunk's events
        // of the same context (+ the zeros among those).  Lanes with the same context are found from per-bit
        // ballots of the context number (no loop over contexts); lane c applies the same match to "context c".
        LANEVAR(uint32_t, w1); LANEVAR(uint32_t, w2); LANEVAR(uint32_t, cnw);
#define ICER_MATCH(KEY, V, B0, B1, B2, B3) \
        ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
        // A blank chunk -- 64 pixels that are and stay insignificant with no significant neighbour, i.e. 64 zero events
        // of context 0 and no sign event; more than half of all chunks, the high planes mostly -- needs no matching:
        // the rank of an event is its lane number and so is the number of zeros before it.
#ifdef ICER_WAVE_EMU
        assert(blank == (BALLOT(!LV(valid1) || LV(ctx1) != 0u || LV(bit1) != 0u || LV(valid2)) == 0ull));     // (tests) the chunk table agrees with the events
#endif
        if (blank) {
            FOR_LANES
            {
                LV(w1) = 0x80u | ((uint32_t)lane << 8) | ((uint32_t)lane << 16);
                LV(w2) = 0;
                LV(cnw) = lane == 0 ? (64u | (64u << 8)) : 0u;
            }
        } else {
            // magnitude-bit events: contexts 0..11; sign events: contexts 12..16, keyed by context - 12
            const uint64_t V = BALLOT(LV(valid1) && LV(ctx1) != 31u);
            const uint64_t B0 = BALLOT(LV(ctx1) & 1u), B1 = BALLOT(LV(ctx1) & 2u), B2 = BALLOT(LV(ctx1) & 4u), B3 = BALLOT(LV(ctx1) & 8u);
            const uint64_t ZM = BALLOT(LV(bit1) == 0u);
            const uint64_t U = BALLOT(LV(valid2) != 0u);
            const uint64_t C0 = BALLOT((LV(ctx2) - 12u) & 1u), C1 = BALLOT((LV(ctx2) - 12u) & 2u), C2 = BALLOT((LV(ctx2) - 12u) & 4u);
            const uint64_t ZN = BALLOT(LV(bit2) == 0u);
            // The events of a context are matched once per CONTEXT, by lane c for context c (it needs them for the context's
            // totals anyway); an event's lane fetches the mask of its context from that lane (two ds_bpermute) instead of
            // matching its own key against the ballots again (22 VALU instructions per event position).
            LANEVAR(uint32_t, om_lo); LANEVAR(uint32_t, om_hi); LANEVAR(uint32_t, g1l); LANEVAR(uint32_t, g1h); LANEVAR(uint32_t, g2l); LANEVAR(uint32_t, g2h);
            LANEVAR(uint32_t, x1); LANEVAR(uint32_t, x2);
            FOR_LANES
            {
                uint64_t m = 0, zs = ZM;
                if (U) {
                    // (magnitude contexts on lanes 0..11, sign contexts on lanes 12..16: one match on selected ballots)
                    const bool sg = lane >= 12;
                    const uint32_t key = sg ? (uint32_t)lane - 12u : (uint32_t)lane;
                    const uint64_t vv = sg ? U : V, b0 = sg ? C0 : B0, b1 = sg ? C1 : B1, b2 = sg ? C2 : B2, b3 = sg ? 0ull : B3;
                    if (lane <= 16) m = ICER_MATCH(key, vv, b0, b1, b2, b3);
                    zs = sg ? ZN : ZM;
                } else if (lane < 12) m = ICER_MATCH((uint32_t)lane, V, B0, B1, B2, B3);
                LV(om_lo) = (uint32_t)m; LV(om_hi) = (uint32_t)(m >> 32);
                LV(cnw) = (uint32_t)popc64(m) | ((uint32_t)popc64(m & zs) << 8);
                LV(x1) = LV(ctx1) & 31u; LV(x2) = LV(ctx2) & 31u;                   // (context 31, uncoded: lane 31 holds an empty mask)
                LV(g1l) = 0; LV(g1h) = 0; LV(g2l) = 0; LV(g2h) = 0;
            }
            WAVE_GATHER(g1l, om_lo, x1) WAVE_GATHER(g1h, om_hi, x1)
            if (U) { WAVE_GATHER(g2l, om_lo, x2) WAVE_GATHER(g2h, om_hi, x2) }
            FOR_LANES
            {
                LV(w1) = 0; LV(w2) = 0;
                if (LV(valid1)) {
                    LV(w1) = 0x80u | (LV(bit1) << 5) | LV(ctx1);
                    if (LV(ctx1) != 31u) {
                        const uint64_t m = (uint64_t)LV(g1l) | ((uint64_t)LV(g1h) << 32);
                        LV(w1) |= ((uint32_t)mbcnt64(m, lane) << 8) | ((uint32_t)mbcnt64(m & ZM, lane) << 16);
                    }
                }
                if (LV(valid2)) {
                    LV(w2) = 0x80u | (LV(bit2) << 5) | LV(ctx2);
                    const uint64_t m = (uint64_t)LV(g2l) | ((uint64_t)LV(g2h) << 32);
                    LV(w2) |= ((uint32_t)mbcnt64(m, lane) << 8) | ((uint32_t)mbcnt64(m & ZN, lane) << 16);
                }
            }
        }
#undef ICER_MATCH
        ICER_TICK(0)
        // queue slot j % D is free once the count wave has consumed chunk j - D
        ICER_WAIT_CNT(s.a_done, ad_, j < ad_ + kQueueDepth, ab_, ICER_BP_SLEEP)
        if (ab_) break;
        ICER_TICK(1)
        PixelSlot &o = s.pq[j % kQueueDepth];
        FOR_LANES
        {
            o.e[lane][0] = LV(w1);
            o.e[lane][1] = LV(w2);
            if (lane < 17) o.cn[lane] = (uint16_t)LV(cnw);
            if (lane == 17) o.cn[17] = blank ? 1u : 0u;            // handed on to the golomb wave (EventSlot::blank)
        }
        ICER_PUBLISH(s.p_done[k], j + 1u)
    }
    ICER_TIMERS_STORE(a.timers)
}

// ==========================================================================================
// count wave
// ==========================================================================================
struct CountWave {              // lane c: adaptive counts of context c (icer_context_model_typedef, icer.h:195-199)
    LANEVAR(uint32_t, czer);
    LANEVAR(uint32_t, ctot);
};

ICER_DEV void count_wave_run(CoderShared &s, const UnitArgs &a, CountWave &cs, uint32_t j0, uint32_t j1, uint32_t npw)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    LANEVAR(uint32_t, czer); LANEVAR(uint32_t, ctot);
    FOR_LANES
    {
        LV(czer) = j0 == 0 ? 2u : LV(cs.czer);                    // icer_context_modeller.c:607-613
        LV(ctot) = j0 == 0 ? 4u : LV(cs.ctot);
    }
    for (uint32_t j = j0; j < j1; j++) {
        ICER_WAIT_CNT(s.p_done[j % npw], pd_, pd_ > j, ab_, 1)
        if (ab_) break;
        ICER_TICK(2)
        const PixelSlot &in = s.pq[j % kQueueDepth];
        LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
        LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
        LANEVAR(uint32_t, z1); LANEVAR(uint32_t, t1); LANEVAR(uint32_t, z2); LANEVAR(uint32_t, t2);
        LANEVAR(uint32_t, rk1); LANEVAR(uint32_t, zb1); LANEVAR(uint32_t, rk2); LANEVAR(uint32_t, zb2);
        LANEVAR(uint32_t, cn); LANEVAR(uint32_t, cnz);
        FOR_LANES
        {
            const uint32_t c1 = in.e[lane][0], c2 = in.e[lane][1];
            LV(valid1) = (c1 >> 7) & 1u; LV(ctx1) = c1 & 31u; LV(bit1) = (c1 >> 5) & 1u; LV(rk1) = (c1 >> 8) & 255u; LV(zb1) = (c1 >> 16) & 255u;
            LV(valid2) = (c2 >> 7) & 1u; LV(ctx2) = c2 & 31u; LV(bit2) = (c2 >> 5) & 1u; LV(rk2) = (c2 >> 8) & 255u; LV(zb2) = (c2 >> 16) & 255u;
            const uint32_t w = lane < 17 ? (uint32_t)in.cn[lane] : 0u;
            LV(cn) = w & 255u; LV(cnz) = w >> 8;
            LV(z1) = 1; LV(t1) = 2;                               // what an uncoded event presents (C2)
            LV(z2) = 0; LV(t2) = 0;
        }

#ifdef ICER_WAVE_THREADS
        assert(ICER_LOAD_CNT(s.p_done[j % npw]) <= j + kQueueDepth);      // (test build) the pixel wave has not recycled this slot
#endif
        // ---- adaptive counts per event (C5) --------------------------------------------------------------
        // counts an event sees = its context's counts at chunk start + the ranks the pixel wave prepared; lane c
        // owns context c and advances its counters by the chunk's totals.  A context that reaches the rescale
        // point inside this chunk (total 500, at most once per chunk) is redone by ICER_CTX_STEP.
        {
            LANEVAR(uint32_t, t0); LANEVAR(uint32_t, zz0); LANEVAR(uint32_t, idx);
            FOR_LANES { LV(idx) = LV(ctx1) & 15u; }
            WAVE_GATHER(t0, ctot, idx)
            WAVE_GATHER(zz0, czer, idx)
            FOR_LANES
            {
                if (LV(valid1) && LV(ctx1) != 31u) { LV(t1) = LV(t0) + LV(rk1); LV(z1) = LV(zz0) + LV(zb1); }
                LV(idx) = LV(ctx2);
            }
            if (BALLOT(LV(valid2) != 0u)) {                       // (no sign events in most chunks of the high planes)
                WAVE_GATHER(t0, ctot, idx)
                WAVE_GATHER(zz0, czer, idx)
            }
            LANEVAR(uint32_t, cross);
            FOR_LANES
            {
                if (LV(valid2)) { LV(t2) = LV(t0) + LV(rk2); LV(z2) = LV(zz0) + LV(zb2); }
                LV(cross) = 0;
                if (lane < 17) {
                    if (LV(ctot) + LV(cn) < kRescaleCap) { LV(ctot) += LV(cn); LV(czer) += LV(cnz); }
                    else LV(cross) = 1;
                }
            }
            for (uint64_t rem = BALLOT(LV(cross) != 0u); rem; rem &= rem - 1ull) {     // rare: contexts that rescale in this chunk
                const uint32_t c = (uint32_t)ffs64(rem);
                if (c < 12u) ICER_CTX_STEP(c, LV(valid1) && LV(ctx1) == c, LV(bit1) == 0u, z1, t1)
                else ICER_CTX_STEP(c, LV(valid2) && LV(ctx2) == c, LV(bit2) == 0u, z2, t2)
            }
        }
        ICER_TICK(3)

        // ---- fold + bin (E1), hand the chunk over ------------------------------------------------------
        LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);
        FOR_LANES
        {
            uint32_t e1 = 0, e2 = 0;
            if (LV(valid1)) {
                uint32_t z = LV(z1), t = LV(t1), b = LV(bit1);
                if (z < (t >> 1)) { z = t - z; b ^= 1u; }
                e1 = 0x80u | (b << 5) | pick_bin(s.tab.binlut, z, t);
            }
            if (LV(valid2)) {
                uint32_t z = LV(z2), t = LV(t2), b = LV(bit2);
                if (z < (t >> 1)) { z = t - z; b ^= 1u; }
                e2 = 0x80u | (b << 5) | pick_bin(s.tab.binlut, z, t);
            }
            LV(ev1) = e1;
            LV(ev2) = e2;
        }
        ICER_TICK(4)
        // queue slot j % D is free once the assembly wave has retired chunk j - D
        ICER_WAIT_CNT(s.b_done, bd_, j < bd_ + kQueueDepth, ab2_, ICER_BP_SLEEP)
        if (ab2_) break;
        ICER_TICK(5)
        EventSlot &q = s.eq[j % kQueueDepth];
        FOR_LANES
        {
            q.ev1[lane] = (uint8_t)LV(ev1);
            q.ev2[lane] = (uint8_t)LV(ev2);
            if (lane == 0) q.blank = in.cn[17];
            if (lane < kNumBins) q.cnt[lane] = LV(czer) | (LV(ctot) << 16);
        }
        ICER_PUBLISH(s.a_done, j + 1u)
    }
    FOR_LANES { LV(cs.czer) = LV(czer); LV(cs.ctot) = LV(ctot); }
    ICER_TIMERS_STORE(a.timers)
}

// ==========================================================================================
// the counts-only prefix pass of a sub-range workgroup ("Sub-ranges" above)
// ==========================================================================================
// The adaptive counts at the workgroup's first chunk j0 depend on the events of [0, j0) alone: per chunk and context the number of
// events and of zeros among them (a sum), and, where a context reaches the rescale point inside a chunk (total 500 -> 250, quirk C5),
// the zeros among its events up to that one.  All waves but one make the per-chunk totals (pixel_prefix_run, chunks dealt round-robin);
// the count wave adds them up in order (count_prefix_run) and looks at a chunk's events only when a context rescales in it.  The waves
// meet through a ring of kPrefixRing chunks -- totals and event bytes, 100 B per chunk, laid over the (idle) event queue -- and the
// count wave takes whatever the ring holds per look at the hand-off counters: its cost per chunk is one LDS read and a dozen
// instructions, not a poll and a publish (round 6; before, 0.3 us per chunk: a fifth of a lone frame's time went into this pass).
constexpr uint32_t kPrefixRing = 32;
struct PrefixRing {
    uint16_t cn[kPrefixRing][20];       // per chunk and context: events | zeros among them << 8
    uint8_t ev[kPrefixRing][64];        // the chunk's event bytes (events.hpp)
};
static_assert(sizeof(PrefixRing) <= sizeof(EventSlot) * kQueueDepth, "the prefix ring lies over the event queue");
ICER_DEV PrefixRing &prefix_ring(CoderShared &s) { return *reinterpret_cast<PrefixRing *>(&s.eq[0]); }

// pixel wave k of npw in the prefix pass: the per-context totals of its chunks (j % npw == k) among [j0, j1)
ICER_DEV void pixel_prefix_run(CoderShared &s, const UnitArgs &a, PixelWave &cw, uint32_t j0, uint32_t j1, uint32_t k, uint32_t npw)
{
    DECL_LANE;
    PrefixRing &ring = prefix_ring(s);
    const uint32_t nchunks = (a.w * a.h + 63u) / 64u;
    const uint32_t lsb = (uint32_t)a.lsb;
    const uint32_t jfirst = j0 + (k + npw - j0 % npw) % npw;
    if (cw.fetched != jfirst && jfirst < nchunks) {
        ICER_BLANK_MASK(jfirst)
        if (!ICER_IS_BLANK(jfirst)) { FOR_LANES { LV(cw.nb) = a.ev[(size_t)jfirst * 64u + (uint32_t)lane]; } }
        cw.fetched = jfirst;
    }
    for (uint32_t j = jfirst; j < j1; j += npw) {
        ICER_BLANK_MASK(j)
        const bool blank = ICER_IS_BLANK(j);
        LANEVAR(uint32_t, eb); LANEVAR(uint32_t, cnw);
        FOR_LANES { LV(eb) = blank ? 0u : LV(cw.nb); }
        if (j + npw < nchunks) {
            ICER_BLANK_MASK(j + npw)
            if (!ICER_IS_BLANK(j + npw)) { FOR_LANES { LV(cw.nb) = a.ev[(size_t)(j + npw) * 64u + (uint32_t)lane]; } }
        }
        cw.fetched = j + npw;
        if (blank) {
            FOR_LANES { LV(cnw) = lane == 0 ? (64u | (64u << 8)) : 0u; }
        } else {
            LANEVAR(uint32_t, valid1); LANEVAR(uint32_t, ctx1); LANEVAR(uint32_t, bit1);
            LANEVAR(uint32_t, valid2); LANEVAR(uint32_t, ctx2); LANEVAR(uint32_t, bit2);
            FOR_LANES { ICER_UNPACK_EVENT_BYTE(LV(eb), LV(valid1), LV(ctx1), LV(bit1), LV(valid2), LV(ctx2), LV(bit2)) }
#define ICER_MATCH(KEY, V, B0, B1, B2, B3) \
            ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
            const uint64_t V = BALLOT(LV(valid1) && LV(ctx1) != 31u);
            const uint64_t B0 = BALLOT(LV(ctx1) & 1u), B1 = BALLOT(LV(ctx1) & 2u), B2 = BALLOT(LV(ctx1) & 4u), B3 = BALLOT(LV(ctx1) & 8u);
            const uint64_t ZM = BALLOT(LV(bit1) == 0u);
            const uint64_t U = BALLOT(LV(valid2) != 0u);
            const uint64_t C0 = BALLOT((LV(ctx2) - 12u) & 1u), C1 = BALLOT((LV(ctx2) - 12u) & 2u), C2 = BALLOT((LV(ctx2) - 12u) & 4u);
            const uint64_t ZN = BALLOT(LV(bit2) == 0u);
            FOR_LANES
            {
                // (magnitude contexts on lanes 0..11, sign contexts on lanes 12..16: one match on selected ballots)
                const bool sg = lane >= 12;
                const uint32_t key = sg ? (uint32_t)lane - 12u : (uint32_t)lane;
                const uint64_t vv = sg ? U : V, b0 = sg ? C0 : B0, b1 = sg ? C1 : B1, b2 = sg ? C2 : B2, b3 = sg ? 0ull : B3;
                const uint64_t m = lane <= 16 ? ICER_MATCH(key, vv, b0, b1, b2, b3) : 0ull;
                LV(cnw) = (uint32_t)popc64(m) | ((uint32_t)popc64(m & (sg ? ZN : ZM)) << 8);
            }
#undef ICER_MATCH
        }
        // ring slot j % R is free once the count wave has consumed chunk j - R
        ICER_WAIT_CNT(s.a_done, ad_, j < ad_ + kPrefixRing, ab_, 2)
        if (ab_) break;
        FOR_LANES
        {
            if (lane < 17) ring.cn[j % kPrefixRing][lane] = (uint16_t)LV(cnw);
            ring.ev[j % kPrefixRing][lane] = (uint8_t)LV(eb);
        }
        ICER_PUBLISH(s.p_done[k], j + 1u)
    }
}

// chunks [0, result) have been handed over by the npw pixel waves of the prefix pass (wave k: chunks k, k + npw, ...; p_done[k] =
// 1 + its last one, 0 before its first)
ICER_DEV uint32_t prefix_frontier(CoderShared &s, uint32_t npw)
{
    DECL_LANE;
    LANEVAR(uint32_t, nxt);
    FOR_LANES
    {
        const uint32_t pd = (uint32_t)lane < npw ? ICER_LOAD_CNT(s.p_done[lane]) : 0u;
        LV(nxt) = (uint32_t)lane < npw ? ~(pd ? pd - 1u + npw : (uint32_t)lane) : 0u;       // (complement: the minimum through WAVE_MAX)
    }
    uint32_t m;
    WAVE_MAX(m, nxt)
    return ~m;
}

// the count wave in the prefix pass: the adaptive counts advanced over [j0, j1) and nothing else -- no bins, no events for the other waves
ICER_DEV void count_prefix_run(CoderShared &s, const UnitArgs &a, CountWave &cs, uint32_t j0, uint32_t j1, uint32_t npw)
{
    DECL_LANE;
    (void)a;
    PrefixRing &ring = prefix_ring(s);
    LANEVAR(uint32_t, czer); LANEVAR(uint32_t, ctot);
    FOR_LANES
    {
        LV(czer) = j0 == 0 ? 2u : LV(cs.czer);                    // icer_context_modeller.c:607-613
        LV(ctot) = j0 == 0 ? 4u : LV(cs.ctot);
    }
    uint32_t avail = j0;                                          // chunks below this one are in the ring
    for (uint32_t j = j0; j < j1; j++) {
        if (j >= avail) {
            avail = prefix_frontier(s, npw);
            if (avail <= j) {
                // (what has been consumed is given back BEFORE waiting: the pixel wave that owes chunk j may be waiting for its slot)
                ICER_PUBLISH(s.a_done, j)
                ICER_WAIT_UNTIL((avail = prefix_frontier(s, npw)) > j || ICER_LOAD_CNT(s.abort))
            }
            if (ICER_LOAD_CNT(s.abort)) break;
            ICER_ACQUIRE()
        }
        // the per-context totals of the chunk; its events are looked at only when a context reaches the rescale point
        LANEVAR(uint32_t, cross);
        FOR_LANES
        {
            const uint32_t w = lane < 17 ? (uint32_t)ring.cn[j % kPrefixRing][lane] : 0u;
            LV(cross) = 0;
            if (lane < 17) {
                if (LV(ctot) + (w & 255u) < kRescaleCap) { LV(ctot) += w & 255u; LV(czer) += w >> 8; }
                else LV(cross) = 1;
            }
        }
        uint64_t rem = BALLOT(LV(cross) != 0u);
        if (rem) {
            LANEVAR(uint32_t, v1); LANEVAR(uint32_t, c1); LANEVAR(uint32_t, b1); LANEVAR(uint32_t, v2); LANEVAR(uint32_t, c2); LANEVAR(uint32_t, b2);
            LANEVAR(uint32_t, zo); LANEVAR(uint32_t, to);
            FOR_LANES
            {
                ICER_UNPACK_EVENT_BYTE((uint32_t)ring.ev[j % kPrefixRing][lane], LV(v1), LV(c1), LV(b1), LV(v2), LV(c2), LV(b2))
                LV(zo) = 0; LV(to) = 0;
            }
            (void)zo; (void)to;             // (ICER_CTX_STEP also leaves what every event sees: not needed here)
            for (; rem; rem &= rem - 1ull) {
                const uint32_t c = (uint32_t)ffs64(rem);
                if (c < 12u) ICER_CTX_STEP(c, LV(v1) && LV(c1) == c, LV(b1) == 0u, zo, to)
                else ICER_CTX_STEP(c, LV(v2) && LV(c2) == c, LV(b2) == 0u, zo, to)
            }
        }
        if ((j & 7u) == 7u) ICER_PUBLISH(s.a_done, j + 1u)
    }
    ICER_PUBLISH(s.a_done, j1)
    FOR_LANES { LV(cs.czer) = LV(czer); LV(cs.ctot) = LV(ctot); }
}

// ==========================================================================================
// compaction wave (bins 1..7): the chunk's events grouped per bin, for the walker and records waves
// ==========================================================================================
ICER_DEV void compact_wave_run(CoderShared &s, const UnitArgs &a, uint32_t j0, uint32_t j1)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    for (uint32_t j = j0; j < j1; j++) {
#if !defined(ICER_WAVE_EMU) || defined(ICER_WAVE_THREADS)
        // progressive mode: has the byte quota been used up by units of higher priority in the meantime?  (Checked by
        // this wave because it has the slack and no other global memory traffic: in the pixel wave the check's loads
        // made the compiler wait for the window prefetch at once.)
        if ((j & 127u) == 127u && quota_already_spent(a)) { ICER_PUBLISH(s.abort, 3u) break; }
#endif
        ICER_WAIT_CNT(s.a_done, ad_, ad_ > j, ab_, 1)
        if (ab_) break;
        ICER_ACQUIRE()
        ICER_TICK(25)
        EventSlot &q = s.eq[j % kQueueDepth];
        LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);
        FOR_LANES
        {
            LV(ev1) = q.ev1[lane];
            LV(ev2) = q.ev2[lane];
        }
        const uint64_t V1 = BALLOT((LV(ev1) & 0x98u) == 0x80u && (LV(ev1) & 7u)), V2 = BALLOT((LV(ev2) & 0x98u) == 0x80u && (LV(ev2) & 7u));
        if ((V1 | V2) == 0ull) {
            // no event of bins 1..7 in this chunk (the rule in the sparse high planes): only the counts are read downstream
            FOR_LANES
            {
                if (lane < 8) q.binn[lane] = 0;
            }
            ICER_PUBLISH(s.c_done, j + 1u)
            ICER_TICK(26)
            continue;
        }
        FOR_LANES
        {
            if (lane < 48) (&q.binbits[0][0])[lane] = 0;
        }
        WAVE_SYNC();
        // every event of bins 1..7 puts its input bit at its rank into the bin's bit string and its position
        // into the bin's position list.  Rank = number of earlier events of the same bin.  The lanes of a bin are found from
        // per-bit ballots of the bin number (no loop over bins) -- once per BIN, by lane b for bin b (which also needs them
        // for the bin's event count); an event's lane then fetches the two masks of its bin (first / second position of a
        // lane) from that lane: four ds_bpermute instead of matching its own key against the ballots again (2 x 16 VALU
        // instructions per event position; the kernel is VALU-bound, its LDS pipe mostly idle).  Without an event of these bins
        // in a second position (V2 == 0: the sign events of most chunks are in bin 0) half of it drops out.
        {
#define ICER_MATCH3(KEY, V, B0, B1, B2) ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)))
            const uint64_t P0 = BALLOT(LV(ev1) & 1u), P1 = BALLOT(LV(ev1) & 2u), P2 = BALLOT(LV(ev1) & 4u);
            LANEVAR(uint32_t, oa_lo); LANEVAR(uint32_t, oa_hi); LANEVAR(uint32_t, ob_lo); LANEVAR(uint32_t, ob_hi);   // lane b: the events of bin b
            LANEVAR(uint32_t, i1); LANEVAR(uint32_t, i2);
            LANEVAR(uint32_t, a1l); LANEVAR(uint32_t, a1h); LANEVAR(uint32_t, b1l); LANEVAR(uint32_t, b1h);
            LANEVAR(uint32_t, a2l); LANEVAR(uint32_t, a2h); LANEVAR(uint32_t, b2l); LANEVAR(uint32_t, b2h);
            if (V2) {
                const uint64_t Q0 = BALLOT(LV(ev2) & 1u), Q1 = BALLOT(LV(ev2) & 2u), Q2 = BALLOT(LV(ev2) & 4u);
                FOR_LANES
                {
                    const uint64_t ma = ICER_MATCH3((uint32_t)lane & 7u, V1, P0, P1, P2), mb = ICER_MATCH3((uint32_t)lane & 7u, V2, Q0, Q1, Q2);
                    LV(oa_lo) = (uint32_t)ma; LV(oa_hi) = (uint32_t)(ma >> 32); LV(ob_lo) = (uint32_t)mb; LV(ob_hi) = (uint32_t)(mb >> 32);
                    if (lane < 8) q.binn[lane] = (uint8_t)(popc64(ma) + popc64(mb));
                    LV(i1) = LV(ev1) & 7u; LV(i2) = LV(ev2) & 7u;
                }
                WAVE_GATHER(a1l, oa_lo, i1) WAVE_GATHER(a1h, oa_hi, i1) WAVE_GATHER(b1l, ob_lo, i1) WAVE_GATHER(b1h, ob_hi, i1)
                WAVE_GATHER(a2l, oa_lo, i2) WAVE_GATHER(a2h, oa_hi, i2) WAVE_GATHER(b2l, ob_lo, i2) WAVE_GATHER(b2h, ob_hi, i2)
            } else {
                FOR_LANES
                {
                    const uint64_t ma = ICER_MATCH3((uint32_t)lane & 7u, V1, P0, P1, P2);
                    LV(oa_lo) = (uint32_t)ma; LV(oa_hi) = (uint32_t)(ma >> 32);
                    if (lane < 8) q.binn[lane] = (uint8_t)popc64(ma);
                    LV(i1) = LV(ev1) & 7u;
                    LV(b1l) = 0; LV(b1h) = 0; LV(a2l) = 0; LV(a2h) = 0; LV(b2l) = 0; LV(b2h) = 0;
                }
                WAVE_GATHER(a1l, oa_lo, i1) WAVE_GATHER(a1h, oa_hi, i1)
            }
            FOR_LANES
            {
                if ((V1 >> lane) & 1ull) {
                    const uint32_t b = LV(ev1) & 7u;
                    const uint64_t m1 = (uint64_t)LV(a1l) | ((uint64_t)LV(a1h) << 32), m2 = (uint64_t)LV(b1l) | ((uint64_t)LV(b1h) << 32);
                    const uint32_t r = (uint32_t)(mbcnt64(m1, lane) + mbcnt64(m2, lane));
                    q.rk1[lane] = (uint8_t)r;
                    q.binseq[b][r] = (uint8_t)(2u * (uint32_t)lane);
                    if (LV(ev1) & 0x20u) LDS_OR(q.binbits[b][(r + 8u) >> 5], 1u << ((r + 8u) & 31u));
                }
                if ((V2 >> lane) & 1ull) {
                    const uint32_t b = LV(ev2) & 7u;
                    const uint64_t m1 = (uint64_t)LV(a2l) | ((uint64_t)LV(a2h) << 32), m2 = (uint64_t)LV(b2l) | ((uint64_t)LV(b2h) << 32);   // this lane's own magnitude event comes first
                    const uint32_t r = (uint32_t)(mbcnt64(m1, lane) + (int)((m1 >> lane) & 1ull) + mbcnt64(m2, lane));
                    q.rk2[lane] = (uint8_t)r;
                    q.binseq[b][r] = (uint8_t)(2u * (uint32_t)lane + 1u);
                    if (LV(ev2) & 0x20u) LDS_OR(q.binbits[b][(r + 8u) >> 5], 1u << ((r + 8u) & 31u));
                }
            }
#undef ICER_MATCH3
        }
        ICER_PUBLISH(s.c_done, j + 1u)
        ICER_TICK(26)
    }
    ICER_TIMERS_STORE(a.timers)
}

// ==========================================================================================
// walker wave (bins 1..7)
// ==========================================================================================
struct WalkWave {
    LANEVAR(uint32_t, node);    // lane b (1..7): code-tree node of bin b's partial input, acc | 1 << bits (1 = root)
    LANEVAR(uint32_t, cb);      // this lane's bin: lane b for b = 1..7, CoderTables::cand_bin for the candidate lanes, else 0
    LANEVAR(uint32_t, ce);      // candidate lanes: the node they assume
    uint32_t next, gen;         // next chunk to walk; generation (= number of exact-path chunks seen)
};

ICER_DEV void walk_wave_init(CoderShared &s, WalkWave &ww, uint32_t j0 = 0)
{
    DECL_LANE;
    FOR_LANES
    {
        LV(ww.node) = 1;
        LV(ww.cb) = (lane >= 1 && lane <= 7) ? (uint32_t)lane : (uint32_t)s.tab.cand_bin[lane];
        LV(ww.ce) = LV(ww.cb) ? (uint32_t)s.tab.node_c[LV(ww.cb) & 7u][s.tab.cand_node[lane]] & 7u : 0u;
    }
    ww.next = j0;
    ww.gen = 0;
}

// bits [start, start + 6) of a bit string stored with an offset of 8 (rank r lives at bit r + 8, so that
// reads a few ranks before rank 0 see zeros); `start` is a rank - 4 >= -4
ICER_DEV uint32_t window6(const uint32_t *words, int start)
{
    const uint32_t pos = (uint32_t)(start + 8);
    const uint64_t two = (uint64_t)words[pos >> 5] | ((uint64_t)words[(pos >> 5) + 1] << 32);
    return (uint32_t)(two >> (pos & 31u)) & 63u;
}

ICER_DEV uint32_t chunk_tag(uint32_t j, uint32_t gen) { return ((j << 8) | (gen & 255u)) + 1u; }

// Processes chunks speculatively (assuming the merge wave takes the fast path) for as long as events are
// available; rolls back to the chunk after the last exact-path chunk whenever the merge wave reports one.
// `max_chunks` bounds the work of one call (the emulation interleaves the waves; the GPU passes ~0u).
// Returns the number of chunks walked.
ICER_DEV uint32_t walk_wave_run(CoderShared &s, const UnitArgs &a, WalkWave &ww, uint32_t nchunks, uint32_t max_chunks)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    ICER_IDLE_DECL
    (void)a;
    uint32_t done = 0;
    for (;;) {
        // (control words read together: one LDS round trip)
        const uint32_t ab_ = ICER_LOAD_CNT(s.abort), seq = ICER_LOAD_CNT(s.exact_seq), cd_ = ICER_LOAD_CNT(s.c_done);
        if (ab_) break;
        if (seq != ww.gen) {
            // everything walked after chunk last_exact is void; the replay left the bins' partial inputs in LDS
            ICER_ACQUIRE()
            ww.gen = seq;
            ww.next = s.last_exact + 1u;
            FOR_LANES
            {
                if (lane >= 1 && lane <= 7) LV(ww.node) = st_acc(s.bin_state[lane]) | (1u << st_nin(s.bin_state[lane]));
            }
        }
        const uint32_t j = ww.next;
        if (j >= nchunks || cd_ <= j) {
            if (ICER_LOAD_CNT(s.b_done) >= nchunks || done >= max_chunks) break;
            ICER_IDLE()
            continue;
        }
        if (done >= max_chunks) break;
        ICER_ACQUIRE()
        ICER_TICK(6)
        const EventSlot &q = s.eq[j % kQueueDepth];
        WalkSlot &o = s.wq[j % kQueueDepth];
        RecSlot &ro = s.rq[j % kQueueDepth];
        if (BALLOT(lane >= 1 && lane <= 7 && q.binn[lane & 7] != 0) == 0ull) {
            // no event of bins 1..7 in this chunk: nothing to walk, the bins keep their state
            FOR_LANES
            {
                if (lane >= 1 && lane <= 7) {
                    const uint32_t node = LV(ww.node), nin = 31u - (uint32_t)clz32(node);
                    o.bincarry[lane] = (uint8_t)node;
                    o.post_nin[lane] = (uint8_t)nin;
                    ro.binst[lane] = st_pack(255u, node ^ (1u << nin), nin);
                }
            }
            ICER_TICK(8)
            ICER_PUBLISH(o.tag, chunk_tag(j, ww.gen))
            ww.next = j + 1u;
            done++;
            ICER_IDLE_RESET
            continue;
        }
        // A bin's walk is a chain of dependent table look-ups, six input bits each.  To halve the chain the bin's
        // n ranks are cut at h = ceil(n / 2): lane b (1..7) walks [0, h) from the node carried in, and one lane
        // per node of the bin's code tree (CoderTables::cand_*) walks [h, n) as if entered at that node; when the
        // first half is done its end node says which of them was right.
        LANEVAR(uint32_t, stl0); LANEVAR(uint32_t, stl1);       // start flags by rank, relative to the segment (a segment has <= 64 ranks)
        LANEVAR(uint32_t, wnode); LANEVAR(uint32_t, wh); LANEVAR(uint32_t, wn);
        FOR_LANES
        {
            const uint32_t b = LV(ww.cb);
            const bool first = lane >= 1 && lane <= 7;
            const uint32_t n = b ? (uint32_t)q.binn[b] : 0u;
            const uint32_t h = n >= 18u ? (n + 1u) >> 1 : n;                     // (short walks are not worth splitting)
            const uint32_t r0 = first ? 0u : h, len = first ? h : n - h;
            // compact node number (0 = root); the candidates' ones were looked up once (walk_wave_init)
            uint32_t node = first ? (uint32_t)s.tab.node_c[b & 7u][LV(ww.node)] : LV(ww.ce);
            uint64_t st_lo = 0;
            if (b && len) {
                uint64_t lo = ((uint64_t)q.binbits[b][0] | ((uint64_t)q.binbits[b][1] << 32)) >> 8;      // ranks 0..55
                uint64_t hi = (uint64_t)q.binbits[b][2] | ((uint64_t)q.binbits[b][3] << 32);             // ranks 56..119
                const uint32_t top8 = q.binbits[b][4];                                                     // ranks 120..127
                lo |= hi << 56;
                hi = (hi >> 8) | ((uint64_t)top8 << 56);                                                   // ranks 64..127
                if (r0 == 64u) { lo = hi; hi = 0; }
                else if (r0) { lo = (lo >> r0) | (hi << (64u - r0)); hi >>= r0; }                            // the segment starts at bit 0
                uint32_t r = 0;
                for (; r + 6u <= len; r += 6) {
                    const uint32_t e = s.tab.v2v_step6[b][node][(uint32_t)(lo >> r) & 63u];
                    st_lo |= (uint64_t)(e >> 4) << r;
                    node = e & 7u;
                }
                if (r < len) {                                  // last 1..5 bits: flags from the zero-padded step, node from the tail table
                    const uint32_t k = len - r, bits = (uint32_t)(lo >> r) & ((1u << k) - 1u);
                    const uint32_t e = s.tab.v2v_step6[b][node][bits];
                    st_lo |= (uint64_t)((e >> 4) & ((1u << k) - 1u)) << r;
                    node = s.tab.v2v_tail[b][node][(1u << k) | bits];
                }
            }
            LV(stl0) = (uint32_t)st_lo; LV(stl1) = (uint32_t)(st_lo >> 32);
            LV(wnode) = b ? (uint32_t)s.tab.node_full[b & 7u][node & 7u] : 1u;   // back to the tree's own numbering
            LV(wh) = h; LV(wn) = n;
        }
        ICER_TICK(7)
        // lane b fetches the second half from the lane that started at the node the first half ended in
        LANEVAR(uint32_t, src); LANEVAR(uint32_t, g0); LANEVAR(uint32_t, g1); LANEVAR(uint32_t, gnode);
        FOR_LANES
        {
            LV(src) = (uint32_t)lane;
            if (lane >= 1 && lane <= 7 && LV(wh) < LV(wn)) LV(src) = s.tab.cand_lane[lane][LV(wnode)];
        }
        WAVE_GATHER(g0, stl0, src)
        WAVE_GATHER(g1, stl1, src)
        WAVE_GATHER(gnode, wnode, src)
        FOR_LANES
        {
            if (lane >= 1 && lane <= 7) {
                const int b = lane;
                const uint32_t n = LV(wn), h = LV(wh);
                uint64_t st_lo = (uint64_t)LV(stl0) | ((uint64_t)LV(stl1) << 32), st_hi = 0;
                uint32_t node = LV(wnode);
                if (h < n) {                                    // 8 <= h <= 64: the second half's flags move up by h ranks
                    const uint64_t s2 = (uint64_t)LV(g0) | ((uint64_t)LV(g1) << 32);
                    if (h == 64u) st_hi = s2;
                    else { st_lo |= s2 << h; st_hi = s2 >> (64u - h); }
                    node = LV(gnode);
                }
                o.bincarry[b] = (uint8_t)LV(ww.node);
                LV(ww.node) = node;
                // post: start flags with the same offset of 8 as the bit string
                o.binstart[b][0] = (uint32_t)(st_lo << 8);
                o.binstart[b][1] = (uint32_t)(st_lo >> 24);
                o.binstart[b][2] = (uint32_t)(st_lo >> 56) | (uint32_t)(st_hi << 8);
                o.binstart[b][3] = (uint32_t)(st_hi >> 24);
                o.binstart[b][4] = (uint32_t)(st_hi >> 56);
                o.binstart[b][5] = 0;
                const uint32_t nin = 31u - (uint32_t)clz32(node);
                o.post_nin[b] = (uint8_t)nin;
                uint32_t op = 255;
                if (n) {
                    // open word after the chunk: the last start, unless everything after it completed
                    int last = st_hi ? 64 + 63 - clz64(st_hi) : (st_lo ? 63 - clz64(st_lo) : -1);
                    op = node == 1u ? 254u : (last >= 0 ? (uint32_t)q.binseq[b][last] : 255u);
                }
                ro.binst[b] = st_pack(op, node ^ (1u << nin), nin);
            }
        }
        ICER_TICK(8)
        ICER_PUBLISH(o.tag, chunk_tag(j, ww.gen))
        ww.next = j + 1u;
        done++;
        ICER_IDLE_RESET
    }
    ICER_TIMERS_STORE(a.timers)
    return done;
}

// ==========================================================================================
// golomb wave (bins 0 and 8..16)
// ==========================================================================================
struct GolombWave {             // golomb state wave and golomb workers (the state wave keeps the bins' zero-run lengths in CoderShared::gk)
    uint32_t next, gen;         // as WalkWave
};

ICER_DEV void golomb_wave_init(GolombWave &gw, uint32_t j0 = 0)
{
    gw.next = j0;
    gw.gen = 0;
}

// The zero-run length of every Golomb bin at the start of every chunk: lane b = bin b.  After a chunk it is
// (run before + the bin's events) mod m if the bin had no one-event in the chunk, else (its zero events after its last
// one-event) mod m -- a word ends at a one-event or when the run reaches m (icer_encoding.c:62-80).
// Same speculation / roll-back scheme as walk_wave_run.
ICER_DEV uint32_t golomb_state_run(CoderShared &s, const UnitArgs &a, GolombWave &gw, uint32_t nchunks, uint32_t max_chunks)
{
    DECL_LANE;
    ICER_IDLE_DECL
    (void)a;
    uint32_t done = 0;
    for (;;) {
        const uint32_t ab_ = ICER_LOAD_CNT(s.abort), seq = ICER_LOAD_CNT(s.exact_seq), ad_ = ICER_LOAD_CNT(s.a_done);
        if (ab_) break;
        if (seq != gw.gen) {
            ICER_ACQUIRE()
            gw.gen = seq;
            gw.next = s.last_exact + 1u;
            FOR_LANES
            {
                if (lane >= 8 && lane <= 16) s.gk[lane] = st_acc(s.bin_state[lane]);
            }
        }
        const uint32_t j = gw.next;
        if (j >= nchunks || ad_ <= j) {
            if (ICER_LOAD_CNT(s.b_done) >= nchunks || done >= max_chunks) break;
            ICER_IDLE()
            continue;
        }
        if (done >= max_chunks) break;
        ICER_ACQUIRE()
        // (slot j % depth is free: the count wave is at most `depth` chunks ahead of the merge wave's retire count, and a
        // chunk is only retired after its worker has handed it over)
        const EventSlot &q = s.eq[j % kQueueDepth];
        RunSlot &o = s.kq[j % kQueueDepth];
        LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2); LANEVAR(uint32_t, run);
        FOR_LANES
        {
            LV(ev1) = q.ev1[lane];
            LV(ev2) = q.ev2[lane];
            LV(run) = lane < kNumBins ? s.gk[lane] : 0u;
            if (lane < kNumBins) o.gk[lane] = LV(run);
        }
        const uint64_t G1 = BALLOT((LV(ev1) & 0x98u) >= 0x88u), G2 = BALLOT((LV(ev2) & 0x98u) >= 0x88u);
        if (G1 | G2) {
#define ICER_MATCH(KEY, V, B0, B1, B2, B3) \
            ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
            const uint64_t K0 = BALLOT(LV(ev1) & 1u), K1 = BALLOT(LV(ev1) & 2u), K2 = BALLOT(LV(ev1) & 4u), K3 = BALLOT((LV(ev1) & 31u) == 16u);
            const uint64_t J0 = BALLOT(LV(ev2) & 1u), J1 = BALLOT(LV(ev2) & 2u), J2 = BALLOT(LV(ev2) & 4u), J3 = BALLOT((LV(ev2) & 31u) == 16u);
            const uint64_t O1 = G1 & BALLOT(LV(ev1) & 0x20u), O2 = G2 & BALLOT(LV(ev2) & 0x20u);       // one-events
            FOR_LANES
            {
                if (lane >= 8 && lane <= 16) {
                    const uint32_t key = ((uint32_t)lane & 7u) | (lane == 16 ? 8u : 0u);
                    const uint64_t m1 = ICER_MATCH(key, G1, K0, K1, K2, K3), m2 = ICER_MATCH(key, G2, J0, J1, J2, J3);
                    const uint32_t all = (uint32_t)(popc64(m1) + popc64(m2));
                    if (all) {
                        const int lo1 = last_le(m1 & O1, m2 & O2, 127u);
                        uint32_t z = lo1 < 0 ? LV(run) + all : all - cnt_lt(m1, m2, (uint32_t)lo1 + 1u);
                        z -= ((z * s.tab.ginv[lane]) >> 20) * s.tab.gm[lane];
                        s.gk[lane] = z;
                    }
                }
            }
#undef ICER_MATCH
        }
        ICER_PUBLISH(o.tag, chunk_tag(j, gw.gen))
        gw.next = j + 1u;
        done++;
        ICER_IDLE_RESET
    }
    return done;
}

// Golomb worker k of ngw: the chunks j with j % ngw == k, each from the run lengths the state wave
// left for it (RunSlot); keeps no state of its own.  Same speculation / roll-back scheme as walk_wave_run.
ICER_DEV uint32_t golomb_wave_run(CoderShared &s, const UnitArgs &a, GolombWave &gw, uint32_t nchunks, uint32_t max_chunks, uint32_t k, uint32_t ngw)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    ICER_IDLE_DECL
    (void)a;
    uint32_t done = 0;
    // ngw == 0: no state wave -- this wave is the whole Golomb stage (the small shape of the workgroup): it takes the run
    // lengths from CoderShared::gk and leaves the new ones there itself
    const bool fused = ngw == 0u;
    const uint32_t step = fused ? 1u : ngw;
    if (!fused && gw.next % ngw != k) gw.next += (k + ngw - gw.next % ngw) % ngw;     // (first call: this worker's first chunk)
    for (;;) {
        const uint32_t ab_ = ICER_LOAD_CNT(s.abort), seq = ICER_LOAD_CNT(s.exact_seq), ad_ = ICER_LOAD_CNT(s.a_done);
        if (ab_) break;
        if (seq != gw.gen) {
            ICER_ACQUIRE()
            gw.gen = seq;
            const uint32_t first = s.last_exact + 1u;
            gw.next = fused ? first : first + (k + ngw - first % ngw) % ngw;
            if (fused) {
                FOR_LANES
                {
                    if (lane >= 8 && lane <= 16) s.gk[lane] = st_acc(s.bin_state[lane]);
                }
            }
        }
        const uint32_t j = gw.next;
        if (j >= nchunks || (fused ? ad_ <= j : ICER_LOAD_CNT(s.kq[j % kQueueDepth].tag) != chunk_tag(j, gw.gen))) {
            if (ICER_LOAD_CNT(s.b_done) >= nchunks || done >= max_chunks) break;
            ICER_IDLE()
            continue;
        }
        if (done >= max_chunks) break;
        ICER_ACQUIRE()
        ICER_TICK(10)
        struct { const uint32_t *gk; } runs_in = {fused ? s.gk : s.kq[j % kQueueDepth].gk};
        const EventSlot &q = s.eq[j % kQueueDepth];
        RecSlot &o = s.rq[j % kQueueDepth];
        LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);
        LANEVAR(uint32_t, fl1); LANEVAR(uint32_t, fl2);     // bit0: a word starts at this event, bit1: a word ends here
        LANEVAR(uint32_t, wd1); LANEVAR(uint32_t, wd2);     // finished ring word of an end event
        LANEVAR(uint32_t, sp1); LANEVAR(uint32_t, sp2);     // start position of the word an end event closes (255: carried in)
        FOR_LANES
        {
            LV(ev1) = q.ev1[lane];
            LV(ev2) = q.ev2[lane];
            LV(fl1) = 0; LV(fl2) = 0; LV(wd1) = 0; LV(wd2) = 0; LV(sp1) = 255; LV(sp2) = 255;
            // a bin without events in this chunk keeps its run length and its open word (open_pos 255)
            if (lane == 0 || (lane >= 8 && lane <= 16)) o.binst[lane] = st_pack(255u, runs_in.gk[lane], 0u);
            // bin 0 (uncoded): every event is a complete one-bit word (E3)
            if ((LV(ev1) & 0x9Fu) == 0x80u) { LV(fl1) = 3; LV(wd1) = kWordDone | (1u << 11) | ((LV(ev1) >> 5) & 1u); LV(sp1) = 2u * (uint32_t)lane; }
            if ((LV(ev2) & 0x9Fu) == 0x80u) { LV(fl2) = 3; LV(wd2) = kWordDone | (1u << 11) | ((LV(ev2) >> 5) & 1u); LV(sp2) = 2u * (uint32_t)lane + 1u; }
        }
        WAVE_SYNC();
        // Golomb bins 8..16, no loop over bins: the lanes of one bin are found from per-bit ballots of (bin - 8); an
        // event's run length = zeros of its bin since the bin's previous one-event (or since the chunk start, plus the
        // run carried in), modulo m; a word starts where that is 0 and ends at a one or when the run reaches m - 1.
        // The lane holding a bin's last event of the chunk also leaves the bin's state (run length, open word).
#define ICER_MATCH(KEY, V, B0, B1, B2, B3) \
        ((V) & (((KEY)&1u) ? (B0) : ~(B0)) & (((KEY)&2u) ? (B1) : ~(B1)) & (((KEY)&4u) ? (B2) : ~(B2)) & (((KEY)&8u) ? (B3) : ~(B3)))
        LANEVAR(uint32_t, ka1); LANEVAR(uint32_t, ka2);     // run length after the event if it is the bin's last one, else ~0
        // 64 zero events of Golomb bins in at most two runs of lanes (first bin b0, then bin b1) and nothing else -- what
        // a blank chunk turns into once its context's estimate has settled; the bin changes where the estimate crosses
        // a cut-off or is rescaled.  The runs simply continue: the r-th event of a run sees run length (k + r) mod m.
        // (only a blank chunk can qualify -- the count wave hands the pixel wave's flag on -- so the test costs the
        // other chunks one LDS read)
        uint32_t c = 64, e1 = 0;                                            // first lane and event of the second run
        bool runs = q.blank != 0u;
        if (runs) {
            const uint32_t e0 = READLANE(ev1, 0);
            const uint64_t D = BALLOT(LV(ev1) != e0);
            c = D ? (uint32_t)ffs64(D) : 64u;
            e1 = READLANE(ev1, c & 63u);
            runs = BALLOT((LV(ev1) & 0xB8u) < 0x88u || (LV(ev1) & 0x20u) != 0u || LV(ev2) != 0u || ((uint32_t)lane >= c && LV(ev1) != e1)) == 0ull;
        }
        if (runs) {
            WAVE_SYNC();
            FOR_LANES
            {
                const uint32_t b = LV(ev1) & 31u, m = s.tab.gm[b], inv = s.tab.ginv[b];
                const uint32_t r = (uint32_t)lane >= c ? (uint32_t)lane - c : (uint32_t)lane;      // rank inside the run
                const uint32_t z = runs_in.gk[b] + r;
                const uint32_t kb = z - ((z * inv) >> 20) * m;
                const uint32_t ends = kb + 1u == m ? 1u : 0u;
                const uint32_t first = kb <= r ? 2u * ((uint32_t)lane - kb) : 255u;               // first event of the word this event is in
                LV(fl1) = (kb == 0u ? 1u : 0u) | (ends << 1);
                LV(wd1) = kWordDone | (1u << 11) | 1u;
                if (ends) LV(sp1) = first;
                LV(ka1) = ((uint32_t)lane == 63u || (uint32_t)lane + 1u == c) ? (ends ? 0u : kb + 1u) : ~0u;   // last event of its bin
                LV(ka2) = first;
            }
            WAVE_SYNC();
            FOR_LANES
            {
                if (LV(ka1) != ~0u) {
                    const uint32_t b = LV(ev1) & 31u;
                    if (fused) s.gk[b] = LV(ka1);
                    o.binst[b] = st_pack(LV(ka1) ? LV(ka2) : 254u, LV(ka1), 0u);
                }
            }
        } else {
            const uint64_t G1 = BALLOT((LV(ev1) & 0x98u) >= 0x88u), G2 = BALLOT((LV(ev2) & 0x98u) >= 0x88u);
            // key of a bin: bin - 8 = 0..8, i.e. the low three bits of the bin number and "bin == 16".
            // TWO = 0: no sign event of the chunk is in a Golomb bin (G2 == 0) -- the rule: a sign is close to a coin toss, its
            // bins are the low ones -- and everything that concerns the second event of a lane drops out at compile time:
            // about a third of this wave's instructions per dense chunk.  TWO = 1: the general form.
#define ICER_GOLOMB_LANE(TWO, EV, SLOT, KA, FL, WD, MA, MB)                                                    \
                if (((EV) & 0x98u) >= 0x88u) {                                                                \
                    const uint32_t b_ = (EV) & 31u, key_ = (b_ & 7u) | (b_ == 16u ? 8u : 0u);                  \
                    const uint64_t m1_ = ICER_MATCH(key_, G1, K0, K1, K2, K3), m2_ = (TWO) ? ICER_MATCH(key_, G2, J0, J1, J2, J3) : 0ull; \
                    MA = m1_; MB = m2_;                                                                         \
                    const uint64_t Z1 = m1_ & ~O1, Z2 = m2_ & ~O2;                                              \
                    const uint32_t zb_ = (TWO) ? cnt_lt_own(Z1, Z2, lane, (SLOT)) : cnt_lt_own1(Z1, lane, (SLOT)); \
                    const int lo_ = (TWO) ? last_lt_own(m1_ & O1, m2_ & O2, lane, (SLOT)) : last_lt_own1(m1_ & O1, lane, (SLOT)); \
                    const uint32_t z_ = lo_ >= 0 ? zb_ - ((TWO) ? cnt_lt(Z1, Z2, (uint32_t)lo_) : cnt_lt1(Z1, (uint32_t)lo_)) : runs_in.gk[b_] + zb_; \
                    const uint32_t m = s.tab.gm[b_], inv = s.tab.ginv[b_];                                      \
                    const uint32_t kb_ = z_ - ((z_ * inv) >> 20) * m;                                           \
                    const uint32_t bit_ = ((EV) >> 5) & 1u;                                                     \
                    FL = (kb_ == 0u ? 1u : 0u) | ((bit_ || kb_ + 1u == m) ? 2u : 0u);                           \
                    WD = bit_ ? golomb_word(s.tab, (int)b_, kb_) : (kWordDone | (1u << 11) | 1u);               \
                    const uint32_t before_ = (TWO) ? cnt_lt_own(m1_, m2_, lane, (SLOT)) : cnt_lt_own1(m1_, lane, (SLOT)); \
                    if (before_ + 1u == (uint32_t)(popc64(m1_) + ((TWO) ? popc64(m2_) : 0))) KA = (FL & 2u) ? 0u : kb_ + 1u;  \
                }
#define ICER_GOLOMB_BINS(TWO)                                                                                  \
            {                                                                                                  \
                const uint64_t K0 = BALLOT(LV(ev1) & 1u), K1 = BALLOT(LV(ev1) & 2u), K2 = BALLOT(LV(ev1) & 4u), K3 = BALLOT((LV(ev1) & 31u) == 16u); \
                const uint64_t J0 = (TWO) ? BALLOT(LV(ev2) & 1u) : 0ull, J1 = (TWO) ? BALLOT(LV(ev2) & 2u) : 0ull;     \
                const uint64_t J2 = (TWO) ? BALLOT(LV(ev2) & 4u) : 0ull, J3 = (TWO) ? BALLOT((LV(ev2) & 31u) == 16u) : 0ull; \
                const uint64_t O1 = G1 & BALLOT(LV(ev1) & 0x20u), O2 = (TWO) ? G2 & BALLOT(LV(ev2) & 0x20u) : 0ull;   /* one-events */ \
                (void)J0; (void)J1; (void)J2; (void)J3; (void)O2;                                              \
                LANEVAR(uint64_t, ma1); LANEVAR(uint64_t, mb1); LANEVAR(uint64_t, ma2); LANEVAR(uint64_t, mb2);   /* events of this lane's bins */ \
                FOR_LANES                                                                                      \
                {                                                                                              \
                    LV(ma1) = 0; LV(mb1) = 0; LV(ma2) = 0; LV(mb2) = 0; LV(ka1) = ~0u; LV(ka2) = ~0u;          \
                    ICER_GOLOMB_LANE(TWO, LV(ev1), 0u, LV(ka1), LV(fl1), LV(wd1), LV(ma1), LV(mb1))            \
                    if (TWO) { ICER_GOLOMB_LANE(TWO, LV(ev2), 1u, LV(ka2), LV(fl2), LV(wd2), LV(ma2), LV(mb2)) } \
                }                                                                                              \
                /* word starts of the Golomb bins; an end event's word began at its bin's latest start, and so did the */ \
                /* word a bin's last event leaves open */                                                       \
                WAVE_SYNC();                                                                                   \
                const uint64_t SB1 = G1 & BALLOT(LV(fl1) & 1u), SB2 = (TWO) ? G2 & BALLOT(LV(fl2) & 1u) : 0ull; \
                (void)SB2;                                                                                     \
                FOR_LANES                                                                                      \
                {                                                                                              \
                    if ((LV(ev1) & 0x98u) >= 0x88u && ((LV(fl1) & 2u) || LV(ka1) != ~0u)) {                     \
                        const int sp = (TWO) ? last_le_own(SB1 & LV(ma1), SB2 & LV(mb1), lane, 0u) : last_le_own1(SB1 & LV(ma1), lane); \
                        if (LV(fl1) & 2u) LV(sp1) = sp < 0 ? 255u : (uint32_t)sp;                               \
                        if (LV(ka1) != ~0u) {                                                                  \
                            const uint32_t b = LV(ev1) & 31u;                                                  \
                            if (fused) s.gk[b] = LV(ka1);                                                      \
                            o.binst[b] = st_pack(LV(ka1) ? (sp < 0 ? 255u : (uint32_t)sp) : 254u, LV(ka1), 0u); \
                        }                                                                                      \
                    }                                                                                          \
                    if ((TWO) && (LV(ev2) & 0x98u) >= 0x88u && ((LV(fl2) & 2u) || LV(ka2) != ~0u)) {            \
                        const int sp = last_le_own(SB1 & LV(ma2), SB2 & LV(mb2), lane, 1u);                    \
                        if (LV(fl2) & 2u) LV(sp2) = sp < 0 ? 255u : (uint32_t)sp;                               \
                        if (LV(ka2) != ~0u) {                                                                  \
                            const uint32_t b = LV(ev2) & 31u;                                                  \
                            if (fused) s.gk[b] = LV(ka2);                                                      \
                            o.binst[b] = st_pack(LV(ka2) ? (sp < 0 ? 255u : (uint32_t)sp) : 254u, LV(ka2), 0u); \
                        }                                                                                      \
                    }                                                                                          \
                }                                                                                              \
            }
            if (G2) { ICER_EMU_COUNT(2); ICER_GOLOMB_BINS(1) }
            else if (G1) { ICER_EMU_COUNT(3); ICER_GOLOMB_BINS(0) }
#undef ICER_GOLOMB_BINS
#undef ICER_GOLOMB_LANE
        }
#undef ICER_MATCH
        ICER_TICK(11)
        FOR_LANES
        {
            // (everything but the events of bins 1..7, which the records wave fills in; empty positions read 0)
            const uint32_t b1 = LV(ev1) & 0x9Fu, b2 = LV(ev2) & 0x9Fu;
            if (!(b1 >= 0x81u && b1 <= 0x87u)) {
                o.rec[2 * lane] = (b1 & 0x80u) ? (LV(fl1) | ((b1 & 31u) << 2) | (LV(sp1) << 8) | (LV(wd1) << 16)) : 0u;
            }
            if (!(b2 >= 0x81u && b2 <= 0x87u)) {
                o.rec[2 * lane + 1] = (b2 & 0x80u) ? (LV(fl2) | ((b2 & 31u) << 2) | (LV(sp2) << 8) | (LV(wd2) << 16)) : 0u;
            }
        }
        ICER_PUBLISH(o.gtag, chunk_tag(j, gw.gen))
        gw.next = j + step;
        done++;
        ICER_IDLE_RESET
        ICER_TICK(12)
    }
    ICER_TIMERS_STORE(a.timers)
    return done;
}

// ==========================================================================================
// merge wave
// ==========================================================================================
struct MergeChunk {             // one chunk's events with their code-word roles, one pixel per lane
    LANEVAR(uint32_t, ev1); LANEVAR(uint32_t, ev2);     // bin of the event (bits 0..4); hybrid_chunk also loads the raw event bytes
    LANEVAR(uint32_t, fl1); LANEVAR(uint32_t, fl2);     // bit0: a word starts at this event, bit1: a word ends here
    LANEVAR(uint32_t, wd1); LANEVAR(uint32_t, wd2);     // finished ring word of an end event
    LANEVAR(uint32_t, sp1); LANEVAR(uint32_t, sp2);     // start position of the word an end event closes (255: carried in)
    LANEVAR(uint32_t, st);                              // lane b: RecSlot::binst of bin b
    uint64_t S1, S2;                                    // word-start flags of the chunk
};

// wait for the (speculative) results of the golomb, walker and records waves for chunk j and unpack them.
// (Every wait of the merge wave also ends when the unit is abandoned: the drain wave may have found the payload
// slot too small and left.)
// Returns false when the unit was abandoned; *popped = the drain wave's pop count as of the same moment.
ICER_DEV bool merge_gather(CoderShared &s, MergeChunk &c, uint32_t j, uint32_t gen, uint32_t *popped ICER_TIMER_PARAMS)
{
    DECL_LANE;
    const RecSlot &rq = s.rq[j % kQueueDepth];
    const uint32_t tag = chunk_tag(j, gen);
    uint32_t ab_ = 0;
#if defined(ICER_WAVE_EMU) && !defined(ICER_WAVE_THREADS)
    assert((rq.gtag == tag && rq.rtag == tag) || s.abort);
    ab_ = s.abort;
    *popped = s.popped;
#else
    {   // both tags, the abort word and the pop count per poll in one LDS round trip
        uint32_t spins_ = 0;
        for (;;) {
            const uint32_t g_ = ICER_LOAD_CNT(rq.gtag), r_ = ICER_LOAD_CNT(rq.rtag);
            ab_ = ICER_LOAD_CNT(s.abort);
            *popped = ICER_LOAD_CNT(s.popped);
            if (((g_ == tag) & (r_ == tag)) | (ab_ != 0u)) break;
            ICER_POLL_PAUSE(1);
            if (++spins_ > kPollLimit) { ICER_SET_ABORT2(); ab_ = 2u; break; }
        }
    }
#endif
    if (ab_) return false;
    ICER_ACQUIRE()
    ICER_TICK(18)
    ICER_MARK("SEC merge_unpack")
    FOR_LANES
    {
        const uint32_t r1 = rq.rec[2 * lane], r2 = rq.rec[2 * lane + 1];
        LV(c.fl1) = r1 & 3u; LV(c.ev1) = (r1 >> 2) & 31u; LV(c.sp1) = (r1 >> 8) & 255u; LV(c.wd1) = r1 >> 16;
        LV(c.fl2) = r2 & 3u; LV(c.ev2) = (r2 >> 2) & 31u; LV(c.sp2) = (r2 >> 8) & 255u; LV(c.wd2) = r2 >> 16;
        LV(c.st) = lane < kNumBins ? rq.binst[lane] : 255u;
    }
    c.S1 = BALLOT(LV(c.fl1) & 1u);
    c.S2 = BALLOT(LV(c.fl2) & 1u);
    ICER_TICK(19)
    return true;
}

// the bins' open words and coder state after the chunk
ICER_DEV void commit_bins(CoderShared &s, MergeChunk &c, uint32_t tail)
{
    DECL_LANE;
    const uint64_t S1 = c.S1, S2 = c.S2;
    FOR_LANES
    {
        if (lane >= 1 && lane < kNumBins) {
            const uint32_t op = LV(c.st) & 255u;
            if (op == 254u) s.bin_slot[lane] = -1;
            else if (op < 128u) s.bin_slot[lane] = (int32_t)((tail + cnt_lt(S1, S2, op)) & (kRingWords - 1));
            s.bin_state[lane] = LV(c.st);
        }
    }
}

// fast path: ring slots in allocation order = order of the words' first events (E2), finished words into
// their slots, bin states as of this (now retired) chunk.  `tail` = allocation count before the chunk.
// A word's slot is tail + (word starts before its first event): every start event knows that rank from two lane-masked
// counts and leaves it in LDS (srank) for the event that ends the word and for the bin's state -- no 64-bit mask shifted
// by another lane's position anywhere.
ICER_DEV void merge_commit(CoderShared &s, MergeChunk &c, uint32_t tail)
{
    DECL_LANE;
    const uint64_t S1 = c.S1, S2 = c.S2;
    FOR_LANES
    {
        if (LV(c.fl1) & 1u) {
            const uint32_t r = cnt_lt_own(S1, S2, lane, 0u);
            s.srank[2 * lane] = (uint8_t)r;
            RING_ST((tail + r) & (kRingWords - 1), LV(c.ev1));
        }
        if (LV(c.fl2) & 1u) {
            const uint32_t r = cnt_lt_own(S1, S2, lane, 1u);
            s.srank[2 * lane + 1] = (uint8_t)r;
            RING_ST((tail + r) & (kRingWords - 1), LV(c.ev2));
        }
    }
    WAVE_SYNC();
    FOR_LANES
    {
        if (LV(c.fl1) & 2u) {
            const uint32_t slot = LV(c.sp1) == 255u ? (uint32_t)s.bin_slot[LV(c.ev1)] : tail + (uint32_t)s.srank[LV(c.sp1) & 127u];
            RING_ST(slot & (kRingWords - 1), LV(c.wd1));
        }
        if (LV(c.fl2) & 2u) {
            const uint32_t slot = LV(c.sp2) == 255u ? (uint32_t)s.bin_slot[LV(c.ev2)] : tail + (uint32_t)s.srank[LV(c.sp2) & 127u];
            RING_ST(slot & (kRingWords - 1), LV(c.wd2));
        }
    }
    WAVE_SYNC();
    // the bins' open words and coder state after the chunk (as commit_bins, ranks from srank)
    FOR_LANES
    {
        if (lane >= 1 && lane < kNumBins) {
            const uint32_t op = LV(c.st) & 255u;
            if (op == 254u) s.bin_slot[lane] = -1;
            else if (op < 128u) s.bin_slot[lane] = (int32_t)((tail + (uint32_t)s.srank[op]) & (kRingWords - 1));
            s.bin_state[lane] = LV(c.st);
        }
    }
}

// ring stores for the chunk's events at positions [lo, hi): open markers of the words that start there, finished
// words of the ones that end there (slot of a word = tail0 + number of word starts before its first event, E2)
ICER_DEV void commit_range(CoderShared &s, MergeChunk &c, uint32_t tail0, uint32_t lo, uint32_t hi)
{
    DECL_LANE;
    const uint64_t S1 = c.S1, S2 = c.S2;
    FOR_LANES
    {
        const uint32_t p1 = 2u * (uint32_t)lane, p2 = p1 + 1u;
        if ((LV(c.fl1) & 1u) && p1 >= lo && p1 < hi) RING_ST((tail0 + cnt_lt_own(S1, S2, lane, 0u)) & (kRingWords - 1), (LV(c.ev1) & 31u));
        if ((LV(c.fl2) & 1u) && p2 >= lo && p2 < hi) RING_ST((tail0 + cnt_lt_own(S1, S2, lane, 1u)) & (kRingWords - 1), (LV(c.ev2) & 31u));
    }
    FOR_LANES
    {
        const uint32_t p1 = 2u * (uint32_t)lane, p2 = p1 + 1u;
        if ((LV(c.fl1) & 2u) && p1 >= lo && p1 < hi) {
            const uint32_t slot = LV(c.sp1) == 255u ? (uint32_t)s.bin_slot[LV(c.ev1) & 31u] : (tail0 + cnt_lt(S1, S2, LV(c.sp1)));
            RING_ST(slot & (kRingWords - 1), LV(c.wd1));
        }
        if ((LV(c.fl2) & 2u) && p2 >= lo && p2 < hi) {
            const uint32_t slot = LV(c.sp2) == 255u ? (uint32_t)s.bin_slot[LV(c.ev2) & 31u] : (tail0 + cnt_lt(S1, S2, LV(c.sp2)));
            RING_ST(slot & (kRingWords - 1), LV(c.wd2));
        }
    }
    WAVE_SYNC();
}

// next event of a bin: lowest position in (Q1: even positions 2 * lane, Q2: odd positions 2 * lane + 1); removes it
#define ICER_NEXT_EVENT(Q1, Q2, POS, V)                                                                    \
    {                                                                                                      \
        const uint32_t q1_ = (Q1) ? 2u * (uint32_t)ffs64(Q1) : 999u, q2_ = (Q2) ? 2u * (uint32_t)ffs64(Q2) + 1u : 999u; \
        if (q1_ < q2_) { POS = q1_; V = READLANE(c.ev1, q1_ >> 1); (Q1) &= (Q1) - 1ull; }                  \
        else { POS = q2_; V = READLANE(c.ev2, q2_ >> 1); (Q2) &= (Q2) - 1ull; }                             \
    }

// A chunk inside which the ring may fill up.  The speculative results (c) are right up to the first word start that
// finds the ring full -- position P, known from the ring occupancy alone -- so events before P are committed as in
// the fast path.  Then everything finished is popped (the reference pops after every event; popping is only
// observable through `used` when a word is allocated) and, if the ring is still full, the oldest word is
// force-completed (E5, icer_flush_encode icer_encoding.c:141-189: it belongs to bin hb, whose state at P follows
// from hb's events since the word's start).  A forced flush changes the word boundaries of bin hb only: hb's
// events from P on are replayed one by one (icer_encode_bit, icer_encoding.c:37-112) with the bin starting afresh,
// all other results stay valid, and the scheme repeats from P.  Returns true if a word was force-completed (the
// results the walker / golomb / records waves have produced for later chunks are then void for bin hb).
ICER_DEV bool hybrid_chunk(CoderShared &s, MergeChunk &c, uint32_t j, uint32_t tail0)
{
    DECL_LANE;
    uint32_t base = 0;
    bool flushed = false;
    {
        const EventSlot &q = s.eq[j % kQueueDepth];        // the raw events: 0x80 | bit << 5 | bin
        FOR_LANES
        {
            LV(c.ev1) = q.ev1[lane];
            LV(c.ev2) = q.ev2[lane];
        }
    }
    for (;;) {
        const uint64_t S1 = c.S1, S2 = c.S2;
        const uint32_t t = (uint32_t)kRingWords - (tail0 - s.popped);           // rank of the first word start that finds the ring full
        if (t >= (uint32_t)(popc64(S1) + popc64(S2))) break;
        const uint64_t h1 = BALLOT((LV(c.fl1) & 1u) && cnt_lt_own(S1, S2, lane, 0u) == t);
        const uint64_t h2 = BALLOT((LV(c.fl2) & 1u) && cnt_lt_own(S1, S2, lane, 1u) == t);
        const uint32_t P = h1 ? 2u * (uint32_t)ffs64(h1) : 2u * (uint32_t)ffs64(h2) + 1u;
        commit_range(s, c, tail0, base, P);
        const uint32_t alloc = tail0 + t;
        wave_drain(s, alloc, 2u);
        if (alloc - s.popped == (uint32_t)kRingWords) {
            // still full: the head word is open.  Its bin, and that bin's state just before P:
            const uint32_t head = s.popped & (kRingWords - 1);
            const uint32_t hb = RING_LD(head) & 31u;
            const uint64_t E1 = BALLOT((LV(c.ev1) & 0x9Fu) == (0x80u | hb)), E2 = BALLOT((LV(c.ev2) & 0x9Fu) == (0x80u | hb));
            const int x = last_lt(S1 & E1, S2 & E2, P);                           // first event of the open word, -1: carried in
            const uint32_t lo = x < 0 ? 0u : (uint32_t)x;
            uint64_t R1 = E1 & below64((P + 1u) >> 1) & ~below64((lo + 1u) >> 1);   // hb's events in [lo, P)
            uint64_t R2 = E2 & below64(P >> 1) & ~below64(lo >> 1);
            uint32_t hacc = x < 0 ? st_acc(s.bin_state[hb]) : 0u, hnin = x < 0 ? st_nin(s.bin_state[hb]) : 0u;
            uint32_t word;
            if (hb >= 8u) {
                hacc += (uint32_t)(popc64(R1) + popc64(R2));                       // all zeros, or the word would have ended
                word = (hacc == (uint32_t)s.tab.gm[hb] - 1u) ? (kWordDone | (1u << 11) | 1u) : golomb_word(s.tab, (int)hb, hacc);
            } else {
                while (R1 | R2) {
                    uint32_t q, v;
                    ICER_NEXT_EVENT(R1, R2, q, v)
                    (void)q;
                    hacc |= ((v >> 5) & 1u) << hnin;
                    hnin++;
                }
                const uint32_t f = s.tab.v2v_flush[hb][hacc > 8u ? 8u : hacc][hnin > 5u ? 5u : hnin];
                const uint32_t en = s.tab.v2v[hb][(hacc | ((f & 15u) << hnin)) & 31u];
                word = kWordDone | (((en >> 4) & 15u) << 11) | (en >> 8);       // QUIRK (kept): not checked to be a code word
            }
            FOR_LANES
            {
                if (lane == 0) RING_ST(head, word);
            }
            WAVE_SYNC();
            // bin hb starts afresh at P: replay its remaining events of the chunk
            uint64_t Q1 = E1 & ~below64((P + 1u) >> 1), Q2 = E2 & ~below64(P >> 1);
            bool open = false;
            uint32_t acc = 0, nin = 0, spos = 255;
            while (Q1 | Q2) {
                uint32_t q, v;
                ICER_NEXT_EVENT(Q1, Q2, q, v)
                const uint32_t bit = (v >> 5) & 1u;
                const uint32_t starts = open ? 0u : 1u;
                if (!open) { open = true; spos = q; }
                uint32_t wd = 0;
                bool close = false;
                if (hb >= 8u) {
                    if (bit) { wd = golomb_word(s.tab, (int)hb, acc); close = true; }
                    else if (acc + 1u >= s.tab.gm[hb]) { wd = kWordDone | (1u << 11) | 1u; close = true; }
                    else acc++;
                } else if (hb >= 1u) {
                    acc |= bit << nin;
                    nin++;
                    const uint32_t en = s.tab.v2v[hb][acc & 31u];
                    if ((en & 15u) == nin) { wd = kWordDone | (((en >> 4) & 15u) << 11) | (en >> 8); close = true; }
                } else {
                    wd = kWordDone | (1u << 11) | bit;
                    close = true;
                }
                const uint32_t fl = starts | (close ? 2u : 0u), sp = close ? spos : 255u;
                FOR_LANES
                {
                    if ((uint32_t)lane == (q >> 1)) {
                        if (q & 1u) { LV(c.fl2) = fl; LV(c.sp2) = sp; LV(c.wd2) = wd; }
                        else { LV(c.fl1) = fl; LV(c.sp1) = sp; LV(c.wd1) = wd; }
                    }
                }
                if (close) { open = false; acc = 0; nin = 0; }
            }
            FOR_LANES
            {
                if ((uint32_t)lane == hb) LV(c.st) = st_pack(open ? spos : 254u, acc, nin);
            }
            c.S1 = BALLOT(LV(c.fl1) & 1u);
            c.S2 = BALLOT(LV(c.fl2) & 1u);
            flushed = true;
            wave_drain(s, alloc, 2u);
        }
        base = P;
    }
    commit_range(s, c, tail0, base, 128u);
    commit_bins(s, c, tail0);
    WAVE_SYNC();
    return flushed;
}
#undef ICER_NEXT_EVENT

// ==========================================================================================
// records wave and drain wave; the merge wave's hand-shake with the drain wave
// ==========================================================================================
// Records wave: once the walker wave has walked chunk r, every event lane of bins 1..7 derives from the bin's
// start flags whether a code word starts / ends at its event and, for an end, the finished ring word and the
// position of the word's first event.  `max_steps` bounds one call in the emulation (the GPU passes ~0u).
struct RecordsWave { uint32_t next = 0, gen = 0; };     // next chunk / generation, as WalkWave (next = the workgroup's first chunk)

ICER_DEV void records_wave_run(CoderShared &s, const UnitArgs &a, RecordsWave &rw, uint32_t max_steps)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    ICER_IDLE_DECL
    const uint32_t nchunks = s.nchunks;
    for (uint32_t step = 0;;) {
        // (control words read together: one LDS round trip; the tag is the walker's for the chunk we expect next --
        // after a roll-back it is simply re-read on the next pass)
        const uint32_t ab_ = ICER_LOAD_CNT(s.abort), seq = ICER_LOAD_CNT(s.exact_seq);
        const uint32_t tg_ = ICER_LOAD_CNT(s.wq[rw.next % kQueueDepth].tag);
        if (ab_) break;
        if (seq != rw.gen) {                            // records for chunks after last_exact are void
            ICER_ACQUIRE()
            rw.gen = seq;
            rw.next = s.last_exact + 1u;
            continue;
        }
        const uint32_t r = rw.next, gen = rw.gen;
        if (r >= nchunks || tg_ != chunk_tag(r, gen)) {
            if (ICER_LOAD_CNT(s.b_done) >= nchunks || step >= max_steps) break;
            ICER_IDLE()
            continue;
        }
        if (step >= max_steps) break;
        ICER_ACQUIRE()
        ICER_TICK(20)
        const EventSlot &q = s.eq[r % kQueueDepth];
        const WalkSlot &o = s.wq[r % kQueueDepth];
        RecSlot &ro = s.rq[r % kQueueDepth];
#define ICER_V2V_RECORD(EV, RK, POS)                                                                   \
            if (((EV)&0x98u) == 0x80u && ((EV)&7u)) {                                                  \
                const uint32_t b_ = (EV)&7u, r_ = (RK), n_ = q.binn[b_];                                \
                const uint32_t sw_ = window6(o.binstart[b_], (int)r_ - 4);  /* starts at ranks r-4 .. r+1 */ \
                const uint32_t bw_ = window6(q.binbits[b_], (int)r_ - 4);   /* input bits, same ranks */    \
                const uint32_t starts_ = (sw_ >> 4) & 1u;                                               \
                const uint32_t carry_ = o.bincarry[b_];                                                 \
                const uint32_t ends_ = (r_ + 1u < n_) ? ((sw_ >> 5) & 1u) : (o.post_nin[b_] == 0u ? 1u : 0u); \
                uint32_t wd_ = 0, sp_ = 255;                                                            \
                if (ends_) {                                                                            \
                    const uint32_t back_ = sw_ & 31u;                        /* starts at r-4 .. r */   \
                    uint32_t acc_;                                                                      \
                    if (back_) {                                                                        \
                        const uint32_t k_ = 31u - (uint32_t)clz32(back_);    /* start at rank r-4+k */  \
                        acc_ = (bw_ & 31u) >> k_;                                                       \
                        sp_ = q.binseq[b_][r_ - 4u + k_];                                               \
                    } else {                                                 /* the carried-in word */  \
                        const uint32_t cn_ = 31u - (uint32_t)clz32(carry_);                             \
                        acc_ = (carry_ ^ (1u << cn_)) | (((bw_ & 31u) >> (4u - r_)) << cn_);            \
                    }                                                                                   \
                    const uint32_t e_ = s.tab.v2v[b_][acc_ & 31u];                                      \
                    wd_ = kWordDone | (((e_ >> 4) & 15u) << 11) | (e_ >> 8);                            \
                }                                                                                       \
                ro.rec[POS] = starts_ | (ends_ << 1) | (b_ << 2) | (sp_ << 8) | (wd_ << 16);            \
            }
            FOR_LANES
            {
                const uint32_t e1 = q.ev1[lane], e2 = q.ev2[lane];
                ICER_V2V_RECORD(e1, (uint32_t)q.rk1[lane], 2 * lane)
                ICER_V2V_RECORD(e2, (uint32_t)q.rk2[lane], 2 * lane + 1)
            }
#undef ICER_V2V_RECORD
#ifdef ICER_WAVE_THREADS
        // (test build) the walker's slot was not recycled under our feet: same chunk, or a roll-back made it void
        assert(ICER_LOAD_CNT(o.tag) == chunk_tag(r, gen) || ICER_LOAD_CNT(s.exact_seq) != gen);
#endif
        ICER_PUBLISH(ro.rtag, chunk_tag(r, gen))
        rw.next = r + 1u;
        ICER_TICK(21)
        step++;
        ICER_IDLE_RESET
    }
    ICER_TIMERS_STORE(a.timers)
}

// Drain wave: pops finished words from the head of the ring and writes the payload, until told to park (hold_seq
// odd) or to exit.
ICER_DEV void drain_wave_run(CoderShared &s, const UnitArgs &a, uint32_t max_steps)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    ICER_IDLE_DECL
    uint32_t idle = 0;
    for (uint32_t step = 0;;) {
        if (ICER_LOAD_CNT(s.abort)) break;
        // drain_exit is read BEFORE hold_seq, and the load is complete before the next one is issued.  The merge wave ends a
        // unit with "drain_exit = 1; hold_seq = odd; wait for hold_ack == hold_seq".  Read the other way round (round 1 and 2),
        // this wave could see a hold it had already acknowledged (hs = 23), then -- the merge wave releasing that hold and
        // ending the unit in between -- drain_exit = 1, acknowledge the stale 23 and leave, the merge wave waiting for 25 until
        // its spin bound: the rare coding-unit time-out (one unit in ~10^5 whose last chunk took the exact path; seen once
        // more at four workgroups per compute unit: profiles/archive/r03_logs/r03_full_bench_timeout.err).  With this order
        // drain_exit = 1 implies that the hold_seq read afterwards is the final request or the even value before it.
        const uint32_t ex_ = ICER_LOAD_CNT(s.drain_exit);
        ICER_ACQUIRE()
        const uint32_t hs = ICER_LOAD_CNT(s.hold_seq);
        if (hs & 1u) {
            // parked: the merge wave owns popped / bitpos / the bit stage until it releases the hold
            // (end of unit: the payload words this wave stored are read back by the merge wave for the CRC; the hand-off
            // fences are LDS-only, so the stores are completed explicitly before the acknowledgement)
            if (ex_) ICER_GLOBAL_RELEASE();
            ICER_PUBLISH(s.hold_ack, hs)
            if (ex_ || step >= max_steps) break;
            ICER_IDLE()
            continue;
        }
        if (step >= max_steps) break;
        const uint32_t limit = ICER_LOAD_CNT(s.alloc);
        // a drain pass has a fixed cost: run one when enough words have piled up or nothing else happened for a while
        if (limit - s.popped >= 128u || (limit != s.popped && idle >= 16u)) {
            ICER_ACQUIRE()
            ICER_TICK(24)
            const uint32_t npop = wave_drain(s, limit, 4u);     // bounded, so that a hold request is answered soon
            ICER_TICK(22)
            if (npop && !flush_stage(s, a, false)) {     // payload slot too small: abandon the unit
                ICER_PUBLISH(s.abort, 1u)
                break;
            }
            ICER_TICK(23)
            idle = 0;
            ICER_IDLE_RESET
            step++;
            continue;
        }
        idle++;
        ICER_IDLE()
    }
    ICER_TIMERS_STORE(a.timers)
}

// merge wave: take over / give back the drain state
#if defined(ICER_WAVE_EMU) && !defined(ICER_WAVE_THREADS)
#define ICER_DRAIN_HOLD(S, A) { (S).hold_seq |= 1u; drain_wave_run((S), (A), 0u); assert((S).hold_ack == (S).hold_seq); }
#else
#define ICER_DRAIN_HOLD(S, A) { const uint32_t hs_ = (S).hold_seq | 1u; ICER_PUBLISH((S).hold_seq, hs_) ICER_WAIT_UNTIL(ICER_LOAD_CNT((S).hold_ack) == hs_ || ICER_LOAD_CNT((S).abort)) }
#endif
#define ICER_DRAIN_RELEASE(S) { ICER_PUBLISH((S).hold_seq, ((S).hold_seq | 1u) + 1u) }

// ---- sub-ranges: snapshots of the coder state and the comparison with them (see "Sub-ranges" above) -----------------
// Is `nj` (the chunk about to start) a point where this workgroup writes a snapshot of its own (*write_m) or compares its
// state with a snapshot of a later sub-range (*match_t, *match_m)?
ICER_DEV bool sub_checkpoint(const SubLayout &L, uint32_t nj, int *write_m, uint32_t *match_t, int *match_m)
{
    *write_m = -1; *match_m = -1; *match_t = 0;
    const uint32_t i = L.index, end = L.first[L.n_sub];
    if (nj >= end) return false;
    if (i >= 1u && nj > L.first[i]) {
        const uint32_t d = nj - L.first[i];
        if (d % kSnapEvery == 0u && d / kSnapEvery <= kMaxSnaps) *write_m = (int)(d / kSnapEvery) - 1;
    }
    uint32_t t = L.n_sub - 1u;
    while (t > i && nj <= L.first[t]) t--;                      // the latest sub-range start this workgroup has passed
    if (t > i) {
        const uint32_t d = nj - L.first[t];
        if (d % kSnapEvery == 0u && d / kSnapEvery <= kMaxSnaps) { *match_t = t; *match_m = (int)(d / kSnapEvery) - 1; }
    }
    return *write_m >= 0 || *match_m >= 0;
}

// The first chunk count nj >= from at which sub_checkpoint fires (~0u: none).  The merge wave keeps it in a register and looks at the
// layout again only there -- the layout sits behind a generic pointer, three dependent loads per look (0.5 k cycles per chunk of the
// merge wave of a split unit, the slowest stage of a lone frame's pipeline, before round 6).
ICER_DEV uint32_t sub_next_checkpoint(const SubLayout &L, uint32_t from)
{
    const uint32_t i = L.index, n = L.n_sub, end = L.first[n];
    uint32_t best = ~0u;
    for (uint32_t t = i ? i : 1u; t < n; t++) {
        // t == i: this workgroup's own snapshots; t > i: those of a later sub-range, compared while t is the latest start passed
        const uint32_t f = L.first[t];
        const uint32_t m = from > f ? (from - f + kSnapEvery - 1u) / kSnapEvery : 1u;
        const uint32_t nj = f + (m ? m : 1u) * kSnapEvery;
        const uint32_t last = (t == i || t + 1u >= n) ? end - 1u : L.first[t + 1u];     // (a later start passed: its snapshots take over)
        if ((m ? m : 1u) <= kMaxSnaps && nj < end && nj <= last && nj < best) best = nj;
    }
    return best;
}

// The coder state before chunk `nj`, with the drain wave parked and everything finished popped: written to `out` (writer)
// or compared with `ref` (returns true when equal).  `q` = the event slot of chunk nj - 1 (its counts), `tail` = the
// allocation count.  One wavefront.
ICER_DEV bool sub_state(CoderShared &s, const EventSlot &q, uint32_t tail, uint32_t nj, Snapshot *out, const Snapshot *ref)
{
    DECL_LANE;
    const uint32_t popped = s.popped, nring = tail - popped;
    LANEVAR(uint32_t, bad);
    FOR_LANES
    {
        LV(bad) = 0;
        if (lane < kNumBins) {
            const uint32_t cnt = q.cnt[lane], st = lane >= 1 ? s.bin_state[lane] >> 8 : 0u;
            const uint32_t off = (lane >= 1 && s.bin_slot[lane] >= 0) ? (((uint32_t)s.bin_slot[lane] - popped) & (uint32_t)(kRingWords - 1)) : ~0u;
            if (out) { out->cnt[lane] = cnt; out->bin_state[lane] = st; out->open_off[lane] = off; }
            else LV(bad) = (ref->cnt[lane] != cnt || ref->bin_state[lane] != st || ref->open_off[lane] != off) ? 1u : 0u;
        }
        if (lane == 0) {
            if (out) { out->chunk = nj; out->bitpos = s.bitpos; out->nring = nring; }
            else if (ref->chunk != nj || ref->nring != nring) LV(bad) = 1u;
        }
    }
    if (!out && BALLOT(LV(bad) != 0u)) return false;
    FOR_LANES
    {
        for (uint32_t k = (uint32_t)lane; k < nring; k += 64u) {
            const uint16_t w = (uint16_t)RING_LD((popped + k) & (uint32_t)(kRingWords - 1));
            if (out) out->ring[k] = w;
            else if (ref->ring[k] != w) LV(bad) = 1u;
        }
    }
    return out ? true : BALLOT(LV(bad) != 0u) == 0ull;
}

constexpr uint32_t kMergeAbandoned = 0, kMergeDone = 1, kMergeMatched = 2;

// chunks [j0, j1); returns kMergeAbandoned when the unit was abandoned (payload slot too small), kMergeMatched when the
// workgroup's state equalled a later sub-range's snapshot (it has written its SubRecord and stopped), else kMergeDone
ICER_DEV uint32_t merge_wave_run(CoderShared &s, const UnitArgs &a, uint32_t j0, uint32_t j1)
{
    DECL_LANE;
    ICER_TIMERS_DECL
    uint32_t tail = s.alloc;                                // allocation count (this wave owns it)
    uint32_t gen = s.exact_seq;                             // generation (this wave bumps it)
    uint32_t next_cp = a.sub ? sub_next_checkpoint(*a.sub, j0 + 1u) : ~0u;   // sub-ranges: the next chunk count with a snapshot to write or to compare
    for (uint32_t j = j0; j < j1; j++) {
        MergeChunk c;
        uint32_t popped_seen;
        if (!merge_gather(s, c, j, gen, &popped_seen ICER_TIMER_PASS)) { ICER_TIMERS_STORE(a.timers) return kMergeAbandoned; }
        // If the ring cannot fill up inside this chunk no forced flush (E5) is possible and word boundaries depend
        // on each bin alone: the speculative results say how many words the chunk opens *if* no flush happens, and
        // if they all fit none happens.  The drain wave's pop count may lag, which only over-estimates the
        // occupancy; when the test fails with it the drain wave is parked, everything finished is popped (the
        // reference's state) and the test repeated.
        ICER_MARK("SEC merge_test")
        const uint32_t nstarts = (uint32_t)(popc64(c.S1) + popc64(c.S2));
        bool held = false, fast = true;
        ICER_COUNT(31)
        if (tail - popped_seen + nstarts > (uint32_t)kRingWords) {
            ICER_DRAIN_HOLD(s, a)
            if (ICER_LOAD_CNT(s.abort)) { ICER_TIMERS_STORE(a.timers) return kMergeAbandoned; }
            held = true;
            // (two rounds = 128 words is all a chunk can need; whatever else is finished is left to the drain wave)
            wave_drain(s, tail, 2u);
            fast = tail - s.popped + nstarts <= (uint32_t)kRingWords;
            ICER_COUNT(30)
        }
        if (fast) {
            ICER_MARK("SEC merge_commit")
            merge_commit(s, c, tail);
            ICER_MARK("SEC merge_commit_end")
            tail += nstarts;
            ICER_EMU_COUNT(0);
            ICER_TICK(14)
        } else {
            // (merge_gather has waited for the walker, golomb and records waves: they are past their speculative pass
            // over this chunk; bin_state holds the bins' state as of the last retired chunk)
            const bool flushed = hybrid_chunk(s, c, j, tail);
            tail += (uint32_t)(popc64(c.S1) + popc64(c.S2));
            if (flushed) {
                ICER_EMU_COUNT(1);
                ICER_COUNT(29)
                // results produced for later chunks assumed no forced flush here: void them
                FOR_LANES
                {
                    if (lane == 0) s.last_exact = j;
                }
                gen++;
                ICER_PUBLISH(s.exact_seq, gen)
            } else {
                ICER_EMU_COUNT(0);
            }
            ICER_TICK(16)
        }
        // sub-ranges: a snapshot of the state before the next chunk, or the comparison with a later sub-range's one
        ICER_MARK("SEC merge_retire")
#ifdef ICER_WAVE_EMU
        if (a.sub) { int wm_, mm_; uint32_t mt_; assert(sub_checkpoint(*a.sub, j + 1u, &wm_, &mt_, &mm_) == (j + 1u == next_cp) || j + 1u > next_cp); }
#endif
        if (a.sub && j + 1u >= next_cp) {
            int write_m, match_m;
            uint32_t match_t;
            next_cp = sub_next_checkpoint(*a.sub, j + 2u);
            if (sub_checkpoint(*a.sub, j + 1u, &write_m, &match_t, &match_m)) {
                WAVE_SYNC();
                if (!held) {
                    ICER_DRAIN_HOLD(s, a)
                    if (ICER_LOAD_CNT(s.abort)) { ICER_TIMERS_STORE(a.timers) return kMergeAbandoned; }
                    held = true;
                }
                wave_drain(s, tail, ~0u);                         // everything finished is popped: the state is canonical
                const EventSlot &q = s.eq[j % kQueueDepth];
                const SubLayout &L = *a.sub;
                if (write_m >= 0) {
                    const uint32_t slot = L.index * kMaxSnaps + (uint32_t)write_m;
                    sub_state(s, q, tail, j + 1u, &L.snaps[slot], nullptr);
                    ICER_AGENT_PUBLISH(&L.snap_valid[slot], 1u)
                }
                if (match_m >= 0) {
                    const uint32_t slot = match_t * kMaxSnaps + (uint32_t)match_m;
                    if (ICER_AGENT_ACQUIRE(&L.snap_valid[slot]) == 1u && sub_state(s, q, tail, j + 1u, nullptr, &L.snaps[slot])) {
                        // same state, same input from here on: the other workgroup's bits continue this one's
                        const bool fits = flush_stage(s, a, true);
                        FOR_LANES
                        {
                            if (lane == 0) {
                                SubRecord &r = L.rec[L.index];
                                r.end_chunk = j + 1u; r.end_bits = fits ? s.bitpos : kUnitTooBig; r.match_sub = match_t; r.match_snap = (uint32_t)match_m;
                            }
                        }
                        ICER_AGENT_PUBLISH(&L.rec[L.index].done, 1u)
                        ICER_PUBLISH(s.abort, 4u)
                        ICER_TIMERS_STORE(a.timers)
                        return kMergeMatched;
                    }
                }
            }
        }
        if (held) {
            // what this wave drained while holding goes out now (the drain wave only flushes after its own pops)
            if (!flush_stage(s, a, false)) {
                ICER_PUBLISH(s.abort, 1u)
                ICER_TIMERS_STORE(a.timers)
                return kMergeAbandoned;
            }
            ICER_PUBLISH(s.alloc, tail)
            ICER_DRAIN_RELEASE(s)
            ICER_PUBLISH(s.b_done, j + 1u)
        } else {
#ifdef ICER_WAVE_THREADS
            {   // (test build) nobody recycled the chunk's slots before it was retired
                const RecSlot &rq_ = s.rq[j % kQueueDepth];
                assert(ICER_LOAD_CNT(rq_.gtag) == chunk_tag(j, gen) && ICER_LOAD_CNT(rq_.rtag) == chunk_tag(j, gen));
                assert(ICER_LOAD_CNT(s.a_done) <= j + kQueueDepth);
            }
#endif
            // the new words (and the finished ones) become visible to the drain wave; the chunk's queue slots are free
            ICER_PUBLISH2(s.alloc, tail, s.b_done, j + 1u)
        }
        ICER_MARK("SEC merge_retire_end")
        ICER_TICK(17)
    }
    ICER_TIMERS_STORE(a.timers)
    return kMergeDone;
}

// end of unit: park the drain wave for good, force-complete whatever is still open (C8,
// icer_context_modeller.c:452-455); returns the payload length in bits, or kUnitTooBig
ICER_DEV uint32_t merge_wave_finish(CoderShared &s, const UnitArgs &a)
{
    DECL_LANE;
    ICER_PUBLISH(s.drain_exit, 1u)
    ICER_DRAIN_HOLD(s, a)
    if (ICER_LOAD_CNT(s.abort)) return kUnitTooBig;
    wave_drain(s, s.alloc, ~0u);
    while (s.alloc != s.popped) {
        FOR_LANES
        {
            if (lane == 0) seq_complete_head(s);
        }
        WAVE_SYNC();
        wave_drain(s, s.alloc, ~0u);
    }
    return flush_stage(s, a, true) ? s.bitpos : kUnitTooBig;
}

// state every wave relies on; run by ONE wave before the others start (a workgroup barrier follows on the GPU)
// (`j0`: the workgroup's first chunk -- 0 unless it codes a later sub-range of the unit)
ICER_DEV void unit_state_init(CoderShared &s, uint32_t j0 = 0)
{
    DECL_LANE;
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)lane; i < kStageWords; i += 64) s.stage[i] = 0;
        if (lane < kNumBins) { s.bin_slot[lane] = -1; s.bin_state[lane] = 0; s.gk[lane] = 0; }
        if (lane == 0) {
            s.alloc = 0; s.popped = 0; s.bitpos = 0; s.flushed_words = 0; s.hold_seq = 0; s.hold_ack = 0; s.drain_exit = 0;
            for (uint32_t i = 0; i < kMaxPixelWaves; i++) s.p_done[i] = j0;
            s.a_done = j0; s.c_done = j0; s.b_done = j0; s.abort = 0; s.abort_site = 0; s.exact_seq = 0; s.last_exact = 0;
            for (uint32_t i = 0; i < kQueueDepth; i++) { s.wq[i].tag = 0; s.rq[i].rtag = 0; s.rq[i].gtag = 0; s.kq[i].tag = 0; }
        }
    }
    WAVE_SYNC();
}

}  // namespace icer
