// coder_wg.hpp -- one workgroup of kWgWaves wavefronts codes one ICER coding unit
// (channel, level, subband, plane, segment); barriers only, no wave ever waits on a flag.
//
// Replaces the reference's per-segment chain (the uint8 twins are the same code on 7 planes)
//   icer_compress_bitplane_uint16   lib_icer/src/icer_context_modeller.c:312-457
//   icer_encode_bit / icer_compute_bin   icer_encoding.c:37-112, icer_util.c:48-56
//   icer_popbuf_while_avail / icer_flush_encode   icer_encoding.c:114-189
// and must emit the identical payload bits.
//
// The segment is coded in WINDOWS of kWgWaves chunks of 64 pixels (raster order inside the segment): wave w owns
// chunk j0 + w of the window and does ALL the work for it; what a chunk needs from the chunks before it is a small
// per-chunk summary that composes, so every wave rebuilds its own start state from the summaries of the waves before
// it (a scan over at most kWgWaves - 1 summaries read from LDS) after a workgroup barrier:
//
//   state                        summary of a chunk                                     composition
//   adaptive counts (17 ctx)     events and zeros per context (+ zeros up to rank k)    add; rescale 500 -> 250 in closed form
//   code-tree node, bins 1..7    end node for every start node (3 bits x 8 nodes)       function composition
//   Golomb run, bins 8..16       zeros after the last one-event / zeros if none         replace / add (mod m)
//   ring slots (E2)              words opened; per bin: closed / opened at rank r       add; last writer wins
//   payload bit position         code-word lengths of the popped ring words            add
//
//   phase A  (per wave)  pixel category, magnitude bit, 8-neighbour context, sign context; rank of every event among
//                        the chunk's events of the same context                                         | barrier
//   phase B  scan counts; counts every event sees; probability fold + bin selection (E1)
//   phase C  bins 1..7: events grouped per bin, the bin's bit string walked through the code tree from EVERY node
//            of the tree (one lane per node, six input bits per table look-up); bins 8..16: run summary      | barrier
//   phase D  scan coder states; pick the walk that started at the right node
//   phase E  per event: does a code word start / end here, the finished word, its first event           | barrier
//   phase F  scan ring slots; forced-flush test (below); open markers and finished words into the ring  | barrier
//   drain    the ring head is the oldest open word (min over the bins' open slots): everything before it is
//            finished; lengths -> prefix sums over lanes and waves -> bits OR-ed into the LDS bit stage;
//            complete 32-bit words stored coalesced to the unit's payload slot                           | 2 barriers
//
// Exactness (E5).  Word boundaries depend on each bin alone unless a word start finds the 2048-word ring full; then
// the oldest word is force-completed (icer_flush_encode).  Only a word that was already open when the window began can
// be hit inside the window (a window opens at most kWgWaves * 128 <= 2048 words), there are at most 16 of those
// (one per bin), and whether word s_b of bin b is hit is a comparison of two positions known from the summaries: the
// word start that allocates slot s_b + 2048 and bin b's first end event.  So the first forced flush of a window is
// found exactly, with no speculation to undo: the chunks before it are committed as computed, the chunk in which it
// falls is replayed by its wave with the reference's event-by-event rule for the flushed bin (exact_chunk), and the
// remaining chunks repeat phases D-F from the new state (their events, bins and per-node walks do not depend on it).
// The physical ring has 4096 entries so that a whole window can be committed before it is drained.
//
// Written with the SPMD macros of wave.hpp (see there for the tests-only CPU build).
#pragma once
#include "icer_tables.hpp"
#include "wave.hpp"

#ifndef ICER_WG_WAVES
#define ICER_WG_WAVES 16
#endif

// A region executed by every wave of the workgroup, closed by a workgroup barrier.  (tests-only lane-loop build: the
// waves' register state is an array and the region a loop over it.)
#ifdef ICER_WAVE_EMU
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#define WG_REGS_PARAM(T, name) T (&name)[ICER_WG_WAVES]
// (the order in which the waves run a region must not matter: g_wg_order = 1 runs them backwards, >= 2 shuffled)
#define WG_EACH_WAVE for (uint32_t wi_ = 0; wi_ < (uint32_t)ICER_WG_WAVES; ++wi_) { const uint32_t w = icer::ICER_WG_NS::wg_wave_order(wi_); auto &R = regs[w];
#define WG_BARRIER }
#define WG_BARRIER_NONE }
#define WG_UNIFORM(field) (regs[0].field)
#define WG_ASSERT(x) do { if (!(x)) { wg_assert_fail(#x, __LINE__); } } while (0)
#define WG_GLOBAL_RELEASE()
#define WG_GLOBAL_ACQUIRE()
#else
#define WG_REGS_PARAM(T, name) T &name
#define WG_EACH_WAVE { const uint32_t w = threadIdx.x >> 6; auto &R = regs;
#define WG_BARRIER } __syncthreads();
#define WG_BARRIER_NONE }                     // closes a region that only touches this wave's registers
#define WG_UNIFORM(field) (regs.field)
#define WG_ASSERT(x)
#define WG_STAT(i)
#define WG_GLOBAL_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#ifdef ICER_PHASE_TIMERS
// profiling build only (libicer_hip_prof.so): cycles since the previous tick go to bucket k of this wave
#define WG_TICK(k) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); R.tacc[k] += (uint32_t)(t_ - R.tlast); R.tlast = t_; }
#define WG_COUNT(k) { R.tacc[k] += 1u; }
#endif
#define WG_GLOBAL_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#endif
#ifndef WG_TICK
#define WG_TICK(k)
#endif
#ifndef WG_COUNT
#define WG_COUNT(k)
#endif

#define ICER_WG_NS wg
#include "coder_wg_impl.hpp"
#undef ICER_WG_NS
