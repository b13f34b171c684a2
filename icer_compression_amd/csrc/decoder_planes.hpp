// decoder_planes.hpp -- the decode kernel of round 4: one WAVEFRONT per packet (one bit plane of one segment), the nine
// planes of a chain as the nine waves of a workgroup over a ring of rows in LDS.
//
// Replaces, for chains whose packets take the fast entropy path (>= kFastPacketBits bits each, decoder_core.hpp), the
// lane-per-plane kernel of decoder_wave.hpp, whose nine lanes execute the union of each other's paths (~ 450 instructions
// per decision round).  Here a decision is WAVE-UNIFORM: the wave's scalar unit runs the state machine of
//   icer_decode_bit                       lib_icer/src/icer_decoding.c:108-194
//   icer_decompress_bitplane_uint16/_8    icer_context_modeller.c:461-602 / :167-310
// with real branches, and its 64 lanes are used for what is data-parallel:
//   * register files indexed by lane: lane b holds bin b's pending bits / last-word index, lane k context k's counts, the
//     code-word tables and the bins' cut-offs sit in four vector registers -- v_readlane / v_writelane with a scalar
//     index, no LDS or scratch access on the decision chain;
//   * the payload: 64 dwords per vector load, taken one v_readlane at a time, the next 256 bytes loaded a chunk ahead;
//   * the context of 64 samples at a time: everything a sample's context needs except its LEFT neighbour's outcome in
//     this very plane depends on rows r - 1 (done), r + 1 and the right neighbour (the plane above, a row ahead): computed
//     by the 64 lanes in one go as two candidate contexts per sample (left insignificant / significant) and two sign
//     contexts (left not negative / negative); the serial part selects.
// Planes run a row (and up to 64 samples) behind the plane above -- the dependency rule of decoder_core.hpp, plane_needs --
// and meet only through progress counters in LDS; the lowest running plane writes finished rows back to the channel plane
// and recycles their ring slots.  Results are those of plane_decision / entropy_decode_fast (same arithmetic, same tables).
//
// Written with the SPMD macros of wave.hpp because this kernel has to be debugged in a container without a GPU: the same
// source runs in the CPU lane-loop build (tests/emu/decoder_emu.cpp, mode 4), where the nine waves are stepped round-robin
// (pw_step returns instead of waiting) and the images are compared with the decoder oracle.
#pragma once
#include "wave.hpp"

#include "decoder_core.hpp"

#ifdef ICER_WAVE_EMU
#define PW_WRITELANE(X, L, V) { (X)[(L)] = (V); }
#define PW_UNIFORM(x) (x)
#define PW_RCP(x) (1.0f / (x))
#define PW_UMIN(a, b) ((a) < (b) ? (a) : (b))
#define PW_SGPR(x)
#define PW_LDS_LOAD(x) (x)
#define PW_LDS_STORE(x, v) ((x) = (v))
#define PW_FENCE_ACQ()
#define PW_FENCE_REL()
#else
// v_writelane_b32: this clang has no builtin for it, but the LLVM intrinsic is reachable by name; the compiler then minds the
// M0 / lane-select hazards itself.  (The first versions used a compare of the lane number and a select: three vector
// instructions per write on gfx9 -- a second scalar operand does not fit a VALU instruction's constant bus, so the value went
// through a v_mov -- and, worse, the compiler then computed the value's whole uniform chain on the vector unit.)
// (a compiler that has the builtin uses it; either way `hipcc -S` of decoder.hip must show v_writelane_b32: 405 of them with ROCm 7.2, whose clang has no such builtin)
#if defined(__has_builtin) && __has_builtin(__builtin_amdgcn_writelane)
#define PW_WRITELANE(X, L, V) { (X) = (uint32_t)__builtin_amdgcn_writelane((int)(V), (int)(L), (int)(X)); }
#else
extern "C" __device__ int pw_writelane_i32(int value, int lane_select, int old) __asm("llvm.amdgcn.writelane.i32");
#define PW_WRITELANE(X, L, V) { (X) = (uint32_t)pw_writelane_i32((int)(V), (int)(L), (int)(X)); }
#endif
#define PW_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define PW_RCP(x) __builtin_amdgcn_rcpf(x)
// (a minimum of three uniform values: the instruction selector takes v_min3_u32 -- there is no scalar one -- and the pass that
// legalises the copy back then moves every scalar user of the result, loop counter included, onto the vector unit)
static __device__ __forceinline__ uint32_t pw_smin(uint32_t a, uint32_t b) { uint32_t r; asm("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
#define PW_UMIN(a, b) pw_smin((a), (b))
// "this value lives in a scalar register here" (no instruction): where the compiler would rather continue a uniform
// computation on the vector unit because its result ends up in a vector instruction anyway -- the decision chain then pays a
// vector instruction's 8 cycles per step and a v_readfirstlane hand-over where the scalar side needs the value again
#define PW_SGPR(x) asm volatile("" : "+s"(x))

#define PW_LDS_LOAD(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PW_LDS_STORE(x, v) __hip_atomic_store(&(x), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// LDS-only fences (lgkmcnt): the ring and the counters live in LDS; the row write-back to global memory needs no ordering
#define PW_FENCE_ACQ() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local")
#define PW_FENCE_REL() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local")
#endif

namespace icer {

constexpr uint32_t kPwBlock = 64;                       // samples per look at the neighbourhood
constexpr uint32_t kPwWaves = (uint32_t)kPlanes;        // wavefronts per workgroup (one per bit plane)
// Rows of the ring: the top plane holds rows r and r + 1, each plane below runs at least a row behind the one above, the
// lowest still reads its row - 1: planes + 2 live rows when every plane is stopped at a row boundary (which is where a
// stalled pipeline ends up, whatever the block size), so planes + 2 rows cannot lock up; two more keep the top plane from
// waiting for every retired row.
ICER_HD uint32_t pw_ring_rows(int planes) { return (uint32_t)planes + 4u; }
ICER_HD uint32_t pw_ring_pitch(uint32_t w) { return (w + 2u + 1u) & ~1u; }       // a zero guard column on either side
struct PwShared {                                       // control words of a chain (LDS)
    uint32_t done[12];                                  // samples finished by plane job j (0 = top plane)
    uint32_t retired;                                   // rows written back and recycled
    uint32_t pad[3];
};
// LDS of a chain: PwShared | one zero row (what lies above row 0 and below row h - 1) | the ring
ICER_HD size_t pw_lds_bytes(uint32_t w, int planes)
{
    return sizeof(PwShared) + (size_t)(pw_ring_rows(planes) + 1u) * pw_ring_pitch(w) * sizeof(uint16_t);
}

struct PlaneWave {
    // wave-uniform
    uint32_t j, nrun, lsb, w, h, pitch, rows, r, c, done, prev, sign_bit, mask, subband, retired;
    uint64_t win;
    uint32_t win_bits, words, pay_k, base, stream_len, bin_half;
    const uint8_t *stream;
    uint16_t *seg;                                      // the segment in the channel plane (global memory)
    size_t stride;
    // register files, one entry per lane
    LANEVAR(uint32_t, fst);                             // lane b: bits pending of bin b (int16) | pattern << 16
    LANEVAR(uint32_t, idx);                             // lane b: `words` when bin b's last code word was read
    LANEVAR(uint32_t, cnt);                             // lane k: context k's zero | total << 16
    LANEVAR(uint32_t, pay); LANEVAR(uint32_t, pay2);    // payload dwords 64 * chunk + lane of this chunk and the next
    LANEVAR(uint32_t, tg);                              // lane b: DecoderTables::gpk[b]
    LANEVAR(uint32_t, tv0); LANEVAR(uint32_t, tv1);     // DecoderTables::v2vlut as 112 dwords
    LANEVAR(uint32_t, tc);                              // lane k < 16: DecoderTables::cut[k]; the lanes above: 0xFFFFFFFF
};

// payload dwords 64 * chunk + lane (bytes behind the stream read as zero, like entropy_byte): one unconditional load from
// a clamped address and a shift -- NO per-lane branch (stream_len >= 4: a chain of this kernel has packets).
// (Per-lane branches are kept out of this file's scalar paths altogether: where such a branch re-joins, the compiler's
// uniformity analysis gives up on every value that merges there, and the decoder's state would leave the scalar unit.)
#define PW_LOAD_CHUNK(DST, CHUNK)                                                                          \
    FOR_LANES                                                                                              \
    {                                                                                                      \
        const uint32_t at_ = p.base + 4u * ((CHUNK) * 64u + (uint32_t)lane), last_ = p.stream_len - 4u;    \
        const uint32_t from_ = at_ < last_ ? at_ : last_, skip_ = at_ < last_ ? 0u : at_ - last_;          \
        uint32_t v_;                                                                                       \
        memcpy(&v_, p.stream + from_, 4);                                                                  \
        LV(DST) = skip_ >= 4u ? 0u : v_ >> (8u * skip_);                                                   \
    }

// icer_compute_bin (icer_util.c:48-56) of a FOLDED estimate (zero >= total / 2): the number of cut-offs that zero / total
// reaches, zero * 65536 >= total * cut[k] (cut-offs ascending; products below 2^25).  Lane k holds cut-off k, so all sixteen
// comparisons are ONE vector multiply and compare, and the bin is the population count of the ballot: four instructions,
// no branch, no table in memory (pick_bin_plain's division + look-up is the per-lane form of the same thing).
// (Lanes 16 and up hold 0xFFFFFFFF: total * that wraps to 2^32 - total, far above any zero << 16 <= 2^25 -- they never
// count, without a per-lane guard.)
#define PW_BIN(FZ, TOTAL) ((uint32_t)popc64(BALLOT((((FZ) << 16) >= (TOTAL) * LV(p.tc)))))

ICER_DEV void pw_init(PlaneWave &p, uint32_t j, uint32_t nrun, const ChainDesc &c, int planes, int sign_bit, uint16_t *plane,
                      size_t stride, const uint8_t *stream, uint32_t stream_len, const DecoderTables *t)
{
    DECL_LANE;
    // (every scalar of the wave's state is made uniform explicitly -- v_readfirstlane -- so that the decision chain below is
    // compiled for the scalar unit with real branches; a value the compiler cannot prove uniform would drag it all onto EXEC
    // masks)
    p.j = PW_UNIFORM(j); p.nrun = PW_UNIFORM(nrun); p.lsb = PW_UNIFORM((uint32_t)planes - 1u - j);
    p.w = PW_UNIFORM((uint32_t)c.w); p.h = PW_UNIFORM((uint32_t)c.h);
    p.pitch = pw_ring_pitch(p.w); p.rows = PW_UNIFORM(pw_ring_rows(planes));
    p.r = 0; p.c = 0; p.done = 0; p.prev = 0; p.sign_bit = PW_UNIFORM((uint32_t)sign_bit); p.mask = (1u << p.sign_bit) - 1u;
    p.subband = PW_UNIFORM(c.subband); p.retired = 0;
    p.stream = stream; p.stream_len = PW_UNIFORM(stream_len);
    p.base = PW_UNIFORM(c.pkt[p.lsb]) + (uint32_t)kHeaderBytes;
    p.win = 0; p.win_bits = 0; p.words = 0; p.pay_k = 0;
    p.seg = plane + PW_UNIFORM(c.first); p.stride = stride;
    FOR_LANES
    {
        LV(p.fst) = 0; LV(p.idx) = 0;
        LV(p.cnt) = 0;
        LV(p.tg) = t->gpk[lane < kNumBins ? lane : 0];                 // (clamped indices, not guarded loads: no per-lane branch)
        const uint16_t *v = &t->v2vlut[0][0];
        const int l1 = lane < 48 ? lane : 0;
        LV(p.tv0) = (uint32_t)v[2 * lane] | ((uint32_t)v[2 * lane + 1] << 16);
        LV(p.tv1) = (uint32_t)v[128 + 2 * l1] | ((uint32_t)v[128 + 2 * l1 + 1] << 16);
        const uint32_t cutv = t->cut[lane < 16 ? lane : 0];
        LV(p.tc) = lane < 16 ? cutv : 0xFFFFFFFFu;
    }
    // every context starts at zero = 2 of total = 4 (icer_init_context_model_vals, icer_context_modeller.c:607-613); an
    // unmodelled decision (category 3) presents 1 of 2
    FOR_LANES { LV(p.cnt) = 2u | (4u << 16); }
    p.bin_half = PW_BIN(1u, 2u);
    PW_LOAD_CHUNK(p.pay, 0u)
    PW_LOAD_CHUNK(p.pay2, 1u)
}

// a < b as a 0 / 1 word, for values below 2^31: the borrow of a - b.  (Written as a comparison the compiler keeps the truth value
// as a lane mask once it meets another one and turns it into a number with v_cndmask -- on the vector unit, where everything
// computed from it then stays: the served bit, the sample word, the counts.  As arithmetic the chain stays scalar.)
ICER_DEV uint32_t pw_lt(uint32_t a, uint32_t b) { uint32_t d = a - b; PW_SGPR(d); return d >> 31; }     // (the barrier: or the optimiser turns it back into a comparison)

// one bit from the entropy decoder for an event of coder bin `bin` (`inv`: the estimate was folded, the served bit is
// flipped): entropy_decode_fast (decoder_core.hpp) behind its bin selection, on wave-uniform values
ICER_DEV uint32_t pw_decode_bin(PlaneWave &p, uint32_t bin, uint32_t inv)
{
    DECL_LANE;
    const uint32_t st = READLANE(p.fst, bin), last_word = READLANE(p.idx, bin);
    int n = (int)(int16_t)(st & 0xFFFFu);
    uint32_t pat = st >> 16;
    if (n <= 0 || p.words - last_word >= (uint32_t)kRingWords) {
        if (p.win_bits < 32u) {
            const uint32_t nxt = READLANE(p.pay, p.pay_k & 63u);
            p.win |= (uint64_t)nxt << p.win_bits;
            p.win_bits += 32u;
            p.pay_k++;
            if ((p.pay_k & 63u) == 0u) {
                FOR_LANES { LV(p.pay) = LV(p.pay2); }
                const uint32_t chunk = (p.pay_k >> 6) + 1u;
                PW_LOAD_CHUNK(p.pay2, chunk)
            }
        }
        const uint32_t x = (uint32_t)p.win & 0x7FFu;
        uint32_t len;
        if (bin >= 8u) {
            const uint32_t g = READLANE(p.tg, bin), m = g & 0xFFFu, l = (g >> 12) & 15u, gi = g >> 16;
            const uint32_t rx = brev32(x);                                   // the low l / l + 1 bits, first bit on top
            const uint32_t k0 = rx >> (32u - l);                             // (l >= 3)
            const uint32_t k1 = ((rx >> (31u - l)) - gi) & 0xFFFFu;
            const bool full = (x & 1u) != 0u, shortw = k0 < gi;
            const uint32_t k = shortw ? k0 : k1;
            len = full ? 1u : (shortw ? l : l + 1u);
            pat = full ? 0u : 1u;
            n = full ? (int)m : (int)(1u + k > 32767u ? 32767u : 1u + k);
        } else if (bin >= 1u) {
            const uint32_t ei = (bin - 1u) * 32u + (x & 31u), dw = ei >> 1;
            const uint32_t two = dw < 64u ? READLANE(p.tv0, dw) : READLANE(p.tv1, dw - 64u);
            const uint32_t ent = (two >> (16u * (ei & 1u))) & 0xFFFFu;
            len = ent & 15u; n = (int)((ent >> 4) & 15u); pat = ent >> 8;
        } else { len = 1u; n = 1; pat = x & 1u; }
        p.win >>= len; p.win_bits -= len;
        p.words++;
        PW_WRITELANE(p.idx, bin, p.words);
    }
    uint32_t nm1 = (uint32_t)n;
    PW_SGPR(nm1);               // (hides that n - 1 and the test n > 0 below are one subtraction-with-borrow: that one the instruction selector puts on the vector unit, and every value behind it)
    nm1 -= 1u;
    const uint32_t top = (pat >> (nm1 & 31u)) & 1u;
    const uint32_t b = n > 0 ? (bin >= 8u ? (n == 1 ? pat : 0u) : top) : 0u;
    PW_WRITELANE(p.fst, bin, (nm1 & 0xFFFFu) | (pat << 16));
    return b ^ inv;
}

// Runs of zero decisions.  In the upper bit planes most samples are insignificant with insignificant neighbours: context 0,
// a near-certain zero, served from the pending zeros of a Golomb code word.  `rl` such samples lie ahead (the block's vector
// pass counted them).  As long as the events stay in one Golomb bin (the estimate only rises with every zero, so it is
// enough that the LAST event of the run still sees it), that bin has zeros pending (without its last-word rule coming
// due: no code word is read, so `words` stands still) and the counts stay below the rescale point, t decisions are
// t zeros and a few additions -- exactly what t single decisions of context 0 would leave.  Returns t (0: does not apply).
constexpr uint32_t kPwRunMin = 3;
#ifdef ICER_WAVE_EMU
static unsigned long long g_pw_run_stats[2];               // tests only: runs taken, decisions they stood for
#define PW_RUN_STAT(t) (g_pw_run_stats[0]++, g_pw_run_stats[1] += (t))
#else
#define PW_RUN_STAT(t)
#endif
ICER_DEV uint32_t pw_zero_run(PlaneWave &p, uint32_t rl)
{
    const uint32_t w = READLANE(p.cnt, 0u);
    const uint32_t zero = w & 0xFFFFu, total = w >> 16;
    if (zero < (total >> 1)) return 0u;                                // folded: a served 0 is a one-event
    const uint32_t bin = PW_BIN(zero, total);
    if (bin < 8u) return 0u;                                           // not a Golomb bin
    const uint32_t st = READLANE(p.fst, bin);
    const int n = (int)(int16_t)(st & 0xFFFFu);
    const uint32_t pat = st >> 16;
    if (n <= 0 || p.words - READLANE(p.idx, bin) >= (uint32_t)kRingWords) return 0u;
    const uint32_t avail = pat ? (uint32_t)n - 1u : (uint32_t)n;      // zeros above the closing one-bit / a full run of m zeros
    const uint32_t room = (kRescaleCap - 1u) - total;                 // events before the one that triggers the rescale
    uint32_t t = PW_UMIN(PW_UMIN(rl, avail), room);
    if (bin < 16u) {
        const uint32_t cut = READLANE(p.tc, bin);
        while (t >= 2u && ((zero + t - 1u) << 16) >= (total + t - 1u) * cut) t >>= 1;
    }
    if (t < 2u) return 0u;
    PW_WRITELANE(p.fst, bin, ((uint32_t)(n - (int)t) & 0xFFFFu) | (pat << 16));
    PW_WRITELANE(p.cnt, 0u, (zero + t) | ((total + t) << 16));
    PW_RUN_STAT(t);
    return t;
}

// samples of the plane above that must be finished before this plane takes the block ending at column c_end of row r
ICER_HD uint32_t pw_needs(uint32_t r, uint32_t c_end, uint32_t w, uint32_t h)
{
    if (r + 1u >= h) return w * h;
    return (r + 1u) * w + (c_end + 1u < w ? c_end + 1u : w - 1u) + 1u;
}

// write rows [p.retired, upto) of the ring back to the channel plane, zero their slots, publish
ICER_DEV void pw_retire(PlaneWave &p, PwShared &s, uint16_t *ring, uint32_t upto)
{
    DECL_LANE;
    while (p.retired < upto) {
        uint16_t *slot = ring + (size_t)(p.retired % p.rows) * p.pitch + 1u;
        for (uint32_t x0 = 0; x0 < p.w; x0 += 64u) {
            FOR_LANES
            {
                // (lanes past the row's end repeat its last sample: same value to the same address, and no per-lane branch)
                const uint32_t x = x0 + (uint32_t)lane < p.w ? x0 + (uint32_t)lane : p.w - 1u;
                p.seg[(size_t)p.retired * p.stride + x] = slot[x];
            }
        }
        WAVE_SYNC();
        for (uint32_t x0 = 0; x0 < p.w; x0 += 64u) {
            FOR_LANES
            {
                const uint32_t x = x0 + (uint32_t)lane < p.w ? x0 + (uint32_t)lane : p.w - 1u;
                slot[x] = 0;
            }
        }
        p.retired++;
    }
    WAVE_SYNC();
    PW_FENCE_REL();
    FOR_LANES { PW_LDS_STORE(s.retired, p.retired); }          // (every lane: the same word, the same value)
}

// One step of a plane's wave: the next block of up to 64 samples of its row, if the plane above and the ring allow it.
// Returns 0 = blocked (nothing done), 1 = progressed, 2 = the plane is finished.
// `zero_row`, `ring`: LDS (pw_lds_bytes); every wave of the chain passes the same ones.
// What the wave's next block waits for: a word of LDS that only grows, and the value it must have reached.
//   plane j > 0     the plane above must have decoded row r + 1 up to one column past the block's end (pw_needs)
//   the top plane   is the first to touch row r + 1: that row's ring slot must have been recycled, r + 1 < retired + rows
// (*need = 0: nothing to wait for.)
ICER_DEV const uint32_t *pw_wait_for(const PlaneWave &p, const PwShared &s, uint32_t *need)
{
    const uint32_t r = p.r, c0 = p.c, n = p.w - c0 < kPwBlock ? p.w - c0 : kPwBlock;
    if (p.j > 0u) { *need = pw_needs(r, c0 + n - 1u, p.w, p.h); return &s.done[p.j - 1u]; }
    *need = (c0 == 0u && r + 1u < p.h && r + 2u > p.rows) ? r + 2u - p.rows : 0u;
    return &s.retired;
}

ICER_DEV int pw_step(PlaneWave &p, PwShared &s, uint16_t *zero_row, uint16_t *ring)
{
    DECL_LANE;
    const uint32_t w = p.w, h = p.h;
    if (p.r >= h) return 2;
    const uint32_t r = p.r, c0 = p.c, n = w - c0 < kPwBlock ? w - c0 : kPwBlock, c_end = c0 + n - 1u;
    {
        uint32_t need;
        const uint32_t *word = pw_wait_for(p, s, &need);
        if (PW_UNIFORM(PW_LDS_LOAD(*word)) < need) return 0;
    }
    PW_FENCE_ACQ();
    const uint32_t lsb = p.lsb, lsb1 = lsb + 1u, mask = p.mask, sb = p.sign_bit;
    const uint16_t *rowU = r > 0u ? ring + (size_t)((r - 1u) % p.rows) * p.pitch : zero_row;
    uint16_t *rowC = ring + (size_t)(r % p.rows) * p.pitch;
    const uint16_t *rowD = r + 1u < h ? ring + (size_t)((r + 1u) % p.rows) * p.pitch : zero_row;
    // ---- 64 lanes: what the contexts of the block's samples need from the rows around them
    LANEVAR(uint32_t, desc); LANEVAR(uint32_t, curv); LANEVAR(uint32_t, outv); LANEVAR(uint32_t, elig);
    const bool is_hl = p.subband == (uint32_t)kHL, is_hh = p.subband == (uint32_t)kHH;
    FOR_LANES
    {
        const bool valid = (uint32_t)lane < n;
        const uint32_t b = valid ? c0 + (uint32_t)lane + 1u : 1u;                  // (guard column at index 0)
        const uint32_t ul = rowU[b - 1u], u = rowU[b], ur = rowU[b + 1u];
        const uint32_t cur = rowC[b], right = rowC[b + 1u];
        const uint32_t dl = rowD[b - 1u], d = rowD[b], dr = rowD[b + 1u];
#define PW_SIG(v, pl) ((((v) & mask) >> (pl)) != 0u ? 1u : 0u)
        const uint32_t su = PW_SIG(u, lsb), sul = PW_SIG(ul, lsb), sur = PW_SIG(ur, lsb);
        const uint32_t sd = PW_SIG(d, lsb1), sdl = PW_SIG(dl, lsb1), sdr = PW_SIG(dr, lsb1), sr = PW_SIG(right, lsb1);
#undef PW_SIG
        const uint32_t nu = su & (u >> sb), nd = sd & (d >> sb), nr = sr & (right >> sb);
        const uint32_t m = cur & mask;
        const int msb = 31 - clz32(m | 1u);
        int cat = msb < (int)lsb ? 0 : msb - (int)lsb;
        if (cat > 3) cat = 3;
        const uint32_t vv0 = su + sd, dd = sul + sur + sdl + sdr;
        uint32_t ctxs[2], ses[2];
        for (uint32_t left = 0; left < 2u; left++) {
            uint32_t hh = left + sr, vv = vv0;
            const bool any_hv = (left | sr | su | sd) != 0u;
            if (is_hl) { const uint32_t x = hh; hh = vv; vv = x; }
            const uint32_t ctx0 = is_hh ? dec_ctx_hh_packed(hh + vv, dd) : dec_ctx_plain_packed(hh, vv, dd);
            ctxs[left] = cat == 0 ? ctx0 : cat == 1 ? (any_hv ? 10u : 9u) : 11u;
            uint32_t sh = 2u - left - nr, sv = 2u - nu - nd;                       // (left: the left neighbour is negative)
            if (is_hl) { const uint32_t x = sh; sh = sv; sv = x; }
            constexpr uint64_t ksign = dec_pack_sign();
            ses[left] = (uint32_t)(ksign >> (4u * (sh * 3u + sv))) & 15u;
        }
        LV(desc) = (uint32_t)cat | (ctxs[0] << 2) | (ctxs[1] << 6) | (ses[0] << 10) | (ses[1] << 14);
        LV(curv) = cur;
        LV(outv) = 0;
        LV(elig) = (valid && cat == 0 && ctxs[0] == 0u) ? 1u : 0u;                 // insignificant among insignificant neighbours
    }
    // (bits 18..24 of a sample's descriptor: how many such samples lie ahead from it on, itself included -- pw_zero_run)
    const uint64_t E = BALLOT(LV(elig) != 0u);
    FOR_LANES
    {
        const uint64_t rest = ~(E >> lane);
        LV(desc) |= (uint32_t)(rest ? ffs64(rest) : 64 - lane) << 18;
    }
    // ---- the wave's scalar side: the block's decisions one after the other
    // (one decision per trip and ONE copy of the entropy decoder in the loop: a sample's magnitude bit, then -- when that made
    // it significant -- its sign in the next trip; the kernel's code stays small and the decision chain short)
    uint32_t prev = c0 == 0u ? 0u : p.prev;
    uint32_t val = 0, se = 0;
    bool sign_next = false;
    for (uint32_t i = 0; i < n;) {
        uint32_t w, ctx = 0, mod = 1u;                                             // mod: 1 = a modelled decision
        if (!sign_next) {
            const uint32_t de = READLANE(desc, i);
            const uint32_t cat = de & 3u;
            const uint32_t leftsig = ((prev & mask) >> lsb) != 0u ? 1u : 0u, leftneg = leftsig & (prev >> sb) & 1u;
            const uint32_t rl = (de >> 18) & 127u;
            if (rl >= kPwRunMin && leftsig == 0u) {
                const uint32_t t = pw_zero_run(p, rl);
                if (t) {                                                           // samples i .. i + t - 1 decode a zero: their words stay
                    FOR_LANES { if ((uint32_t)lane >= i && (uint32_t)lane < i + t) LV(outv) = LV(curv); }
                    prev = READLANE(curv, i + t - 1u);
                    i += t;
                    continue;
                }
            }
            val = READLANE(curv, i);
            mod = (6u - cat) >> 2;                                                 // category 3: unmodelled, 1 of 2 (C2) -- cat != 3 as a 0 / 1 word
            ctx = (de >> (leftsig ? 6u : 2u)) & 15u;
            // (the sign's context, should the sample become significant: QUIRK C6, only negative significant neighbours count)
            se = (cat == 0u ? 0x10u : 0u) | ((de >> (leftneg ? 14u : 10u)) & 15u);
        } else ctx = 12u + (se & 7u);
        // the estimate folded to >= 1/2, its bin, one bit from that bin's code words, the counts through dec_model_update
        // (QUIRK C5 included) -- selects, no branches, every step a scalar instruction (PW_FLAG)
        w = READLANE(p.cnt, ctx);
        uint32_t zero = w & 0xFFFFu, total = w >> 16;
        const uint32_t inv = pw_lt(zero, total >> 1) & mod;
        const uint32_t fz = zero ^ ((zero ^ (total - zero)) & (0u - inv));
        const uint32_t bin = (PW_BIN(fz, total) & (0u - mod)) | (p.bin_half & (mod - 1u));
        const uint32_t bit = pw_decode_bin(p, bin, inv);
        total++;
        zero += bit ^ 1u;
        const uint32_t resc = pw_lt(kRescaleCap - 1u, total);
        total >>= resc;
        zero >>= resc & pw_lt(total, zero);
        PW_WRITELANE(p.cnt, ctx | (63u & (mod - 1u)), zero | (total << 16));       // (unmodelled: lane 63, no context lives there)
        if (!sign_next) {
            val |= bit << lsb;
            if ((se & 0x10u) != 0u && bit != 0u) { sign_next = true; continue; }
        } else {
            val |= ((bit ^ (se >> 3)) & 1u) << sb;
            sign_next = false;
        }
        PW_WRITELANE(outv, i, val);
        prev = PW_UNIFORM(val);
        i++;
    }
    p.prev = prev;
    FOR_LANES
    {
        // (lanes past the block write a zero into the row's right guard column, which holds one anyway)
        const bool valid = (uint32_t)lane < n;
        rowC[valid ? c0 + (uint32_t)lane + 1u : w + 1u] = valid ? (uint16_t)LV(outv) : (uint16_t)0;
    }
    p.done += n;
    const bool row_end = c_end + 1u >= w;
    if (row_end) { p.r = r + 1u; p.c = 0; } else p.c = c0 + n;
    WAVE_SYNC();
    PW_FENCE_REL();
    FOR_LANES { PW_LDS_STORE(s.done[p.j], p.done); }           // (every lane: the same word, the same value)
    // the lowest running plane is the last one to look at a row: after its row r, row r - 1 is dead
    if (row_end && p.j + 1u == p.nrun) pw_retire(p, s, ring, p.r >= h ? h : (p.r >= 2u ? p.r - 1u : 0u));
    return p.r >= h ? 2 : 1;
}


// All planes of one chain: the body of decode_chains_planes_kernel.  GPU build: called by every wavefront of the workgroup
// (wave `wave` of kPwWaves; `lds` = pw_lds_bytes(c.w, planes) bytes, zeroed by the workgroup before); a wave that waits
// longer than kPwSpinLimit polls gives up and raises *err (the host then fails the call loudly) -- never a hang.
// CPU builds (tests only): one call runs the chain's waves in turns.  Returns false on a lock-up / time-out.
constexpr uint32_t kPwSpinLimit = 1u << 20;       // (polls; the late ones sleep 8 k cycles each: seconds)
#ifdef ICER_WAVE_EMU
ICER_DEV bool pw_run_chain(uint8_t *lds, uint32_t /*wave*/, const ChainDesc &c, int planes, int sign_bit, uint16_t *plane, size_t stride,
                           const uint8_t *stream, uint32_t stream_len, const DecoderTables *t, uint32_t * /*err*/)
{
    memset(lds, 0, pw_lds_bytes(c.w, planes));
    PwShared &sh = *reinterpret_cast<PwShared *>(lds);
    uint16_t *zero_row = reinterpret_cast<uint16_t *>(lds + sizeof(PwShared)), *ring = zero_row + pw_ring_pitch(c.w);
    uint32_t nrun = 0;
    while ((int)nrun < planes && c.pkt[planes - 1 - (int)nrun] != kNoPacket) nrun++;
    PlaneWave pw[kPlanes];
    for (uint32_t j = 0; j < nrun; j++) pw_init(pw[j], j, nrun, c, planes, sign_bit, plane, stride, stream, stream_len, t);
    for (;;) {
        bool progress = false, all_done = true;
        for (uint32_t j = 0; j < nrun; j++) {
            const int st = pw_step(pw[j], sh, zero_row, ring);
            progress = progress || st == 1;
            all_done = all_done && (st == 2 || pw[j].r >= pw[j].h);
        }
        if (all_done) return true;
        if (!progress) return false;
    }
}
#else
ICER_DEV bool pw_run_chain(uint8_t *lds, uint32_t wave, const ChainDesc &c, int planes, int sign_bit, uint16_t *plane, size_t stride,
                           const uint8_t *stream, uint32_t stream_len, const DecoderTables *t, uint32_t *err)
{
    // (`c` refers to global memory: a private copy indexed by a run-time plane number would live in scratch)
    PwShared &sh = *reinterpret_cast<PwShared *>(lds);
    uint16_t *zero_row = reinterpret_cast<uint16_t *>(lds + sizeof(PwShared)), *ring = zero_row + pw_ring_pitch(PW_UNIFORM((uint32_t)c.w));
    uint32_t nrun = 0;
    while ((int)nrun < planes && PW_UNIFORM(c.pkt[planes - 1 - (int)nrun]) != kNoPacket) nrun++;
    if (wave >= nrun) return true;
    PlaneWave p;
    pw_init(p, wave, nrun, c, planes, sign_bit, plane, stride, stream, stream_len, t);
    // (the loop must leave by wave-UNIFORM conditions only -- the lane-0 store of the error word comes after it: a per-lane
    // branch inside the loop would make every value the loop carries, i.e. the whole decoder state, divergent for the compiler)
    // A waiting wave must not eat the compute unit's ONE scalar unit: with two long chains on a compute unit that unit is
    // what they share (profiles/r04_logs/r04_l_planes_sq_counters.log: 0.6-0.75 scalar instructions per cycle and compute
    // unit, and the five upper planes of a chain mostly wait).  So the wait is a loop of its own over ONE word of LDS --
    // what the block needs is computed once (pw_wait_for), a poll is a load, a compare and a sleep -- and the sleep grows
    // with the time waited: 128 cycles for the first polls (a neighbour about to publish), then 1 k, then 8 k cycles.  A
    // long sleep costs a waiting wave nothing: it waits because it is the faster one of a pair and has a row of slack.
    bool ok = true;
    while (p.r < p.h) {
        uint32_t need;
        const uint32_t *word = pw_wait_for(p, sh, &need);
        uint32_t spins = 0;
        while (PW_UNIFORM(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need) {
            if (spins < 6u) __builtin_amdgcn_s_sleep(2);
            else if (spins < 12u) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(127);
            spins++;
            if (spins > kPwSpinLimit || ((spins & 63u) == 0u && PW_UNIFORM(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u)) { ok = false; break; }
        }
        if (!ok) break;
        if (pw_step(p, sh, zero_row, ring) == 0) { ok = false; break; }     // (cannot be: the condition just held and only grows)
    }
    if (!ok && (threadIdx.x & 63u) == 0u) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}
#endif

}  // namespace icer
