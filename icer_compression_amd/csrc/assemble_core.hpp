// assemble_core.hpp -- packet framing and stream assembly, one wavefront each.
//
//   finish_unit_wave   header + two CRC-32s of a coded unit
//                      (icer_allocate_data_packet lib_icer/src/icer_encoding.c:210-234,
//                       icer_calculate_segment_crc32 / _packet_crc32 icer_util.c:59-66,
//                       crc32buf crc32.c:157-169; layout icer.h:293-305)
//   scan_frame_wave    byte-quota walk in priority order + offsets in final stream order
//                      (icer_partition.c:315-336 quota accounting, icer_compress.c:404-423)
// Written with the SPMD macros of wave.hpp.
#pragma once
#include "coder_core.hpp"

// all wavefronts of the workgroup (the multi-wave forms of finish_unit_wave / splice_unit_wave; the CPU builds run one wave)
#ifdef ICER_WAVE_EMU
#define ICER_WG_BARRIER()
#else
#define ICER_WG_BARRIER() __syncthreads()
#endif

namespace icer {

constexpr uint32_t kCrcPoly = 0xEDB88320u;     // reflected CRC-32, init/final 0xFFFFFFFF (= zlib crc32)

// a(x) * b(x) mod P(x) in the reflected representation (bit 31 = x^0)
ICER_DEV uint32_t gf_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m != 0; m >>= 1) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0) break;
        }
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// x^(8*nbytes) mod P, from the table x2n[k] = x^(2^k) mod P (period 32 in k)
ICER_DEV uint32_t gf_xpow_bytes(const uint32_t *x2n, uint32_t nbytes)
{
    uint32_t p = 0x80000000u;                  // x^0
    for (uint32_t k = 3; nbytes != 0; nbytes >>= 1, k++)
        if (nbytes & 1u) p = gf_mulmod(x2n[k & 31u], p);
    return p;
}

template <class SharedT> ICER_DEV void build_crc_table(SharedT &s)
{
    DECL_LANE;
    FOR_LANES
    {
        for (uint32_t i = (uint32_t)lane; i < 256; i += 64) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
            s.crc_tab[i] = c;
        }
    }
    WAVE_SYNC();
}

struct FinishArgs {
    uint32_t *slot_words;       // slot start: 7 header words, then the payload words
    uint32_t bits;              // payload length in bits
    uint32_t mean, level, subband, seg, lsb, chan, image_w, image_h;
};

// CRC-32 of the payload: every lane takes a contiguous piece, the 64 piece CRCs are combined with
// crc(A||B) = crc(A) * x^(8|B|) + crc(B)  (valid for the init/final-xor form, cf. zlib crc32_combine)
// `nwv` > 1: called by all nwv wavefronts of a workgroup (wave `wv`), which share the payload; `wg_parts`: nwv words of LDS.
template <class SharedT> ICER_DEV void finish_unit_wave(SharedT &s, const FinishArgs &f, uint32_t wv = 0, uint32_t nwv = 1, uint32_t *wg_parts = nullptr)
{
    DECL_LANE;
    const uint32_t n = (f.bits + 7u) >> 3;
    const uint32_t nth = 64u * nwv;                               // threads sharing the payload
    const uint32_t piece = (((n + nth - 1u) / nth) + 15u) & ~15u;
    const uint32_t *payload = f.slot_words + kHeaderBytes / 4;
    const uint32_t nwords = (n + 3u) >> 2;                       // words that hold payload bytes (nothing behind them is read)
    LANEVAR(uint32_t, part);
    FOR_LANES
    {
        const uint32_t start = (wv * 64u + (uint32_t)lane) * piece;
        const uint32_t end = start + piece < n ? start + piece : n;
        uint32_t c = 0;
        if (start < end) {
            c = 0xFFFFFFFFu;
            // four words per step, the next four already on their way: the loop is a chain of table look-ups per byte and
            // would otherwise also wait for one load per word (a lane's piece is not in any cache: ~ 0.7 us each -- 190 us for
            // the 70 KB of a level-1 unit of the headline frame, on the unit's critical path)
            const uint32_t w0 = start >> 2;
            uint32_t q0 = payload[w0], q1 = w0 + 1u < nwords ? payload[w0 + 1u] : 0u, q2 = w0 + 2u < nwords ? payload[w0 + 2u] : 0u,
                     q3 = w0 + 3u < nwords ? payload[w0 + 3u] : 0u;
            for (uint32_t b = start; b < end; b += 16u) {
                uint32_t v[4] = {q0, q1, q2, q3};
                const uint32_t wn = (b >> 2) + 4u;
                if (b + 16u < end) {
                    q0 = payload[wn];
                    q1 = wn + 1u < nwords ? payload[wn + 1u] : 0u; q2 = wn + 2u < nwords ? payload[wn + 2u] : 0u; q3 = wn + 3u < nwords ? payload[wn + 3u] : 0u;
                }
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint32_t at = b + 4u * k;
                    if (at >= end) break;
                    uint32_t w = v[k];
                    const uint32_t nb = end - at < 4u ? end - at : 4u;
                    for (uint32_t j = 0; j < nb; j++, w >>= 8) c = s.crc_tab[(c ^ w) & 0xFFu] ^ (c >> 8);
                }
            }
            c = ~c;
            if (n > end) c = gf_mulmod(gf_xpow_bytes(s.tab.x2n, n - end), c);
        }
        LV(part) = c;
    }
    uint32_t data_crc;
    WAVE_XOR(data_crc, part);
    if (nwv > 1u) {
        FOR_LANES
        {
            if (lane == 0) wg_parts[wv] = data_crc;
        }
        ICER_WG_BARRIER();
        data_crc = 0;
        for (uint32_t i = 0; i < nwv; i++) data_crc ^= wg_parts[i];
    }
    FOR_LANES
    {
        if (lane == 0 && wv == 0u) {
            uint32_t hw[6];
            hw[0] = 0x605Bu | ((f.mean & 0xFFu) << 16);            // preamble, ll_mean (QUIRK D1: via uint8_t)
            hw[1] = f.level | (f.subband << 8) | (f.seg << 16) | ((f.lsb | (f.chan << 4)) << 24);
            hw[2] = f.image_w;
            hw[3] = f.image_h;
            hw[4] = f.bits;
            hw[5] = data_crc;
            uint32_t c = 0xFFFFFFFFu;
            for (int i = 0; i < 6; i++) {
                uint32_t w = hw[i];
                for (int j = 0; j < 4; j++, w >>= 8) c = s.crc_tab[(c ^ w) & 0xFFu] ^ (c >> 8);
                f.slot_words[i] = hw[i];
            }
            f.slot_words[6] = ~c;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sub-ranges (coder_core.hpp): the payload of a unit that was coded by several workgroups.
// Workgroup 0's bits are in the unit's own slot from bit 0; every workgroup that was matched into continues the
// payload from the bit position of the matched snapshot in ITS slot.  Appends those pieces (bit-granular copies) to the
// unit's slot and returns the payload length in bits, or kUnitTooBig / kUnitFailed as the coder reports them.
//   rec[i]          what workgroup i left (SubRecord)
//   snaps           [K][kMaxSnaps]
//   sub_words[i]    payload words of workgroup i's private slot (i >= 1), sub_words[0] = the unit's payload words
// One wavefront.
// ------------------------------------------------------------------------------------------
// `nwv` > 1: called by all nwv wavefronts of a workgroup (wave `wv`), which share the copies.
ICER_DEV uint32_t splice_unit_wave(uint32_t n_sub, const SubRecord *rec, const Snapshot *snaps, uint32_t *const *sub_words, uint32_t cap_words,
                                   uint32_t wv = 0, uint32_t nwv = 1)
{
    DECL_LANE;
    uint32_t *dst = sub_words[0];
    uint32_t cur = 0, out_bits = 0;
    // walk the chain: piece of workgroup `cur` = its bits [from, rec[cur].end_bits)
    uint32_t from = 0;
    for (uint32_t hop = 0; hop < n_sub; hop++) {
        const SubRecord r = rec[cur];
        if (!r.done) return kUnitFailed;
        if (r.end_bits == kUnitTooBig || r.end_bits == kUnitFailed) return r.end_bits;
        if (r.end_bits < from) return kUnitFailed;
        const uint32_t len = r.end_bits - from;
        if (((uint64_t)out_bits + len + 31u) / 32u > cap_words) return kUnitTooBig;
        if (cur != 0 && len) {
            // dst bits [out_bits, out_bits + len) = src bits [from, from + len)
            const uint32_t *src = sub_words[cur];
            const uint32_t w0 = out_bits >> 5, w1 = (out_bits + len + 31u) >> 5;         // dst words [w0, w1)
            const uint32_t keep = out_bits & 31u;                                        // bits of dst[w0] that are already there
            const uint32_t src_words = (r.end_bits + 31u) >> 5;
            FOR_LANES
            {
                // (two words per step and lane, their source words loaded before either is used: as in finish_unit_wave the loop
                // would otherwise wait for memory once per word)
                const uint32_t nth = 64u * nwv;
                for (uint32_t wb = w0 + wv * 64u + (uint32_t)lane; wb < w1; wb += 2u * nth) {
                    uint32_t lo[2] = {0u, 0u}, hi[2] = {0u, 0u}, sh[2] = {0u, 0u}, old0 = 0u;
                    bool neg[2] = {false, false};
                    for (uint32_t k = 0; k < 2u; k++) {
                        const uint32_t w = wb + nth * k;
                        if (w >= w1) break;
                        // source bit that lands on bit 0 of dst word w (negative for the first word when keep != 0)
                        const int64_t q = (int64_t)from + ((int64_t)w * 32 - (int64_t)out_bits);
                        if (q < 0) { neg[k] = true; lo[k] = src[0]; sh[k] = (uint32_t)(-q); }
                        else {
                            const uint32_t qi = (uint32_t)(q >> 5);
                            sh[k] = (uint32_t)q & 31u;
                            lo[k] = src[qi];
                            hi[k] = (sh[k] && qi + 1u < src_words) ? src[qi + 1u] : 0u;
                        }
                    }
                    if (wb == w0 && keep) old0 = dst[wb];
                    for (uint32_t k = 0; k < 2u; k++) {
                        const uint32_t w = wb + nth * k;
                        if (w >= w1) break;
                        uint32_t v = neg[k] ? lo[k] << sh[k] : (sh[k] ? (lo[k] >> sh[k]) | (hi[k] << (32u - sh[k])) : lo[k]);
                        // bits beyond the end of the piece are zero (so that the next piece can be OR-ed in)
                        const uint32_t endbit = out_bits + len;
                        if ((w + 1u) * 32u > endbit) v &= (endbit & 31u) ? ((1u << (endbit & 31u)) - 1u) : (w * 32u < endbit ? ~0u : 0u);
                        if (w == w0 && keep) v = (v & ~((1u << keep) - 1u)) | (old0 & ((1u << keep) - 1u));
                        dst[w] = v;
                    }
                }
            }
            WAVE_SYNC();
            if (nwv > 1u) ICER_WG_BARRIER();                                // (the next piece starts in the word this one ended in)
        }
        out_bits += len;
        if (r.match_sub == 0u) return out_bits;                             // this workgroup coded to the unit's end
        from = snaps[r.match_sub * kMaxSnaps + r.match_snap].bitpos;
        cur = r.match_sub;
    }
    return kUnitFailed;                                                     // (a chain longer than the unit has sub-ranges: corrupt records)
}

// ------------------------------------------------------------------------------------------
// Quota walk + final offsets for one frame.
//   bits[u]         payload bits of unit u (priority order), kUnitTooBig if it overflowed its slot
//   final_order[j]  unit index of the j-th unit in final stream order
//   final_off[u]    out: byte offset in the final stream, ~0 if the unit is dropped
// Returns kept-unit count through *kept, stream length through *size_used and the reference
// return code (0 or ICER_BYTE_QUOTA_EXCEEDED).
// Rule (P3): walking units in priority order with `used` bytes so far, a unit is kept iff
// 28 header bytes fit and floor(bits/8) < quota - used - 28; the first failing unit stops everything.
// ------------------------------------------------------------------------------------------
// index of the first unit (priority order) that does not fit the quota, n_units if all fit
ICER_DEV uint32_t quota_cut_wave(const uint32_t *bits, uint32_t n_units, uint64_t quota)
{
    DECL_LANE;
    uint64_t used = 0;
    uint32_t K = n_units;
    for (uint32_t base = 0; base < n_units && K == n_units; base += 64) {
        LANEVAR(uint64_t, sz); LANEVAR(uint64_t, before); LANEVAR(uint32_t, b);
        FOR_LANES
        {
            const uint32_t u = base + (uint32_t)lane;
            const uint32_t v = u < n_units ? bits[u] : 0u;
            LV(b) = v;
            LV(sz) = u < n_units ? (uint64_t)kHeaderBytes + (((uint64_t)v + 7u) >> 3) : 0u;
        }
        uint64_t total;
        WAVE_EXCL_SCAN(uint64_t, before, sz, total);
        const uint64_t fail = BALLOT(
            (base + (uint32_t)lane < n_units) &&
            ((LV(b) == kUnitTooBig) || (used + LV(before) + kHeaderBytes > quota) ||
             (LV(b) > 0u && (uint64_t)(LV(b) >> 3) + used + LV(before) + kHeaderBytes >= quota)));
        if (fail) K = base + (uint32_t)ffs64(fail);
        else used += total;
    }
    return K;
}

ICER_DEV int scan_frame_wave(const uint32_t *bits, const uint32_t *final_order, uint32_t n_units, uint64_t quota,
                             uint64_t *final_off, uint32_t *kept, uint64_t *size_used)
{
    DECL_LANE;
    const uint32_t K = quota_cut_wave(bits, n_units, quota);
    // final stream offsets
    uint64_t off = 0;
    for (uint32_t base = 0; base < n_units; base += 64) {
        LANEVAR(uint64_t, sz); LANEVAR(uint64_t, before); LANEVAR(uint32_t, unit);
        FOR_LANES
        {
            const uint32_t j = base + (uint32_t)lane;
            const uint32_t u = j < n_units ? final_order[j] : 0xFFFFFFFFu;
            LV(unit) = u;
            LV(sz) = (u < K) ? (uint64_t)kHeaderBytes + (((uint64_t)bits[u] + 7u) >> 3) : 0u;
        }
        uint64_t total;
        WAVE_EXCL_SCAN(uint64_t, before, sz, total);
        FOR_LANES
        {
            if (LV(unit) != 0xFFFFFFFFu) final_off[LV(unit)] = (LV(unit) < K) ? off + LV(before) : ~0ull;
        }
        off += total;
    }
    *kept = K;
    *size_used = off;
    return K < n_units ? kByteQuotaExceeded : kOk;
}

}  // namespace icer
