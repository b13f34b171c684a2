// decoder_core.hpp -- device side of the ICER *decoder* (SURVEY.md 8f, row next-1): the consumer of the streams the
// encoder hot path writes.  Plain per-thread code, shared between the gfx950 build (decoder.hip) and a g++ build
// for tests/emu (the authoring container has no GPU); the wave-level cooperation is in decoder_wave.hpp.
//
// STATUS: bit-exact against the decoder oracle on the CPU build (tests/test_emu_decoder.py) and on an MI355X
// (tests/test_gpu_decoder.py, default-on in the GPU gate); two decode kernels use it: one thread per chain (decoder.hip) and
// one wavefront per chain with a lane per bit plane (decoder_wave.hpp, the default).  Numbers: HISTORY.md 6.2 / 6b.  It is
// not part of libicer_hip.so.
//
// Restates, per segment ("chain": the bit planes of one segment of one subband of one channel, top plane first):
//   entropy decoder        icer_decode_bit + bit readers            lib_icer/src/icer_decoding.c:12-194
//   bit-plane decoder      icer_decompress_bitplane_uint16/_uint8   icer_context_modeller.c:461-602 / :167-310
//   plane loop             icer_decompress_partition_uint16/_uint8  icer_partition.c:427-443
// and per line of a level:
//   inverse lifting step   icer_inverse_wavelet_transform_1d_uint16/_uint8   icer_wavelet.c:467-550 / :298-383
// The quirks listed in HISTORY.md 6b (summary: DESIGN.md 8) (no end-of-packet check, one-code-word packets refused, "2048 code words ago"
// flush rule, filter C, the uint8 interleave of odd lines) are reproduced.
#pragma once
#include <stdint.h>
#include <string.h>

#include "icer_tables.hpp"

#ifndef ICER_HD
#if defined(__HIPCC__)
#define ICER_HD __host__ __device__ __forceinline__
#else
#define ICER_HD inline
#endif
#endif

namespace icer {

constexpr int kDecoderOutOfData = -7;      // ICER_DECODER_OUT_OF_DATA, icer.h:100
constexpr int kDecodedInvalidData = -8;    // ICER_DECODED_INVALID_DATA, icer.h:101
constexpr uint32_t kNoPacket = 0xFFFFFFFFu;

// tables of the entropy decoder (built on the host from the coder's tables, ~1.7 KiB; the decode kernels keep a copy in LDS)
struct DecoderTables {
    // bins 1..7: [bin][code word value] -> code bits | pattern bits << 4 | reversed source pattern << 8 (0 = no entry):
    // the inverse of CoderTables::v2v (icer_init.c:38-120; reversed so that the pattern pops in input order)
    uint16_t dec[8][32];
    uint16_t gm[17], gl[17], gi[17];       // Golomb parameters of bins 8..16
    uint32_t cut[16];                      // probability cut-offs x65536 between the bins (icer_config.c:69-87)
    uint32_t binlut[257];                  // CoderTables::binlut: the bin of floor(zero * 65536 / total) by one look-up
    // bins 1..7: [bin - 1][the next 5 bits of the payload] -> what the code-word search of icer_decode_bit (:154-172) arrives
    // at: bits consumed | pattern bits << 4 | reversed pattern << 8.  The reference's codes are complete prefix codes of at
    // most 5 bits, so the search never runs on to its other exits (ten bits without a match; a value >= 32:
    // ICER_DECODED_INVALID_DATA) -- build_decoder_tables checks that on all 1024 ten-bit inputs and clears `lut_ok`
    // otherwise, which sends every packet through the general routine.
    uint16_t v2vlut[7][32];
    uint32_t lut_ok;
    uint32_t gpk[17];                      // bins 8..16: gm | gl << 12 | gi << 16 (one look-up instead of three)
};

inline void build_decoder_tables(DecoderTables *d, const CoderTables &t)
{
    memset(d, 0, sizeof *d);
    for (int b = 1; b <= 7; b++)
        for (uint32_t v = 0; v < 32; v++) {
            const uint32_t e = t.v2v[b][v];
            if (!e) continue;
            const uint32_t nin = e & 15u, nout = (e >> 4) & 15u, code = e >> 8;
            uint32_t rev = 0;
            for (uint32_t k = 0; k < nin; k++) rev |= ((v >> k) & 1u) << (nin - 1u - k);
            d->dec[b][code] = (uint16_t)(nout | (nin << 4) | (rev << 8));
        }
    for (int b = 0; b < 17; b++) { d->gm[b] = t.gm[b]; d->gl[b] = t.gl[b]; d->gi[b] = t.gi[b]; }
    for (int k = 0; k < 16; k++) d->cut[k] = t.cut[k];
    for (int k = 0; k <= 256; k++) d->binlut[k] = t.binlut[k];
    for (int b = 0; b < 17; b++) d->gpk[b] = (uint32_t)d->gm[b] | ((uint32_t)d->gl[b] << 12) | ((uint32_t)d->gi[b] << 16);
    d->lut_ok = 1;
    for (int b = 8; b < 17; b++) if (d->gm[b] >= 4096u || d->gl[b] >= 16u) d->lut_ok = 0;
    for (int b = 1; b <= 7; b++)
        for (uint32_t x = 0; x < 1024u; x++) {
            uint32_t r = 0xFFFFFFFFu;                               // (no match in ten bits / invalid: not representable)
            for (uint32_t nb = 1; nb <= 10u; nb++) {
                const uint32_t code = x & ((1u << nb) - 1u);
                if (code >= 32u) break;
                const uint32_t e = d->dec[b][code];
                if ((e & 15u) == nb) { r = nb | (e & 0xFFF0u); break; }
            }
            if (x < 32u) d->v2vlut[b - 1][x] = (uint16_t)r;
            if (r == 0xFFFFFFFFu || (r & 15u) > 5u || r != d->v2vlut[b - 1][x & 31u]) d->lut_ok = 0;
        }
}

// one chain = one segment of one subband of one channel
struct ChainDesc {
    uint32_t frame;             // which stream / image of a batch
    uint32_t subband;           // kLL .. kHH: selects the context tables
    uint32_t chan;              // channel plane the segment lives in
    uint32_t first;             // index of the segment's first sample in that plane
    uint16_t w, h;              // segment size
    uint32_t pkt[kPlanes];      // per bit plane: byte offset of the packet (its header) in the stream, kNoPacket = absent
    uint32_t fast;              // every packet the chain runs has >= kFastPacketBits bits: none of the reference's length tests
                                // can fire, the chain may take the wave-per-plane kernel (decoder_planes.hpp)
};

// ------------------------------------------------------------------------------------------ entropy decoder
struct EntropyDecoder {
    const uint8_t *stream;      // the whole stream
    uint32_t stream_len;        // bytes behind it read as zero (the reference reads whatever lies there)
    uint32_t base;              // first payload byte of the packet
    uint32_t total_bits;        // data_length of the packet
    uint32_t pos;               // bit cursor relative to base
    uint32_t words;             // code words read so far
    uint64_t win;               // (packets of >= 8 bits) the next win_bits bits of the payload, from `pos` on
    uint32_t win_bits;
    uint32_t win_next;          // next payload byte to load into the window
    uint32_t ahead;             // (entropy_decode_fast) the four bytes at win_next, loaded one refill early
    // per-bin state, entry b at [b * ss]: in the thread's own memory (ss = 1) or a column of an LDS block (plane_attach_*)
    int16_t *n;                 // bits pending per bin, served from the top
    uint8_t *bits;              // bins 0..7: the pending pattern (bit k = k-th from the bottom); bins 8..16: bottom bit
    uint32_t *index;            // `words` when the bin's last code word was read
    uint32_t ss;
    uint32_t *fst;              // (entropy_decode_fast; plane_attach_chunk) per bin: bits pending (int16) | pattern << 16, entry b at [b]
};

ICER_HD void entropy_init(EntropyDecoder &d, const uint8_t *stream, uint32_t stream_len, uint32_t base, uint32_t total_bits)
{
    d.stream = stream; d.stream_len = stream_len; d.base = base; d.total_bits = total_bits; d.pos = 0; d.words = 0;
    d.win = 0; d.win_bits = 0; d.win_next = 0;
    for (uint32_t b = 0; b < (uint32_t)kNumBins; b++) { d.n[b * d.ss] = 0; d.bits[b * d.ss] = 0; d.index[b * d.ss] = 0; }
}
ICER_HD uint32_t entropy_byte(const EntropyDecoder &d, uint32_t i)
{
    const uint32_t at = d.base + i;
    return at < d.stream_len ? d.stream[at] : 0u;
}
// The bit readers below keep the reference's semantics (icer_get_bit_from_codeword :46-57, icer_get_bits_from_codeword
// :59-82, icer_pop_bits_from_codeword :84-105) but serve the bits from a 64-bit window that is refilled four bytes at a
// time.  QUIRK kept: the reference never advances its count of decoded bits, so its out-of-data test compares each
// byte-bounded piece of a read (at most 8 bits) with the packet's whole length -- it can only fire in packets shorter
// than 8 bits, and those take the literal path.
ICER_HD void entropy_fill(EntropyDecoder &d)               // afterwards the window holds at least 32 bits
{
    while (d.win_bits < 32u) {
        uint32_t v = 0;
        const uint32_t at = d.base + d.win_next;
        if (at + 4u <= d.stream_len) {                       // (the usual case: four bytes inside the stream, one after the other)
            const uint8_t *q = d.stream + at;
            v = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        } else
        for (uint32_t i = 0; i < 4u; i++) v |= entropy_byte(d, d.win_next + i) << (8u * i);
        d.win_next += 4u;
        d.win |= (uint64_t)v << d.win_bits;
        d.win_bits += 32u;
    }
}
// the k-th bit (1-based, k <= 11) ahead of the cursor
ICER_HD uint32_t entropy_peek(EntropyDecoder &d, uint32_t k)
{
    if (d.total_bits < 8u) {
        const uint32_t p = d.pos + (k - 1u);
        return (entropy_byte(d, p >> 3) >> (p & 7u)) & 1u;
    }
    entropy_fill(d);
    return (uint32_t)(d.win >> (k - 1u)) & 1u;
}
// `nb` (<= 11) bits LSB first
ICER_HD int entropy_read(EntropyDecoder &d, uint32_t nb, bool consume)
{
    if (d.total_bits < 8u) {                                // literal: byte-bounded pieces, each tested against the length
        int num = 0;
        uint32_t got = 0, p = d.pos;
        while (nb) {
            const uint32_t room = 8u - (p & 7u), take = room < nb ? room : nb;
            if (take > d.total_bits) return kDecoderOutOfData;
            num |= (int)(((entropy_byte(d, p >> 3) >> (p & 7u)) & ((1u << take) - 1u)) << got);
            nb -= take; got += take; p += take;
            if (consume) d.pos = p;
        }
        return num;
    }
    entropy_fill(d);
    const int num = (int)((uint32_t)d.win & ((1u << nb) - 1u));
    if (consume) { d.win >>= nb; d.win_bits -= nb; d.pos += nb; }
    return num;
}
ICER_HD uint32_t reverse_low_bits(uint32_t v, uint32_t n)            // icer_reverse_bits, icer.h:601-610 (16-bit)
{
#if defined(__clang__)
    return n ? (__builtin_bitreverse32(v) >> (32u - n)) & 0xFFFFu : 0u;
#else
    uint32_t r = 0;
    for (uint32_t k = 0; k < n; k++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r & 0xFFFFu;
#endif
}
// icer_compute_bin, icer_util.c:48-56: the highest bin whose cut-off the probability of a zero reaches.
// zero * 65536 >= total * cut  <=>  floor(zero * 65536 / total) >= cut, so one exact division (total <= 500: a float
// reciprocal estimate is within 1 of the quotient, then corrected) and one table look-up replace the search over the
// 16 cut-offs (the same scheme as the encoder's pick_bin; tests/test_tables.py and the decoder tests pin it).
ICER_HD int pick_bin_plain(const DecoderTables &t, uint32_t zero, uint32_t total)
{
    const uint32_t a = zero << 16;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)total));
#else
    uint32_t q = (uint32_t)((float)a * (1.0f / (float)total));
#endif
    int32_t rem = (int32_t)a - (int32_t)(q * total);
    if (rem < 0) { q--; rem += (int32_t)total; }
    if (rem < 0) { q--; rem += (int32_t)total; }
    if (rem >= (int32_t)total) { q++; rem -= (int32_t)total; }
    if (rem >= (int32_t)total) q++;
    const uint32_t e = t.binlut[q >> 8];
    return (int)((e & 255u) + (q >= (e >> 8) ? 1u : 0u));
}

// icer_decode_bit, icer_decoding.c:108-194
ICER_HD int entropy_decode(EntropyDecoder &d, const DecoderTables &t, uint32_t *bit, uint32_t zero, uint32_t total)
{
    bool inv = false;
    if (zero < (total >> 1)) { zero = total - zero; inv = true; }
    const int bin = pick_bin_plain(t, zero, total);
    const uint32_t at = (uint32_t)bin * d.ss;
    // a new code word is due when nothing is pending, or when kRingWords code words have been read since this bin's
    // last one: the encoder's ring was full then and it force-completed the bin's word (:128)
    if (d.n[at] <= 0 || d.words - d.index[at] >= (uint32_t)kRingWords) {
        d.n[at] = 0;
        d.bits[at] = 0;
        if (bin >= 8) {
            const uint32_t m = t.gm[bin], l = t.gl[bin], gi = t.gi[bin];
            if (entropy_peek(d, 1)) {                              // "1": a full run of m zeros
                entropy_read(d, 1, true);
                d.n[at] = (int16_t)m;
            } else {
                // (QUIRK: an out-of-data result is used as a number; only reachable with a packet shorter than one
                // code word, which no encoder writes)
                uint32_t k = reverse_low_bits((uint32_t)entropy_read(d, l, false) & 0xFFFFu, l);
                if (k < gi) {
                    entropy_read(d, l, true);
                } else {
                    k = reverse_low_bits((uint32_t)entropy_read(d, l + 1u, true) & 0xFFFFu, l + 1u);
                    k = (k - gi) & 0xFFFFu;
                }
                d.bits[at] = 1;                                   // a one at the bottom, k zeros on top of it
                const uint32_t cnt = 1u + k;
                d.n[at] = (int16_t)(cnt > 32767u ? 32767u : cnt);
            }
        } else if (bin >= 1) {
            uint32_t code = 0, nb = 0;
            do {
                // QUIRK (:159): with the count never advanced this refuses exactly the packets that are no longer than
                // the code word being read
                if (nb + 1u >= d.total_bits) return kDecoderOutOfData;
                code |= entropy_peek(d, nb + 1u) << nb;
                nb++;
                if (code >= 32u) return kDecodedInvalidData;
                const uint32_t e = t.dec[bin][code];
                if ((e & 15u) == nb) {
                    d.bits[at] = (uint8_t)(e >> 8);
                    d.n[at] = (int16_t)((e >> 4) & 15u);
                    if ((int)code != entropy_read(d, nb, true)) return kDecodedInvalidData;
                    break;
                }
            } while (nb < 10u);
        } else {
            const int b = entropy_read(d, 1, true);
            if (b == kDecoderOutOfData) return kDecoderOutOfData;
            d.bits[at] = (uint8_t)(b != 0);
            d.n[at] = 1;
        }
        d.words++;
        d.index[at] = d.words;
    }
    // serve the top pending bit (:186-190).  Golomb bins hold zeros above one bottom bit; the reference's shift by -1 at
    // multiples of 32 pending bits reads a zero, which is what lies there; after a code word that matched nothing the
    // count goes to -1 and a zero is served.
    uint32_t b = 0;
    const int n = d.n[at];
    if (n > 0) {
        if (bin >= 8) b = (n == 1) ? d.bits[at] : 0u;
        else b = (d.bits[at] >> (n - 1)) & 1u;
    }
    d.n[at] = (int16_t)(n - 1);
    *bit = inv ? (b ^ 1u) : b;
    return kOk;
}

// ---- the same for packets of >= kFastPacketBits bits, where none of the reference's length tests can fire (they compare
// at most 11 with the packet's length): the code word of bins 1..7 by ONE look-up of the next five bits (v2vlut), the
// window refilled from a word that was loaded a refill earlier (its latency is off the decoding chain), and no loop -- the
// lanes of a wavefront that decode side by side stay together.  State and results are those of entropy_decode.
constexpr uint32_t kFastPacketBits = 16;
// Four payload bytes from byte i on, bytes behind the stream reading as zero -- as ONE unconditional load (from a clamped
// address; needs stream_len >= 4, which a packet of kFastPacketBits bits implies) whose result is only touched when the
// window takes it in, a refill later (entropy_ahead): no branch and no wait at the load.
ICER_HD uint32_t entropy_load4(const EntropyDecoder &d, uint32_t i)
{
    const uint32_t at = d.base + i, last = d.stream_len - 4u, from = at < last ? at : last;
    uint32_t v;
    memcpy(&v, d.stream + from, 4);
    return v;
}
ICER_HD uint32_t entropy_ahead(const EntropyDecoder &d)         // the bytes at win_next, from the word loaded for them
{
    const uint32_t at = d.base + d.win_next, last = d.stream_len - 4u, skip = at < last ? 0u : at - last;
    return skip >= 4u ? 0u : d.ahead >> (8u * skip);
}
ICER_HD void entropy_fast_begin(EntropyDecoder &d)                  // after entropy_init, storage from plane_attach_chunk
{
    d.ahead = entropy_load4(d, 0);
    for (uint32_t b = 0; b < (uint32_t)kNumBins; b++) { d.fst[b] = 0; d.index[b] = 0; }
}
ICER_HD int entropy_decode_fast(EntropyDecoder &d, const DecoderTables &t, uint32_t *bit, uint32_t zero, uint32_t total)
{
    const bool inv = zero < (total >> 1);
    if (inv) zero = total - zero;
    const int bin = pick_bin_plain(t, zero, total);
    const uint32_t st = d.fst[bin], last_word = d.index[bin];
    int n = (int)(int16_t)(st & 0xFFFFu);
    uint32_t pat = st >> 16;
    if ((n <= 0) | (d.words - last_word >= (uint32_t)kRingWords)) {
        if (d.win_bits < 32u) {
            d.win |= (uint64_t)entropy_ahead(d) << d.win_bits;
            d.win_bits += 32u;
            d.win_next += 4u;
            d.ahead = entropy_load4(d, d.win_next);
        }
        const uint32_t x = (uint32_t)d.win & 0x7FFu;
        uint32_t len;
        if (bin >= 8) {
            const uint32_t g = t.gpk[bin], m = g & 0xFFFu, l = (g >> 12) & 15u, gi = g >> 16;
#if defined(__clang__)
            const uint32_t rx = __builtin_bitreverse32(x);                    // the low l / l + 1 bits, first bit on top
#else
            uint32_t rx = 0;
            for (uint32_t k = 0; k < 32u; k++) rx |= ((x >> k) & 1u) << (31u - k);
#endif
            const uint32_t k0 = rx >> (32u - l);                              // (l >= 3)
            const uint32_t k1 = ((rx >> (31u - l)) - gi) & 0xFFFFu;
            const bool full = (x & 1u) != 0, shortw = k0 < gi;
            const uint32_t k = shortw ? k0 : k1;
            len = full ? 1u : (shortw ? l : l + 1u);
            pat = full ? 0u : 1u;
            n = full ? (int)m : (int)(1u + k > 32767u ? 32767u : 1u + k);
        } else if (bin >= 1) {
            const uint32_t e = t.v2vlut[bin - 1][x & 31u];
            len = e & 15u; n = (int)((e >> 4) & 15u); pat = e >> 8;
        } else { len = 1u; n = 1; pat = x & 1u; }
        d.win >>= len; d.win_bits -= len;
        d.words++;
        d.index[bin] = d.words;
    }
    // (selects, no branch: the shift count is masked so that it is defined for n <= 0 too, where a zero is served)
    const uint32_t top = (pat >> ((uint32_t)(n - 1) & 31u)) & 1u;
    const uint32_t b = n > 0 ? (bin >= 8 ? (n == 1 ? pat : 0u) : top) : 0u;
    d.fst[bin] = ((uint32_t)(n - 1) & 0xFFFFu) | (pat << 16);
    *bit = inv ? (b ^ 1u) : b;
    return kOk;
}

// ------------------------------------------------------------------------------------------ bit-plane decoder
// context tables (icer_config.c:26-67)
constexpr ICER_HD int dec_ctx_plain(int h, int v, int d)
{
    if (h == 2) return 8;
    if (h == 1) return v == 0 ? (d == 0 ? 5 : d == 1 ? 6 : 7) : 7;
    if (v == 0) return d > 2 ? 2 : d;
    return v == 1 ? 3 : 4;
}
constexpr ICER_HD int dec_ctx_hh(int hv, int d)
{
    if (d >= 3) return 8;
    if (d == 0) return hv > 2 ? 2 : hv;
    if (d == 1) return 3 + (hv > 2 ? 2 : hv);
    return hv == 0 ? 6 : 7;
}
constexpr ICER_HD int dec_sign_ctx(int sh, int sv)      // icer_sign_context_table
{
    if (sh == 2) return sv == 2 ? 12 : 13;
    if (sv == 2) return 15;
    return ((sh < 2) == (sv < 2)) ? 14 : 16;
}
constexpr ICER_HD int dec_sign_pred(int sh, int sv)     // icer_sign_prediction_table
{
    if (sh < 2) return 1;
    if (sh == 2) return sv > 2 ? 1 : 0;
    return 0;
}
// the same tables as nibbles of 64-bit constants (made from the functions above at compile time), for code that must not
// branch: entry i of table word(s) K is (K[i / 16] >> 4 * (i % 16)) & 15
constexpr uint64_t dec_pack_plain(int word)
{
    uint64_t k = 0;
    for (int i = word * 16; i < word * 16 + 16 && i < 45; i++) k |= (uint64_t)dec_ctx_plain(i / 15, (i / 5) % 3, i % 5) << (4 * (i % 16));
    return k;
}
constexpr uint64_t dec_pack_hh(int word)
{
    uint64_t k = 0;
    for (int i = word * 16; i < word * 16 + 16 && i < 25; i++) k |= (uint64_t)dec_ctx_hh(i / 5, i % 5) << (4 * (i % 16));
    return k;
}
constexpr uint64_t dec_pack_sign()          // per (sh, sv): sign context - 12, predicted sign << 3
{
    uint64_t k = 0;
    for (int i = 0; i < 9; i++) k |= (uint64_t)((dec_sign_ctx(i / 3, i % 3) - 12) | (dec_sign_pred(i / 3, i % 3) << 3)) << (4 * i);
    return k;
}
ICER_HD uint32_t dec_ctx_plain_packed(uint32_t h, uint32_t v, uint32_t d)      // h, v <= 2, d <= 4
{
    constexpr uint64_t k0 = dec_pack_plain(0), k1 = dec_pack_plain(1), k2 = dec_pack_plain(2);
    const uint32_t i = h * 15u + v * 5u + d;
    const uint64_t k = i < 16u ? k0 : i < 32u ? k1 : k2;
    return (uint32_t)(k >> (4u * (i & 15u))) & 15u;
}
ICER_HD uint32_t dec_ctx_hh_packed(uint32_t hv, uint32_t d)                   // hv, d <= 4
{
    constexpr uint64_t k0 = dec_pack_hh(0), k1 = dec_pack_hh(1);
    const uint32_t i = hv * 5u + d;
    return (uint32_t)((i < 16u ? k0 : k1) >> (4u * (i & 15u))) & 15u;
}
ICER_HD void dec_model_update(uint16_t &zero, uint16_t &total, bool was_zero)
{
    // icer_context_modeller.c:546-552 (QUIRK C5: a halved zero count is computed and dropped when zero <= total)
    total++;
    zero = (uint16_t)(zero + (was_zero ? 1 : 0));
    if (total >= kRescaleCap) {
        total >>= 1;
        if (zero > total) zero >>= 1;
    }
}

// One bit plane of one segment as a resumable job: plane_step decodes the next sample (row-major) and adds its bit of
// plane `lsb` -- and its sign, when the sample becomes significant -- to the sign-magnitude word (sign at `sign_bit`).
struct PlaneDecoder {
    EntropyDecoder d;
    uint16_t *zero, *total;     // context model (icer_init_context_model_vals :607-613), entry k at [k * d.ss]
    uint32_t r, c;              // next sample
    uint32_t left;              // the sample to the left, this plane already decoded
    uint32_t done;              // samples finished (= r * w + c)
    int lsb;
    int status;                 // 1 = running, 0 = finished (kOk), < 0 = failed with that code, 2 = not started
    // (plane_decision) a sample that became significant and whose sign is the next decision: its word, sign context and
    // predicted sign
    uint32_t pend, pval, psctx, ppred;
    // (plane_decision) the neighbourhood slides along the row in registers: significant / negative flags of the words
    // around the next sample, that sample's word, and where the three rows start in the image
    uint32_t nf, ncur, ou, oc, od;
    uint32_t *fzt;              // (plane_decision; plane_attach_chunk) context k: zero | total << 16 at [k]
};

// where a plane job keeps its per-bin / per-context arrays: its own memory ...
struct PlaneStorage {
    uint32_t index[kNumBins];
    int16_t n[kNumBins];
    uint16_t zero[kNumContexts], total[kNumContexts];
    uint8_t bits[kNumBins];
};
ICER_HD void plane_attach_local(PlaneDecoder &p, PlaneStorage &st)
{
    p.d.n = st.n; p.d.bits = st.bits; p.d.index = st.index; p.zero = st.zero; p.total = st.total; p.d.ss = 1;
}
// ... or column `col` of a block shared by `columns` jobs (LDS: dynamic indexing there beats the scratch memory that
// per-thread arrays indexed by a run-time bin number end up in).  Block layout: index, n, zero, total, bits.
ICER_HD size_t plane_block_bytes(uint32_t columns) { return (size_t)columns * (kNumBins * 7u + kNumContexts * 4u); }
ICER_HD void plane_attach_columns(PlaneDecoder &p, uint8_t *block, uint32_t columns, uint32_t col)
{
    uint8_t *q = block;
    p.d.index = reinterpret_cast<uint32_t *>(q) + col; q += (size_t)columns * kNumBins * 4u;
    p.d.n = reinterpret_cast<int16_t *>(q) + col;      q += (size_t)columns * kNumBins * 2u;
    p.zero = reinterpret_cast<uint16_t *>(q) + col;    q += (size_t)columns * kNumContexts * 2u;
    p.total = reinterpret_cast<uint16_t *>(q) + col;   q += (size_t)columns * kNumContexts * 2u;
    p.d.bits = q + col;
    p.d.ss = columns;
}

// ... or a chunk of its own (kPlaneChunkBytes, 4-byte aligned; an odd number of words apart, so the same entry of
// different lanes lies in different LDS banks) that holds either the arrays above or the packed ones of plane_decision /
// entropy_decode_fast -- a plane uses one form from its first sample to its last.
constexpr uint32_t kPlaneChunkBytes = 53u * 4u;
ICER_HD void plane_attach_chunk(PlaneDecoder &p, uint8_t *chunk)
{
    p.d.index = reinterpret_cast<uint32_t *>(chunk);                         // 17 x 4 (both forms)
    p.d.n = reinterpret_cast<int16_t *>(chunk + 68);                         // 17 x 2
    p.zero = reinterpret_cast<uint16_t *>(chunk + 102);                      // 17 x 2
    p.total = reinterpret_cast<uint16_t *>(chunk + 136);                     // 17 x 2
    p.d.bits = chunk + 170;                                                  // 17
    p.d.ss = 1;
    p.d.fst = reinterpret_cast<uint32_t *>(chunk + 68);                      // 17 x 4 (packed form)
    p.fzt = reinterpret_cast<uint32_t *>(chunk + 136);                       // 17 x 4
}
// after plane_begin, for a plane that takes plane_decision
ICER_HD void plane_fast_begin(PlaneDecoder &p)
{
    entropy_fast_begin(p.d);
    for (uint32_t k = 0; k < (uint32_t)kNumContexts; k++) p.fzt[k] = 2u | (4u << 16);
}

// (attach the storage, then entropy_init, then plane_begin)
ICER_HD void plane_begin(PlaneDecoder &p, int lsb, int sign_bit, uint32_t w, uint32_t h)
{
    for (uint32_t k = 0; k < (uint32_t)kNumContexts; k++) { p.zero[k * p.d.ss] = 2; p.total[k * p.d.ss] = 4; }
    p.r = 0; p.c = 0; p.left = 0; p.done = 0; p.lsb = lsb; p.pend = 0; p.pval = 0; p.psctx = 0; p.ppred = 0;
    p.status = (lsb + 1 >= sign_bit + 1) ? kBitplaneOutOfRange : ((w == 0 || h == 0) ? kOk : 1);
}

// one sample (icer_context_modeller.c:495-598); sets p.status when the plane ends or fails
// `img`: the segment's samples, at(r, c) / put(r, c, v) (global memory, or a ring of rows in LDS)
template <class Img>
ICER_HD void plane_step_img(PlaneDecoder &p, Img &img, uint32_t w, uint32_t h, int subband, int sign_bit, const DecoderTables &t)
{
    const int lsb = p.lsb;
    const uint32_t mask = (1u << sign_bit) - 1u;
    const uint32_t r = p.r, c = p.c;
    const bool up = r > 0, dn = r + 1 < h;
    const uint32_t left = p.left;
    const uint32_t cur = img.at(r, c);
    const uint32_t m = cur & mask;
    const int msb = 31 - __builtin_clz(m | 1u);
    int cat = msb < lsb ? 0 : msb - lsb;
    if (cat > 3) cat = 3;
    uint32_t bit, val;
    int res;
    if (cat == 3) {
        if ((res = entropy_decode(p.d, t, &bit, 1, 2)) != kOk) { p.status = res; return; }
        val = cur | (bit << lsb);
    } else {
        const bool has_r = c + 1 < w;
        const uint32_t right = has_r ? img.at(r, c + 1) : 0u;
        const uint32_t u0 = up ? img.at(r - 1, c) : 0u, d0 = dn ? img.at(r + 1, c) : 0u;
        int ctx;
        if (cat == 2) ctx = 11;
        else {
            // neighbours already visited count at this plane, the others at the plane above (:509-521)
            const uint32_t ul = (up && c > 0) ? img.at(r - 1, c - 1) : 0u, ur = (up && has_r) ? img.at(r - 1, c + 1) : 0u;
            const uint32_t dl = (dn && c > 0) ? img.at(r + 1, c - 1) : 0u, dr = (dn && has_r) ? img.at(r + 1, c + 1) : 0u;
            int hh = (c > 0 && ((left & mask) >> lsb)) + (((right & mask) >> (lsb + 1)) != 0);
            int vv = (((u0 & mask) >> lsb) != 0) + (((d0 & mask) >> (lsb + 1)) != 0);
            const int dd = (((ul & mask) >> lsb) != 0) + (((ur & mask) >> lsb) != 0) +
                           (((dl & mask) >> (lsb + 1)) != 0) + (((dr & mask) >> (lsb + 1)) != 0);
            if (cat == 1) ctx = (hh + vv == 0) ? 9 : 10;
            else {
                if (subband == kHL) { const int x = hh; hh = vv; vv = x; }
                ctx = subband == kHH ? dec_ctx_hh(hh + vv, dd) : dec_ctx_plain(hh, vv, dd);
            }
        }
        if ((res = entropy_decode(p.d, t, &bit, p.zero[(uint32_t)ctx * p.d.ss], p.total[(uint32_t)ctx * p.d.ss])) != kOk) { p.status = res; return; }
        val = cur | (bit << lsb);
        dec_model_update(p.zero[(uint32_t)ctx * p.d.ss], p.total[(uint32_t)ctx * p.d.ss], bit == 0);
        if (cat == 0 && bit) {
            // sign: only negative significant neighbours count (QUIRK C6)
            auto sgn = [&](uint32_t v, int plane) { return (((v & mask) >> plane) != 0 && ((v >> sign_bit) & 1u)) ? -1 : 0; };
            int sh = (c > 0 ? sgn(left, lsb) : 0) + sgn(right, lsb + 1) + 2;
            int sv = sgn(u0, lsb) + sgn(d0, lsb + 1) + 2;
            if (subband == kHL) { const int x = sh; sh = sv; sv = x; }
            const int sctx = dec_sign_ctx(sh, sv);
            uint32_t agree;
            if ((res = entropy_decode(p.d, t, &agree, p.zero[(uint32_t)sctx * p.d.ss], p.total[(uint32_t)sctx * p.d.ss])) != kOk) { img.put(r, c, val); p.status = res; return; }
            val |= ((agree ^ (uint32_t)dec_sign_pred(sh, sv)) & 1u) << sign_bit;
            dec_model_update(p.zero[(uint32_t)sctx * p.d.ss], p.total[(uint32_t)sctx * p.d.ss], agree == 0);
        }
    }
    img.put(r, c, val);
    p.done++;
    if (c + 1 < w) { p.c = c + 1; p.left = val; }
    else { p.c = 0; p.left = 0; p.r = r + 1; if (r + 1 >= h) p.status = kOk; }
}

// ---- one DECISION per call instead of one sample: the sign of a sample that just became significant is left for the
// next call.  Lanes of a wavefront that decode planes side by side (decoder_wave.hpp) then run the same instructions call
// after call -- one context, one entropy_decode_fast, one model update -- instead of taking turns through the branches
// of plane_step_img; everything a sample reads is loaded up front (clamped addresses, masked by validity) and the
// context is selected without branches.  Packets shorter than kFastPacketBits take plane_step_img.
template <class Img>
ICER_HD void plane_decision(PlaneDecoder &p, Img &img, uint32_t w, uint32_t h, int subband, int sign_bit, const DecoderTables &t)
{
    if (p.d.total_bits < kFastPacketBits || !t.lut_ok) { plane_step_img(p, img, w, h, subband, sign_bit, t); return; }
    const int lsb = p.lsb;
    const uint32_t mask = (1u << sign_bit) - 1u;
    const uint32_t r = p.r, c = p.c;
    uint32_t cur = p.pval, ctx = p.psctx, sctx = p.psctx, pred = p.ppred;
    int cat = 0;
    const bool magnitude = p.pend == 0;
    if (magnitude) {
        const bool up = r > 0, dn = r + 1 < h, hr = c + 1 < w;
        const int lsb1 = lsb + 1;
        // what the context needs of a neighbour: is it significant (at this plane if already visited, else at the plane
        // above), and if so is it negative -- two flags, taken once when the word enters the neighbourhood
        const auto flags_of = [mask, sign_bit](uint32_t v, int plane) {
            const uint32_t sg = ((v & mask) >> plane) != 0u ? 1u : 0u;
            return sg | ((sg & (v >> sign_bit)) << 8);
        };
        if (c == 0) {                              // a new row: where its three rows lie, and their first column
            // (row r - 1 and row r are where the last row had its rows r and r + 1)
            if (r == 0) { p.oc = img.row_at(0); p.ou = p.oc; } else { p.ou = p.oc; p.oc = p.od; }
            p.od = dn ? img.row_after(p.oc) : p.oc;
            const uint32_t u0 = up ? img.at_row(p.ou, 0) : 0u, d0 = dn ? img.at_row(p.od, 0) : 0u;
            p.ncur = img.at_row(p.oc, 0);
            p.nf = (flags_of(u0, lsb) << 1) | (flags_of(d0, lsb1) << 4);
        }
        // three new words per sample (the column to the right); nothing that is read here can still change in the bits
        // this plane looks at: the planes above are past it, the planes below only add lower bits -- and signs of
        // samples that are insignificant from this plane up, which are not looked at
        const uint32_t cr = hr ? c + 1u : c;
        const uint32_t ur0 = img.at_row(p.ou, cr), right0 = img.at_row(p.oc, cr), dr0 = img.at_row(p.od, cr);
        const uint32_t ur = (up && hr) ? ur0 : 0u, right = hr ? right0 : 0u, dr = (dn && hr) ? dr0 : 0u;
        // flag bits: 0..2 above-left / above / above-right, 3..5 the same below, 6 left, 7 right; + 8: negative
        const uint32_t f = p.nf | (flags_of(ur, lsb) << 2) | (flags_of(dr, lsb1) << 5) | (flags_of(right, lsb1) << 7);
        cur = p.ncur;
        p.nf = (f >> 1) & 0x1B1Bu;                 // the window moves on: above -> above-left, above-right -> above, ...
        p.ncur = right;
        const uint32_t m = cur & mask;
        const int msb = 31 - __builtin_clz(m | 1u);
        cat = msb < lsb ? 0 : msb - lsb;
        if (cat > 3) cat = 3;
        // significance at this plane (visited) / the plane above (not yet visited), icer_context_modeller.c:509-521;
        // sign context: only negative significant neighbours count (QUIRK C6)
        uint32_t hh = (uint32_t)__builtin_popcount(f & 0x00C0u), vv = (uint32_t)__builtin_popcount(f & 0x0012u);
        const uint32_t dd = (uint32_t)__builtin_popcount(f & 0x002Du);
        uint32_t sh = 2u - (uint32_t)__builtin_popcount(f & 0xC000u), sv = 2u - (uint32_t)__builtin_popcount(f & 0x1200u);
        const bool any_hv = (f & 0x00D2u) != 0u;
        if (subband == kHL) { uint32_t x = hh; hh = vv; vv = x; x = sh; sh = sv; sv = x; }
        const uint32_t c0 = subband == kHH ? dec_ctx_hh_packed(hh + vv, dd) : dec_ctx_plain_packed(hh, vv, dd);
        ctx = cat == 0 ? c0 : cat == 1 ? (any_hv ? 10u : 9u) : 11u;
        constexpr uint64_t ksign = dec_pack_sign();
        const uint32_t se = (uint32_t)(ksign >> (4u * (sh * 3u + sv))) & 15u;
        sctx = 12u + (se & 7u);
        pred = se >> 3;
    }
    // (no branches around the model: an unmodelled decision -- category 3 -- reads and writes back context 11's counts)
    const bool modelled = !(magnitude && cat == 3);
    const uint32_t zt = p.fzt[ctx];
    uint16_t zero = modelled ? (uint16_t)(zt & 0xFFFFu) : (uint16_t)1, total = modelled ? (uint16_t)(zt >> 16) : (uint16_t)2;
    uint32_t bit;
    (void)entropy_decode_fast(p.d, t, &bit, zero, total);          // (always kOk: packets of >= kFastPacketBits bits)
    dec_model_update(zero, total, bit == 0);
    p.fzt[ctx] = modelled ? ((uint32_t)zero | ((uint32_t)total << 16)) : zt;
    const bool to_sign = magnitude && cat == 0 && bit != 0u;        // the sample became significant: its sign is the next decision
    const uint32_t val = magnitude ? (cur | (bit << lsb)) : (cur | (((bit ^ pred) & 1u) << sign_bit));
    p.pend = to_sign ? 1u : 0u; p.pval = val; p.psctx = sctx; p.ppred = pred;
    if (to_sign) return;
    img.put_row(p.oc, c, val);
    p.done++;
    const bool more = c + 1 < w, last = !more && r + 1 >= h;
    const uint32_t sg = ((val & mask) >> lsb) != 0u ? 1u : 0u;                 // this sample is the next one's left neighbour
    p.nf |= (sg << 6) | ((sg & (val >> sign_bit)) << 14);                      // (a new row starts its flags afresh)
    p.c = more ? c + 1u : 0u;
    p.r = more ? r : r + 1u;
    p.status = last ? kOk : p.status;
}

// the segment in place, in the channel plane
struct GlobalImage {
    uint16_t *seg; size_t stride;
    ICER_HD uint32_t at(uint32_t r, uint32_t c) const { return seg[(size_t)r * stride + c]; }
    ICER_HD void put(uint32_t r, uint32_t c, uint32_t v) { seg[(size_t)r * stride + c] = (uint16_t)v; }
    ICER_HD uint32_t row_at(uint32_t r) const { return (uint32_t)(r * stride); }
    ICER_HD uint32_t row_after(uint32_t row) const { return row + (uint32_t)stride; }
    ICER_HD uint32_t at_row(uint32_t row, uint32_t c) const { return seg[(size_t)row + c]; }
    ICER_HD void put_row(uint32_t row, uint32_t c, uint32_t v) { seg[(size_t)row + c] = (uint16_t)v; }
};
ICER_HD void plane_step(PlaneDecoder &p, uint16_t *seg, uint32_t w, uint32_t h, size_t stride, int subband, int sign_bit,
                        const DecoderTables &t)
{
    GlobalImage img{seg, stride};
    plane_step_img(p, img, w, h, subband, sign_bit, t);
}

ICER_HD uint32_t packet_bits(const uint8_t *stream, uint32_t at)
{
    const uint8_t *p = stream + at;
    return (uint32_t)p[16] | ((uint32_t)p[17] << 8) | ((uint32_t)p[18] << 16) | ((uint32_t)p[19] << 24);
}

// all planes of one chain, top plane first, until one is missing or fails (icer_partition.c:427-443)
// `p`: a plane job with its storage attached (plane_attach_local / plane_attach_columns); it is reused plane after plane
ICER_HD void decode_chain(PlaneDecoder &p, uint16_t *plane, size_t stride, const ChainDesc &c, int subband, const uint8_t *stream,
                          uint32_t stream_len, const DecoderTables &t, int planes, int sign_bit)
{
    for (int lsb = planes - 1; lsb >= 0; lsb--) {
        const uint32_t at = c.pkt[lsb];
        if (at == kNoPacket) break;
        entropy_init(p.d, stream, stream_len, at + (uint32_t)kHeaderBytes, packet_bits(stream, at));
        plane_begin(p, lsb, sign_bit, c.w, c.h);
        while (p.status == 1) plane_step(p, plane + c.first, c.w, c.h, stride, subband, sign_bit, t);
        if (p.status != kOk) break;
    }
}

// ---- the planes of a chain side by side.  Plane lsb reads, of the plane above, the samples to the right and in the
// row below (significance and sign at lsb + 1), so it can run while the plane above is still at work as long as that one
// has finished sample (r + 1, c + 1) -- one row and one sample of lag per plane; nothing a plane writes is visible to the
// planes above it (they shift those bits out and have passed the sample).  Every packet becomes a job of its own.
// samples of the plane above that must be finished before this plane decodes sample (r, c)
ICER_HD uint32_t plane_needs(uint32_t r, uint32_t c, uint32_t w, uint32_t h)
{
    if (r + 1 >= h) return w * h;
    return (r + 1) * w + (c + 1 < w ? c + 1 : w - 1) + 1;
}
// may the plane below `above` decode its next sample?  (false for good once `above` has failed or was never started)
ICER_HD bool plane_ready(const PlaneDecoder &me, int above_status, uint32_t above_done, uint32_t w, uint32_t h)
{
    if (me.status != 1) return false;
    if (above_status == kOk) return true;
    if (above_status != 1) return false;
    return above_done >= plane_needs(me.r, me.c, w, h);
}
// When plane q fails, the reference has not started the planes below it: what they wrote is taken back -- their
// magnitude bits, and the sign of a sample whose remaining magnitude is zero (a sign is only ever set by the plane that
// makes the sample significant).
ICER_HD void chain_rollback(uint16_t *seg, uint32_t w, uint32_t h, size_t stride, int failed_lsb, int sign_bit)
{
    const uint32_t keep = ((1u << sign_bit) - 1u) & ~((1u << failed_lsb) - 1u);
    for (uint32_t r = 0; r < h; r++)
        for (uint32_t c = 0; c < w; c++) {
            const uint32_t v = seg[(size_t)r * stride + c], m = v & keep;
            seg[(size_t)r * stride + c] = (uint16_t)(m ? (m | (v & (1u << sign_bit))) : 0u);
        }
}

// ------------------------------------------------------------------------------------------ sample post-processing
// icer_from_sign_magnitude_int16 / _int8 (icer_wavelet.c:880-886 / :860-866); a set sign with zero magnitude gives 0
ICER_HD int16_t from_sign_magnitude(uint32_t v, int sign_bit)
{
    const int32_t mag = (int32_t)(v & ((1u << sign_bit) - 1u));
    return (int16_t)(((v >> sign_bit) & 1u) ? -mag : mag);
}
// the LL mean comes back in modulo the sample width (icer_compress.c:522-531 / :260-269)
ICER_HD int16_t add_ll_mean(int16_t s, uint16_t mean, int bits)
{
    return bits == 8 ? (int16_t)(int8_t)(s + (int8_t)mean) : (int16_t)(s + (int16_t)mean);
}

// ------------------------------------------------------------------------------------------ inverse DWT, one line
ICER_HD int32_t dec_floordiv(int32_t a, int32_t b)            // icer_floor_div_int32, icer.h:562-566 (b > 0)
{
    int32_t q = a / b;
    if ((a % b) != 0 && a < 0) q--;
    return q;
}
// src: [lows | highs] of a line of n samples (stride in samples); dst: the samples, value v of the [lows | highs]
// layout going to position pos_of[v] (the plain interleave, except for the uint8 routine's odd lines -- see
// interleave_positions).  `bits` = 8: int8 storage (truncating stores), icer_wavelet.c:298-383.
ICER_HD void idwt_line(const int16_t *src, int16_t *dst, uint32_t n, size_t stride, const FilterTaps f, int bits,
                       const uint32_t *pos_of)
{
    const uint32_t nl = (n + 1u) / 2u, nh = n / 2u;
    const bool odd = (n & 1u) != 0;
#define LO(k) ((int32_t)src[(size_t)(k) * stride])
#define HI(k) ((int32_t)src[(size_t)(nl + (k)) * stride])
#define RR(k) ((int32_t)(int16_t)(LO((k) - 1) - LO(k)))             /* get_r_int16 :206-208 (QUIRK W1b: wraps) */
#define TR(v) (bits == 8 ? (int16_t)(int8_t)(v) : (int16_t)(v))
    int32_t next_hi = 0;                                             // restored high k + 1
    for (uint32_t it = 0; it < nh; it++) {
        const uint32_t k = nh - 1u - it;
        int32_t add;
        if (k == 0) add = dec_floordiv(RR(1), 4);
        else if (k == 1 && f.am1 != 0) {
            // QUIRK (filter C, mirror of W3): the reference reads high 1 itself, still unrestored
            const int32_t x = (odd && nl == 3u) ? 0 : HI(1);
            add = dec_floordiv(2 * RR(1) + 3 * RR(2) - 2 * x + 4, 8);
        } else if (!odd && k == nh - 1u) add = dec_floordiv(RR(nh - 1u), 4);
        else {
            const int32_t rm = k >= 2u ? RR(k - 1u) : 1;
            const int32_t dn = (odd && k + 1u == nl - 1u) ? 0 : next_hi;
            add = dec_floordiv(f.am1 * rm + f.a0 * RR(k) + f.a1 * RR(k + 1u) - f.be * dn + 8, 16);
        }
        const int32_t hi = TR(HI(k) + add);
        next_hi = hi;
        const int32_t a = LO(k) + dec_floordiv(hi + 1, 2);
        dst[(size_t)pos_of[k] * stride] = TR(a);
        dst[(size_t)pos_of[nl + k] * stride] = TR(a - hi);
    }
    if (odd) dst[(size_t)pos_of[nl - 1u] * stride] = (int16_t)LO(nl - 1u);
#undef LO
#undef HI
#undef RR
#undef TR
}

// ------------------------------------------------------------------------------------------ host-side helpers
// icer_find_k (icer_wavelet.c:823-848): the reference's search for a slice 3^k + 1 <= len (not always the largest)
inline unsigned shuffle_slice_k(size_t len)
{
    unsigned lo_k = 0, hi_k = 11, res = 0;
    while (lo_k < hi_k) {
        const unsigned mid = (hi_k + lo_k) / 2;
        size_t slice = 1;
        for (unsigned e = 0; e < mid; e++) slice *= 3;
        slice += 1;
        if (len > slice) { lo_k = mid + 1; res = mid; }
        else if (len < slice) hi_k = (mid - 1) & 0xFFu;
        else break;
    }
    return res;
}
// Where each value of the [lows | highs] layout ends up after icer_interleave_uint16 / _uint8
// (icer_wavelet.c:705-763 / :570-628).  The in-place shuffle is followed on an index array; for the uint16 routine and
// for even lengths the result is the plain interleave.  QUIRK: the uint8 routine rotates with a different bound on odd
// lengths (:614 vs :749) and scrambles such lines.  pos_of has `len` entries.
inline void interleave_positions(size_t len, int bits, uint32_t *pos_of)
{
    if (len == 0) return;
    uint32_t *src = new uint32_t[len + 1];
    const bool odd = (len & 1) != 0;
    const size_t n = len - (odd ? 1 : 0);
    for (size_t i = 0; i < len; i++) src[i] = (uint32_t)i;
    auto rev = [&](size_t a, size_t b) { while (a < b) { const uint32_t x = src[a]; src[a] = src[b]; src[b] = x; a++; b--; } };
    if (odd) {
        const uint32_t x = src[n / 2];
        for (size_t i = n / 2; i < n; i++) src[i] = src[i + 1];
        src[len - 1] = x;
    }
    for (size_t done = 0; done < n;) {
        const unsigned k = shuffle_slice_k(n - done);
        size_t slice = 1;
        for (unsigned e = 0; e < k; e++) slice *= 3;
        slice += 1;
        const size_t half = slice / 2, left = n - done, halfleft = left / 2 - ((bits == 8 && odd) ? 0 : 1);
        rev(done + half, done + halfleft + half);
        rev(done + half, done + slice - 1);
        rev(done + slice, done + halfleft + half);
        for (size_t i = 1; i < slice; i *= 3) {
            size_t j = i;
            uint32_t carry = src[done + j];
            do {
                j = j < half ? 2 * j : (j - half) * 2 + 1;
                const uint32_t x = src[done + j]; src[done + j] = carry; carry = x;
            } while (j != i);
        }
        done += slice;
    }
    for (size_t i = 0; i < len; i++) pos_of[src[i]] = (uint32_t)i;
    delete[] src;
}

}  // namespace icer
