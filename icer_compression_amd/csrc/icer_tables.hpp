// icer_tables.hpp -- constants of the ICER encoder as used by the gfx950 kernels and the host planner.
//
// Values are data of the codec (filter taps, probability cut-offs, Golomb parameters, the
// variable-to-variable codes of bins 1..7 and their forced-completion bits); they restate
// lib_icer/src/icer_config.c:18-107 and lib_icer/src/icer_init.c:124-256 in our own layout and
// are verified entry by entry against the reference build in tests/test_tables.py.
#pragma once
#include <stdint.h>
#include <string.h>

namespace icer {

constexpr int kPlanes = 9;          // ICER_BITPLANES_TO_COMPRESS_16  (icer.h:44-46)
constexpr int kPlanes8 = 7;         // ICER_BITPLANES_TO_COMPRESS_8   (icer.h:41-43)
constexpr int kRingWords = 2048;    // ICER_CIRC_BUF_SIZE             (icer.h:27)
constexpr int kHeaderBytes = 28;    // sizeof(icer_image_segment_typedef) (icer.h:293-305)
constexpr int kMaxSegments = 32;    // ICER_MAX_SEGMENTS              (icer.h:29-31)
constexpr int kMaxStages = 6;       // ICER_MAX_DECOMP_STAGES         (icer.h:32-34)
constexpr int kMaxPackets = 800;    // ICER_MAX_PACKETS_16            (icer.h:38-40)
constexpr int kMaxPackets8 = 300;   // ICER_MAX_PACKETS               (icer.h:35-37)
constexpr int kNumBins = 17;
constexpr int kNumContexts = 17;
constexpr uint32_t kRescaleCap = 500;   // ICER_CONTEXT_RESCALING_CAP (icer.h:151)

// reference return codes (enum icer_status, lib_icer/inc/icer.h:92-105)
enum Status : int {
    kOk = 0, kIntegerOverflow = -1, kOutputBufTooSmall = -2, kTooManySegments = -3, kTooManyStages = -4,
    kByteQuotaExceeded = -5, kBitplaneOutOfRange = -6, kPacketCountExceeded = -9, kFatalError = -10,
    kInvalidInput = -11
};

enum Subband : int { kLL = 0, kHL = 1, kLH = 2, kHH = 3 };   // icer.h:181-187

// lifting filter taps {alpha_-1, alpha_0, alpha_1, beta} x16, filters A..F, Q  (icer_config.c:18-24)
struct FilterTaps { int am1, a0, a1, be; };
inline FilterTaps filter_taps(int filt)
{
    static const FilterTaps t[7] = {{0, 4, 4, 0}, {0, 4, 6, 4}, {-1, 4, 8, 6}, {0, 4, 5, 2},
                                    {0, 3, 8, 6}, {0, 3, 9, 8}, {0, 4, 4, 4}};
    return t[filt];
}

// Tables the coding-unit kernel stages into LDS.  Built once on the host (icer_init /
// encoder creation) and uploaded; ~1.3 KiB.
struct CoderTables {
    // bins 1..7: [bin][partial input value] -> in_bits | out_bits<<4 | out_code<<8   (0 = no entry)
    uint16_t v2v[8][32];
    // bins 1..7: [bin][n] = set of input values that are complete code words of n input bits (bit v set)
    uint32_t v2v_term[8][8];
    // bins 1..7: SIX input bits at a time.  A node of the code tree is numbered (partial input | 1 << bits so far),
    // the root is 1; the trees have at most 8 nodes, numbered compactly 0..7 (root = 0) by node_c / node_full.
    // v2v_step6[bin][compact node][6 bits] -> compact node after the 6 bits | (bit k set: a code word starts at the
    // k-th of the 6 bits) << 4.  The last 1..5 bits of a walk use the same entry for the start flags (padded with
    // zeros: a flag depends on earlier bits only) and v2v_tail[bin][compact node][1 << k | bits] for the node after k bits.
    uint16_t v2v_step6[8][8][64];
    uint8_t v2v_tail[8][8][64];
    uint8_t node_c[8][32];      // node -> compact number (0xFF: not a node of this bin's tree)
    uint8_t node_full[8][8];    // compact number -> node
    // the walker wave splits a bin's walk in two: lanes 1..7 walk the first half from the known node, lane L >= 8
    // walks the second half of bin cand_bin[L] assuming it is entered at node cand_node[L] (one lane per node of
    // the bin's code tree, 46 in all; cand_bin = 0: lane unused); cand_lane[bin][node] = that lane
    uint8_t cand_bin[64], cand_node[64], cand_lane[8][32];
    // bins 1..7: [bin][partial value 0..8][bits so far 0..5] -> appended bits | count<<4
    uint8_t v2v_flush[8][9][6];
    // bins 8..16: Golomb m, l = ceil(log2 m), i = 2^l - m
    uint16_t gm[17], gl[17], gi[17];
    // floor(2^20 / m) + 1: z / m == (z * ginv) >> 20 for every run length z < 2048 (checked at build time)
    uint32_t ginv[17];
    // probability cut-offs x65536 separating bin b-1 from bin b, b = 1..16   (icer_config.c:69-87)
    uint32_t cut[16];
    // the same as a look-up: for r = floor(zero * 65536 / total), entry r >> 8 holds the number of cut-offs below
    // 256 * (r >> 8) in bits 0..7 and, in bits 8..31, the one cut-off inside [256 * (r >> 8), +256) if there is one
    // (0xFFFFFF otherwise; the cut-offs are more than 256 apart, checked at build time): bin = base + (r >= that)
    uint32_t binlut[257];
    // x^(2^k) mod P for the CRC-32 polynomial (reflected), k = 0..31: lets a wave combine piece CRCs
    uint32_t x2n[32];
};

inline void build_coder_tables(CoderTables *t)
{
    memset(t, 0, sizeof *t);
    struct V { uint8_t bin, val, nin, code, nout; };
    // variable-to-variable codes; input bits are consumed LSB first
    static const V codes[] = {
        {1, 1, 2, 2, 2},  {1, 2, 2, 1, 2},  {1, 3, 3, 3, 3},   {1, 4, 3, 4, 3},  {1, 7, 4, 15, 4},
        {1, 8, 4, 8, 4},  {1, 15, 4, 16, 5}, {1, 0, 5, 7, 4},  {1, 16, 5, 0, 5},
        {2, 1, 2, 6, 3},  {2, 2, 2, 1, 2},  {2, 4, 3, 0, 2},   {2, 7, 3, 10, 4}, {2, 0, 4, 3, 3},
        {2, 3, 4, 7, 4},  {2, 11, 4, 2, 5}, {2, 8, 5, 15, 4},  {2, 24, 5, 18, 5},
        {3, 1, 2, 1, 2},  {3, 2, 2, 2, 2},  {3, 3, 2, 7, 3},   {3, 0, 3, 0, 2},  {3, 4, 3, 3, 3},
        {4, 0, 2, 1, 1},  {4, 2, 3, 0, 3},  {4, 3, 3, 12, 4},  {4, 5, 3, 2, 4},  {4, 6, 3, 10, 4},
        {4, 7, 3, 22, 5}, {4, 9, 4, 14, 4}, {4, 1, 5, 4, 4},   {4, 17, 5, 6, 5},
        {5, 1, 1, 2, 2},  {5, 2, 3, 3, 3},  {5, 4, 3, 5, 3},   {5, 6, 3, 15, 4}, {5, 8, 4, 1, 3},
        {5, 0, 5, 0, 2},  {5, 16, 5, 7, 4},
        {6, 3, 2, 7, 4},  {6, 0, 3, 0, 1},  {6, 1, 3, 3, 3},   {6, 2, 3, 5, 3},  {6, 4, 3, 1, 3},
        {6, 5, 3, 31, 5}, {6, 6, 3, 15, 5},
        {7, 1, 2, 3, 3},  {7, 2, 2, 5, 3},  {7, 3, 2, 31, 5},  {7, 4, 3, 1, 3},  {7, 0, 4, 0, 1},
        {7, 8, 5, 7, 4},  {7, 24, 5, 15, 5},
    };
    for (const V &c : codes) {
        t->v2v[c.bin][c.val] = (uint16_t)(c.nin | (c.nout << 4) | (c.code << 8));
        t->v2v_term[c.bin][c.nin] |= 1u << c.val;
    }
    memset(t->node_c, 0xFF, sizeof t->node_c);
    for (uint32_t b = 1; b <= 7; b++) {
        // the tree's nodes: the root and every proper prefix of a code word's input
        uint32_t count = 0;
        for (uint32_t node = 1; node < 32; node++) {
            uint32_t nin = 0;
            while ((2u << nin) <= node) nin++;                     // node = acc | 1 << nin
            const uint32_t acc = node ^ (1u << nin);
            bool is_node = node == 1;
            for (const V &c : codes)
                if (c.bin == b && nin && nin < c.nin && (c.val & ((1u << nin) - 1u)) == acc) is_node = true;
            if (is_node && count < 8) { t->node_c[b][node] = (uint8_t)count; t->node_full[b][count] = (uint8_t)node; count++; }
        }
        // one input bit from a node (icer_encoding.c:87-98): back to the root when the input is a code word
        auto step1 = [&](uint32_t node, uint32_t bit) {
            uint32_t nin = 0;
            while ((2u << nin) <= node) nin++;
            uint32_t acc = (node ^ (1u << nin)) | (bit << nin);
            nin++;
            return (nin == 5 || ((t->v2v_term[b][nin] >> acc) & 1u)) ? 1u : (acc | (1u << nin));
        };
        for (uint32_t cn = 0; cn < count; cn++) {
            for (uint32_t bits = 0; bits < 64; bits++) {
                uint32_t node = t->node_full[b][cn], starts = 0;
                for (int k = 0; k < 6; k++) {
                    if (node == 1) starts |= 1u << k;
                    node = step1(node, (bits >> k) & 1u);
                }
                t->v2v_step6[b][cn][bits] = (uint16_t)(t->node_c[b][node] | (starts << 4));
            }
            for (uint32_t k = 1; k <= 5; k++)
                for (uint32_t bits = 0; bits < (1u << k); bits++) {
                    uint32_t node = t->node_full[b][cn];
                    for (uint32_t i = 0; i < k; i++) node = step1(node, (bits >> i) & 1u);
                    t->v2v_tail[b][cn][(1u << k) | bits] = t->node_c[b][node];
                }
        }
    }
    {
        uint32_t lane = 8;
        for (uint32_t b = 1; b <= 7; b++)
            for (uint32_t node = 1; node < 32; node++) {
                uint32_t nin = 0;
                while ((2u << nin) <= node) nin++;
                const uint32_t acc = node ^ (1u << nin);
                bool is_node = node == 1;                        // proper prefix of some code word's input?
                for (const V &c : codes)
                    if (c.bin == b && nin && nin < c.nin && (c.val & ((1u << nin) - 1u)) == acc) is_node = true;
                if (is_node && lane < 64) {
                    t->cand_bin[lane] = (uint8_t)b;
                    t->cand_node[lane] = (uint8_t)node;
                    t->cand_lane[b][node] = (uint8_t)lane;
                    lane++;
                }
            }
    }
    struct F { uint8_t bin, val, nin, add, nadd; };
    static const F fl[] = {
        {1, 1, 1, 0, 1}, {1, 3, 2, 0, 1}, {1, 7, 3, 0, 1}, {1, 0, 1, 1, 1}, {1, 0, 2, 1, 1}, {1, 0, 3, 1, 1}, {1, 0, 4, 0, 1},
        {2, 0, 1, 1, 1}, {2, 0, 2, 1, 1}, {2, 0, 3, 0, 1}, {2, 8, 4, 0, 1}, {2, 1, 1, 0, 1}, {2, 3, 2, 1, 1}, {2, 3, 3, 0, 1},
        {3, 0, 1, 1, 1}, {3, 0, 2, 0, 1}, {3, 1, 1, 0, 1},
        {4, 0, 1, 0, 1}, {4, 2, 2, 0, 1}, {4, 1, 2, 1, 1}, {4, 1, 3, 1, 1}, {4, 1, 4, 0, 1}, {4, 1, 1, 1, 2}, {4, 3, 2, 0, 1},
        {5, 0, 1, 1, 2}, {5, 1, 2, 0, 1}, {5, 0, 2, 1, 1}, {5, 0, 3, 1, 1}, {5, 0, 4, 0, 1},
        {6, 0, 1, 0, 2}, {6, 0, 2, 0, 1}, {6, 2, 2, 0, 1}, {6, 1, 1, 1, 1}, {6, 1, 2, 0, 1},
        {7, 0, 1, 1, 1}, {7, 0, 2, 1, 1}, {7, 0, 3, 0, 1}, {7, 8, 4, 0, 1}, {7, 1, 1, 0, 1},
    };
    for (const F &f : fl) t->v2v_flush[f.bin][f.val][f.nin] = (uint8_t)(f.add | (f.nadd << 4));
    static const uint16_t m[17] = {0, 0, 0, 0, 0, 0, 0, 0, 5, 6, 7, 11, 17, 31, 70, 200, 512};
    for (int b = 8; b <= 16; b++) {
        unsigned l = 0;
        while ((1u << l) < m[b]) l++;
        t->gm[b] = m[b];
        t->gl[b] = (uint16_t)l;
        t->gi[b] = (uint16_t)((1u << l) - m[b]);
        t->ginv[b] = (1u << 20) / m[b] + 1;
        for (uint32_t z = 0; z < 2048; z++)
            if (((z * t->ginv[b]) >> 20) != z / m[b]) t->ginv[b] = 0;     // would be caught by tests/test_tables.py
    }
    static const uint32_t cut[16] = {35298, 37345, 40503, 43591, 47480, 50133, 53645, 55902,
                                     57755, 58894, 60437, 62267, 63613, 64557, 65134, 65392};
    memcpy(t->cut, cut, sizeof cut);
    for (uint32_t k = 0; k <= 256; k++) {
        uint32_t base = 0, inside = 0xFFFFFFu, n_inside = 0;
        for (int b = 0; b < 16; b++) {
            if (cut[b] < 256u * k) base++;
            else if (cut[b] < 256u * (k + 1)) { inside = cut[b]; n_inside++; }
        }
        t->binlut[k] = n_inside <= 1 ? (base | (inside << 8)) : 0xFFFFFFFFu;   // (two in one bucket: caught by tests/test_tables.py)
    }
    // x2n[0] = x^1; squaring chain (bit 31 = x^0 in the reflected representation)
    auto mulmod = [](uint32_t a, uint32_t b) {
        uint32_t p = 0;
        for (uint32_t m = 0x80000000u; m != 0; m >>= 1) {
            if (a & m) { p ^= b; if ((a & (m - 1u)) == 0) break; }
            b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
        }
        return p;
    };
    t->x2n[0] = 0x40000000u;
    for (int k = 1; k < 32; k++) t->x2n[k] = mulmod(t->x2n[k - 1], t->x2n[k - 1]);
}

}  // namespace icer
