// decoder_plan.hpp -- host-side planner of the decoder: from the verified packets of a stream to
// the list of chains the decode kernel runs and the levels the inverse transform undoes.  Pure host C++, shared by
// decoder.hip and the CPU build in tests/emu.
//
// Restates   icer_find_packet_in_bytestream (the walk; the CRCs are checked per candidate, on the device)
//                                                        lib_icer/src/icer_compress.c:569-588
//            the packet table + image size + LL mean     icer_compress.c:443-463, icer_color.c:550-570
//            the subband / segment loops                 icer_compress.c:472-518, icer_color.c:585-634
#pragma once
#include <algorithm>
#include <stdint.h>
#include <vector>

#include "decoder_core.hpp"
#include "plan.hpp"

namespace icer {

// a position in the stream where a packet header with the preamble and a matching header CRC starts
struct PacketCandidate {
    uint32_t off;           // byte offset of the header in its stream
    uint32_t payload_bytes; // ceil(data_length / 8)
    uint32_t fits;          // the payload lies inside the stream
    uint32_t payload_ok;    // ... and its CRC-32 matches
    uint32_t frame;         // which stream of a batch
    uint32_t crc_acc;       // device: CRC-32 of the payload, XOR-accumulated from the pieces of check_payloads_kernel
    uint8_t hdr[kHeaderBytes];  // copy of the header, so that the host planner never reads the (device-resident) stream
};

// reflected CRC-32 (zlib), byte at a time; `tab` = the usual 256-entry table
inline void build_crc32_table(uint32_t *tab)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        tab[i] = c;
    }
}
ICER_HD uint32_t crc32_bytes(const uint32_t *tab, const uint8_t *p, uint32_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 255u] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
ICER_HD uint32_t load_le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// does a header candidate start at `off`?  (preamble + header CRC; icer_compress.c:574-575)
ICER_HD bool header_candidate(const uint32_t *tab, const uint8_t *s, uint32_t len, uint32_t off, PacketCandidate *out)
{
    if (len - off < (uint32_t)kHeaderBytes) return false;           // (the reference reads the header regardless)
    const uint8_t *p = s + off;
    if (p[0] != 0x5Bu || p[1] != 0x60u) return false;
    if (load_le32(p + 24) != crc32_bytes(tab, p, 24)) return false;
    const uint32_t bits = load_le32(p + 16);
    out->off = off;
    out->frame = 0;
    for (int i = 0; i < kHeaderBytes; i++) out->hdr[i] = p[i];
    out->payload_bytes = bits / 8u + ((bits % 8u) ? 1u : 0u);
    out->fits = out->payload_bytes <= len - off - (uint32_t)kHeaderBytes;
    out->payload_ok = 0;
    out->crc_acc = 0;
    return true;
}
// payload check of one candidate (icer_compress.c:576-577)
ICER_HD void check_payload(const uint32_t *tab, const uint8_t *s, PacketCandidate *c)
{
    if (!c->fits) return;
    const uint8_t *p = s + c->off;
    c->payload_ok = load_le32(p + 20) == crc32_bytes(tab, p + kHeaderBytes, c->payload_bytes);
}

// ---- the payload CRC in pieces (device: check_payloads_kernel, 64 threads per candidate).  For the init / final-xor form
// crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P (zlib's crc32_combine), so the CRC of a payload is the XOR of
// its pieces' CRCs, each multiplied by x^(8 * bytes behind the piece).
ICER_HD uint32_t dec_gf_mulmod(uint32_t a, uint32_t b)         // a(x) * b(x) mod P(x), reflected representation (bit 31 = x^0)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m != 0; m >>= 1) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1u)) == 0) break;
        }
        b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
}
ICER_HD uint32_t dec_gf_xpow_bytes(uint32_t nbytes)            // x^(8 * nbytes) mod P by repeated squaring
{
    uint32_t p = 0x80000000u, sq = 0x00800000u;                   // x^0, x^8
    for (; nbytes != 0; nbytes >>= 1) {
        if (nbytes & 1u) p = dec_gf_mulmod(sq, p);
        sq = dec_gf_mulmod(sq, sq);
    }
    return p;
}
// piece `k` of `pieces` of candidate c's payload: its contribution to the payload CRC
ICER_HD uint32_t payload_piece_crc(const uint32_t *tab, const uint8_t *s, const PacketCandidate &c, uint32_t k, uint32_t pieces)
{
    if (!c.fits) return 0;
    const uint32_t n = c.payload_bytes, piece = (((n + pieces - 1u) / pieces) + 15u) & ~15u;
    const uint32_t start = k * piece, end = start + piece < n ? start + piece : n;
    if (start >= end) return 0;
    uint32_t v = crc32_bytes(tab, s + c.off + (uint32_t)kHeaderBytes + start, end - start);
    if (n > end) v = dec_gf_mulmod(dec_gf_xpow_bytes(n - end), v);
    return v;
}

struct DecodeLevel { uint32_t cw, ch; };        // region of one inverse-transform level (deepest first)

struct DecodePlan {
    int rc = kOk;                               // return code of the reference for this call
    bool transform = false;                     // the run reaches sign-magnitude removal / mean / inverse DWT / clamping
    size_t w = 0, h = 0;
    uint16_t mean[3] = {0, 0, 0};
    std::vector<ChainDesc> chains;
    std::vector<DecodeLevel> levels;            // empty when the deepest LL is thinner than 3 (ICER_TOO_MANY_STAGES, ignored)
};

// `cands`: the candidates of ONE stream, sorted by offset.  *w / *h: in = the caller's values (kept when the stream holds no valid packet).
inline void plan_decode(DecodePlan *pl, const std::vector<PacketCandidate> &cands, int channels,
                        int stages, unsigned segments, int sample_bits, size_t w_in, size_t h_in, size_t bufsize)
{
    const int planes = sample_bits == 8 ? kPlanes8 : kPlanes;
    *pl = DecodePlan();
    pl->w = w_in; pl->h = h_in;
    if (channels != 1 && channels != 3) { pl->rc = kInvalidInput; return; }
    if (stages < 1 || stages > kMaxStages) { pl->rc = kTooManyStages; return; }      // (reference: out-of-bounds table)
    // [chan][level][subband][segment][lsb] -> packet offset; the last packet of a kind wins
    std::vector<uint32_t> table((size_t)3 * (kMaxStages + 1) * 4 * (kMaxSegments + 1) * kPlanes, kNoPacket);
    std::vector<uint32_t> bits_tab(table.size(), 0u);              // ... and that packet's data_length
    auto slot_index = [&](int ch, int lv, int sb, int sg, int lsb) {
        return ((((size_t)ch * (kMaxStages + 1) + lv) * 4 + sb) * (kMaxSegments + 1) + sg) * kPlanes + lsb;
    };
    auto slot = [&](int ch, int lv, int sb, int sg, int lsb) -> uint32_t & { return table[slot_index(ch, lv, sb, sg, lsb)]; };
    // the scan accepts a candidate when it starts at or behind the end of the previous packet and both CRCs hold;
    // anything else is stepped over byte by byte
    uint32_t cursor = 0;
    for (const PacketCandidate &c : cands) {
        if (c.off < cursor || !c.fits || !c.payload_ok) continue;
        const uint8_t *p = c.hdr;
        const int lv = p[4], sb = p[5], sg = p[6], lsb = p[7] & 15, ch = channels == 3 ? (p[7] >> 4) : 0;
        if (lv <= kMaxStages && sb < 4 && sg <= kMaxSegments && lsb < kPlanes && ch < 3) {
            slot(ch, lv, sb, sg, lsb) = c.off;
            bits_tab[slot_index(ch, lv, sb, sg, lsb)] = load_le32(p + 16);
        }
        pl->w = load_le32(p + 8);
        pl->h = load_le32(p + 12);
        if (ch < 3) pl->mean[ch] = (uint16_t)(p[2] | (p[3] << 8));
        cursor = c.off + (uint32_t)kHeaderBytes + c.payload_bytes;
    }
    if (bufsize < pl->w * pl->h) { pl->rc = kByteQuotaExceeded; return; }
    const size_t w = pl->w, h = pl->h;
    std::vector<Rect> rects;
    for (int lv = 1; lv <= stages && pl->rc == kOk; lv++)
        for (int ch = 0; ch < channels && pl->rc == kOk; ch++)
            for (int sb = (lv == stages ? 0 : 1); sb < 4; sb++) {
                size_t sw, sh, ox, oy;
                switch (sb) {
                case kLL: sw = dim_low(w, lv);  sh = dim_low(h, lv);  ox = 0; oy = 0; break;
                case kHL: sw = dim_high(w, lv); sh = dim_low(h, lv);  ox = dim_low(w, lv); oy = 0; break;
                case kLH: sw = dim_low(w, lv);  sh = dim_high(h, lv); ox = 0; oy = dim_low(h, lv); break;
                default:  sw = dim_high(w, lv); sh = dim_high(h, lv); ox = dim_low(w, lv); oy = dim_low(h, lv); break;
                }
                SegmentGrid g;
                // (unlike the encoder, P1, the decoder stops on a grid error: whatever was decoded before stays, as
                // sign-magnitude words)
                if ((pl->rc = make_grid(&g, sw, sh, segments)) != kOk) break;
                grid_rects(g, &rects);
                for (size_t sg = 0; sg < rects.size() && sg <= (size_t)kMaxSegments; sg++) {
                    if (slot(ch, lv, sb, (int)sg, planes - 1) == kNoPacket) continue;
                    ChainDesc c;
                    c.frame = 0;
                    c.subband = (uint32_t)sb;
                    c.chan = (uint32_t)ch;
                    c.first = (uint32_t)((oy + rects[sg].y) * w + ox + rects[sg].x);
                    c.w = (uint16_t)rects[sg].w; c.h = (uint16_t)rects[sg].h;
                    for (int lsb = 0; lsb < kPlanes; lsb++) c.pkt[lsb] = lsb < planes ? slot(ch, lv, sb, (int)sg, lsb) : kNoPacket;
                    c.fast = (c.w > 0 && c.h > 0) ? 1u : 0u;
                    for (int lsb = planes - 1; lsb >= 0 && c.pkt[lsb] != kNoPacket; lsb--)
                        if (bits_tab[slot_index(ch, lv, sb, (int)sg, lsb)] < kFastPacketBits) c.fast = 0u;
                    pl->chains.push_back(c);
                }
            }
    if (pl->rc != kOk) return;
    pl->transform = true;
    if (dim_low(w, stages) >= 3 && dim_low(h, stages) >= 3)
        for (int it = 1; it <= stages; it++)
            pl->levels.push_back(DecodeLevel{(uint32_t)dim_low(w, stages - it), (uint32_t)dim_low(h, stages - it)});
}

}  // namespace icer
