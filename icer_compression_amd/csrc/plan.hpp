// plan.hpp -- host-side planner: everything about a frame geometry that does not depend on pixel
// values.  Built once per (w, h, channels, stages, segments) and uploaded as flat tables that the
// kernels index by blockIdx.
//
// Restates, for the uint16 encoder:
//   subband geometry            icer_get_dim_n_low/high_stages   lib_icer/src/icer_wavelet.c:107-113
//   segment grid                icer_generate_partition_parameters   icer_partition.c:7-54
//   segment walk order          icer_compress_partition_uint16       icer_partition.c:299-385
//   packet list + priorities    icer_compress.c:315-363 (gray), icer_color.c:398-456 (YUV, quirk D4)
//   priority order              comp_packet + qsort, icer_compress.c:8-15,365 (stable: glibc merge sort)
//   final stream order          icer_compress.c:409-423, icer_color.c:508-527
#pragma once
#include <algorithm>
#include <array>
#include <map>
#include <stdint.h>
#include <utility>
#include <vector>

#include "icer_tables.hpp"

namespace icer {

constexpr int kXcds = 8;             // MI355X: 8 XCDs, workgroup b of a launch is placed on XCD b % 8

struct Packet {
    uint8_t level, subband, lsb, chan;
    uint64_t priority;
};

// One coding unit = one (packet, segment).  Device-visible; identical for every frame of a batch.
struct UnitDesc {
    uint32_t x0, y0, w, h;          // rectangle in plane coordinates
    uint32_t chan, level, subband, lsb, seg;
    uint32_t cap_words;             // payload capacity of the slot (32-bit words)
    uint32_t cap_is_bound;          // 1: capacity came from the bits-per-pixel bound, 0: from the byte quota
    uint32_t prio;                  // wave priority (s_setprio): the largest units form the critical path of a frame
    uint32_t sig_off;               // byte offset of the chunk table of the unit's family -- (channel, level, subband, segment), shared by
                                    // its bit planes -- in the frame's chunk-table area: one byte per 64-pixel chunk (chunk_blank_plane)
    uint64_t slot_off;              // byte offset of the slot (28-byte header + payload) in the frame's slot area
    uint32_t n_sub;                 // sub-ranges the unit is cut into when a launch is too small to fill the chip (coder_core.hpp "Sub-ranges"); <= 1: none
    uint32_t sub_first;             // index of its sub-range 1 in Plan::subs (sub-ranges 1 .. n_sub-1 are consecutive)
    uint32_t sub_entry;             // index of its sub-range 0 in the per-frame arrays of records / snapshots (n_sub consecutive entries)
    uint32_t family;                // index of the unit's family (its chunk table, its histogram of blank planes: Plan::n_families per frame)
};

// One extra workgroup of a split unit: sub-range `index` >= 1 (device-visible)
struct SubDesc {
    uint32_t unit, index;
    uint32_t cap_words;             // capacity of its private payload area (32-bit words)
    uint32_t pad_;
    uint64_t slot_off;              // byte offset of that area in the frame's slot area (after the units' slots)
};

struct SegmentGrid {
    // field order of partition_param_typdef (icer.h:126-142)
    uint16_t w, h, r, c, r_t, h_t, x_t, c_t0, y_t, r_t0, x_b, c_b0, y_b, r_b0, s;
};

// XCD-aware launch orders: entries are dealt to one list per XCD by family (families in turn, in the order they are met) and the
// lists interleaved, position 8k + x = the k-th entry of list x (workgroup b runs on XCD b % 8).  Once a list has run out its
// positions are filled from the longest remaining one, so position % 8 keeps naming the XCD of every entry that is not borrowed.
// (Dealing a new family to the list that is shortest so far, as a review suggested, measured badly: the lists' lengths at the
// moment a family is first met say little about their final lengths -- 216 of C2's 1 440 units ended up off their family's
// XCD against 0 with families dealt in turn, and the launch order lost its cost order: C2 10.9 ms against 6.1.)
inline void interleave_lists(const std::vector<std::vector<uint32_t>> &lists, std::vector<uint32_t> *out)
{
    size_t total = 0;
    for (const auto &l : lists) total += l.size();
    std::vector<size_t> pos(lists.size(), 0);
    out->clear();
    while (out->size() < total)
        for (size_t x = 0; x < lists.size() && out->size() < total; x++) {
            size_t y = x;
            if (pos[y] >= lists[y].size())
                for (size_t z = 0; z < lists.size(); z++) if (lists[z].size() - pos[z] > lists[y].size() - pos[y]) y = z;
            out->push_back(lists[y][pos[y]++]);
        }
}

inline size_t dim_low(size_t d, int level) { return (d + ((size_t(1) << level) - 1)) >> level; }
inline size_t dim_high(size_t d, int level) { return dim_low(d, level - 1) / 2; }

inline int make_grid(SegmentGrid *g, size_t w, size_t h, unsigned segments)
{
    if (segments > w * h || segments > (unsigned)kMaxSegments) return kTooManySegments;
    const size_t s = segments;
    size_t r;
    if (h > (s - 1) * w) r = s;
    else for (r = 1; r < s && (r + 1) * r * w < h * s; r++) {}
    const size_t c = s / r, r_t = (c + 1) * r - s;
    size_t h_t = ((2 * h * c * r_t + s) / 2) / s;
    if (h_t < r_t) h_t = r_t;
    const size_t x_t = w / c, c_t0 = (x_t + 1) * c - w, y_t = h_t / r_t, r_t0 = (y_t + 1) * r_t - h_t;
    size_t x_b = 0, c_b0 = 0, y_b = 0, r_b0 = 0;
    if (r_t < r) {
        x_b = w / (c + 1);
        c_b0 = (x_b + 1) * (c + 1) - w;
        y_b = (h - h_t) / (r - r_t);
        r_b0 = (y_b + 1) * (r - r_t) - (h - h_t);
    }
    *g = SegmentGrid{(uint16_t)w, (uint16_t)h, (uint16_t)r, (uint16_t)c, (uint16_t)r_t, (uint16_t)h_t,
                     (uint16_t)x_t, (uint16_t)c_t0, (uint16_t)y_t, (uint16_t)r_t0, (uint16_t)x_b,
                     (uint16_t)c_b0, (uint16_t)y_b, (uint16_t)r_b0, (uint16_t)s};
    return kOk;
}

struct Rect { uint32_t x, y, w, h; };

// rectangles in coding order: top region row-major, then bottom region row-major
inline void grid_rects(const SegmentGrid &g, std::vector<Rect> *out)
{
    out->clear();
    uint32_t y = 0;
    for (unsigned row = 0; row < g.r_t; row++) {
        const uint32_t sh = g.y_t + (row >= g.r_t0 ? 1u : 0u);
        uint32_t x = 0;
        for (unsigned col = 0; col < g.c; col++) {
            const uint32_t sw = g.x_t + (col >= g.c_t0 ? 1u : 0u);
            out->push_back(Rect{x, y, sw, sh});
            x += sw;
        }
        y += sh;
    }
    for (unsigned row = 0; row < (unsigned)(g.r - g.r_t); row++) {
        const uint32_t sh = g.y_b + (row >= g.r_b0 ? 1u : 0u);
        uint32_t x = 0;
        for (unsigned col = 0; col < (unsigned)(g.c + 1); col++) {
            const uint32_t sw = g.x_b + (col >= g.c_b0 ? 1u : 0u);
            out->push_back(Rect{x, y, sw, sh});
            x += sw;
        }
        y += sh;
    }
}

// `planes` = bit planes coded: 9 for the uint16 entry points, 7 for the uint8 twins (icer_compress.c:53-103, icer_color.c:73-131)
inline void make_packets(std::vector<Packet> *pk, int stages, int channels, int planes = kPlanes)
{
    pk->clear();
    if (channels == 1) {
        for (int st = 1; st <= stages; st++) {
            const uint64_t pr = uint64_t(1) << st;
            for (int lsb = 0; lsb < planes; lsb++) {
                pk->push_back(Packet{(uint8_t)st, kHL, (uint8_t)lsb, 0, pr << lsb});
                pk->push_back(Packet{(uint8_t)st, kLH, (uint8_t)lsb, 0, pr << lsb});
                pk->push_back(Packet{(uint8_t)st, kHH, (uint8_t)lsb, 0, ((pr / 2) << lsb) + 1});
            }
        }
        const uint64_t pr = uint64_t(1) << stages;
        for (int lsb = 0; lsb < planes; lsb++) pk->push_back(Packet{(uint8_t)stages, kLL, (uint8_t)lsb, 0, (2 * pr) << lsb});
    } else {
        // QUIRK D4: a 32-bit priority doubled once per (plane, Y) and never reset per plane
        for (int st = 1; st <= stages; st++) {
            uint32_t pr = 1u << st;
            for (int lsb = 0; lsb < planes; lsb++)
                for (int ch = 0; ch < channels; ch++) {
                    if (ch == 0) pr *= 2;
                    pk->push_back(Packet{(uint8_t)st, kHL, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(pr << lsb)});
                    pk->push_back(Packet{(uint8_t)st, kLH, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(pr << lsb)});
                    pk->push_back(Packet{(uint8_t)st, kHH, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(((pr / 2) << lsb) + 1)});
                }
        }
        uint32_t pr = 1u << stages;
        for (int lsb = 0; lsb < planes; lsb++)
            for (int ch = 0; ch < channels; ch++) {
                if (ch == 0) pr *= 2;
                pk->push_back(Packet{(uint8_t)stages, kLL, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)((2 * pr) << lsb)});
            }
    }
    std::stable_sort(pk->begin(), pk->end(), [](const Packet &a, const Packet &b) {
        if (a.priority != b.priority) return a.priority > b.priority;
        return a.subband < b.subband;
    });
}

struct Plan {
    size_t w = 0, h = 0;
    int channels = 0, stages = 0, segments = 0;
    int sample_bits = 16;                  // 16: uint16 entry points; 8: the uint8 twins (int8 storage, 7 planes)
    int error = kOk;                       // non-zero: the reference would refuse this geometry
    std::vector<Packet> packets;           // priority order
    std::vector<UnitDesc> units;           // priority order: packet order, then segment number
    std::vector<uint32_t> final_order;     // D7 order -> index into units
    std::vector<uint32_t> work_order;      // launch order (largest units first) -> index into units
    size_t slot_bytes = 0;                 // per-frame slot area for the current capacity rule
    size_t sig_bytes = 0;                  // per-frame chunk-table area (UnitDesc::sig_off)
    uint32_t n_families = 0;               // families per frame (UnitDesc::family): (channel, level, subband, segment)
    std::vector<uint32_t> sig_blocks;      // the work list of family_events_kernel: pairs (unit, block of 64 chunks), one member of every family
    std::vector<SubDesc> subs;             // sub-range workgroups of split launches: unit order, a unit's sub-ranges 1 .. n_sub-1 consecutive
    std::vector<uint32_t> split_launch;    // launch order of a split launch: bit 31 set = sub-range workgroup (index into subs), else a unit;
                                           // longest expected run first (a sub-range = its chunks + a tenth of its prefix pass)
    uint32_t sub_entries = 0;              // per-frame entries of the record / snapshot arrays (sum of n_sub over the split units)
};

// first chunk of sub-range i of a unit of `nchunks` chunks cut into `n_sub` (i = n_sub: the end)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t sub_first_chunk(uint32_t nchunks, uint32_t n_sub, uint32_t i) { return (uint32_t)((uint64_t)nchunks * i / n_sub); }

// The pipeline kernel of a batch is a ONE-dimensional grid of per_frame x n_frames workgroups walked position-major: eight launch
// positions (one per XCD: position % 8 names the XCD while a group is full, so a family keeps its L2) of every frame, then the next
// eight -- the largest units of ALL frames start first, the smallest of all frames end the launch.  Workgroup b -> (frame, launch
// position); a bijection for every per_frame >= 1 (the last group of a frame may have fewer than eight positions).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void position_major(uint32_t b, uint32_t per_frame, uint32_t n_frames, uint32_t *frame, uint32_t *lpos)
{
    const uint32_t g = b / (8u * n_frames), base = g * 8u;
    const uint32_t width = per_frame - base < 8u ? per_frame - base : 8u, r = b - g * 8u * n_frames;
    *frame = r / width;
    *lpos = base + r % width;
}

// quarter-octave size class of a unit (launch order treats units of one class as equally large)
inline int size_class(uint64_t pixels)
{
    int c = 0;
    while ((pixels >> (c + 1)) != 0) c++;                            // floor(log2)
    const uint64_t frac = c >= 2 ? (pixels >> (c - 2)) & 3u : 0u;    // next two bits
    return 4 * c + (int)frac;
}

inline int build_plan(Plan *p, size_t w, size_t h, int channels, int stages, int segments, int sample_bits = 16)
{
    p->w = w; p->h = h; p->channels = channels; p->stages = stages; p->segments = segments; p->sample_bits = sample_bits;
    // uint8 twins: ICER_BITPLANES_TO_COMPRESS_8 planes, the smaller packet table (ICER_MAX_PACKETS, icer.h:35-37)
    const int planes = sample_bits == 8 ? kPlanes8 : kPlanes, max_packets = sample_bits == 8 ? kMaxPackets8 : kMaxPackets;
    p->units.clear(); p->final_order.clear(); p->work_order.clear();
    if (channels != 1 && channels != 3) return p->error = kInvalidInput;
    if (w == 0 || h == 0 || w > 65535 || h > 65535 || segments < 1) return p->error = kInvalidInput;
    // beyond 6 stages / 32 segments the reference indexes past its static tables; we refuse instead
    if (stages < 1 || stages > kMaxStages) return p->error = kTooManyStages;
    if (segments > kMaxSegments) return p->error = kTooManySegments;
    if (dim_low(w, stages) < 3 || dim_low(h, stages) < 3) return p->error = kTooManyStages;   // icer_wavelet.c:63-68
    if ((3 * stages + 1) * planes * channels >= max_packets) return p->error = kPacketCountExceeded;

    make_packets(&p->packets, stages, channels, planes);
    SegmentGrid grid{};
    bool grid_valid = false;
    std::vector<Rect> rects;
    // slot index for the final re-ordering
    std::vector<int64_t> where((size_t)3 * (kMaxStages + 1) * 4 * kPlanes * (kMaxSegments + 1), -1);
    auto key = [](int ch, int lv, int sb, int lsb, int sg) {
        return ((((size_t)ch * (kMaxStages + 1) + lv) * 4 + sb) * kPlanes + lsb) * (kMaxSegments + 1) + sg;
    };
    for (const Packet &pk : p->packets) {
        size_t sw, sh, ox, oy;
        switch (pk.subband) {
        case kLL: sw = dim_low(w, pk.level);  sh = dim_low(h, pk.level);  ox = 0; oy = 0; break;
        case kHL: sw = dim_high(w, pk.level); sh = dim_low(h, pk.level);  ox = dim_low(w, pk.level); oy = 0; break;
        case kLH: sw = dim_low(w, pk.level);  sh = dim_high(h, pk.level); ox = 0; oy = dim_low(h, pk.level); break;
        default:  sw = dim_high(w, pk.level); sh = dim_high(h, pk.level); ox = dim_low(w, pk.level); oy = dim_low(h, pk.level); break;
        }
        // QUIRK P1: the reference ignores a failed grid and keeps using the previous packet's one
        if (make_grid(&grid, sw, sh, (unsigned)segments) == kOk) grid_valid = true;
        else if (!grid_valid) return p->error = kTooManySegments;   // reference reads an uninitialised struct here
        grid_rects(grid, &rects);
        for (size_t sg = 0; sg < rects.size(); sg++) {
            UnitDesc u{};
            u.x0 = (uint32_t)ox + rects[sg].x; u.y0 = (uint32_t)oy + rects[sg].y;
            u.w = rects[sg].w; u.h = rects[sg].h;
            u.chan = pk.chan; u.level = pk.level; u.subband = pk.subband; u.lsb = pk.lsb; u.seg = (uint32_t)sg;
            where[key(pk.chan, pk.level, pk.subband, pk.lsb, (int)sg)] = (int64_t)p->units.size();
            p->units.push_back(u);
        }
    }
    // D7: segment up, subband down, level down, plane down, channel up (icer_compress.c:409-423 / :148-162,
    // icer_color.c:508-527); the uint8 YUV variant walks subband, level and plane UP instead (icer_color.c:184-202)
    const bool up = sample_bits == 8 && channels == 3;
    for (int sg = 0; sg <= kMaxSegments; sg++)
        for (int isb = 0; isb < 4; isb++)
            for (int ilv = 0; ilv <= kMaxStages; ilv++)
                for (int il = 0; il < planes; il++)
                    for (int ch = 0; ch < channels; ch++) {
                        const int sb = up ? isb : 3 - isb, lv = up ? ilv : kMaxStages - ilv, lsb = up ? il : planes - 1 - il;
                        const int64_t u = where[key(ch, lv, sb, lsb, sg)];
                        if (u >= 0) p->final_order.push_back((uint32_t)u);
                    }
    // Launch order.  Largest units first (they form the critical path); and XCD-aware: workgroup b of a launch
    // runs on XCD b % 8 (observed placement, used for speed only), each XCD has its own 4 MiB L2, and the 9 bit
    // planes of one (channel, level, subband, segment) read the same coefficients.  So units are dealt to 8 lists
    // by family -- all planes of a family to the same list -- and the lists are interleaved: position 8k + x holds
    // the k-th unit of list x.
    {
        const size_t n = p->units.size();
        std::vector<uint32_t> by_size(n);
        for (size_t i = 0; i < n; i++) by_size[i] = (uint32_t)i;
        // (equal sizes: low bit planes first -- they are the dense, slow ones, and when a launch has more large units
        // than the chip has CUs the ones that double up should be the cheap high planes)
        std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t b) {
            const uint64_t pa = (uint64_t)p->units[a].w * p->units[a].h, pb = (uint64_t)p->units[b].w * p->units[b].h;
            const int ca = size_class(pa), cb = size_class(pb);        // "equal" = within about 20 %
            if (ca != cb) return ca > cb;
            if (p->units[a].lsb != p->units[b].lsb) return p->units[a].lsb < p->units[b].lsb;
            return pa > pb;
        });
        auto family = [&](const UnitDesc &u) { return ((u.chan * (kMaxStages + 1) + u.level) * 4 + u.subband) * (kMaxSegments + 1) + u.seg; };
        std::vector<int> xcd_of_family((size_t)3 * (kMaxStages + 1) * 4 * (kMaxSegments + 1), -1);
        std::vector<std::vector<uint32_t>> lists(kXcds);
        int next_xcd = 0;
        for (uint32_t u : by_size) {                                 // families meet their XCD in size order
            int &x = xcd_of_family[family(p->units[u])];
            if (x < 0) { x = next_xcd; next_xcd = (next_xcd + 1) % kXcds; }
            lists[x].push_back(u);
        }
        interleave_lists(lists, &p->work_order);
    }
    // chunk tables and event bytes: one set per family, in unit order.  A family is the units that code the SAME rectangle of the same
    // (channel, level, subband) at different bit planes -- normally all planes of a (channel, level, subband, segment); under quirk P1
    // (a failed grid keeps the previous PACKET's one, and which packet came before depends on the plane) the planes of one segment
    // number can have different rectangles: those are families of their own.
    {
        std::map<std::array<uint32_t, 8>, std::pair<size_t, uint32_t>> fam;       // rectangle -> (chunk offset, family index)
        size_t off = 0;
        uint32_t families = 0;
        p->sig_blocks.clear();
        for (size_t i = 0; i < p->units.size(); i++) {
            UnitDesc &u = p->units[i];
            const std::array<uint32_t, 8> key{u.chan, u.level, u.subband, u.seg, u.x0, u.y0, u.w, u.h};
            auto it = fam.find(key);
            if (it == fam.end()) {
                it = fam.emplace(key, std::make_pair(off, families++)).first;
                const uint32_t nchunks = ((uint32_t)u.w * u.h + 63u) / 64u;
                off += ((size_t)nchunks + 3) & ~(size_t)3;
                // the work list of family_events_kernel: this unit stands for its family, one entry per block of 64 chunks
                for (uint32_t b = 0; b < (nchunks + 63u) / 64u; b++) { p->sig_blocks.push_back((uint32_t)i); p->sig_blocks.push_back(b); }
            }
            u.sig_off = (uint32_t)it->second.first;
            u.family = it->second.second;
        }
        p->sig_bytes = off;
        p->n_families = families;
    }
    // the launch is latency-bound by its largest units: give their waves issue priority over the small ones
    const uint64_t biggest = p->units.empty() ? 1 : (uint64_t)p->units[p->work_order[0]].w * p->units[p->work_order[0]].h;
    for (UnitDesc &u : p->units) {
        const uint64_t px = (uint64_t)u.w * u.h;
        u.prio = px * 2 >= biggest ? 3u : px * 8 >= biggest ? 2u : px * 32 >= biggest ? 1u : 0u;
    }
    return p->error = kOk;
}

// Chunks per sub-range for a launch of ONE frame (coder_core.hpp "Sub-ranges"): the smallest piece size whose long pieces -- K per
// unit of at least two pieces, over the units of the lower half of the bit planes (the dense ones on image content) -- still start
// together: at most two per compute unit, the pipeline workgroups a compute unit holds in such a launch.  Smaller pieces shorten the
// chains that end the launch; more pieces than slots queue, and every piece pays its counts-only prefix and healing overlap.
// Measured (round 6, 256 compute units, profiles/r06_logs/r06x_split_by_geometry.log, r06ad_split_budget.log): 4096^2 with 10 / 12 / 16 segments
// 2 048 / 2 048 / 1 536 (5.43 / 5.66 / 4.45 ms; 3 072: 5.55 / 7.05 / -), 4096 x 2048 / 6 segments 1 024 (3.7 ms; 1 536: 4.1, 3 072: 6.8), 2048^2 / 4
// segments 1 024 (2.8 ms; 3 072: 5.1), 3000 x 2000 / 10 segments 1 024 (2.45 ms; 3 072: 3.1), 2048^2 / 16 and 1024^2 / 8 segments: no split (512 / 256: slower).
inline uint32_t auto_split_chunks(const std::vector<UnitDesc> &units, int n_cus, int planes)
{
    static const uint32_t sizes[] = {1024u, 1536u, 2048u, 3072u, 4096u, 6144u, 8192u, 12288u, 16384u, 32768u};
    const uint64_t budget = (uint64_t)n_cus * 2u;
    for (uint32_t p : sizes) {
        uint64_t pieces = 0;
        for (const UnitDesc &u : units) {
            if ((int)u.lsb * 2 > planes) continue;
            uint32_t k = ((u.w * u.h + 63u) / 64u) / p;
            if (k > 8u) k = 8u;
            if (k >= 2u) pieces += k;
        }
        if (pieces <= budget) return p;
    }
    return 32768u;
}

// Slot capacities: a unit's payload can never be useful beyond (quota - 28) bytes (P3), and we
// provision `bits_per_pixel` bits per pixel otherwise (overflow of that bound is detected and the
// batch is re-run with a larger bound).
inline void assign_slots(Plan *p, size_t quota, unsigned bits_per_pixel, uint32_t split_chunks = 0)
{
    const uint64_t quota_cap = ((quota > (size_t)kHeaderBytes ? quota - kHeaderBytes : 0) + 3) / 4 + 1;   // words
    uint64_t off = 0;
    for (UnitDesc &u : p->units) {
        const uint64_t bound = ((uint64_t)u.w * u.h * bits_per_pixel + 31) / 32 + 16;                     // words
        const uint64_t cap = std::min(bound, quota_cap);
        u.cap_words = (uint32_t)cap;
        u.cap_is_bound = bound < quota_cap ? 1u : 0u;
        u.slot_off = off;
        off += kHeaderBytes + cap * 4;
    }
    p->slot_bytes = (size_t)off;
    // sub-ranges (split launches): a unit of at least 2 * `split_chunks` chunks is cut into pieces of about that many (at most
    // kMaxSubs = 8); piece i >= 1 gets a private payload area for the chunks from its first one to the unit's end (it may
    // have to code all of them), behind the units' slots
    p->subs.clear();
    p->sub_entries = 0;
    for (size_t ui = 0; ui < p->units.size(); ui++) {
        UnitDesc &u = p->units[ui];
        const uint32_t nchunks = (u.w * u.h + 63u) / 64u;
        uint32_t k = split_chunks ? nchunks / split_chunks : 0u;
        if (k > 8u) k = 8u;
        u.n_sub = k >= 2u ? k : 0u;
        u.sub_first = (uint32_t)p->subs.size();
        u.sub_entry = p->sub_entries;
        if (u.n_sub) {
            p->sub_entries += u.n_sub;
            for (uint32_t i = 1; i < u.n_sub; i++) {
                const uint32_t c = sub_first_chunk(nchunks, u.n_sub, i);
                const uint64_t cap = (uint64_t)u.cap_words * (nchunks - c) / nchunks + 64u;
                p->subs.push_back(SubDesc{(uint32_t)ui, i, (uint32_t)cap, 0u, off});
                off += cap * 4;
            }
        }
    }
    p->slot_bytes = (size_t)off;              // (the frame's slot area includes the sub-ranges' private areas)
    {
        // Launch order of a split launch: longest expected run first, AND XCD-aware like work_order: workgroup b of a launch
        // runs on XCD b % 8, every XCD has an L2 of its own, and the bit planes of a family -- (channel, level, subband,
        // segment): units and sub-range workgroups alike -- read the same coefficients.  Entries are dealt to 8 lists by
        // family (a family meets its XCD when its most expensive entry comes up), each list in cost order, and the lists are
        // interleaved.  (Round 3 sorted by cost alone: the nine planes of a family then sat on nine L2s and the coder kernel
        // fetched 80 MiB per C2 launch instead of 23.)
        std::vector<std::pair<uint64_t, uint32_t>> cost;          // (expected chunks of work, launch entry)
        for (size_t i = 0; i < p->subs.size(); i++) {
            const UnitDesc &u = p->units[p->subs[i].unit];
            const uint32_t nchunks = (u.w * u.h + 63u) / 64u;
            const uint32_t c0 = sub_first_chunk(nchunks, u.n_sub, p->subs[i].index), c1 = sub_first_chunk(nchunks, u.n_sub, p->subs[i].index + 1u);
            cost.push_back({(uint64_t)(c1 - c0) * 10u + c0, 0x80000000u | (uint32_t)i});
        }
        for (uint32_t ui : p->work_order) {
            const UnitDesc &u = p->units[ui];
            const uint32_t nchunks = (u.w * u.h + 63u) / 64u;
            cost.push_back({(uint64_t)(u.n_sub > 1u ? sub_first_chunk(nchunks, u.n_sub, 1u) : nchunks) * 10u, ui});
        }
        std::stable_sort(cost.begin(), cost.end(), [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a.first > b.first; });
        auto family = [&](uint32_t entry) {
            const UnitDesc &u = p->units[(entry >> 31) ? p->subs[entry & 0x7FFFFFFFu].unit : entry];
            return ((u.chan * (kMaxStages + 1) + u.level) * 4 + u.subband) * (kMaxSegments + 1) + u.seg;
        };
        std::vector<int> xcd_of_family((size_t)3 * (kMaxStages + 1) * 4 * (kMaxSegments + 1), -1);
        std::vector<std::vector<uint32_t>> lists(kXcds);
        int next_xcd = 0;
        for (const auto &c : cost) {
            int &x = xcd_of_family[family(c.second)];
            if (x < 0) { x = next_xcd; next_xcd = (next_xcd + 1) % kXcds; }
            lists[x].push_back(c.second);
        }
        interleave_lists(lists, &p->split_launch);
    }
}

}  // namespace icer
