/*
 * icer_oracle.c -- TEST INFRASTRUCTURE ONLY (see icer_oracle.h).
 *
 * A from-scratch, plain-C restatement of the encoder hot path of lib_icer
 * (TheRealOrange/icer_compression).  Every function cites the reference file:line whose
 * behaviour it restates.  It is written for clarity, not speed: out-of-place 1-D lifting
 * instead of the in-place shuffle, an explicit event/codeword model of the interleaved
 * entropy coder, and a closed-form byte-quota walk.  All reference quirks that influence the
 * byte stream are reproduced on purpose (see the comments marked QUIRK).
 *
 * Parity status: PINNED by tests/test_oracle_vs_ref.py (against oracle/_ref/libicer_ref.so,
 * built from the untouched sources) and by the digests in tests/golden/.
 */
#include "icer_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MAX_PLANES 9        /* ICER_BITPLANES_TO_COMPRESS_16, icer.h:44-46 (ICER_BITPLANES_TO_COMPRESS_8 = 7, icer.h:41-43) */
#define RING_WORDS 2048     /* ICER_CIRC_BUF_SIZE, icer.h:27 */
#define HEADER_BYTES 28     /* sizeof(icer_image_segment_typedef), icer.h:293-305 */
#define MAX_SEGMENTS 32     /* ICER_MAX_SEGMENTS, icer.h:29-31 */
#define MAX_STAGES 6        /* ICER_MAX_DECOMP_STAGES, icer.h:32-34 */
#define MAX_PACKETS 800     /* ICER_MAX_PACKETS_16, icer.h:38-40 */

enum { SB_LL = 0, SB_HL = 1, SB_LH = 2, SB_HH = 3 };   /* icer.h:181-187 */

/* ------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------- */
static int32_t floordiv(int32_t a, int32_t b)          /* icer.h:562-566 (b > 0 here) */
{
    int32_t q = a / b;
    if ((a % b) != 0 && a < 0) q--;
    return q;
}
static size_t ceil_shift(size_t v, int s) { return (v + (((size_t)1 << s) - 1)) >> s; }
/* icer_get_dim_n_low_stages / _high_stages, icer_wavelet.c:107-113 */
static size_t dim_low(size_t d, int level) { return ceil_shift(d, level); }
static size_t dim_high(size_t d, int level) { return ceil_shift(d, level - 1) / 2; }

/* ------------------------------------------------------------------------------------------
 * a-1..a-4  forward lifting DWT (icer_wavelet.c:57-77, 155-171, 385-465; filter table
 * icer_config.c:18-24).  The reference lifts pairs in place, un-shuffles in place and then
 * runs the prediction step in place; only the resulting layout [lows | highs] and values
 * matter, so this restatement works on copies.
 * ---------------------------------------------------------------------------------------- */
static const int FILT[7][4] = {   /* alpha_-1, alpha_0, alpha_1, beta   (x16) */
    {0, 4, 4, 0}, {0, 4, 6, 4}, {-1, 4, 8, 6}, {0, 4, 5, 2}, {0, 3, 8, 6}, {0, 3, 9, 8}, {0, 4, 4, 4}};

static int fits16(int32_t v) { return v >= -32768 && v <= 32767; }
static int fits8(int32_t v) { return v >= -128 && v <= 127; }

/* `bits` = 16: icer_wavelet_transform_1d_uint16 (icer_wavelet.c:385-465); `bits` = 8: its int8 twin
 * (icer_wavelet.c:215-296), whose samples live sign-extended in the int16 line: stores truncate to int8 and
 * flag overflow at the int8 bounds, r[] is formed in int16 without wrapping (get_r_int8 :196-198). */
static int dwt_1d_bits(int16_t *line, size_t n, size_t stride, int filt, int bits)
{
#define FITS(v) (bits == 8 ? fits8(v) : fits16(v))
#define TRUNC(v) (bits == 8 ? (int16_t)(int8_t)(v) : (int16_t)(v))
    size_t nl = (n + 1) / 2, nh = n / 2;
    int odd = (int)(n & 1);
    int overflow = 0;
    int16_t *lo = (int16_t *)malloc(sizeof(int16_t) * (nl + 1));
    int16_t *hi = (int16_t *)malloc(sizeof(int16_t) * (nh + 2));
    int16_t *out = (int16_t *)malloc(sizeof(int16_t) * (nh + 1));

    /* step 1 (icer_wavelet.c:402-426): pair average (floor) and difference, truncating stores */
    for (size_t k = 0; k < nh; k++) {
        int32_t a = line[(2 * k) * stride], b = line[(2 * k + 1) * stride];
        int32_t l = floordiv(a + b, 2), h = a - b;
        if (!FITS(l) || !FITS(h)) overflow = 1;
        lo[k] = TRUNC(l);
        hi[k] = TRUNC(h);
    }
    if (odd) lo[nl - 1] = line[(n - 1) * stride];        /* lone last sample is a low (always in range) */
    hi[nh] = 0;                                          /* get_d_int16 :210-212: missing last high reads as 0 */

    /* step 2 (icer_wavelet.c:430-462).  r[k] = (int16)(lo[k-1]-lo[k]) -- QUIRK W1b: the
     * difference is wrapped to 16 bits without an overflow flag (get_r_int16 :206-208). */
#define R(k) ((int32_t)(int16_t)(lo[(k) - 1] - lo[(k)]))
    const int am1 = FILT[filt][0], a0 = FILT[filt][1], a1 = FILT[filt][2], be = FILT[filt][3];
    for (size_t k = 0; k < nh; k++) {
        int32_t sub;
        if (k == 0) {
            sub = floordiv(R(1), 4);
        } else if (k == 1 && am1 != 0) {
            /* QUIRK W3 (filter C only): the reference passes offset=low_N, so the "d[2]" of the
             * ICER paper is really d[1] (still unmodified), and 0 when n is odd and nl == 3. */
            int32_t x = (odd && nl == 3) ? 0 : hi[1];
            sub = floordiv(2 * R(1) + 3 * R(2) - 2 * x + 4, 8);
        } else if (!odd && k == nh - 1) {
            sub = floordiv(R(nh - 1), 4);
        } else {
            int32_t rm = (k >= 2) ? R(k - 1) : 1;        /* r[0] reads as 1; only ever multiplied by alpha_-1 == 0 */
            int32_t dn = (odd && k + 1 == nl - 1) ? 0 : hi[k + 1];
            sub = floordiv(am1 * rm + a0 * R(k) + a1 * R(k + 1) - be * dn + 8, 16);
        }
        int32_t h = (int32_t)hi[k] - sub;
        if (!FITS(h)) overflow = 1;
        out[k] = TRUNC(h);
    }
#undef R
#undef FITS
#undef TRUNC
    for (size_t k = 0; k < nl; k++) line[k * stride] = lo[k];
    for (size_t k = 0; k < nh; k++) line[(nl + k) * stride] = out[k];
    free(lo); free(hi); free(out);
    return overflow ? ORC_INTEGER_OVERFLOW : ORC_OK;
}

int orc_dwt_1d(int16_t *line, size_t n, size_t stride, int filt) { return dwt_1d_bits(line, n, stride, filt, 16); }

static int dwt_stages_bits(int16_t *s, size_t w, size_t h, int stages, int filt, int bits)
{
    if (dim_low(w, stages) < 3 || dim_low(h, stages) < 3) return ORC_TOO_MANY_STAGES;
    int overflow = 0;
    size_t cw = w, ch = h;
    for (int st = 0; st < stages; st++) {
        for (size_t r = 0; r < ch; r++) overflow |= (dwt_1d_bits(s + r * w, cw, 1, filt, bits) != ORC_OK);
        for (size_t c = 0; c < cw; c++) overflow |= (dwt_1d_bits(s + c, ch, w, filt, bits) != ORC_OK);
        cw = (cw + 1) / 2;
        ch = (ch + 1) / 2;
    }
    return overflow ? ORC_INTEGER_OVERFLOW : ORC_OK;
}

int orc_dwt_stages_u16(uint16_t *img, size_t w, size_t h, int stages, int filt)
{
    /* icer_wavelet.c:57-77: refuse when the final LL would be thinner than 3 */
    if (dim_low(w, stages) < 3 || dim_low(h, stages) < 3) return ORC_TOO_MANY_STAGES;
    int overflow = 0;
    size_t cw = w, ch = h;
    int16_t *s = (int16_t *)img;
    for (int st = 0; st < stages; st++) {
        /* icer_wavelet.c:155-171: every row of the current LL region, then every column */
        for (size_t r = 0; r < ch; r++) overflow |= (orc_dwt_1d(s + r * w, cw, 1, filt) != ORC_OK);
        for (size_t c = 0; c < cw; c++) overflow |= (orc_dwt_1d(s + c, ch, w, filt) != ORC_OK);
        cw = (cw + 1) / 2;
        ch = (ch + 1) / 2;
    }
    return overflow ? ORC_INTEGER_OVERFLOW : ORC_OK;
}

/* a-6  icer_to_sign_magnitude_int16, icer_wavelet.c:871-877 */
void orc_sign_magnitude(uint16_t *data, size_t len)
{
    for (size_t i = 0; i < len; i++) {
        int16_t v = (int16_t)data[i];
        if (v < 0) data[i] = (uint16_t)(0x8000u | (uint16_t)(-(int32_t)v));   /* -32768 -> 0x8000 */
    }
}

/* ------------------------------------------------------------------------------------------
 * a-8  segment grid, icer_partition.c:7-54 (ICER paper's partition rule)
 * ---------------------------------------------------------------------------------------- */
int orc_partition_make(orc_partition *p, size_t w, size_t h, unsigned segments)
{
    if (segments > w * h || segments > MAX_SEGMENTS) return ORC_TOO_MANY_SEGMENTS;
    size_t s = segments;
    size_t r;
    if (h > (s - 1) * w) r = s;
    else for (r = 1; r < s && (r + 1) * r * w < h * s; r++) {}
    size_t c = s / r;
    size_t r_t = (c + 1) * r - s;
    size_t h_t = ((2 * h * c * r_t + s) / 2) / s;
    if (h_t < r_t) h_t = r_t;
    size_t x_t = w / c, c_t0 = (x_t + 1) * c - w;
    size_t y_t = h_t / r_t, r_t0 = (y_t + 1) * r_t - h_t;
    size_t x_b = 0, c_b0 = 0, y_b = 0, r_b0 = 0;
    if (r_t < r) {
        x_b = w / (c + 1);
        c_b0 = (x_b + 1) * (c + 1) - w;
        y_b = (h - h_t) / (r - r_t);
        r_b0 = (y_b + 1) * (r - r_t) - (h - h_t);
    }
    /* all fields are uint16_t in the reference (icer.h:126-142): truncating stores */
    p->w = (uint16_t)w; p->h = (uint16_t)h; p->s = (uint16_t)s;
    p->r = (uint16_t)r; p->c = (uint16_t)c; p->r_t = (uint16_t)r_t; p->h_t = (uint16_t)h_t;
    p->x_t = (uint16_t)x_t; p->c_t0 = (uint16_t)c_t0; p->y_t = (uint16_t)y_t; p->r_t0 = (uint16_t)r_t0;
    p->x_b = (uint16_t)x_b; p->c_b0 = (uint16_t)c_b0; p->y_b = (uint16_t)y_b; p->r_b0 = (uint16_t)r_b0;
    return ORC_OK;
}

/* segment rectangles in coding order: top region row-major, then bottom region row-major
 * (icer_partition.c:299-385).  Returns the number of rectangles (== p->s for a valid grid). */
int orc_partition_rects(const orc_partition *p, orc_rect *rects)
{
    int n = 0;
    uint32_t y = 0;
    for (unsigned row = 0; row < p->r_t; row++) {
        uint32_t sh = p->y_t + (row >= p->r_t0 ? 1u : 0u), x = 0;
        for (unsigned col = 0; col < p->c; col++) {
            uint32_t sw = p->x_t + (col >= p->c_t0 ? 1u : 0u);
            rects[n].x = x; rects[n].y = y; rects[n].w = sw; rects[n].h = sh; n++;
            x += sw;
        }
        y += sh;
    }
    for (unsigned row = 0; row < (unsigned)(p->r - p->r_t); row++) {
        uint32_t sh = p->y_b + (row >= p->r_b0 ? 1u : 0u), x = 0;
        for (unsigned col = 0; col < (unsigned)(p->c + 1); col++) {
            uint32_t sw = p->x_b + (col >= p->c_b0 ? 1u : 0u);
            rects[n].x = x; rects[n].y = y; rects[n].w = sw; rects[n].h = sh; n++;
            x += sw;
        }
        y += sh;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * a-7  packet list + priority order (gray: icer_compress.c:315-365; YUV: icer_color.c:398-458;
 * comparator icer_compress.c:8-15).  glibc's qsort is a stable merge sort, so ties keep
 * generation order: restated here as a stable insertion sort.
 * ---------------------------------------------------------------------------------------- */
static int packet_list_planes(orc_packet *out, int stages, int channels, int PLANES);
int orc_packet_list(orc_packet *out, int stages, int channels) { return packet_list_planes(out, stages, channels, 9); }

/* `PLANES` = ICER_BITPLANES_TO_COMPRESS_16 (9) or ICER_BITPLANES_TO_COMPRESS_8 (7); the uint8 variants
 * (icer_compress.c:53-103, icer_color.c:73-131) are otherwise the same code */
static int packet_list_planes(orc_packet *out, int stages, int channels, int PLANES)
{
    int n = 0;
    if (channels == 1) {
        for (int st = 1; st <= stages; st++) {
            uint64_t pr = (uint64_t)1 << st;
            for (int lsb = 0; lsb < PLANES; lsb++) {
                out[n++] = (orc_packet){(uint8_t)st, SB_HL, (uint8_t)lsb, 0, pr << lsb};
                out[n++] = (orc_packet){(uint8_t)st, SB_LH, (uint8_t)lsb, 0, pr << lsb};
                out[n++] = (orc_packet){(uint8_t)st, SB_HH, (uint8_t)lsb, 0, ((pr / 2) << lsb) + 1};
            }
        }
        uint64_t pr = (uint64_t)1 << stages;
        for (int lsb = 0; lsb < PLANES; lsb++)
            out[n++] = (orc_packet){(uint8_t)stages, SB_LL, (uint8_t)lsb, 0, (2 * pr) << lsb};
    } else {
        /* QUIRK D4: the YUV variant doubles a uint32 `priority` once per (lsb, Y) and never
         * resets it inside the lsb loop, so it grows by 2 per plane *in addition* to << lsb. */
        for (int st = 1; st <= stages; st++) {
            uint32_t pr = (uint32_t)1 << st;
            for (int lsb = 0; lsb < PLANES; lsb++) {
                for (int ch = 0; ch < channels; ch++) {
                    if (ch == 0) pr *= 2;
                    out[n++] = (orc_packet){(uint8_t)st, SB_HL, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(pr << lsb)};
                    out[n++] = (orc_packet){(uint8_t)st, SB_LH, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(pr << lsb)};
                    out[n++] = (orc_packet){(uint8_t)st, SB_HH, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)(((pr / 2) << lsb) + 1)};
                }
            }
        }
        uint32_t pr = (uint32_t)1 << stages;
        for (int lsb = 0; lsb < PLANES; lsb++)
            for (int ch = 0; ch < channels; ch++) {
                if (ch == 0) pr *= 2;
                out[n++] = (orc_packet){(uint8_t)stages, SB_LL, (uint8_t)lsb, (uint8_t)ch, (uint64_t)(uint32_t)((2 * pr) << lsb)};
            }
    }
    /* stable: priority descending, then subband ascending */
    for (int i = 1; i < n; i++) {
        orc_packet key = out[i];
        int j = i - 1;
        while (j >= 0 && (out[j].priority < key.priority ||
                          (out[j].priority == key.priority && out[j].subband > key.subband))) {
            out[j + 1] = out[j];
            j--;
        }
        out[j + 1] = key;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * a-15  CRC-32 (reflected 0xEDB88320, init/final 0xFFFFFFFF), crc32.c:72-116,157-169
 * ---------------------------------------------------------------------------------------- */
static uint32_t crc_table[256];
static int crc_ready = 0;
uint32_t orc_crc32(const uint8_t *buf, size_t len)
{
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = 1;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; i++) c = crc_table[(c ^ buf[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}

/* ------------------------------------------------------------------------------------------
 * a-12..a-14  interleaved entropy coder (icer_encoding.c:15-206; tables icer_config.c:69-107,
 * icer_init.c:124-256).
 * ---------------------------------------------------------------------------------------- */
/* probability cut-offs x65536 between bins (icer_config.c:69-87) */
static const uint32_t CUT[17] = {35298, 37345, 40503, 43591, 47480, 50133, 53645, 55902, 57755,
                                 58894, 60437, 62267, 63613, 64557, 65134, 65392, 65536};
/* Golomb parameter m for bins 8..16 (icer_config.c:89-107) */
static const uint16_t GOLOMB_M[17] = {0, 0, 0, 0, 0, 0, 0, 0, 5, 6, 7, 11, 17, 31, 70, 200, 512};

/* variable-to-variable codes of bins 1..7 (icer_init.c:124-188): {input value, input bits,
 * output code, output bits}; input bits arrive LSB first. */
typedef struct { uint8_t in_val, in_bits, out_code, out_bits; } v2v;
static const v2v V2V[8][9] = {
    {{0, 0, 0, 0}},
    {{1, 2, 2, 2}, {2, 2, 1, 2}, {3, 3, 3, 3}, {4, 3, 4, 3}, {7, 4, 15, 4}, {8, 4, 8, 4}, {15, 4, 16, 5}, {0, 5, 7, 4}, {16, 5, 0, 5}},
    {{1, 2, 6, 3}, {2, 2, 1, 2}, {4, 3, 0, 2}, {7, 3, 10, 4}, {0, 4, 3, 3}, {3, 4, 7, 4}, {11, 4, 2, 5}, {8, 5, 15, 4}, {24, 5, 18, 5}},
    {{1, 2, 1, 2}, {2, 2, 2, 2}, {3, 2, 7, 3}, {0, 3, 0, 2}, {4, 3, 3, 3}},
    {{0, 2, 1, 1}, {2, 3, 0, 3}, {3, 3, 12, 4}, {5, 3, 2, 4}, {6, 3, 10, 4}, {7, 3, 22, 5}, {9, 4, 14, 4}, {1, 5, 4, 4}, {17, 5, 6, 5}},
    {{1, 1, 2, 2}, {2, 3, 3, 3}, {4, 3, 5, 3}, {6, 3, 15, 4}, {8, 4, 1, 3}, {0, 5, 0, 2}, {16, 5, 7, 4}},
    {{3, 2, 7, 4}, {0, 3, 0, 1}, {1, 3, 3, 3}, {2, 3, 5, 3}, {4, 3, 1, 3}, {5, 3, 31, 5}, {6, 3, 15, 5}},
    {{1, 2, 3, 3}, {2, 2, 5, 3}, {3, 2, 31, 5}, {4, 3, 1, 3}, {0, 4, 0, 1}, {8, 5, 7, 4}, {24, 5, 15, 5}},
};
/* bits appended to a partial input of bins 1..7 when it is force-completed
 * (icer_init.c:191-237): {partial value, bits so far, appended bits, #appended} */
typedef struct { uint8_t val, nbits, add, nadd; } v2v_flush;
static const v2v_flush V2VF[8][7] = {
    {{0, 0, 0, 0}},
    {{1, 1, 0, 1}, {3, 2, 0, 1}, {7, 3, 0, 1}, {0, 1, 1, 1}, {0, 2, 1, 1}, {0, 3, 1, 1}, {0, 4, 0, 1}},
    {{0, 1, 1, 1}, {0, 2, 1, 1}, {0, 3, 0, 1}, {8, 4, 0, 1}, {1, 1, 0, 1}, {3, 2, 1, 1}, {3, 3, 0, 1}},
    {{0, 1, 1, 1}, {0, 2, 0, 1}, {1, 1, 0, 1}},
    {{0, 1, 0, 1}, {2, 2, 0, 1}, {1, 2, 1, 1}, {1, 3, 1, 1}, {1, 4, 0, 1}, {1, 1, 1, 2}, {3, 2, 0, 1}},
    {{0, 1, 1, 2}, {1, 2, 0, 1}, {0, 2, 1, 1}, {0, 3, 1, 1}, {0, 4, 0, 1}},
    {{0, 1, 0, 2}, {0, 2, 0, 1}, {2, 2, 0, 1}, {1, 1, 1, 1}, {1, 2, 0, 1}},
    {{0, 1, 1, 1}, {0, 2, 1, 1}, {0, 3, 0, 1}, {8, 4, 0, 1}, {1, 1, 0, 1}},
};

/* dense look-ups in the reference's indexing (value-indexed, zero = "no entry") */
static struct { uint8_t in_bits, out_bits, out_code; } code_lut[17][32];
static struct { uint8_t add, nadd; } flush_lut[17][9][6];
static struct { uint16_t m, l, i; } golomb[17];
static int coder_ready = 0;

static void coder_tables_init(void)
{
    if (coder_ready) return;
    memset(code_lut, 0, sizeof code_lut);
    memset(flush_lut, 0, sizeof flush_lut);
    memset(golomb, 0, sizeof golomb);
    for (int b = 1; b <= 7; b++) {
        for (int k = 0; k < 9; k++) {
            v2v e = V2V[b][k];
            if (e.in_bits == 0) continue;
            code_lut[b][e.in_val].in_bits = e.in_bits;
            code_lut[b][e.in_val].out_bits = e.out_bits;
            code_lut[b][e.in_val].out_code = e.out_code;
        }
        for (int k = 0; k < 7; k++) {
            v2v_flush f = V2VF[b][k];
            if (f.nbits == 0) continue;
            flush_lut[b][f.val][f.nbits].add = f.add;
            flush_lut[b][f.val][f.nbits].nadd = f.nadd;
        }
    }
    for (int b = 8; b <= 16; b++) {                      /* icer_init.c:239-256 */
        unsigned m = GOLOMB_M[b], l = 0;
        while ((1u << l) < m) l++;                       /* ceil(log2 m) */
        golomb[b].m = (uint16_t)m; golomb[b].l = (uint16_t)l; golomb[b].i = (uint16_t)((1u << l) - m);
    }
    coder_ready = 1;
}

/* icer_compute_bin, icer_util.c:48-56 */
int orc_pick_bin(uint32_t zero, uint32_t total)
{
    uint32_t lhs = zero * 65536u;
    for (int b = 16; b >= 1; b--)
        if (lhs >= total * CUT[b - 1]) return b;
    return 0;
}

typedef struct {
    uint8_t bin;        /* owner bin while open */
    uint8_t done;
    uint8_t nbits;      /* output bits once done */
    uint16_t value;     /* open: Golomb run length / v2v partial input; done: output code */
} ring_word;

typedef struct {
    ring_word ring[RING_WORDS];
    unsigned head, used;
    int open_slot[17];          /* -1 = bin has no open word */
    int in_bits[17];            /* v2v bins: input bits accumulated so far */
    uint8_t *out;
    size_t out_cap;
    uint64_t bitpos;            /* bits emitted so far */
    int full;                   /* ran out of out_cap */
} coder;

static void coder_init(coder *c, uint8_t *out, size_t cap)
{
    c->head = c->used = 0;
    for (int b = 0; b < 17; b++) { c->open_slot[b] = -1; c->in_bits[b] = 0; }
    c->out = out; c->out_cap = cap; c->bitpos = 0; c->full = 0;
    if (cap) memset(out, 0, cap);
}

static unsigned reverse_bits(unsigned v, int n)          /* icer.h:601-610 */
{
    unsigned r = 0;
    for (int k = 0; k < n; k++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* icer_popbuf_while_avail, icer_encoding.c:114-139: drain finished words from the head,
 * LSB-first bit packing */
static void coder_drain(coder *c)
{
    while (c->used > 0 && c->ring[c->head].done) {
        ring_word w = c->ring[c->head];
        c->head = (c->head + 1) % RING_WORDS;
        c->used--;
        for (int k = 0; k < w.nbits; k++) {
            size_t byte = (size_t)(c->bitpos >> 3);
            if (byte >= c->out_cap) { c->full = 1; return; }
            if ((w.value >> k) & 1) c->out[byte] |= (uint8_t)(1u << (c->bitpos & 7));
            c->bitpos++;
        }
    }
}

static void golomb_close(ring_word *w, int bin, unsigned k)     /* icer_encoding.c:73-80 */
{
    unsigned code = k + (k >= golomb[bin].i ? golomb[bin].i : 0);
    int n = golomb[bin].l + (k >= golomb[bin].i ? 1 : 0);
    w->value = (uint16_t)(reverse_bits(code, n) & 0x3FF);
    w->nbits = (uint8_t)n;
    w->done = 1;
}

/* icer_flush_encode, icer_encoding.c:141-189: force-complete the oldest word */
static void coder_flush_head(coder *c)
{
    ring_word *w = &c->ring[c->head];
    if (!w->done) {
        int bin = w->bin;
        if (bin >= 8) {
            unsigned k = w->value;
            if (k == (unsigned)golomb[bin].m - 1) { w->value = 1; w->nbits = 1; w->done = 1; }
            else golomb_close(w, bin, k);
        } else if (bin >= 1) {
            unsigned add = flush_lut[bin][w->value][c->in_bits[bin]].add;
            unsigned pre = w->value | (add << c->in_bits[bin]);
            /* QUIRK: no check that the completed input is a real code; the LUT entry is used as is */
            w->value = code_lut[bin][pre & 31].out_code;
            w->nbits = code_lut[bin][pre & 31].out_bits;
            w->done = 1;
            c->in_bits[bin] = 0;
        }
        if (bin >= 1) c->open_slot[bin] = -1;
    }
    coder_drain(c);
}

/* icer_encode_bit, icer_encoding.c:37-112 */
static void coder_put(coder *c, int bit, uint32_t zero, uint32_t total)
{
    if (zero < (total >> 1)) { zero = total - zero; bit ^= 1; }
    int bin = orc_pick_bin(zero, total);
    if (c->open_slot[bin] < 0) {
        if (c->used == RING_WORDS) coder_flush_head(c);          /* E5: ring full */
        unsigned slot = (c->head + c->used) % RING_WORDS;
        c->used++;
        c->ring[slot].bin = (uint8_t)bin; c->ring[slot].done = 0; c->ring[slot].nbits = 0; c->ring[slot].value = 0;
        c->open_slot[bin] = (int)slot;
    }
    ring_word *w = &c->ring[c->open_slot[bin]];
    if (bin >= 8) {
        if (bit) { golomb_close(w, bin, w->value); c->open_slot[bin] = -1; }
        else if (++w->value >= golomb[bin].m) { w->value = 1; w->nbits = 1; w->done = 1; c->open_slot[bin] = -1; }
    } else if (bin >= 1) {
        w->value |= (uint16_t)(bit << c->in_bits[bin]);
        c->in_bits[bin]++;
        if (code_lut[bin][w->value & 31].in_bits == c->in_bits[bin]) {
            unsigned pre = w->value & 31;
            w->value = code_lut[bin][pre].out_code; w->nbits = code_lut[bin][pre].out_bits; w->done = 1;
            c->open_slot[bin] = -1; c->in_bits[bin] = 0;
        }
    } else {
        w->value = (uint16_t)bit; w->nbits = 1; w->done = 1; c->open_slot[0] = -1;
    }
    coder_drain(c);
}

/* ------------------------------------------------------------------------------------------
 * a-10, a-11  context modeller for one bit plane of one segment
 * (icer_context_modeller.c:312-457, 607-613, 631-642; tables icer_config.c:26-67)
 * ---------------------------------------------------------------------------------------- */
static int ctx_plain(int h, int v, int d)                /* icer_context_table_ll_lh_hl */
{
    if (h == 2) return 8;
    if (h == 1) return (v == 0) ? (d == 0 ? 5 : d == 1 ? 6 : 7) : 7;
    if (v == 0) return d > 2 ? 2 : d;
    return v == 1 ? 3 : 4;
}
static int ctx_hh(int hv, int d)                         /* icer_context_table_hh */
{
    if (d >= 3) return 8;
    int k = hv > 2 ? 2 : hv;
    if (d == 0) return k;
    if (d == 1) return 3 + k;
    return hv == 0 ? 6 : 7;
}
static const uint8_t SIGN_CTX[5][5] = {{14, 14, 15, 16, 16}, {14, 14, 15, 16, 16}, {13, 13, 12, 13, 13},
                                       {16, 16, 15, 14, 14}, {16, 16, 15, 14, 14}};
static const uint8_t SIGN_PRED[5][5] = {{1, 1, 1, 1, 1}, {1, 1, 1, 1, 1}, {0, 0, 0, 1, 1}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};

static void model_update(uint32_t *zero, uint32_t *total, int was_zero)
{
    /* icer_context_modeller.c:396-402.  QUIRK C5: when zero <= total/2 at a rescale the
     * reference computes a halved value and discards it, so zero is left untouched. */
    (*total)++;
    *zero += (uint32_t)was_zero;
    if (*total >= 500) {
        *total >>= 1;
        if (*zero > *total) *zero >>= 1;
    }
}

long orc_code_unit(const uint16_t *seg, size_t w, size_t h, size_t rowstride,
                   int subband, int lsb, uint8_t *out, size_t out_cap)
{
    coder_tables_init();
    if (lsb + 1 >= 16) return ORC_BITPLANE_OUT_OF_RANGE;
    coder *c = (coder *)malloc(sizeof(coder));
    coder_init(c, out, out_cap);
    uint32_t zero[17], total[17];
    for (int k = 0; k < 17; k++) { zero[k] = 2; total[k] = 4; }   /* :607-613 */

#define MAG(r, cc) (seg[(r) * rowstride + (cc)] & 0x7FFFu)
#define NEG(r, cc) (seg[(r) * rowstride + (cc)] >> 15)
    /* significance of a neighbour at plane `l`; outside the segment = insignificant (:361-372) */
#define SIG(r, cc, l) (((r) < 0 || (cc) < 0 || (r) >= (long)h || (cc) >= (long)w) ? 0 : ((MAG(r, cc) >> (l)) != 0))
    /* QUIRK C6: only negative significant neighbours contribute (-1); positive ones give 0 */
#define SGN(r, cc, l) ((SIG(r, cc, l) && NEG(r, cc)) ? -1 : 0)
    for (long r = 0; r < (long)h && !c->full; r++) {
        for (long cc = 0; cc < (long)w && !c->full; cc++) {
            unsigned m = MAG(r, cc);
            int msb = 0;
            for (unsigned t = m | 1; t > 1; t >>= 1) msb++;
            int cat = msb - lsb;
            if (cat < 0) cat = 0;
            if (cat > 3) cat = 3;
            int bit = (int)((m >> lsb) & 1);
            if (cat == 3) { coder_put(c, bit, 1, 2); continue; }          /* uncoded, no model update */
            int ctx;
            if (cat == 2) ctx = 11;
            else {
                int hh = SIG(r, cc - 1, lsb) + SIG(r, cc + 1, lsb + 1);
                int vv = SIG(r - 1, cc, lsb) + SIG(r + 1, cc, lsb + 1);
                int dd = SIG(r - 1, cc - 1, lsb) + SIG(r - 1, cc + 1, lsb) + SIG(r + 1, cc - 1, lsb + 1) + SIG(r + 1, cc + 1, lsb + 1);
                if (cat == 1) ctx = (hh + vv == 0) ? 9 : 10;
                else {
                    if (subband == SB_HL) { int t = hh; hh = vv; vv = t; }
                    ctx = (subband == SB_HH) ? ctx_hh(hh + vv, dd) : ctx_plain(hh, vv, dd);
                }
            }
            coder_put(c, bit, zero[ctx], total[ctx]);
            model_update(&zero[ctx], &total[ctx], !bit);
            if (cat == 0 && bit) {
                int sh = SGN(r, cc - 1, lsb) + SGN(r, cc + 1, lsb + 1) + 2;
                int sv = SGN(r - 1, cc, lsb) + SGN(r + 1, cc, lsb + 1) + 2;
                if (subband == SB_HL) { int t = sh; sh = sv; sv = t; }
                int sctx = SIGN_CTX[sh][sv];
                int agree = (SIGN_PRED[sh][sv] ^ (int)NEG(r, cc)) & 1;
                coder_put(c, agree, zero[sctx], total[sctx]);
                model_update(&zero[sctx], &total[sctx], agree == 0);
            }
        }
    }
#undef MAG
#undef NEG
#undef SIG
#undef SGN
    while (c->used > 0 && !c->full) coder_flush_head(c);            /* :452-455 */
    long bits = c->full ? (long)ORC_BYTE_QUOTA_EXCEEDED : (long)c->bitpos;
    free(c);
    return bits;
}

/* ------------------------------------------------------------------------------------------
 * a-5, a-9, a-15..a-17  frame driver (icer_compress.c:279-426, icer_color.c:343-530,
 * icer_partition.c:279-388, icer_encoding.c:210-234)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *bytes;     /* header + payload */
    size_t len;
} unit_blob;

static void put16(uint8_t *p, unsigned v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

static int compress_core(uint16_t *const planes[], const uint16_t mean[3], int channels, size_t w, size_t h, int stages,
                         unsigned segments, size_t quota, uint8_t *out, size_t *size_used,
                         int PLANES, int max_packets, int yuv_order_up);

int orc_compress_u16(uint16_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                     unsigned segments, size_t quota, uint8_t *out, size_t *size_used)
{
    *size_used = 0;
    if (channels != 1 && channels != 3) return ORC_INVALID_INPUT;
    if (stages < 1 || stages > MAX_STAGES) return ORC_TOO_MANY_STAGES;   /* reference is UB beyond 6 */

    /* DWT per channel, aborting at the first failing channel (icer_color.c:347-354) */
    for (int ch = 0; ch < channels; ch++) {
        int res = orc_dwt_stages_u16(planes[ch], w, h, stages, filt);
        if (res != ORC_OK) return res;
    }
    /* LL mean (icer_compress.c:286-302): unsigned sum, integer divide, must fit int16 */
    size_t llw = dim_low(w, stages), llh = dim_low(h, stages);
    uint16_t mean[3] = {0, 0, 0};
    for (int ch = 0; ch < channels; ch++) {
        uint64_t sum = 0;
        for (size_t r = 0; r < llh; r++)
            for (size_t c = 0; c < llw; c++) sum += planes[ch][r * w + c];
        mean[ch] = (uint16_t)(sum / (llw * llh));
    }
    for (int ch = 0; ch < channels; ch++)
        if (mean[ch] > 32767) return ORC_INTEGER_OVERFLOW;
    for (int ch = 0; ch < channels; ch++) {
        int16_t *s = (int16_t *)planes[ch];
        for (size_t r = 0; r < llh; r++)
            for (size_t c = 0; c < llw; c++) s[r * w + c] = (int16_t)(s[r * w + c] - (int16_t)mean[ch]);
        orc_sign_magnitude(planes[ch], w * h);
    }
    return compress_core(planes, mean, channels, w, h, stages, segments, quota, out, size_used, 9, MAX_PACKETS, 0);
}

/* packets in priority order -> coding units -> quota rule -> final order.  `planes` hold sign-magnitude words
 * (bit 15 sign).  Shared by the uint16 (icer_compress.c:304-426, icer_color.c:388-530) and uint8
 * (icer_compress.c:43-166, icer_color.c:62-206) entry points, which differ in the plane count, the packet-table
 * size and -- YUV only -- the direction of the final re-ordering loops. */
static int compress_core(uint16_t *const planes[], const uint16_t mean[3], int channels, size_t w, size_t h, int stages,
                         unsigned segments, size_t quota, uint8_t *out, size_t *size_used,
                         int PLANES, int max_packets, int yuv_order_up)
{
    orc_packet pk[MAX_PACKETS];
    if ((3 * stages + 1) * PLANES * channels >= max_packets) return ORC_PACKET_COUNT_EXCEEDED;
    int npk = packet_list_planes(pk, stages, channels, PLANES);

    /* kept units, addressed [chan][level][subband][lsb][segment] for the final re-ordering */
    static unit_blob kept[3][MAX_STAGES + 1][4][MAX_PLANES][MAX_SEGMENTS + 1];
    memset(kept, 0, sizeof kept);

    int rc = ORC_OK;
    size_t used = 0;
    orc_partition part;
    int part_valid = 0;
    orc_rect rects[MAX_SEGMENTS + 64];
    for (int ip = 0; ip < npk && rc == ORC_OK; ip++) {
        const orc_packet *p = &pk[ip];
        size_t sw, sh, ox, oy;                                   /* icer_compress.c:378-399 */
        switch (p->subband) {
        case SB_LL: sw = dim_low(w, p->level);  sh = dim_low(h, p->level);  ox = 0; oy = 0; break;
        case SB_HL: sw = dim_high(w, p->level); sh = dim_low(h, p->level);  ox = dim_low(w, p->level); oy = 0; break;
        case SB_LH: sw = dim_low(w, p->level);  sh = dim_high(h, p->level); ox = 0; oy = dim_low(h, p->level); break;
        default:    sw = dim_high(w, p->level); sh = dim_high(h, p->level); ox = dim_low(w, p->level); oy = dim_low(h, p->level); break;
        }
        /* QUIRK P1: the reference ignores the error return, re-using the previous packet's grid.
         * A failure on the very first packet reads an uninitialised struct there; we refuse. */
        if (orc_partition_make(&part, sw, sh, segments) == ORC_OK) part_valid = 1;
        else if (!part_valid) return ORC_TOO_MANY_SEGMENTS;
        int nseg = orc_partition_rects(&part, rects);
        const uint16_t *base = planes[p->chan] + oy * w + ox;
        for (int sg = 0; sg < nseg; sg++) {
            /* P2/P3: header must fit, then the payload must end strictly inside the quota */
            size_t rem = quota - used;
            if (rem < HEADER_BYTES) { rc = ORC_BYTE_QUOTA_EXCEEDED; break; }
            size_t cap = (size_t)rects[sg].w * rects[sg].h * 3 + 64;
            uint8_t *blob = (uint8_t *)malloc(HEADER_BYTES + cap);
            long bits = orc_code_unit(base + (size_t)rects[sg].y * w + rects[sg].x, rects[sg].w, rects[sg].h, w,
                                      p->subband, p->lsb, blob + HEADER_BYTES, cap);
            if (bits < 0) { free(blob); return ORC_FATAL_ERROR; }   /* cap is a strict upper bound */
            if ((size_t)bits / 8 >= rem - HEADER_BYTES && bits > 0) { free(blob); rc = ORC_BYTE_QUOTA_EXCEEDED; break; }
            size_t nbytes = ((size_t)bits + 7) / 8;
            /* header, icer.h:293-305 / icer_encoding.c:210-234 (F1) */
            put16(blob + 0, 0x605B);
            put16(blob + 2, (uint8_t)mean[p->chan]);             /* QUIRK D1: mean passes through a uint8_t */
            blob[4] = p->level; blob[5] = p->subband; blob[6] = (uint8_t)sg;
            blob[7] = (uint8_t)(p->lsb | (p->chan << 4));
            put32(blob + 8, (uint32_t)w); put32(blob + 12, (uint32_t)h);
            put32(blob + 16, (uint32_t)bits);
            put32(blob + 20, orc_crc32(blob + HEADER_BYTES, nbytes));
            put32(blob + 24, orc_crc32(blob, 24));
            kept[p->chan][p->level][p->subband][p->lsb][sg].bytes = blob;
            kept[p->chan][p->level][p->subband][p->lsb][sg].len = HEADER_BYTES + nbytes;
            used += HEADER_BYTES + nbytes;
        }
    }

    /* D7: final order  segment up, subband down, level down, plane down, channel up
     * (icer_compress.c:409-423 and :148-162, icer_color.c:508-527); the uint8 YUV variant walks subband, level and
     * plane UP instead (icer_color.c:184-202) */
    size_t off = 0;
    for (int sg = 0; sg <= MAX_SEGMENTS; sg++)
        for (int isb = 0; isb < 4; isb++)
            for (int ilv = 0; ilv <= MAX_STAGES; ilv++)
                for (int il = 0; il < PLANES; il++)
                    for (int ch = 0; ch < channels; ch++) {
                        const int sb = yuv_order_up ? isb : 3 - isb, lv = yuv_order_up ? ilv : MAX_STAGES - ilv;
                        const int lsb = yuv_order_up ? il : PLANES - 1 - il;
                        unit_blob *u = &kept[ch][lv][sb][lsb][sg];
                        if (!u->bytes) continue;
                        memcpy(out + off, u->bytes, u->len);
                        off += u->len;
                        free(u->bytes);
                        u->bytes = NULL;
                    }
    *size_used = off;
    return rc;
}

/* uint8 twins (icer_compress.c:17-166, icer_color.c:18-206): samples are int8 storage.  `planes8[c]` are mutated
 * in place like the reference does (int8 DWT, LL-mean removal modulo 256, int8 sign-magnitude :852-858). */
int orc_compress_u8(uint8_t *const planes8[], int channels, size_t w, size_t h, int stages, int filt,
                    unsigned segments, size_t quota, uint8_t *out, size_t *size_used)
{
    *size_used = 0;
    if (channels != 1 && channels != 3) return ORC_INVALID_INPUT;
    if (stages < 1 || stages > MAX_STAGES) return ORC_TOO_MANY_STAGES;
    uint16_t *wide[3] = {NULL, NULL, NULL};
    int rc = ORC_OK;
    for (int ch = 0; ch < channels; ch++) {
        wide[ch] = (uint16_t *)malloc(sizeof(uint16_t) * w * h);
        for (size_t i = 0; i < w * h; i++) wide[ch][i] = (uint16_t)(int16_t)(int8_t)planes8[ch][i];
    }
    for (int ch = 0; ch < channels && rc == ORC_OK; ch++)
        rc = dwt_stages_bits((int16_t *)wide[ch], w, h, stages, filt, 8);
    size_t llw = dim_low(w, stages), llh = dim_low(h, stages);
    uint16_t mean[3] = {0, 0, 0};
    if (rc == ORC_OK) {
        /* icer_compress.c:24-38: the LL samples are summed as uint8, the mean must fit int8 */
        for (int ch = 0; ch < channels; ch++) {
            uint64_t sum = 0;
            for (size_t r = 0; r < llh; r++)
                for (size_t c = 0; c < llw; c++) sum += (uint8_t)wide[ch][r * w + c];
            mean[ch] = (uint8_t)(sum / (llw * llh));
        }
        for (int ch = 0; ch < channels; ch++)
            if (mean[ch] > 127) rc = ORC_INTEGER_OVERFLOW;
    }
    if (rc == ORC_OK) {
        for (int ch = 0; ch < channels; ch++) {
            int16_t *s16 = (int16_t *)wide[ch];
            for (size_t r = 0; r < llh; r++)
                for (size_t c = 0; c < llw; c++) s16[r * w + c] = (int16_t)(int8_t)(s16[r * w + c] - (int16_t)(int8_t)mean[ch]);
            for (size_t i = 0; i < w * h; i++) {
                /* icer_to_sign_magnitude_int8: -128 becomes 0x80 (sign set, magnitude 0) */
                const int8_t v = (int8_t)s16[i];
                const uint8_t m = v < 0 ? (uint8_t)(0x80u | ((uint8_t)(-(int)v) & 0x7Fu)) : (uint8_t)v;
                planes8[ch][i] = m;
                wide[ch][i] = (uint16_t)(((m & 0x80u) << 8) | (m & 0x7Fu));
            }
        }
        rc = compress_core(wide, mean, channels, w, h, stages, segments, quota, out, size_used, 7, 300, channels == 3);
    }
    for (int ch = 0; ch < channels; ch++) free(wide[ch]);
    return rc;
}

/* ==========================================================================================
 * DECODER  (SURVEY.md section 8f, row "next-1": the consumer of the streams the hot path writes)
 *
 * Restates   icer_find_packet_in_bytestream          icer_compress.c:569-588
 *            icer_decode_bit and its bit readers      icer_decoding.c:12-194
 *            icer_decompress_bitplane_uint16/_uint8   icer_context_modeller.c:461-602 / :167-310
 *            icer_decompress_partition_uint16/_uint8  icer_partition.c:393-494 / :171-275
 *            icer_decompress_image_[yuv_]uint16/8     icer_compress.c:430-536 / :168-277, icer_color.c:534-663 / :208-340
 *            the inverse lifting DWT                  icer_wavelet.c:81-103, 175-191, 467-550 (int8 twin :298-383)
 * for streams made of CRC-valid packets written by an ICER encoder.  Where the reference would read or write out
 * of bounds on hostile input (packet fields used as array indices unchecked, header reads past the end of the
 * buffer, bit reads past the end of the stream) this restatement ignores the packet / reads zeros instead; those
 * cases are not part of the parity claim.
 * ---------------------------------------------------------------------------------------- */
#define ORC_DECODER_OUT_OF_DATA (-7)       /* ICER_DECODER_OUT_OF_DATA, icer.h:100 */
#define ORC_DECODED_INVALID_DATA (-8)      /* ICER_DECODED_INVALID_DATA, icer.h:101 */

/* one packet = 28-byte header + payload; NULL payload = absent */
typedef struct { const uint8_t *hdr; } packet_ref;

/* icer_find_packet_in_bytestream, icer_compress.c:569-588: first offset at which a header with the preamble, a
 * matching header CRC, a payload that fits and a matching payload CRC starts.  Returns 1 and the offset just
 * behind the packet, or 0 with *next = len. */
static int find_packet(const uint8_t *data, size_t len, const uint8_t **pkt, size_t *next)
{
    for (size_t off = 0; off < len; off++) {
        if (len - off < HEADER_BYTES) continue;                 /* (the reference reads the header regardless) */
        const uint8_t *p = data + off;
        if (p[0] != 0x5B || p[1] != 0x60) continue;
        const uint32_t hcrc = (uint32_t)p[24] | ((uint32_t)p[25] << 8) | ((uint32_t)p[26] << 16) | ((uint32_t)p[27] << 24);
        if (hcrc != orc_crc32(p, 24)) continue;
        const uint32_t bits = (uint32_t)p[16] | ((uint32_t)p[17] << 8) | ((uint32_t)p[18] << 16) | ((uint32_t)p[19] << 24);
        const size_t nbytes = (size_t)(bits / 8u) + ((bits % 8u) ? 1u : 0u);
        if (nbytes > len - off - HEADER_BYTES) continue;
        const uint32_t dcrc = (uint32_t)p[20] | ((uint32_t)p[21] << 8) | ((uint32_t)p[22] << 16) | ((uint32_t)p[23] << 24);
        if (dcrc != orc_crc32(p + HEADER_BYTES, nbytes)) continue;
        *pkt = p;
        *next = off + HEADER_BYTES + nbytes;
        return 1;
    }
    *pkt = NULL;
    *next = len;
    return 0;
}

/* entropy decoder state of one packet (icer_decoder_context_typedef, icer.h:330-340) */
#define PEND_MAX 1100
typedef struct {
    const uint8_t *bytes;       /* payload */
    size_t avail;               /* bytes readable from `bytes` up to the end of the whole stream */
    uint32_t total_bits;        /* data_length of the packet */
    size_t ind;                 /* read cursor: byte, bit */
    unsigned off;
    size_t words;               /* code words decoded so far */
    int n[17];                  /* bits pending per bin */
    size_t index[17];           /* `words` when the bin's last code word was read */
    uint8_t pend[17][PEND_MAX]; /* pending bits, served from the TOP (icer_decoding.c:186-190) */
} dcoder;

static void dcoder_init(dcoder *d, const uint8_t *payload, size_t avail, uint32_t total_bits)   /* :12-26 */
{
    memset(d, 0, sizeof *d);
    d->bytes = payload; d->avail = avail; d->total_bits = total_bits;
}
static unsigned dbyte(const dcoder *d, size_t i) { return i < d->avail ? d->bytes[i] : 0u; }

/* icer_get_bit_from_codeword :46-57: the k-th bit (1-based) ahead of the cursor, cursor unchanged */
static int peek_bit(const dcoder *d, unsigned k)
{
    const unsigned o = d->off + (k - 1);
    return (int)((dbyte(d, d->ind + o / 8) >> (o % 8)) & 1u);
}
/* icer_get_bits_from_codeword :59-82 / icer_pop_bits_from_codeword :84-105: `nb` bits LSB first, byte-sized pieces.
 * QUIRK: decoded_bits_total is never advanced in the reference, so the out-of-data test only ever compares the
 * piece size with the packet's total length. */
static int read_bits(dcoder *d, unsigned nb, int consume)
{
    int num = 0;
    unsigned got = 0, off = d->off;
    size_t ind = d->ind;
    while (nb) {
        const unsigned take = (8 - off) < nb ? (8 - off) : nb;
        if (take > d->total_bits) return ORC_DECODER_OUT_OF_DATA;
        num |= (int)(((dbyte(d, ind) >> off) & ((1u << take) - 1u)) << got);
        nb -= take; got += take; off += take;
        if (off / 8) ind++;
        off %= 8;
        if (consume) { d->off = off; d->ind = ind; }
    }
    return num;
}
static void push_bits(dcoder *d, int bin, unsigned value, unsigned nb)       /* icer_push_bin_bits :28-44 */
{
    for (unsigned k = 0; k < nb; k++) {
        if (d->n[bin] < PEND_MAX) d->pend[bin][d->n[bin]] = (uint8_t)((k < 16 ? (value >> k) : 0u) & 1u);
        d->n[bin]++;
    }
}

/* decode table of bins 1..7: code word value -> {code length, source pattern reversed, pattern length}
 * (icer_init.c:38-120: the inverse of the coding scheme, pattern stored reversed so that it pops in input order) */
static struct { uint8_t code_bits, pattern_rev, pattern_bits; } decode_lut[17][32];
static int decode_ready = 0;
static void decode_tables_init(void)
{
    if (decode_ready) return;
    coder_tables_init();
    memset(decode_lut, 0, sizeof decode_lut);
    for (int b = 1; b <= 7; b++)
        for (int k = 0; k < 9; k++) {
            const v2v e = V2V[b][k];
            if (e.in_bits == 0) continue;
            decode_lut[b][e.out_code].code_bits = e.out_bits;
            decode_lut[b][e.out_code].pattern_rev = (uint8_t)reverse_bits(e.in_val, e.in_bits);
            decode_lut[b][e.out_code].pattern_bits = e.in_bits;
        }
    decode_ready = 1;
}
/* test tap: decode table entry */
void orc_decode_entry(int bin, int code, int *code_bits, int *pattern_rev, int *pattern_bits)
{
    decode_tables_init();
    *code_bits = decode_lut[bin][code].code_bits;
    *pattern_rev = decode_lut[bin][code].pattern_rev;
    *pattern_bits = decode_lut[bin][code].pattern_bits;
}

/* icer_decode_bit, icer_decoding.c:108-194 */
static int decode_bit(dcoder *d, int *bit, uint32_t zero, uint32_t total)
{
    int inv = 0;
    if (zero < (total >> 1)) { zero = total - zero; inv = 1; }
    const int bin = orc_pick_bin(zero, total);
    /* a new code word is due when the bin has nothing pending, or when RING_WORDS code words have been read since
     * the bin's last one: the encoder's ring was full then and it force-completed that word (:128) */
    if (d->n[bin] <= 0 || d->words - d->index[bin] >= RING_WORDS) {
        d->n[bin] = 0;
        if (bin >= 8) {
            if (peek_bit(d, 1)) {                                    /* "1": a full run of m zeros */
                read_bits(d, 1, 1);
                push_bits(d, bin, 0, golomb[bin].m);
            } else {
                /* QUIRK: an out-of-data result of these reads is used as a number (only possible when the
                 * packet is shorter than one code word, which an encoder never writes) */
                uint16_t k = (uint16_t)read_bits(d, golomb[bin].l, 0);
                k = (uint16_t)reverse_bits(k, golomb[bin].l);
                if (k < golomb[bin].i) {
                    read_bits(d, golomb[bin].l, 1);
                    push_bits(d, bin, 1, 1);
                    push_bits(d, bin, 0, k);
                } else {
                    k = (uint16_t)read_bits(d, golomb[bin].l + 1u, 1);
                    k = (uint16_t)reverse_bits(k, golomb[bin].l + 1);
                    push_bits(d, bin, 1, 1);
                    push_bits(d, bin, 0, (unsigned)(uint16_t)(k - golomb[bin].i));
                }
            }
        } else if (bin >= 1) {
            unsigned code = 0, nb = 0;
            do {
                /* QUIRK: refuses a code word that ends on the packet's last bit but one or later when ... the
                 * running count is never advanced, so this only bites packets no longer than one code word */
                if (nb + 1 >= d->total_bits) return ORC_DECODER_OUT_OF_DATA;
                code |= (unsigned)peek_bit(d, nb + 1) << nb;
                nb++;
                if (code >= 32) return ORC_DECODED_INVALID_DATA;
                if (decode_lut[bin][code].code_bits == nb) {
                    push_bits(d, bin, decode_lut[bin][code].pattern_rev, decode_lut[bin][code].pattern_bits);
                    if ((int)code != read_bits(d, nb, 1)) return ORC_DECODED_INVALID_DATA;
                    break;
                }
            } while (nb < 10);
        } else {
            const int b = read_bits(d, 1, 1);
            if (b == ORC_DECODER_OUT_OF_DATA) return ORC_DECODER_OUT_OF_DATA;
            push_bits(d, bin, b != 0, 1);
        }
        d->words++;
        d->index[bin] = d->words;
    }
    /* serve the top pending bit.  QUIRK: with a multiple of 32 bits pending (or none, after a code word that matched
     * nothing) the reference shifts by -1; as compiled for x86-64 that reads a zero bit and leaves the real one in
     * place -- which is a zero of a Golomb run whenever an encoder wrote the packet */
    int b = 0;
    if ((d->n[bin] & 31) != 0 && d->n[bin] > 0 && d->n[bin] <= PEND_MAX) b = d->pend[bin][d->n[bin] - 1];
    d->n[bin]--;
    *bit = inv ? !b : b;
    return ORC_OK;
}

/* icer_decompress_bitplane_uint16 / _uint8 (icer_context_modeller.c:461-602 / :167-310): adds bit plane `lsb` of one
 * segment to the sign-magnitude words in `seg` (sign at bit `sign_bit`), which hold the planes above it. */
static int decode_unit(uint16_t *seg, size_t w, size_t h, size_t rowstride, int subband, int lsb, int sign_bit, dcoder *d)
{
    if (lsb + 1 >= sign_bit + 1) return ORC_BITPLANE_OUT_OF_RANGE;
    const unsigned mask = (1u << sign_bit) - 1u;
    uint32_t zero[17], total[17];
    for (int k = 0; k < 17; k++) { zero[k] = 2; total[k] = 4; }
#define MAG(r, cc) (seg[(r) * rowstride + (cc)] & mask)
#define NEG(r, cc) ((seg[(r) * rowstride + (cc)] >> sign_bit) & 1u)
#define SIG(r, cc, l) (((r) < 0 || (cc) < 0 || (r) >= (long)h || (cc) >= (long)w) ? 0 : ((MAG(r, cc) >> (l)) != 0))
#define SGN(r, cc, l) ((SIG(r, cc, l) && NEG(r, cc)) ? -1 : 0)
    for (long r = 0; r < (long)h; r++) {
        for (long cc = 0; cc < (long)w; cc++) {
            uint16_t *pos = &seg[r * rowstride + cc];
            const unsigned m = MAG(r, cc);
            int msb = 0;
            for (unsigned t = m | 1; t > 1; t >>= 1) msb++;
            int cat = msb < lsb ? 0 : msb - lsb;
            if (cat > 3) cat = 3;
            int bit, res;
            if (cat == 3) {
                if ((res = decode_bit(d, &bit, 1, 2)) != ORC_OK) return res;
                *pos |= (uint16_t)(bit << lsb);
                continue;
            }
            int ctx;
            if (cat == 2) ctx = 11;
            else {
                int hh = SIG(r, cc - 1, lsb) + SIG(r, cc + 1, lsb + 1);
                int vv = SIG(r - 1, cc, lsb) + SIG(r + 1, cc, lsb + 1);
                int dd = SIG(r - 1, cc - 1, lsb) + SIG(r - 1, cc + 1, lsb) + SIG(r + 1, cc - 1, lsb + 1) + SIG(r + 1, cc + 1, lsb + 1);
                if (cat == 1) ctx = (hh + vv == 0) ? 9 : 10;
                else {
                    if (subband == SB_HL) { int t = hh; hh = vv; vv = t; }
                    ctx = (subband == SB_HH) ? ctx_hh(hh + vv, dd) : ctx_plain(hh, vv, dd);
                }
            }
            if ((res = decode_bit(d, &bit, zero[ctx], total[ctx])) != ORC_OK) return res;
            *pos |= (uint16_t)(bit << lsb);
            model_update(&zero[ctx], &total[ctx], !bit);
            if (cat == 0 && bit) {
                int sh = SGN(r, cc - 1, lsb) + SGN(r, cc + 1, lsb + 1) + 2;
                int sv = SGN(r - 1, cc, lsb) + SGN(r + 1, cc, lsb + 1) + 2;
                if (subband == SB_HL) { int t = sh; sh = sv; sv = t; }
                const int sctx = SIGN_CTX[sh][sv];
                int agree;
                if ((res = decode_bit(d, &agree, zero[sctx], total[sctx])) != ORC_OK) return res;
                *pos |= (uint16_t)(((agree ^ SIGN_PRED[sh][sv]) & 1) << sign_bit);
                model_update(&zero[sctx], &total[sctx], agree == 0);
            }
        }
    }
#undef MAG
#undef NEG
#undef SIG
#undef SGN
    return ORC_OK;
}

/* icer_find_k (icer_wavelet.c:823-848): the reference's search for a k with 3^k + 1 <= len; it does not always
 * return the largest one, which is harmless, but the slices it picks decide the order below */
static unsigned shuffle_k(size_t len)
{
    unsigned lo_k = 0, hi_k = 11, res = 0;                  /* MAX_K - 1, icer.h:28; uint8_t arithmetic in the reference */
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
static void idx_reverse(size_t *a, size_t from, size_t to)
{
    while (from < to) { const size_t t = a[from]; a[from] = a[to]; a[to] = t; from++; to--; }
}
/* icer_interleave_uint16 / _uint8 (icer_wavelet.c:705-763 / :570-628) as a permutation: src[i] = position in the
 * [lows | highs] layout of the value that ends up at position i.  The reference shuffles in place (rotations by
 * reversal, then cycle leaders over slices of 3^k + 1); for the uint16 routine, and for even lengths, the result is
 * the plain interleave.  QUIRK: the uint8 routine uses a different rotation bound for ODD lengths (:614 vs :749),
 * which scrambles the line -- the reference's uint8 decoder is only usable when every level has even sides. */
static void interleave_order(size_t len, int bits, size_t *src)
{
    const int odd = (int)(len & 1);
    const size_t n = len - (size_t)odd;
    for (size_t i = 0; i < len; i++) src[i] = i;
    if (odd) {                                              /* the lone last low goes to the end */
        const size_t t = src[n / 2];
        for (size_t i = n / 2; i < n; i++) src[i] = src[i + 1];
        src[len - 1] = t;
    }
    size_t done = 0;
    while (done < n) {
        const unsigned k = shuffle_k(n - done);
        size_t slice = 1;
        for (unsigned e = 0; e < k; e++) slice *= 3;
        slice += 1;
        const size_t half = slice / 2, left = n - done;
        const size_t halfleft = left / 2 - ((bits == 8 && odd) ? 0u : 1u);
        idx_reverse(src, done + half, done + halfleft + half);
        idx_reverse(src, done + half, done + slice - 1);
        idx_reverse(src, done + slice, done + halfleft + half);
        for (size_t i = 1; i < slice; i *= 3) {
            size_t j = i, carry = src[done + j];
            do {
                j = j < half ? 2 * j : (j - half) * 2 + 1;
                const size_t t = src[done + j]; src[done + j] = carry; carry = t;
            } while (j != i);
        }
        done += slice;
    }
}

/* inverse of dwt_1d_bits (icer_wavelet.c:467-550, int8 twin :298-383): [lows | highs] -> samples.  `order` =
 * interleave_order(n, bits). */
static void idwt_1d_bits(int16_t *line, size_t n, size_t stride, int filt, int bits, const size_t *order)
{
#define TRUNC(v) (bits == 8 ? (int16_t)(int8_t)(v) : (int16_t)(v))
    const size_t nl = (n + 1) / 2, nh = n / 2;
    const int odd = (int)(n & 1);
    int16_t *lo = (int16_t *)malloc(sizeof(int16_t) * (nl + 1));
    int16_t *hi = (int16_t *)malloc(sizeof(int16_t) * (nh + 2));
    int16_t *v = (int16_t *)malloc(sizeof(int16_t) * (n + 1));
    for (size_t k = 0; k < nl; k++) lo[k] = line[k * stride];
    for (size_t k = 0; k < nh; k++) hi[k] = line[(nl + k) * stride];
    /* step 1 (:484-515): undo the prediction from the last high to the first, so hi[k + 1] is already restored
     * when hi[k] needs it -- as the forward step read it unmodified */
#define R(k) ((int32_t)(int16_t)(lo[(k) - 1] - lo[(k)]))
    const int am1 = FILT[filt][0], a0 = FILT[filt][1], a1 = FILT[filt][2], be = FILT[filt][3];
    for (size_t it = 0; it < nh; it++) {
        const size_t k = nh - 1 - it;
        int32_t add;
        if (k == 0) {
            add = floordiv(R(1), 4);
        } else if (k == 1 && am1 != 0) {
            /* QUIRK (filter C, mirror of W3): reads hi[1] itself, which still holds the transformed value here
             * while the forward step subtracted a term of the original one -- the reference's own round trip is
             * therefore not exact for filter C */
            const int32_t x = (odd && nl == 3) ? 0 : hi[1];
            add = floordiv(2 * R(1) + 3 * R(2) - 2 * x + 4, 8);
        } else if (!odd && k == nh - 1) {
            add = floordiv(R(nh - 1), 4);
        } else {
            const int32_t rm = (k >= 2) ? R(k - 1) : 1;
            const int32_t dn = (odd && k + 1 == nl - 1) ? 0 : hi[k + 1];
            add = floordiv(am1 * rm + a0 * R(k) + a1 * R(k + 1) - be * dn + 8, 16);
        }
        hi[k] = TRUNC((int32_t)hi[k] + add);
    }
#undef R
    /* step 2 (:517-545): pair (mean, difference) -> samples, still in the [lows | highs] layout; then the interleave (:547) */
    for (size_t k = 0; k < nh; k++) {
        const int32_t low = lo[k], high = hi[k];
        const int32_t a = low + floordiv(high + 1, 2);
        v[k] = TRUNC(a);
        v[nl + k] = TRUNC(a - high);
    }
    if (odd) v[nl - 1] = lo[nl - 1];
    for (size_t i = 0; i < n; i++) line[i * stride] = v[order[i]];
#undef TRUNC
    free(lo); free(hi); free(v);
}

/* icer_inverse_wavelet_transform_stages_uint16 (:81-103) over icer_inverse_wavelet_transform_2d_uint16 (:175-191):
 * deepest level first, every column of the level's region, then every row.  (Overflow is only reported through a
 * return value that the decoders ignore.) */
static void idwt_stages_bits(int16_t *s, size_t w, size_t h, int stages, int filt, int bits)
{
    if (dim_low(w, stages) < 3 || dim_low(h, stages) < 3) return;      /* ICER_TOO_MANY_STAGES, ignored by the callers */
    size_t *order = (size_t *)malloc(sizeof(size_t) * ((w > h ? w : h) + 1));
    for (int it = 1; it <= stages; it++) {
        const size_t cw = dim_low(w, stages - it), ch = dim_low(h, stages - it);
        interleave_order(ch, bits, order);
        for (size_t c = 0; c < cw; c++) idwt_1d_bits(s + c, ch, w, filt, bits, order);
        interleave_order(cw, bits, order);
        for (size_t r = 0; r < ch; r++) idwt_1d_bits(s + r * w, cw, 1, filt, bits, order);
    }
    free(order);
}

/* icer_decompress_partition_uint16 / _uint8 (icer_partition.c:393-494 / :171-275): every segment of one subband,
 * planes from the top down until one is missing or fails */
typedef const uint8_t *packet_table[MAX_STAGES + 1][4][MAX_SEGMENTS + 1][MAX_PLANES];

static void decode_subband(uint16_t *base, const orc_partition *part, size_t rowstride, const uint8_t *pk[][MAX_PLANES],
                           int planes, int sign_bit, const uint8_t *stream_end, dcoder *d)
{
    orc_rect rects[MAX_SEGMENTS + 64];
    const int nseg = orc_partition_rects(part, rects);
    for (int sg = 0; sg < nseg && sg <= MAX_SEGMENTS; sg++) {
        for (int lsb = planes - 1; lsb >= 0 && pk[sg][lsb] != NULL; lsb--) {
            const uint8_t *top = pk[sg][planes - 1], *p = pk[sg][lsb];
            const uint32_t bits = (uint32_t)p[16] | ((uint32_t)p[17] << 8) | ((uint32_t)p[18] << 16) | ((uint32_t)p[19] << 24);
            dcoder_init(d, p + HEADER_BYTES, (size_t)(stream_end - (p + HEADER_BYTES)), bits);
            /* (subband type from the top plane's header, :432-434) */
            if (decode_unit(base + (size_t)rects[sg].y * rowstride + rects[sg].x, rects[sg].w, rects[sg].h, rowstride,
                            top[5], lsb, sign_bit, d) != ORC_OK) break;
        }
    }
}

/* icer_decompress_image_uint16 / _yuv_uint16 (icer_compress.c:430-536, icer_color.c:534-663) and the uint8 twins
 * (icer_compress.c:168-277, icer_color.c:208-340) on widened samples: `planes[c]` (w*h words each) receive
 * the decoded image; *w / *h are taken from the last valid packet (left as passed in when there is none). */
static int decompress_core(uint16_t *const planes[], int channels, size_t *w, size_t *h, size_t bufsize,
                           const uint8_t *data, size_t len, int stages, int filt, unsigned segments, int bits)
{
    decode_tables_init();
    if (channels != 1 && channels != 3) return ORC_INVALID_INPUT;
    if (stages < 1 || stages > MAX_STAGES) return ORC_TOO_MANY_STAGES;       /* (reference: out-of-bounds table) */
    const int nplanes = bits == 8 ? 7 : 9, sign_bit = bits == 8 ? 7 : 15;
    static packet_table table[3];
    memset(table, 0, sizeof table);
    uint16_t mean[3] = {0, 0, 0};                   /* (YUV: a channel without packets reads an uninitialised mean) */
    for (size_t off = 0; off < len;) {
        const uint8_t *p;
        size_t next;
        if (find_packet(data + off, len - off, &p, &next)) {
            const int lv = p[4], sb = p[5], sg = p[6], lsb = p[7] & 15, ch = channels == 3 ? (p[7] >> 4) : 0;
            if (lv <= MAX_STAGES && sb < 4 && sg <= MAX_SEGMENTS && lsb < MAX_PLANES && ch < 3) table[ch][lv][sb][sg][lsb] = p;
            *w = (size_t)p[8] | ((size_t)p[9] << 8) | ((size_t)p[10] << 16) | ((size_t)p[11] << 24);
            *h = (size_t)p[12] | ((size_t)p[13] << 8) | ((size_t)p[14] << 16) | ((size_t)p[15] << 24);
            if (ch < 3) mean[ch] = (uint16_t)(p[2] | (p[3] << 8));
        }
        off += next;
    }
    if (bufsize < (*w) * (*h)) return ORC_BYTE_QUOTA_EXCEEDED;
    const size_t iw = *w, ih = *h;
    for (int c = 0; c < channels; c++) memset(planes[c], 0, sizeof(uint16_t) * iw * ih);
    dcoder *d = (dcoder *)malloc(sizeof(dcoder));
    int rc = ORC_OK;
    for (int lv = 1; lv <= stages && rc == ORC_OK; lv++) {
        for (int c = 0; c < channels && rc == ORC_OK; c++) {
            for (int pass = (lv == stages ? 0 : 1); pass < 4 && rc == ORC_OK; pass++) {
                const int sb = pass;                                     /* LL (deepest level only), HL, LH, HH */
                size_t sw, sh, ox, oy;
                switch (sb) {
                case SB_LL: sw = dim_low(iw, lv);  sh = dim_low(ih, lv);  ox = 0; oy = 0; break;
                case SB_HL: sw = dim_high(iw, lv); sh = dim_low(ih, lv);  ox = dim_low(iw, lv); oy = 0; break;
                case SB_LH: sw = dim_low(iw, lv);  sh = dim_high(ih, lv); ox = 0; oy = dim_low(ih, lv); break;
                default:    sw = dim_high(iw, lv); sh = dim_high(ih, lv); ox = dim_low(iw, lv); oy = dim_low(ih, lv); break;
                }
                orc_partition part;
                /* (unlike the encoder, P1, the decoder does stop on a grid error) */
                if ((rc = orc_partition_make(&part, sw, sh, segments)) != ORC_OK) break;
                decode_subband(planes[c] + oy * iw + ox, &part, iw, table[c][lv][sb], nplanes, sign_bit, data + len, d);
            }
        }
    }
    free(d);
    if (rc != ORC_OK) return rc;
    const size_t llw = dim_low(iw, stages), llh = dim_low(ih, stages);
    for (int c = 0; c < channels; c++) {
        int16_t *s = (int16_t *)planes[c];
        /* icer_from_sign_magnitude_int16 / _int8 (icer_wavelet.c:880-886 / :860-866); sign with zero magnitude -> 0 */
        for (size_t i = 0; i < iw * ih; i++) {
            const unsigned v = planes[c][i];
            const int neg = (v >> sign_bit) & 1u, mag = (int)(v & ((1u << sign_bit) - 1u));
            s[i] = (int16_t)(neg ? -mag : mag);
        }
        /* LL mean back in, modulo the sample width (icer_compress.c:522-531) */
        for (size_t r = 0; r < llh; r++)
            for (size_t cc = 0; cc < llw; cc++) {
                if (bits == 8) s[r * iw + cc] = (int16_t)(int8_t)(s[r * iw + cc] + (int8_t)mean[c]);
                else s[r * iw + cc] = (int16_t)(s[r * iw + cc] + (int16_t)mean[c]);
            }
        idwt_stages_bits(s, iw, ih, stages, filt, bits);
        for (size_t i = 0; i < iw * ih; i++)                                 /* icer_remove_negative_*, icer_util.c:70-91 */
            if (s[i] < 0) s[i] = 0;
    }
    return ORC_OK;
}

int orc_decompress_u16(uint16_t *const planes[], int channels, size_t *w, size_t *h, size_t bufsize,
                       const uint8_t *data, size_t len, int stages, int filt, unsigned segments)
{
    return decompress_core(planes, channels, w, h, bufsize, data, len, stages, filt, segments, 16);
}

int orc_decompress_u8(uint8_t *const planes8[], int channels, size_t *w, size_t *h, size_t bufsize,
                      const uint8_t *data, size_t len, int stages, int filt, unsigned segments)
{
    if (channels != 1 && channels != 3) return ORC_INVALID_INPUT;
    /* the image size is only known after the packet scan: widen into scratch planes of the buffer's size */
    uint16_t *wide[3] = {NULL, NULL, NULL};
    size_t cap = bufsize;
    for (int c = 0; c < channels; c++) wide[c] = (uint16_t *)calloc(cap ? cap : 1, sizeof(uint16_t));
    const int rc = decompress_core(wide, channels, w, h, bufsize, data, len, stages, filt, segments, 8);
    if (rc != ORC_BYTE_QUOTA_EXCEEDED && rc != ORC_INVALID_INPUT && rc != ORC_TOO_MANY_STAGES)
        for (int c = 0; c < channels; c++)
            for (size_t i = 0; i < (*w) * (*h) && i < cap; i++) planes8[c][i] = (uint8_t)wide[c][i];
    for (int c = 0; c < channels; c++) free(wide[c]);
    return rc;
}
