/*
 * icer_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the ICER *encoder* hot path of TheRealOrange/icer_compression
 * (lib_icer), and of its decoder (the consumer of those streams).  It exists to check the MI355X/HIP product path; nothing in the product may
 * include, link or call it (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py compares every function here with the
 * untouched reference compiled from /root/reference into oracle/_ref/libicer_ref.so, and
 * tests/golden/ holds stream digests generated from that reference build.
 */
#ifndef ICER_ORACLE_H
#define ICER_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes, numerically equal to enum icer_status (lib_icer/inc/icer.h:92-105) */
enum {
    ORC_OK = 0, ORC_INTEGER_OVERFLOW = -1, ORC_OUTPUT_BUF_TOO_SMALL = -2, ORC_TOO_MANY_SEGMENTS = -3,
    ORC_TOO_MANY_STAGES = -4, ORC_BYTE_QUOTA_EXCEEDED = -5, ORC_BITPLANE_OUT_OF_RANGE = -6,
    ORC_PACKET_COUNT_EXCEEDED = -9, ORC_FATAL_ERROR = -10, ORC_INVALID_INPUT = -11
};

/* segment grid of one subband (icer.h:126-142 / icer_partition.c:7-54), same field order */
typedef struct {
    uint16_t w, h, r, c, r_t, h_t, x_t, c_t0, y_t, r_t0, x_b, c_b0, y_b, r_b0, s;
} orc_partition;

/* one rectangle of the grid, in subband-local coordinates */
typedef struct { uint32_t x, y, w, h; } orc_rect;

typedef struct {
    uint8_t level, subband, lsb, chan;
    uint64_t priority;
} orc_packet;

int      orc_dwt_1d(int16_t *line, size_t n, size_t stride, int filt);
int      orc_dwt_stages_u16(uint16_t *img, size_t w, size_t h, int stages, int filt);
void     orc_sign_magnitude(uint16_t *data, size_t len);
int      orc_partition_make(orc_partition *p, size_t w, size_t h, unsigned segments);
int      orc_partition_rects(const orc_partition *p, orc_rect *rects /* [p->s] */);
int      orc_packet_list(orc_packet *out /* [>= (3*stages+1)*9*channels] */, int stages, int channels);
long     orc_code_unit(const uint16_t *seg, size_t w, size_t h, size_t rowstride,
                       int subband, int lsb, uint8_t *out, size_t out_cap);
uint32_t orc_crc32(const uint8_t *buf, size_t len);
int      orc_pick_bin(uint32_t zero, uint32_t total);

/* Whole-frame encoders.  `planes[c]` are mutated in place exactly like the reference
 * (DWT, LL-mean removal, sign-magnitude).  The final stream (rearranged order) is written to
 * `out` (capacity >= quota), its length to *size_used.  Return value = reference return code. */
int orc_compress_u16(uint16_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                     unsigned segments, size_t quota, uint8_t *out, size_t *size_used);

/* uint8 twins (icer_compress_image_uint8 / icer_compress_image_yuv_uint8): int8 storage, 7 bit planes */
int orc_compress_u8(uint8_t *const planes[], int channels, size_t w, size_t h, int stages, int filt,
                    unsigned segments, size_t quota, uint8_t *out, size_t *size_used);

/* Whole-frame DECODERS (icer_decompress_image_[yuv_]uint16 / _uint8, icer.h:447-466): the stream's packets ->
 * `planes[c]` (each >= bufsize samples).  *w / *h are set from the stream (and are inputs when it holds no valid
 * packet, as in the reference).  Return value = reference return code. */
int orc_decompress_u16(uint16_t *const planes[], int channels, size_t *w, size_t *h, size_t bufsize,
                       const uint8_t *data, size_t len, int stages, int filt, unsigned segments);
int orc_decompress_u8(uint8_t *const planes[], int channels, size_t *w, size_t *h, size_t bufsize,
                      const uint8_t *data, size_t len, int stages, int filt, unsigned segments);
void orc_decode_entry(int bin, int code, int *code_bits, int *pattern_rev, int *pattern_bits);

#ifdef __cplusplus
}
#endif
#endif
