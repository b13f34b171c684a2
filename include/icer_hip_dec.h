/*
 * icer_hip_dec.h -- C ABI of libicer_hip_dec.so, the MI355X (gfx950) ICER *decoder* (SURVEY.md 8f, row next-1).
 *
 * STATUS: bit-exact against the decoder oracle in its CPU builds (tests/test_emu_decoder.py) and on an MI355X
 * (tests/test_gpu_decoder.py: gray / YUV, 16 / 8 bit, damaged and truncated streams, wrong decode parameters, the golden
 * decoder digests up to 4096 x 4096, the batch object, all three decode kernels; the reference-held fixtures and the
 * reference's own example programs linked against this library, tests/test_gpu_parity.py / test_gpu_examples.py).
 * Speed (round 4, bench.py `decode` object and tools/decode_bench.py; HISTORY.md 6b (summary: DESIGN.md 8)): a chain (segment of a subband) is a serial
 * adaptive decode, one decision at a time per bit plane.  One 4096 x 4096 headline stream: 55 Mpix/s (303 ms; one wavefront per
 * bit plane with wave-uniform decisions, decoder_planes.hpp) = 14 x the reference decoder on one core of the same box; the
 * kernels around the chains (payload CRCs, inverse DWT, sample post-processing) take 0.6 ms together.  Batches (chains of all
 * streams launched longest first): 4 streams per call 220 Mpix/s, 8: 437 (eight streams in the time of one), 16: 586, 24: 686,
 * 32: 922, 64: 890, 128: 1 036 -- beyond twelve chains per compute unit the lane-per-plane kernel of decoder_wave.hpp takes
 * over, launched once per size class of row ring (chosen per call; ICER_DEC_WAVE=0|1|2 pins a kernel, ICER_DEC_PLANES_PER_CU
 * moves the cross-over, ICER_DEC_ORDER=0 keeps stream order).  A separate library, so that libicer_hip.so (the measured
 * encoder) is unaffected.
 *
 * Same names, argument meaning and return codes as the decoding entry points of lib_icer
 * (TheRealOrange/icer_compression, lib_icer/inc/icer.h); the work runs on the GPU and there is no CPU fallback
 * (without a usable HIP device the image functions return ICER_FATAL_ERROR and print the reason to stderr).
 * Results equal the reference's for streams made of CRC-valid packets; where the reference reads memory it does not
 * own (bits behind the end of the stream, packet fields used as indices unchecked, the mean of a YUV channel without
 * packets) this library reads zeros / ignores the packet / uses 0.
 */
#ifndef ICER_HIP_DEC_H
#define ICER_HIP_DEC_H

#include "icer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* replaces icer_get_image_dimensions, icer.h:374 (lib_icer/src/icer_compress.c:541-566): size fields of the first
 * CRC-valid packet.  Host only.  ICER_INVALID_INPUT for null arguments, ICER_DECODER_OUT_OF_DATA when no packet is found. */
int icer_get_image_dimensions(const uint8_t *datastream, size_t data_length, size_t *image_w, size_t *image_h);

/* replaces icer_decompress_image_uint16, icer.h:461-462 (lib_icer/src/icer_compress.c:430-536).  `image` (host memory,
 * image_bufsize samples) receives the decoded image; *image_w / *image_h are set from the stream.  `stages`, `filt`
 * and `segments` must be the ones the stream was made with.  ICER_BYTE_QUOTA_EXCEEDED when the buffer is too small,
 * ICER_TOO_MANY_SEGMENTS when a subband is too small for the segment grid (the image then holds the sign-magnitude
 * words decoded up to that point, as in the reference). */
int icer_decompress_image_uint16(uint16_t *image, size_t *image_w, size_t *image_h, size_t image_bufsize,
                                 const uint8_t *datastream, size_t data_length, uint8_t stages,
                                 enum icer_filter_types filt, uint8_t segments);

/* replaces icer_decompress_image_yuv_uint16, icer.h:463-465 (lib_icer/src/icer_color.c:534-663) */
int icer_decompress_image_yuv_uint16(uint16_t *y_channel, uint16_t *u_channel, uint16_t *v_channel, size_t *image_w,
                                     size_t *image_h, size_t image_bufsize, const uint8_t *datastream,
                                     size_t data_length, uint8_t stages, enum icer_filter_types filt, uint8_t segments);

/* replace icer_decompress_image_uint8 / icer_decompress_image_yuv_uint8, icer.h:407-411
 * (lib_icer/src/icer_compress.c:168-277, icer_color.c:208-340): int8 storage, 7 bit planes */
int icer_decompress_image_uint8(uint8_t *image, size_t *image_w, size_t *image_h, size_t image_bufsize,
                                const uint8_t *datastream, size_t data_length, uint8_t stages,
                                enum icer_filter_types filt, uint8_t segments);
int icer_decompress_image_yuv_uint8(uint8_t *y_channel, uint8_t *u_channel, uint8_t *v_channel, size_t *image_w,
                                    size_t *image_h, size_t image_bufsize, const uint8_t *datastream,
                                    size_t data_length, uint8_t stages, enum icer_filter_types filt, uint8_t segments);

/* ---- Part 2: batches and device-resident buffers (our extension; what a decode benchmark times) -------------------
 * One decoder per (channels, stages, filter, segments, sample width); its device buffers grow on demand and are kept.
 * Every frame's image, size and return code equal those of a per-frame call of the Part-1 function. */
typedef struct icerx_decoder icerx_decoder;

/* device < 0: the current HIP device.  sample_bits: 16 or 8. */
int icerx_decoder_create(icerx_decoder **out, int device, int channels, int stages, int filt, unsigned segments,
                         int sample_bits);
void icerx_decoder_destroy(icerx_decoder *dec);

/* n streams in one host buffer: stream k = data[offsets[k] .. offsets[k] + lens[k]).  Frame k's channel c is written to
 * planes_out[k * channels + c] (host memory, frame_stride samples each: uint16 or uint8 by sample_bits).  Per frame:
 * rcs[k] = the Part-1 return code, ws[k] / hs[k] = the image size (in: the values kept when the stream holds no valid
 * packet).  Returns ICER_RESULT_OK, or ICER_FATAL_ERROR / ICER_INVALID_INPUT for the call as a whole. */
int icerx_decode_host(icerx_decoder *dec, int n, const uint8_t *data, const size_t *offsets, const size_t *lens,
                      void *const *planes_out, size_t frame_stride, int *rcs, size_t *ws, size_t *hs);

/* the same with the streams and the images in device memory: frame k's channel c at
 * d_out + (k * channels + c) * frame_stride samples; only its first ws[k] * hs[k] samples are results.  Synchronous. */
int icerx_decode_device(icerx_decoder *dec, int n, const void *d_data, const size_t *offsets, const size_t *lens,
                        void *d_out, size_t frame_stride, int *rcs, size_t *ws, size_t *hs);

/* last error message of this thread's most recent failing call ("" if none) */
const char *icerx_decoder_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
