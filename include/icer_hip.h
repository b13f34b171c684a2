/*
 * icer_hip.h -- C ABI of libicer_hip.so, the MI355X (gfx950) ICER encoder.
 *
 * Part 1 restates, with identical names, argument meaning, struct layout and return codes, the
 * entry points of lib_icer that an application needs for ENCODING: the uint16 path and its uint8 twins
 * (TheRealOrange/icer_compression, lib_icer/inc/icer.h).  A program written against lib_icer links
 * against libicer_hip.so instead of libicer.a and produces byte-identical streams; the work runs
 * on the GPU (there is no CPU fallback: without a usable HIP device every compress call returns
 * ICER_FATAL_ERROR and prints the reason to stderr).
 *
 * Part 2 (prefix icerx_) is our extension for batches of frames and device-resident buffers,
 * which is what bench.py measures.  Each frame's stream, length and return code equal those of a
 * per-frame call of the Part-1 function on the same data.
 *
 * Plain C, LP64; no HIP or torch types appear in any signature (streams and device pointers are
 * passed as void*).
 */
#ifndef ICER_HIP_H
#define ICER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Part 1: lib_icer drop-in surface ------------------------------------------------------ */

/* enum icer_status, lib_icer/inc/icer.h:92-105 (same numeric values) */
enum icer_status {
    ICER_RESULT_OK = 0,
    ICER_INTEGER_OVERFLOW = -1,
    ICER_OUTPUT_BUF_TOO_SMALL = -2,
    ICER_TOO_MANY_SEGMENTS = -3,
    ICER_TOO_MANY_STAGES = -4,
    ICER_BYTE_QUOTA_EXCEEDED = -5,
    ICER_BITPLANE_OUT_OF_RANGE = -6,
    ICER_DECODER_OUT_OF_DATA = -7,
    ICER_DECODED_INVALID_DATA = -8,
    ICER_PACKET_COUNT_EXCEEDED = -9,
    ICER_FATAL_ERROR = -10,
    ICER_INVALID_INPUT = -11
};

/* enum icer_filter_types, lib_icer/inc/icer.h:107-115 */
enum icer_filter_types {
    ICER_FILTER_A = 0, ICER_FILTER_B, ICER_FILTER_C, ICER_FILTER_D, ICER_FILTER_E, ICER_FILTER_F, ICER_FILTER_Q
};

/* icer_output_data_buf_typedef, lib_icer/inc/icer.h:307-312 (32 bytes, same layout) */
typedef struct {
    size_t size_used;            /* out: length of the final stream at rearrange_start */
    size_t size_allocated;       /* byte quota */
    uint8_t *data_start;         /* staging half [0, quota): scratch, contents unspecified */
    uint8_t *rearrange_start;    /* final stream */
} icer_output_data_buf_typedef;

/* replaces icer_init, lib_icer/inc/icer.h:370 (lib_icer/src/icer_init.c:24-35): builds the coder
 * tables.  Idempotent.  Does not touch the GPU. */
int icer_init(void);

/* replaces icer_init_output_struct, icer.h:524 (lib_icer/src/icer_util.c:38-45).
 * Returns ICER_OUTPUT_BUF_TOO_SMALL when 2*byte_quota > buf_len. */
int icer_init_output_struct(icer_output_data_buf_typedef *out, uint8_t *data, size_t buf_len, size_t byte_quota);

/* replaces icer_compress_image_uint16, icer.h:440-441 (lib_icer/src/icer_compress.c:279-426).
 * `image` (host memory, w*h uint16, row-major) is overwritten with the sign-magnitude wavelet
 * coefficients exactly as the reference leaves it (written back while the coder is still running: of the
 * call's three transfers only the upload and the stream download are not hidden; bench.py `dropin`).  Returns ICER_RESULT_OK or
 * ICER_BYTE_QUOTA_EXCEEDED with a valid stream in output_data->rearrange_start[0..size_used),
 * or an error code with size_used == 0. */
int icer_compress_image_uint16(uint16_t *image, size_t image_w, size_t image_h, uint8_t stages,
                               enum icer_filter_types filt, uint8_t segments,
                               icer_output_data_buf_typedef *output_data);

/* replaces icer_compress_image_yuv_uint16, icer.h:442-444 (lib_icer/src/icer_color.c:343-530). */
int icer_compress_image_yuv_uint16(uint16_t *y_channel, uint16_t *u_channel, uint16_t *v_channel,
                                   size_t image_w, size_t image_h, uint8_t stages,
                                   enum icer_filter_types filt, uint8_t segments,
                                   icer_output_data_buf_typedef *output_data);

/* replaces icer_compress_image_uint8, icer.h:386-387 (lib_icer/src/icer_compress.c:17-166).  The uint8 twins treat
 * the samples as int8 STORAGE (a pixel value above 127 is a negative number, lib_icer/src/icer_wavelet.c:231) and code
 * 7 bit planes: only data of at most 7 bits survives the round trip, anything that leaves the int8 range during the
 * transform makes the call return ICER_INTEGER_OVERFLOW exactly as the reference does.  `image` (host memory, w*h
 * bytes) is overwritten with the int8 sign-magnitude wavelet coefficients; on ICER_INTEGER_OVERFLOW its contents are
 * left untouched (the reference leaves partially transformed data there).  ICER_PACKET_COUNT_EXCEEDED is returned for
 * (3*stages+1)*7*channels >= 300 packets as in the reference, without touching the image. */
int icer_compress_image_uint8(uint8_t *image, size_t image_w, size_t image_h, uint8_t stages,
                              enum icer_filter_types filt, uint8_t segments,
                              icer_output_data_buf_typedef *output_data);

/* replaces icer_compress_image_yuv_uint8, icer.h:388-390 (lib_icer/src/icer_color.c:18-206); note that its final
 * re-ordering walks subbands, levels and bit planes upwards, unlike the other three entry points. */
int icer_compress_image_yuv_uint8(uint8_t *y_channel, uint8_t *u_channel, uint8_t *v_channel,
                                  size_t image_w, size_t image_h, uint8_t stages,
                                  enum icer_filter_types filt, uint8_t segments,
                                  icer_output_data_buf_typedef *output_data);

/* ---- Part 2: batched / device-resident extension -------------------------------------------- */

typedef struct icerx_encoder icerx_encoder;

/* Create an encoder for frames of w x h with `channels` (1 = gray, 3 = Y,U,V planes) on HIP device
 * `device`.  All device memory for up to `max_frames` frames per call is allocated here.
 * Returns 0, a (negative) icer_status the reference would return for this geometry
 * (ICER_TOO_MANY_STAGES, ...), or ICER_FATAL_ERROR when no usable device exists.
 * Memory: about 14 bytes per sample and frame (coefficients, the stage-to-stage LL buffer, one event byte per sample and bit plane) plus
 * the coding units' slots.  Streams: an encoder owns a SIDE stream (a small kernel runs beside the main coder kernel of every launch)
 * -- a HIGH-priority stream unless GPU_MAX_HW_QUEUES >= 6 (it must not share the caller's hardware queue: INTEGRATION.md "Hardware
 * queues"; its kernels therefore go ahead of the program's normal-priority ones; ICER_HIP_STREAM_PRIO=0 makes it a plain stream) -- and,
 * with max_frames >= 4, a second plain stream for the second half of a synchronous batch call (icerx_encoder_parts). */
int icerx_encoder_create(icerx_encoder **enc, int device, size_t w, size_t h, int channels, int stages,
                         int filt, int segments, int max_frames);
/* The same with the sample width: sample_bits = 16 (as icerx_encoder_create) or 8 for the uint8 twins. */
int icerx_encoder_create_ex(icerx_encoder **enc, int device, size_t w, size_t h, int channels, int stages,
                            int filt, int segments, int max_frames, int sample_bits);
void icerx_encoder_destroy(icerx_encoder *enc);

/* Encode n_frames frames that already live in device memory.
 *   d_frames   device pointer, n_frames * channels planes of w*h uint16 (frame-major, then channel);
 *              not modified
 *   byte_quota per-frame byte quota (the reference's icer_output_data_buf_typedef.size_allocated)
 *   d_out      device pointer, n_frames * out_stride bytes; frame f's stream starts at f*out_stride
 *              (out_stride >= byte_quota)
 *   d_sizes    device pointer, n_frames uint64: stream lengths
 *   d_rcs      device pointer, n_frames int32: per-frame reference return codes
 *   stream     hipStream_t (as void*), NULL = default stream.  All work is enqueued on it; the call returns
 *              after it has completed there (it has to read back one word: whether a coding unit outgrew
 *              its provisioned slot, in which case the batch is redone with larger slots).
 * Returns 0 or ICER_FATAL_ERROR (HIP failure) / ICER_INVALID_INPUT. */
int icerx_encode_device(icerx_encoder *enc, const uint16_t *d_frames, int n_frames, size_t byte_quota,
                        uint8_t *d_out, size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream);

/* The same call in two halves.  icerx_encode_device_async returns as soon as all work is enqueued on `stream`;
 * icerx_encoder_wait returns once it has completed there (re-running the batch in the rare cases the synchronous call
 * does: a coding unit that outgrew its slot, a unit time-out).  Between the two the caller may enqueue its own copies
 * on other streams or drive other encoders; the buffers passed in must stay valid and untouched until the wait returns.
 * At most one pending call per encoder (a second async call, or a synchronous one, returns ICER_INVALID_INPUT);
 * icerx_encoder_wait without a pending call returns 0. */
int icerx_encode_device_async(icerx_encoder *enc, const uint16_t *d_frames, int n_frames, size_t byte_quota,
                              uint8_t *d_out, size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream);
int icerx_encoder_wait(icerx_encoder *enc);

/* Front-end fusion: inputs as the reference's callers hold them BEFORE their app-side conversion,
 * converted on the device, so only 1 byte per sample crosses PCIe.
 *   icerx_encode_device_u8    8-bit gray frames (n_frames * w*h bytes), widened to the uint16 planes the uint16 API
 *                             takes -- what example/src/icer_util.c:163-168 does on the host.  channels must be 1.
 *   icerx_encode_device_rgb8  packed RGB888 frames (n_frames * w*h*3 bytes), converted to Y, Cb, Cr planes with the
 *                             integer formulas of the reference's callers (rgb888_packed_to_yuv,
 *                             example/src/icer_util.c:69-94 with CRGB2Y/Cb/Cr of example/inc/color_util.h:27-29).
 *                             channels must be 3.
 * Same outputs and return values as icerx_encode_device on the converted planes. */
int icerx_encode_device_u8(icerx_encoder *enc, const uint8_t *d_frames, int n_frames, size_t byte_quota, uint8_t *d_out,
                           size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream);
int icerx_encode_device_rgb8(icerx_encoder *enc, const uint8_t *d_rgb, int n_frames, size_t byte_quota, uint8_t *d_out,
                             size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream);

/* uint8 twins on device-resident planes (encoder created with sample_bits = 8): n_frames * channels planes of w*h
 * bytes (int8 storage), otherwise as icerx_encode_device. */
int icerx_encode_device_s8(icerx_encoder *enc, const uint8_t *d_planes, int n_frames, size_t byte_quota, uint8_t *d_out,
                           size_t out_stride, uint64_t *d_sizes, int32_t *d_rcs, void *stream);

/* Host-buffer convenience wrapper: H2D, icerx_encode_device, D2H, synchronous.  `out_stride` is the room of every
 * frame's row in `out`: a stream longer than that is not copied and the call returns ICER_OUTPUT_BUF_TOO_SMALL (sizes /
 * rcs are valid then); byte_quota or more always suffices. */
int icerx_encode_host(icerx_encoder *enc, const uint16_t *frames, int n_frames, size_t byte_quota,
                      uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs);

/* A batch of frames of one geometry from host memory over the GPUs of the node (BASELINE configs 4 and 5): contiguous
 * blocks of frames per device, one host thread and one encoder per device, no communication between devices.  Every
 * device codes its block in sub-batches through three streams -- upload of sub-batch k+1, kernels of k, download of the
 * streams of k-1 at the same time -- so page-lock `frames` and `out` (icerx_pin_host) to let the copies run as DMA beside
 * the kernels.  The per-device encoders and staging buffers are kept between calls (re-made when the geometry changes);
 * icerx_batch_release frees them.
 * frames: n_frames x channels planes of w*h uint16; out: n_frames rows of out_stride bytes (a stream longer than
 * out_stride makes the call fail with ICER_OUTPUT_BUF_TOO_SMALL; byte_quota always suffices);
 * n_gpus: devices to use (0 = all present; clamped to the number present and to n_frames) -- devices 0 .. n_gpus-1; the
 * _devices variant names them (one process per GPU: pass that process's device).  sizes / rcs per frame equal a
 * per-frame call of icer_compress_image_[yuv_]uint16 (reference: icer.h:440-444).  Returns 0, or the first failing
 * device's error code (icerx_last_error lists every failing device).
 * Env: ICER_HIP_BATCH_SUB=<frames per sub-batch>, ICER_HIP_BATCH_RAMP=<0|1|2> (smaller sub-batches at the start / and the
 * end of a block; 0 = off is the default), ICER_HIP_NUMA=0 (do not pin the per-device host threads to their GPU's NUMA node).
 * The pipeline keeps six streams per device busy.  With GPU_MAX_HW_QUEUES=8 exported before the process initialises HIP they
 * are plain streams (0.94 x the device-resident rate); with fewer hardware queues the encoders' streams are created at the low
 * priority level -- a queue pool of their own: the same 0.94 x whatever else the process has alive, and the program's own kernels
 * go first (ICER_HIP_STREAM_PRIO=0|1 pins the choice; the library does not touch the environment) -- INTEGRATION.md "Hardware queues".
 *
 * icerx_device_count(): devices the batch calls and icerx_encoder_create accept (0 .. count-1).  ICER_HIP_VIRTUAL_DEVICES=<N>
 * makes that N LOGICAL devices mapped round-robin onto the physical ones: a dry run of every multi-device code path (N host
 * threads, N pooled pipelines, error aggregation) on a node with fewer GPUs; streams are unaffected. */
int icerx_device_count(void);
int icerx_compress_batch_uint16(const uint16_t *frames, int n_frames, size_t w, size_t h, int channels, int stages, int filt,
                                int segments, size_t byte_quota, uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs,
                                int n_gpus);
int icerx_compress_batch_uint16_devices(const uint16_t *frames, int n_frames, size_t w, size_t h, int channels, int stages, int filt,
                                        int segments, size_t byte_quota, uint8_t *out, size_t out_stride, uint64_t *sizes, int32_t *rcs,
                                        const int *devices, int n_devices);
void icerx_batch_release(void);

/* Optional: page-lock a caller buffer that icerx_encode_host / the lib_icer-shaped entry points read frames from or
 * write streams to, so that it crosses PCIe by DMA at link speed (otherwise the runtime stages pageable memory through
 * its own pinned buffers, about 4x slower).  Unpin before freeing the memory.  Returns 0 or ICER_FATAL_ERROR. */
int icerx_pin_host(void *ptr, size_t bytes);
int icerx_unpin_host(void *ptr);

/* Copy the sign-magnitude coefficient plane of (frame, channel) of the last encode to host
 * memory (what the reference leaves in the caller's image buffer). */
int icerx_get_coefficients(icerx_encoder *enc, int frame, int channel, uint16_t *dst);

/* Kernel timing with HIP events on the encode stream.  When enabled, every icerx_encode_device
 * call records events around each pipeline stage; icerx_timing_read synchronises and accumulates.
 * stage ids: 0 = DWT (all stages), 1 = LL mean + sign-magnitude, 2 = coding units (dominant),
 * 3 = quota scan + stream gather.  ms[i] = accumulated milliseconds, calls = number of encodes. */
#define ICERX_NUM_STAGES 4
int icerx_timing_enable(icerx_encoder *enc, int on);
int icerx_timing_read(icerx_encoder *enc, double ms[ICERX_NUM_STAGES], uint64_t *calls, int reset);

/* Number of coding units per frame and the bits-per-pixel slot bound currently in use. */
int icerx_info(icerx_encoder *enc, uint32_t *units_per_frame, uint32_t *slot_bits_per_pixel, uint64_t *slot_bytes_per_frame);

/* Event counters of an encoder since its creation: out[0] = coding units that gave up waiting for a hand-off of the
 * eight-wave pipeline (bounded spins; expected 0), out[1] = batches coded again by the barrier-only workgroup coder
 * because of that (the caller still gets its result), out[2] = batches re-run with larger per-unit slots,
 * out[3] = coder selection in force (0 automatic, 1 pipeline only, 2 workgroup coder only; env ICER_HIP_CODER=pipe|wg).
 * Automatic: the wave pipeline; the workgroup coder for byte quotas below half a byte per sample (progressive mode);
 * in launches of two or more planes (frames x channels), and of one gray frame whose dense units are cut into sub-ranges
 * (icerx_encoder_launch_info), the coding units with >= 95 % (90 %) blank chunks go to a small instance of the workgroup coder
 * (one wave per workgroup in a batch, four for a lone frame; ICER_HIP_LIST_WAVES=1|2|4), which runs beside the pipeline kernel (env ICER_HIP_HYBRID=<percent, 0 = off>, ICER_HIP_HYBRID_FRAMES=<n>).
 * None of this changes a byte of the streams. */
int icerx_encoder_stats(icerx_encoder *enc, uint64_t out[4]);
/* out[0] = coding units (summed over frames and calls) that went to the workgroup coder's small instance beside the
 * pipeline kernel, out[1] = encode calls in which that routing was active (see above: launches of >= 2 planes). */
int icerx_encoder_routing(icerx_encoder *enc, uint64_t out[2]);
/* The shape of the encoder's last launch: out[0] = 1 if its dense coding units were cut into sub-ranges coded by a workgroup
 * each (launches of a single gray frame: the chip has more compute units than such a frame has large units;
 * env ICER_HIP_SPLIT=<chunks per sub-range, 0 = off>), out[1] = those extra workgroups, out[2] = wavefronts per workgroup of
 * the pipeline kernel (8 or 11; 0: the workgroup coder alone), out[3] = 1 if all-but-blank units went to the window coder
 * beside it.  None of this changes a byte of the streams. */
int icerx_encoder_launch_info(icerx_encoder *enc, uint32_t out[4]);
/* Parts the encoder's last call was enqueued in.  A SYNCHRONOUS batch call of four frames or more (icerx_encode_device and the _u8 / _rgb8 /
 * _s8 twins; not progressive mode) enqueues its frames in two parts, the second on a stream of the encoder's own that starts behind
 * whatever `stream` holds and is joined back before the call's own work on `stream` ends: a part's transform and event pass run beside
 * the other part's coder kernels.  The asynchronous calls enqueue one part (their caller overlaps whole batches).  env
 * ICER_HIP_OVERLAP_PARTS=<1..4> (1: off).  None of this changes a byte of the streams. */
int icerx_encoder_parts(icerx_encoder *enc);
/* out[0..2] summed over all encoders of the process, including the one behind the lib_icer-shaped entry points */
int icerx_process_stats(uint64_t out[4]);

const char *icerx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ICER_HIP_H */
