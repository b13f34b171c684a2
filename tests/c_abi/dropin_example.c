/*
 * dropin_example.c -- a plain C program written against the lib_icer call sequence
 * (icer_init / icer_init_output_struct / icer_compress_image_uint16 or _yuv_uint16 / write
 * rearrange_start[0..size_used), cf. example/src/example_encode.c:36-77 of the reference), compiled with
 * gcc against include/icer_hip.h and linked with libicer_hip.so.  Used by tests/test_gpu_parity.py to show
 * that the C ABI is a drop-in from C, not only through ctypes.
 *
 * usage: dropin_example <in.raw> <w> <h> <channels 1|3> <stages> <filter 0..6> <segments> <quota> <out.bin> <out_coef.raw>
 *   in.raw        channels planes of w*h little-endian uint16
 *   out.bin       the compressed stream
 *   out_coef.raw  the planes as the call left them (sign-magnitude coefficients)
 * exit code: 0 on ICER_RESULT_OK / ICER_BYTE_QUOTA_EXCEEDED (like the reference CLI), 100 - rc otherwise.
 */
#include <stdio.h>
#include <stdlib.h>

#include "icer_hip.h"

int main(int argc, char **argv)
{
    if (argc != 11) { fprintf(stderr, "bad usage\n"); return 2; }
    const size_t w = strtoul(argv[2], 0, 10), h = strtoul(argv[3], 0, 10);
    const int channels = atoi(argv[4]), stages = atoi(argv[5]), filt = atoi(argv[6]), segments = atoi(argv[7]);
    const size_t quota = strtoul(argv[8], 0, 10);
    uint16_t *planes = malloc(w * h * channels * sizeof(uint16_t));
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(planes, 2, w * h * channels, f) != w * h * channels) { fprintf(stderr, "cannot read input\n"); return 2; }
    fclose(f);

    icer_init();
    const size_t buf_len = quota * 2 + 50;                 /* like example/src/icer_util.c:182-186 */
    uint8_t *datastream = malloc(buf_len);
    icer_output_data_buf_typedef output;
    if (icer_init_output_struct(&output, datastream, buf_len, quota) != ICER_RESULT_OK) return 3;

    int rc;
    if (channels == 1)
        rc = icer_compress_image_uint16(planes, w, h, (uint8_t)stages, (enum icer_filter_types)filt, (uint8_t)segments, &output);
    else
        rc = icer_compress_image_yuv_uint16(planes, planes + w * h, planes + 2 * w * h, w, h, (uint8_t)stages,
                                            (enum icer_filter_types)filt, (uint8_t)segments, &output);
    printf("rc=%d size_used=%zu\n", rc, output.size_used);
    if (rc != ICER_RESULT_OK && rc != ICER_BYTE_QUOTA_EXCEEDED) return 100 - rc;

    f = fopen(argv[9], "wb");
    fwrite(output.rearrange_start, 1, output.size_used, f);
    fclose(f);
    f = fopen(argv[10], "wb");
    fwrite(planes, 2, w * h * channels, f);
    fclose(f);
    free(datastream);
    free(planes);
    return rc == ICER_RESULT_OK ? 0 : 5;
}
