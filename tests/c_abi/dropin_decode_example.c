/* A program written against lib_icer's DECODING call sequence (example/src/example_decode.c in the reference:
 * icer_get_image_dimensions -> malloc -> icer_decompress_image_uint16 / _yuv_uint16), compiled against
 * include/icer_hip_dec.h and linked with libicer_hip_dec.so instead of libicer.a.  Test program.
 *   dropin_decode_example <in.bin> <channels> <stages> <filter> <segments> <out.raw> [repetitions]
 * prints "rc=<code> w=<w> h=<h>"; exit code 0 for ICER_RESULT_OK, 10 for ICER_FATAL_ERROR (no GPU), 5 otherwise. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "icer_hip_dec.h"

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const int channels = atoi(argv[2]), stages = atoi(argv[3]), filt = atoi(argv[4]), segments = atoi(argv[5]);
    const int reps = argc > 7 ? atoi(argv[7]) : 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *stream = malloc(len > 0 ? (size_t)len : 1);
    if (fread(stream, 1, (size_t)len, f) != (size_t)len) return 3;
    fclose(f);

    size_t w = 0, h = 0;
    int rc = icer_get_image_dimensions(stream, (size_t)len, &w, &h);
    if (rc != ICER_RESULT_OK) { printf("rc=%d w=0 h=0\n", rc); return 5; }
    const size_t n = w * h;
    uint16_t *planes[3] = {NULL, NULL, NULL};
    for (int c = 0; c < channels; c++) planes[c] = calloc(n, sizeof(uint16_t));
    for (int r = 0; r < reps; r++) {               /* (repetitions: the call is timed, host buffers and all) */
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        if (channels == 1)
            rc = icer_decompress_image_uint16(planes[0], &w, &h, n, stream, (size_t)len, (uint8_t)stages,
                                              (enum icer_filter_types)filt, (uint8_t)segments);
        else
            rc = icer_decompress_image_yuv_uint16(planes[0], planes[1], planes[2], &w, &h, n, stream, (size_t)len, (uint8_t)stages,
                                                  (enum icer_filter_types)filt, (uint8_t)segments);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (reps > 1) printf("call %d: %.1f ms\n", r, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6);
    }
    printf("rc=%d w=%zu h=%zu\n", rc, w, h);
    if (rc == ICER_FATAL_ERROR) { fprintf(stderr, "%s\n", icerx_decoder_last_error()); return 10; }
    f = fopen(argv[6], "wb");
    for (int c = 0; c < channels; c++) fwrite(planes[c], sizeof(uint16_t), n, f);
    fclose(f);
    return rc == ICER_RESULT_OK ? 0 : 5;
}
