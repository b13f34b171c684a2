/*
 * batch_devices.c -- plain C (pthreads) against include/icer_hip.h:
 *   1. icerx_compress_batch_uint16 over all devices of the node and over one device give the same bytes;
 *   2. two icerx_encoders driven from two host threads at the same time -- on two devices when the node has two, on the
 *      same device otherwise -- give the same bytes as (1): what a C caller that shards a batch by hand does.
 * usage: batch_devices <in.raw> <n> <w> <h> <stages> <filter> <segments> <quota> <out.bin>
 *   in.raw   n frames of w*h little-endian uint16;  out.bin: n records of (uint64 size, int32 rc, size bytes of stream)
 * exit code 0 when all three ways agree.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "icer_hip.h"

typedef struct {
    int device, n;
    const uint16_t *frames;
    size_t w, h, quota;
    int stages, filt, segments;
    uint8_t *out; uint64_t *sizes; int32_t *rcs;
    int rc;
} job_t;

static void *run(void *p)
{
    job_t *j = (job_t *)p;
    icerx_encoder *e = NULL;
    j->rc = icerx_encoder_create(&e, j->device, j->w, j->h, 1, j->stages, j->filt, j->segments, j->n);
    if (j->rc == 0) {
        for (int rep = 0; rep < 3 && j->rc == 0; rep++)          /* (a few rounds, so that the two threads really overlap) */
            j->rc = icerx_encode_host(e, j->frames, j->n, j->quota, j->out, j->quota, j->sizes, j->rcs);
        icerx_encoder_destroy(e);
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc != 10) { fprintf(stderr, "bad usage\n"); return 2; }
    const int n = atoi(argv[2]);
    const size_t w = strtoul(argv[3], 0, 10), h = strtoul(argv[4], 0, 10);
    const int stages = atoi(argv[5]), filt = atoi(argv[6]), segments = atoi(argv[7]);
    const size_t quota = strtoul(argv[8], 0, 10);
    uint16_t *frames = malloc((size_t)n * w * h * 2);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(frames, 2, (size_t)n * w * h, f) != (size_t)n * w * h) { fprintf(stderr, "cannot read input\n"); return 2; }
    fclose(f);
    const int devices = icerx_device_count();
    printf("devices=%d\n", devices);
    if (devices < 1) { fprintf(stderr, "no device: %s\n", icerx_last_error()); return 10; }

    uint8_t *out[3]; uint64_t *sizes[3]; int32_t *rcs[3];
    for (int k = 0; k < 3; k++) { out[k] = calloc((size_t)n, quota); sizes[k] = calloc((size_t)n, 8); rcs[k] = calloc((size_t)n, 4); }
    int rc = icerx_compress_batch_uint16(frames, n, w, h, 1, stages, filt, segments, quota, out[0], quota, sizes[0], rcs[0], 0);
    if (rc) { fprintf(stderr, "batch over all devices: rc=%d %s\n", rc, icerx_last_error()); return 3; }
    rc = icerx_compress_batch_uint16(frames, n, w, h, 1, stages, filt, segments, quota, out[1], quota, sizes[1], rcs[1], 1);
    if (rc) { fprintf(stderr, "batch on one device: rc=%d %s\n", rc, icerx_last_error()); return 3; }

    const int n0 = (n + 1) / 2;
    job_t j[2] = {{0, n0, frames, w, h, quota, stages, filt, segments, out[2], sizes[2], rcs[2], 0},
                  {devices > 1 ? 1 : 0, n - n0, frames + (size_t)n0 * w * h, w, h, quota, stages, filt, segments,
                   out[2] + (size_t)n0 * quota, sizes[2] + n0, rcs[2] + n0, 0}};
    pthread_t t[2];
    const int nt = n - n0 > 0 ? 2 : 1;
    for (int k = 0; k < nt; k++) pthread_create(&t[k], NULL, run, &j[k]);
    for (int k = 0; k < nt; k++) pthread_join(t[k], NULL);
    for (int k = 0; k < nt; k++) if (j[k].rc) { fprintf(stderr, "thread %d: rc=%d\n", k, j[k].rc); return 4; }

    for (int k = 1; k < 3; k++)
        for (int i = 0; i < n; i++)
            if (sizes[k][i] != sizes[0][i] || rcs[k][i] != rcs[0][i] || memcmp(out[k] + (size_t)i * quota, out[0] + (size_t)i * quota, sizes[0][i])) {
                fprintf(stderr, "way %d differs from the batch call at frame %d\n", k, i);
                return 5;
            }
    f = fopen(argv[9], "wb");
    for (int i = 0; i < n; i++) { fwrite(&sizes[0][i], 8, 1, f); fwrite(&rcs[0][i], 4, 1, f); fwrite(out[0] + (size_t)i * quota, 1, sizes[0][i], f); }
    fclose(f);
    printf("ok frames=%d second_thread_device=%d\n", n, j[1].device);
    return 0;
}
