/*
 * async_overlap.c -- plain C against include/icer_hip.h and the HIP runtime's C API: what a caller does that feeds one
 * GPU from host memory and wants PCIe and the coder busy at the same time, by hand:
 *   two icerx_encoders, each with its own stream and device buffers; for every sub-batch
 *       hipMemcpyAsync (frames in) -> icerx_encode_device_async -> [the other encoder's turn] -> icerx_encoder_wait
 *       -> hipMemcpyAsync (sizes, streams out)
 * and compares every frame with the synchronous host call icerx_encode_host.
 * usage: async_overlap <in.raw> <n> <w> <h> <stages> <filter> <segments> <quota> <out.bin>
 *   in.raw   n frames of w*h little-endian uint16;  out.bin: n records of (uint64 size, int32 rc, size bytes of stream)
 * build: gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude async_overlap.c -licer_hip -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "icer_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 20; } } while (0)

typedef struct {
    icerx_encoder *enc;
    hipStream_t st;
    uint16_t *d_in; uint8_t *d_out; uint64_t *d_sizes; int32_t *d_rcs;
    int first, n;                      /* frames of the sub-batch in flight */
} lane_t;

int main(int argc, char **argv)
{
    if (argc != 10) { fprintf(stderr, "bad usage\n"); return 2; }
    const int n = atoi(argv[2]);
    const size_t w = strtoul(argv[3], 0, 10), h = strtoul(argv[4], 0, 10);
    const int stages = atoi(argv[5]), filt = atoi(argv[6]), segments = atoi(argv[7]);
    const size_t quota = strtoul(argv[8], 0, 10), fe = w * h;
    const int sub = 2;
    uint16_t *frames; uint8_t *out; uint64_t *sizes; int32_t *rcs;
    if (icerx_device_count() < 1) { fprintf(stderr, "no device: %s\n", icerx_last_error()); return 10; }
    CHECK(hipHostMalloc((void **)&frames, (size_t)n * fe * 2, 0));
    CHECK(hipHostMalloc((void **)&out, (size_t)n * quota, 0));
    CHECK(hipHostMalloc((void **)&sizes, (size_t)n * 8, 0));
    CHECK(hipHostMalloc((void **)&rcs, (size_t)n * 4, 0));
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(frames, 2, (size_t)n * fe, f) != (size_t)n * fe) { fprintf(stderr, "cannot read input\n"); return 2; }
    fclose(f);

    lane_t L[2];
    for (int k = 0; k < 2; k++) {
        int rc = icerx_encoder_create(&L[k].enc, 0, w, h, 1, stages, filt, segments, sub);
        if (rc) { fprintf(stderr, "create: rc=%d %s\n", rc, icerx_last_error()); return 3; }
        CHECK(hipStreamCreateWithFlags(&L[k].st, hipStreamNonBlocking));
        CHECK(hipMalloc((void **)&L[k].d_in, (size_t)sub * fe * 2));
        CHECK(hipMalloc((void **)&L[k].d_out, (size_t)sub * quota));
        CHECK(hipMalloc((void **)&L[k].d_sizes, (size_t)sub * 8));
        CHECK(hipMalloc((void **)&L[k].d_rcs, (size_t)sub * 4));
        L[k].n = 0;
    }
    /* a wait without a pending call is a no-op */
    if (icerx_encoder_wait(L[0].enc) != 0) return 4;
    int next = 0;
    for (int turn = 0;; turn++) {
        lane_t *l = &L[turn & 1];
        if (l->n) {                       /* complete what this encoder has in flight, fetch its streams */
            int rc = icerx_encoder_wait(l->enc);
            if (rc) { fprintf(stderr, "wait: rc=%d %s\n", rc, icerx_last_error()); return 5; }
            CHECK(hipMemcpyAsync(sizes + l->first, l->d_sizes, (size_t)l->n * 8, hipMemcpyDeviceToHost, l->st));
            CHECK(hipMemcpyAsync(rcs + l->first, l->d_rcs, (size_t)l->n * 4, hipMemcpyDeviceToHost, l->st));
            CHECK(hipStreamSynchronize(l->st));
            for (int i = 0; i < l->n; i++)
                CHECK(hipMemcpyAsync(out + (size_t)(l->first + i) * quota, l->d_out + (size_t)i * quota, sizes[l->first + i], hipMemcpyDeviceToHost, l->st));
            l->n = 0;
        }
        if (next < n) {                   /* next sub-batch: copy in and enqueue, do not wait */
            l->first = next;
            l->n = n - next < sub ? n - next : sub;
            next += l->n;
            CHECK(hipMemcpyAsync(l->d_in, frames + (size_t)l->first * fe, (size_t)l->n * fe * 2, hipMemcpyHostToDevice, l->st));
            int rc = icerx_encode_device_async(l->enc, l->d_in, l->n, quota, l->d_out, quota, l->d_sizes, l->d_rcs, l->st);
            if (rc) { fprintf(stderr, "async: rc=%d %s\n", rc, icerx_last_error()); return 6; }
            /* a second call on an encoder with one pending is refused */
            if (icerx_encode_device_async(l->enc, l->d_in, l->n, quota, l->d_out, quota, l->d_sizes, l->d_rcs, l->st) != ICER_INVALID_INPUT) return 7;
        }
        if (next >= n && !L[0].n && !L[1].n) break;
    }
    CHECK(hipStreamSynchronize(L[0].st));
    CHECK(hipStreamSynchronize(L[1].st));

    /* the synchronous host call on the same frames */
    uint8_t *ref = malloc((size_t)sub * quota);
    uint64_t rs[2]; int32_t rr[2];
    for (int first = 0; first < n; first += sub) {
        const int m = n - first < sub ? n - first : sub;
        int rc = icerx_encode_host(L[0].enc, frames + (size_t)first * fe, m, quota, ref, quota, rs, rr);
        if (rc) { fprintf(stderr, "host: rc=%d %s\n", rc, icerx_last_error()); return 8; }
        for (int i = 0; i < m; i++)
            if (rs[i] != sizes[first + i] || rr[i] != rcs[first + i] || memcmp(ref + (size_t)i * quota, out + (size_t)(first + i) * quota, rs[i])) {
                fprintf(stderr, "frame %d: asynchronous and synchronous calls differ\n", first + i);
                return 9;
            }
    }
    f = fopen(argv[9], "wb");
    for (int i = 0; i < n; i++) { fwrite(&sizes[i], 8, 1, f); fwrite(&rcs[i], 4, 1, f); fwrite(out + (size_t)i * quota, 1, sizes[i], f); }
    fclose(f);
    for (int k = 0; k < 2; k++) icerx_encoder_destroy(L[k].enc);
    printf("ok frames=%d\n", n);
    return 0;
}
