/*
 * icer_util_hip -- command-line codec on top of libicer_hip.so / libicer_hip_dec.so with the options of the reference's
 * `icer_util` (example/src/icer_util.c:35-54 usage text, :367-477 option handling, :96-240 compress flow, :248-365
 * decompress flow):
 *
 *     icer_util_hip compress   <input> <output> [-s stages] [-f A..Q] [-g segments] [-t bytes] [-c | -G]
 *     icer_util_hip decompress <input> <output> [-s stages] [-f A..Q] [-g segments] (-c | -G)
 *
 * and the same .bin for the same pixels: 8-bit gray is widened to uint16 (icer_util.c:163-168), colour goes through the
 * integer RGB -> Y/Cb/Cr formulas of example/inc/color_util.h:27-29 (icer_util.c:69-94), the byte quota is the target
 * size or w*h*(3|1) (icer_util.c:171-179), the output buffer 2*quota+50 bytes.
 *
 * Own code: the reference loads images through stb_image, which is not part of this repository; this tool reads the
 * uncompressed formats it can parse itself -- binary PGM/PPM (P5/P6, maxval 255) and BMP (8-bit paletted, 24- and
 * 32-bit, BI_RGB) -- and applies stb_image's channel conversions where the requested mode differs from the file's
 * (gray -> colour: replicate; colour -> gray: (77 r + 150 g + 29 b) >> 8), which is what the reference's CLI gets.
 * `decompress` (the reference: icer_util.c:248-365) needs -c or -G like the reference's, decodes with the stages / filter /
 * segments given (they must be the ones of the stream), clamps gray samples to 255, converts colour back with
 * CYCbCr2R/G/B (example/inc/color_util.h:31-33) and writes -- where the reference writes a BMP through stb_image_write --
 * a 24-bit BMP when the output name ends in .bmp, else a binary PGM / PPM.
 *
 * Build:  gcc -O2 -I include tools/icer_util_hip.c -L icer_compression_amd -licer_hip -licer_hip_dec -Wl,-rpath,$PWD/icer_compression_amd
 */
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>

#include "icer_hip.h"
#include "icer_hip_dec.h"

typedef struct {
    int w, h, channels;       /* channels: 1 or 3 as stored in the file */
    uint8_t *px;              /* w*h*channels, row-major top-down, RGB order */
} image_t;

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

static int pnm_token(const uint8_t *d, size_t n, size_t *pos, long *out)
{
    size_t p = *pos;
    for (;;) {                                  /* whitespace and # comments */
        while (p < n && (d[p] == ' ' || d[p] == '\t' || d[p] == '\n' || d[p] == '\r')) p++;
        if (p < n && d[p] == '#') { while (p < n && d[p] != '\n') p++; continue; }
        break;
    }
    if (p >= n || d[p] < '0' || d[p] > '9') return -1;
    long v = 0;
    while (p < n && d[p] >= '0' && d[p] <= '9') v = v * 10 + (d[p++] - '0');
    *pos = p;
    *out = v;
    return 0;
}

static int load_image(const char *path, image_t *im)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *d = (uint8_t *)malloc((size_t)len + 1);
    if (!d || fread(d, 1, (size_t)len, f) != (size_t)len) { fclose(f); free(d); return -1; }
    fclose(f);
    int rc = -1;
    if (len > 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) {
        size_t pos = 2;
        long w, h, maxv;
        if (!pnm_token(d, (size_t)len, &pos, &w) && !pnm_token(d, (size_t)len, &pos, &h) && !pnm_token(d, (size_t)len, &pos, &maxv) &&
            maxv == 255 && w > 0 && h > 0) {
            pos++;                              /* the single whitespace after maxval */
            const int ch = d[1] == '6' ? 3 : 1;
            if (pos + (size_t)w * h * ch <= (size_t)len) {
                im->w = (int)w; im->h = (int)h; im->channels = ch;
                im->px = (uint8_t *)malloc((size_t)w * h * ch);
                memcpy(im->px, d + pos, (size_t)w * h * ch);
                rc = 0;
            }
        }
    } else if (len > 54 && d[0] == 'B' && d[1] == 'M') {
        const uint32_t off = rd32(d + 10), hdr = rd32(d + 14);
        const int32_t w = (int32_t)rd32(d + 18), hs = (int32_t)rd32(d + 22);
        const uint32_t bpp = rd16(d + 28), comp = rd32(d + 30);
        const int h = hs < 0 ? -hs : hs;
        if (w > 0 && h > 0 && comp == 0 && (bpp == 8 || bpp == 24 || bpp == 32)) {
            const size_t stride = (((size_t)w * bpp / 8) + 3) & ~(size_t)3;
            if (off + stride * h <= (size_t)len) {
                const uint8_t *pal = d + 14 + hdr;      /* BGRA entries (8-bit files) */
                /* (stb_image reports every BMP without alpha as 3 channels, paletted ones included) */
                im->w = w; im->h = h; im->channels = 3;
                im->px = (uint8_t *)malloc((size_t)w * h * 3);
                for (int y = 0; y < h; y++) {
                    const uint8_t *row = d + off + stride * (size_t)(hs < 0 ? y : h - 1 - y);     /* bottom-up unless height < 0 */
                    uint8_t *o = im->px + (size_t)y * w * 3;
                    for (int x = 0; x < w; x++) {
                        const uint8_t *e = bpp == 8 ? pal + 4 * row[x] : row + (size_t)x * (bpp / 8);
                        o[3 * x] = e[2]; o[3 * x + 1] = e[1]; o[3 * x + 2] = e[0];
                    }
                }
                rc = 0;
            }
        }
    }
    free(d);
    return rc;
}

static int clip255(int v) { return v > 255 ? 255 : v < 0 ? 0 : v; }

static enum icer_filter_types parse_filter(const char *s)
{
    static const char names[] = "ABCDEFQ";
    if (s[0] && !s[1]) {
        const char *p = strchr(names, s[0] >= 'a' && s[0] <= 'z' ? s[0] - 32 : s[0]);
        if (p) return (enum icer_filter_types)(p - names);
    }
    fprintf(stderr, "unknown wavelet filter '%s' (one of A B C D E F Q); falling back to A\n", s);
    return ICER_FILTER_A;
}

/* 24-bit bottom-up BMP (what stb_image_write makes of 1- and 3-channel data alike), or binary PGM / PPM */
static int save_image(const char *path, const uint8_t *px, size_t w, size_t h, int channels)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const size_t len = strlen(path);
    int ok = 1;
    if (len >= 4 && strcasecmp(path + len - 4, ".bmp") == 0) {
        const size_t stride = (3 * w + 3) & ~(size_t)3, total = 54 + stride * h;
        uint8_t hdr[54] = {'B', 'M'};
        const uint32_t f32[] = {(uint32_t)total, 0, 54, 40, (uint32_t)w, (uint32_t)h};
        for (int i = 0; i < 6; i++) for (int b = 0; b < 4; b++) hdr[2 + 4 * i + b] = (uint8_t)(f32[i] >> (8 * b));
        hdr[26] = 1; hdr[28] = 24;
        ok = fwrite(hdr, 1, 54, f) == 54;
        uint8_t *row = (uint8_t *)calloc(stride, 1);
        for (size_t y = h; y-- > 0 && ok;) {
            for (size_t x = 0; x < w; x++) {
                const uint8_t *p = px + (y * w + x) * (size_t)channels;
                row[3 * x] = channels == 3 ? p[2] : p[0]; row[3 * x + 1] = channels == 3 ? p[1] : p[0]; row[3 * x + 2] = p[0];
            }
            ok = fwrite(row, 1, stride, f) == stride;
        }
        free(row);
    } else {
        fprintf(f, "P%d %zu %zu 255\n", channels == 3 ? 6 : 5, w, h);
        ok = fwrite(px, 1, w * h * (size_t)channels, f) == w * h * (size_t)channels;
    }
    return fclose(f) == 0 && ok ? 0 : -1;
}

/* the reference's decompress flow, example/src/icer_util.c:248-365 */
static int decompress_file(const char *prog, const char *in, const char *out, int stages, enum icer_filter_types filt, int segments,
                           int force_color, int force_gray)
{
    if (!force_color && !force_gray) { fprintf(stderr, "%s: decompress needs --color or --grayscale (the stream does not say)\n", prog); return 1; }
    FILE *f = fopen(in, "rb");
    if (!f) { fprintf(stderr, "%s: cannot open %s\n", prog, in); return 1; }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *data = (uint8_t *)malloc((size_t)len + 1);
    if (!data || fread(data, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "%s: cannot read %s\n", prog, in); fclose(f); return 1; }
    fclose(f);
    printf("stream %s: %ld bytes\n", in, len);
    size_t w = 0, h = 0, aw = 0, ah = 0;
    if (icer_get_image_dimensions(data, (size_t)len, &w, &h) != ICER_RESULT_OK) { fprintf(stderr, "%s: no valid packet in %s\n", prog, in); return 1; }
    printf("image %zu x %zu, decoding as %s with %d stages, filter %d, %d segments\n", w, h, force_color ? "Y Cb Cr" : "gray", stages, (int)filt, segments);
    const size_t n = w * h;
    uint16_t *pl[3] = {NULL, NULL, NULL};
    for (int k = 0; k < 3; k++) pl[k] = (uint16_t *)calloc(n, sizeof(uint16_t));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = force_color ? icer_decompress_image_yuv_uint16(pl[0], pl[1], pl[2], &aw, &ah, n, data, (size_t)len, (uint8_t)stages, filt, (uint8_t)segments)
                               : icer_decompress_image_uint16(pl[0], &aw, &ah, n, data, (size_t)len, (uint8_t)stages, filt, (uint8_t)segments);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc != ICER_RESULT_OK) {
        fprintf(stderr, "decode failed: status %d (%s); were -s -f -g the ones the stream was made with?\n", rc, icerx_decoder_last_error());
        return 1;
    }
    printf("decode call took %.3f s\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    uint8_t *px = (uint8_t *)malloc(n * (force_color ? 3 : 1));
    for (size_t i = 0; i < aw * ah; i++) {
        if (force_color) {
            const int y = pl[0][i], cb = pl[1][i], cr = pl[2][i];
            px[3 * i] = (uint8_t)clip255(y + ((91881 * cr) >> 16) - 179);
            px[3 * i + 1] = (uint8_t)clip255(y - ((22544 * cb + 46793 * cr) >> 16) + 135);
            px[3 * i + 2] = (uint8_t)clip255(y + ((116129 * cb) >> 16) - 226);
        } else px[i] = pl[0][i] > 255 ? 255 : (uint8_t)pl[0][i];
    }
    if (save_image(out, px, aw, ah, force_color ? 3 : 1)) { fprintf(stderr, "cannot write %s\n", out); return 1; }
    printf("wrote %s (%zu x %zu, %s)\n", out, aw, ah, force_color ? "colour" : "gray");
    for (int k = 0; k < 3; k++) free(pl[k]);
    free(px);
    free(data);
    return 0;
}

static void usage(const char *prog)
{
    printf("usage: %s compress|decompress <input> <output> [options]\n\n", prog);
    printf("options:\n");
    printf("  -s, --stages <n>      DWT decomposition levels, 1..6 [4]\n");
    printf("  -f, --filter <type>   lifting filter: A B C D E F Q [A]\n");
    printf("  -g, --segments <n>    error-containment segments per subband, 1..32 [6]\n");
    printf("  -c, --color           encode three planes (Y, Cb, Cr)\n");
    printf("  -G, --grayscale       encode one luminance plane\n");
    printf("  -t, --size <bytes>    byte quota of the stream [0 = lossless: one byte per sample]\n");
    printf("      --help            this text\n\n");
    printf("compress: input = binary PGM/PPM (maxval 255) or uncompressed 8/24/32-bit BMP; runs on the GPU (libicer_hip.so).\n");
    printf("decompress: needs -c or -G and the -s -f -g the stream was made with; runs on the GPU (libicer_hip_dec.so);\n");
    printf("            writes a 24-bit BMP (output name *.bmp) or a binary PGM / PPM.\n");
}

int main(int argc, char **argv)
{
    int stages = 4, segments = 6, target = 0, force_color = 0, force_gray = 0;
    enum icer_filter_types filt = ICER_FILTER_A;
    static struct option lo[] = {{"stages", required_argument, 0, 's'}, {"filter", required_argument, 0, 'f'},
                                 {"segments", required_argument, 0, 'g'}, {"size", required_argument, 0, 't'},
                                 {"color", no_argument, 0, 'c'}, {"grayscale", no_argument, 0, 'G'},
                                 {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "s:f:g:t:cG", lo, NULL)) != -1) {
        switch (c) {
        case 's': stages = atoi(optarg); if (stages < 1 || stages > 6) { fprintf(stderr, "%s: --stages takes 1..6\n", argv[0]); return 1; } break;
        case 'f': filt = parse_filter(optarg); break;
        case 'g': segments = atoi(optarg); if (segments < 1 || segments > 32) { fprintf(stderr, "%s: --segments takes 1..32\n", argv[0]); return 1; } break;
        case 't': target = atoi(optarg); if (target < 0) { fprintf(stderr, "%s: --size takes a byte count >= 0 (0 = lossless)\n", argv[0]); return 1; } break;
        case 'c': force_color = 1; break;
        case 'G': force_gray = 1; break;
        case 'h': usage(argv[0]); return 0;
        default: return 1;
        }
    }
    if (force_color && force_gray) { fprintf(stderr, "%s: --color and --grayscale exclude each other\n", argv[0]); return 1; }
    if (optind + 2 >= argc) { fprintf(stderr, "%s: expected <operation> <input> <output>\n", argv[0]); usage(argv[0]); return 1; }
    const char *op = argv[optind], *in = argv[optind + 1], *out = argv[optind + 2];
    if (strcmp(op, "decompress") == 0) return decompress_file(argv[0], in, out, stages, filt, segments, force_color, force_gray);
    if (strcmp(op, "compress") != 0) { fprintf(stderr, "%s: the operations are 'compress' and 'decompress'\n", argv[0]); return 1; }
    if (icer_init() != ICER_RESULT_OK) { fprintf(stderr, "%s: icer_init failed\n", argv[0]); return 1; }

    image_t im = {0, 0, 0, NULL};
    if (load_image(in, &im)) { fprintf(stderr, "%s: cannot read %s (binary PGM/PPM or uncompressed BMP expected)\n", argv[0], in); return 1; }
    printf("input %s: %d x %d, %d channel(s)\n", in, im.w, im.h, im.channels);
    const int use_color = force_color ? 1 : force_gray ? 0 : im.channels == 3;
    printf("planes: %s\n", use_color ? "Y Cb Cr" : "gray");

    const size_t n = (size_t)im.w * im.h;
    uint16_t *pl[3] = {NULL, NULL, NULL};
    for (int k = 0; k < (use_color ? 3 : 1); k++) pl[k] = (uint16_t *)malloc(n * sizeof(uint16_t));
    for (size_t i = 0; i < n; i++) {
        int r, g, b;
        if (im.channels == 3) { r = im.px[3 * i]; g = im.px[3 * i + 1]; b = im.px[3 * i + 2]; }
        else r = g = b = im.px[i];
        if (use_color) {
            const int y = clip255((19595 * r + 38470 * g + 7471 * b) >> 16);
            pl[0][i] = (uint16_t)y;
            pl[1][i] = (uint16_t)clip255(((36962 * (b - y)) >> 16) + 128);
            pl[2][i] = (uint16_t)clip255(((46727 * (r - y)) >> 16) + 128);
        } else {
            pl[0][i] = (uint16_t)(im.channels == 3 ? (77 * r + 150 * g + 29 * b) >> 8 : r);
        }
    }

    const int quota = target > 0 ? target : (int)(n * (use_color ? 3 : 1));
    const int buffer_size = quota * 2 + 50;
    uint8_t *stream = (uint8_t *)malloc((size_t)buffer_size);
    icer_output_data_buf_typedef od;
    icer_init_output_struct(&od, stream, (size_t)buffer_size, (size_t)quota);
    printf("encoding on the GPU: %d stages, filter %d, %d segments", stages, (int)filt, segments);
    if (target > 0) printf(", byte quota %.2f KiB\n", target / 1024.0);
    else printf(", lossless (quota %.2f KiB)\n", quota / 1024.0);

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = use_color ? icer_compress_image_yuv_uint16(pl[0], pl[1], pl[2], (size_t)im.w, (size_t)im.h, (uint8_t)stages, filt, (uint8_t)segments, &od)
                             : icer_compress_image_uint16(pl[0], (size_t)im.w, (size_t)im.h, (uint8_t)stages, filt, (uint8_t)segments, &od);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc != ICER_RESULT_OK && rc != ICER_BYTE_QUOTA_EXCEEDED) {
        fprintf(stderr, "encode failed: status %d (%s)\n", rc, icerx_last_error());
        return 1;
    }
    printf("encode call took %.3f s\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    printf("stream: %zu bytes, %.1f %% of the samples' size\n", od.size_used, 100.0 * (double)od.size_used / (double)(n * (use_color ? 3 : 1)));
    FILE *f = fopen(out, "wb");
    if (!f || fwrite(od.rearrange_start, 1, od.size_used, f) != od.size_used) { fprintf(stderr, "cannot write %s\n", out); return 1; }
    fclose(f);
    printf("wrote %s (%zu bytes)\n", out, od.size_used);
    for (int k = 0; k < 3; k++) free(pl[k]);
    free(im.px);
    free(stream);
    return 0;
}
