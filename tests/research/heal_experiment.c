/*
 * heal_experiment.c -- TEST / DESIGN RESEARCH ONLY (includes the oracle restatement; nothing in the product uses it).
 *
 * Question ("sub-range parallelism"): if a coding unit's chunk range is cut into K sub-ranges and sub-range i
 * is started COLD -- exact adaptive counts (those depend on the coefficients alone), but no open code words and an empty
 * ring -- how many 64-pixel chunks does it take until its complete coder state (every bin's open word and partial input,
 * the ring from its oldest word to its tail) equals the state of the coder that started at the unit's first pixel?
 * From that chunk on both produce the same bits, so the cold coder's output can be spliced onto the exact prefix.
 *
 *   gcc -O2 -I oracle tests/research/heal_experiment.c -o /tmp/heal && /tmp/heal [w h stages segments mode K]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/icer_oracle.c"

static uint32_t lcg_s;
static uint32_t lcg(void) { lcg_s = lcg_s * 1664525u + 1013904223u; return lcg_s >> 8; }

/* one event source over a segment: calls put(bit, zero, total) on one or two coders */
typedef struct { coder *c; uint32_t zero[17], total[17]; int active; } runner;

static int state_equal(const coder *a, const coder *b)
{
    if (a->used != b->used) return 0;
    for (int k = 0; k < 17; k++) {
        if (a->in_bits[k] != b->in_bits[k]) return 0;
        const int oa = a->open_slot[k] < 0 ? -1 : (int)((a->open_slot[k] + RING_WORDS - a->head) % RING_WORDS);
        const int ob = b->open_slot[k] < 0 ? -1 : (int)((b->open_slot[k] + RING_WORDS - b->head) % RING_WORDS);
        if (oa != ob) return 0;
    }
    for (unsigned k = 0; k < a->used; k++) {
        const ring_word *x = &a->ring[(a->head + k) % RING_WORDS], *y = &b->ring[(b->head + k) % RING_WORDS];
        if (x->bin != y->bin || x->done != y->done || x->nbits != y->nbits || x->value != y->value) return 0;
    }
    return 1;
}

int main(int argc, char **argv)
{
    const size_t W = argc > 1 ? strtoul(argv[1], 0, 10) : 4096, H = argc > 2 ? strtoul(argv[2], 0, 10) : 4096;
    const int stages = argc > 3 ? atoi(argv[3]) : 5, segments = argc > 4 ? atoi(argv[4]) : 10, mode = argc > 5 ? atoi(argv[5]) : 1;
    const int K = argc > 6 ? atoi(argv[6]) : 8;
    uint16_t *img = malloc(W * H * 2);
    lcg_s = 12345;
    for (size_t y = 0; y < H; y++)
        for (size_t x = 0; x < W; x++) {
            int v = mode == 0 ? (int)(lcg() & 255) : (int)(128 + 60 * ((double)x / W - 0.5) + 50 * ((double)y / H - 0.5)) + (int)(lcg() % 17) - 8;
            img[y * W + x] = (uint16_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    orc_dwt_stages_u16(img, W, H, stages, 0);
    orc_sign_magnitude(img, W * H);
    coder_tables_init();
    const size_t cap = 1 << 24;
    uint8_t *o1 = malloc(cap), *o2 = malloc(cap);
    coder *T = malloc(sizeof(coder)), *C = malloc(sizeof(coder));
    long hist[12] = {0};       /* healing distance in chunks: <8, <16, <32, <64, <128, <256, <512, <1024, <2048, <4096, >=4096, never */
    long n_cases = 0, sum_heal = 0, max_heal = 0;
    for (int level = 1; level <= 2; level++) {
        for (int sb = 1; sb <= 3; sb++) {
            const size_t lw = dim_low(W, level), lh = dim_low(H, level), hw = dim_high(W, level), hh = dim_high(H, level);
            const size_t sw = (sb == SB_LH) ? lw : hw, sh = (sb == SB_HL) ? lh : hh;
            const size_t ox = (sb == SB_LH) ? 0 : lw, oy = (sb == SB_HL) ? 0 : lh;
            orc_partition part;
            orc_rect rects[33];
            orc_partition_make(&part, sw, sh, (unsigned)segments);
            const int ns = orc_partition_rects(&part, rects);
            for (int sg = 0; sg < ns; sg += 3) {                 /* every third segment: enough for statistics */
                for (int lsb = 0; lsb <= 5; lsb++) {
                    const uint16_t *seg = img + (oy + rects[sg].y) * W + ox + rects[sg].x;
                    const long w = rects[sg].w, h = rects[sg].h, npix = w * h, nchunks = (npix + 63) / 64;
                    if (nchunks < 16 * K) continue;
                    /* boundaries at i * nchunks / K; run the true coder over the unit; at every boundary fork a cold coder */
                    for (int bi = 1; bi < K; bi++) {
                        const long cstart = bi * nchunks / K;
                        coder_init(T, o1, cap);
                        uint32_t zero[17], total[17];
                        for (int k = 0; k < 17; k++) { zero[k] = 2; total[k] = 4; }
                        int cold_on = 0;
                        long healed = -1, flushes_seen = 0;
                        (void)flushes_seen;
                        for (long p = 0; p < npix; p++) {
                            if ((p & 63) == 0) {
                                const long chunk = p >> 6;
                                if (chunk == cstart) { coder_init(C, o2, cap); cold_on = 1; }
                                else if (cold_on && state_equal(T, C)) { healed = chunk - cstart; break; }
                                if (cold_on && chunk - cstart > 6000) break;
                            }
                            const long r = p / w, cc = p % w;
#define MAG(r, cc) (seg[(r) * W + (cc)] & 0x7FFFu)
#define NEG(r, cc) (seg[(r) * W + (cc)] >> 15)
#define SIG(r, cc, l) (((r) < 0 || (cc) < 0 || (r) >= (long)h || (cc) >= (long)w) ? 0 : ((MAG(r, cc) >> (l)) != 0))
#define SGN(r, cc, l) ((SIG(r, cc, l) && NEG(r, cc)) ? -1 : 0)
#define PUT(bit, z, t) { coder_put(T, bit, z, t); if (cold_on) coder_put(C, bit, z, t); }
                            unsigned m = MAG(r, cc);
                            int msb = 0;
                            for (unsigned t = m | 1; t > 1; t >>= 1) msb++;
                            int cat = msb - lsb;
                            if (cat < 0) cat = 0;
                            if (cat > 3) cat = 3;
                            int bit = (int)((m >> lsb) & 1);
                            if (cat == 3) { PUT(bit, 1, 2); continue; }
                            int ctx;
                            if (cat == 2) ctx = 11;
                            else {
                                int hh2 = SIG(r, cc - 1, lsb) + SIG(r, cc + 1, lsb + 1);
                                int vv = SIG(r - 1, cc, lsb) + SIG(r + 1, cc, lsb + 1);
                                int dd = SIG(r - 1, cc - 1, lsb) + SIG(r - 1, cc + 1, lsb) + SIG(r + 1, cc - 1, lsb + 1) + SIG(r + 1, cc + 1, lsb + 1);
                                if (cat == 1) ctx = (hh2 + vv == 0) ? 9 : 10;
                                else {
                                    if (sb == SB_HL) { int t = hh2; hh2 = vv; vv = t; }
                                    ctx = (sb == SB_HH) ? ctx_hh(hh2 + vv, dd) : ctx_plain(hh2, vv, dd);
                                }
                            }
                            PUT(bit, zero[ctx], total[ctx]);
                            model_update(&zero[ctx], &total[ctx], !bit);
                            if (cat == 0 && bit) {
                                int s_h = SGN(r, cc - 1, lsb) + SGN(r, cc + 1, lsb + 1) + 2;
                                int s_v = SGN(r - 1, cc, lsb) + SGN(r + 1, cc, lsb + 1) + 2;
                                if (sb == SB_HL) { int t = s_h; s_h = s_v; s_v = t; }
                                int sctx = SIGN_CTX[s_h][s_v];
                                int agree = (SIGN_PRED[s_h][s_v] ^ (int)NEG(r, cc)) & 1;
                                PUT(agree, zero[sctx], total[sctx]);
                                model_update(&zero[sctx], &total[sctx], agree == 0);
                            }
                        }
                        n_cases++;
                        int b = 11;
                        if (healed >= 0) {
                            sum_heal += healed;
                            if (healed > max_heal) max_heal = healed;
                            b = 0;
                            for (long lim = 8; b < 10 && healed >= lim; lim *= 2) b++;
                        }
                        hist[b]++;
                        if (healed < 0 || healed > 1024)
                            printf("  level %d sb %d seg %d lsb %d boundary %d/%d (chunk %ld of %ld): healed after %ld chunks\n", level, sb, sg, lsb, bi, K, cstart, nchunks, healed);
                    }
                }
            }
        }
    }
    printf("%zux%zu st=%d seg=%d mode=%d K=%d: %ld boundaries; healing distance (chunks) histogram\n", W, H, stages, segments, mode, K, n_cases);
    const char *names[12] = {"<8", "<16", "<32", "<64", "<128", "<256", "<512", "<1024", "<2048", "<4096", ">=4096", "never(6000)"};
    for (int b = 0; b < 12; b++) printf("  %-12s %ld\n", names[b], hist[b]);
    printf("  mean %.1f  max %ld\n", n_cases ? (double)sum_heal / (double)(n_cases - hist[11]) : 0.0, max_heal);
    return 0;
}
