/*
 * ref_tap.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A tiny translation unit of OUR OWN that is compiled together with the untouched
 * reference sources (where they lie under /root/reference/lib_icer/src) into
 * oracle/_ref/libicer_ref.so.  It only calls public reference functions so that
 * the tests can fetch intermediates of the reference encoder:
 *   - the coefficient buffer after icer_wavelet_transform_stages_uint16
 *   - one coding unit's payload through icer_compress_bitplane_uint16
 *   - the sorted packet list (global icer_packets_16) after a compress call
 *   - the partition geometry from icer_generate_partition_parameters
 * No reference source is copied; only the header is #included at build time.
 */
#include "icer.h"
#include <string.h>

/* one coding unit: returns payload bit length, or a negative icer_status */
long ref_tap_code_unit(const uint16_t *seg, size_t w, size_t h, size_t rowstride,
                       int subband, int lsb, uint8_t *out, size_t out_cap)
{
    icer_context_model_typedef model;
    icer_encoder_context_typedef enc;
    icer_packet_context pkt;
    memset(&pkt, 0, sizeof pkt);
    pkt.subband_type = (uint8_t)subband;
    pkt.lsb = (uint8_t)lsb;
    icer_init_context_model_vals(&model, (enum icer_subband_types)subband);
    icer_init_entropy_coder_context(&enc, icer_encode_circ_buf, ICER_CIRC_BUF_SIZE, out, out_cap);
    int res = icer_compress_bitplane_uint16(seg, w, h, rowstride, &model, &enc, &pkt);
    if (res != ICER_RESULT_OK) return res;
    return (long)(enc.output_ind * 8 + enc.output_bit_offset);
}

/* sorted packet list left behind by the last icer_compress_image_*_uint16 call */
int ref_tap_get_packet(int idx, int *level, int *subband, int *lsb, int *chan, unsigned long long *prio)
{
    if (idx < 0 || idx >= ICER_MAX_PACKETS_16) return -1;
    *level = icer_packets_16[idx].decomp_level;
    *subband = icer_packets_16[idx].subband_type;
    *lsb = icer_packets_16[idx].lsb;
    *chan = icer_packets_16[idx].channel;
    *prio = icer_packets_16[idx].priority;
    return 0;
}

/* partition geometry as 15 uint16 in struct order */
int ref_tap_partition(size_t w, size_t h, unsigned segments, uint16_t out15[15])
{
    partition_param_typdef p;
    memset(&p, 0, sizeof p);
    int res = icer_generate_partition_parameters(&p, w, h, (uint16_t)segments);
    memcpy(out15, &p, 15 * sizeof(uint16_t));
    return res;
}

/* table taps: let tests compare our re-derived constant tables with the reference's */
int ref_tap_custom_code(int bin, int prefix, int *in_bits, int *out_bits, int *out_code)
{
    *in_bits = icer_custom_coding_scheme[bin][prefix].input_code_bits;
    *out_bits = icer_custom_coding_scheme[bin][prefix].output_code_bits;
    *out_code = icer_custom_coding_scheme[bin][prefix].output_code;
    return 0;
}
int ref_tap_flush(int bin, int prefix, int nbits, int *fbit, int *fnum)
{
    *fbit = icer_custom_code_flush_bits[bin][prefix][nbits].flush_bit;
    *fnum = icer_custom_code_flush_bits[bin][prefix][nbits].flush_bit_numbers;
    return 0;
}
int ref_tap_golomb(int bin, int *m, int *l, int *i)
{
    *m = icer_golomb_coders[bin].m; *l = icer_golomb_coders[bin].l; *i = icer_golomb_coders[bin].i;
    return 0;
}
unsigned ref_tap_cutoff(int idx) { return icer_bin_probability_cutoffs[idx]; }
