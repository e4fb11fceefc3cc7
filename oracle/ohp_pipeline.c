/*
 * ohp_pipeline.c -- message-level compositions of the oracle rows, mirroring what one descriptor of
 * include/ohgpu.h asks the device to do.  TEST INFRASTRUCTURE ONLY (see ohp_oracle.h).
 *
 *   ohp_msg_process      = a1 (ConstructPcm) -> a6 (ApplyAttenuation) -> a7 (RampApplicator)
 *                          | a9 (silence) -> a11 o a12 (depth conversion) [-> endian]
 *   ohp_src_msg_process  = a1 -> a11 (to S24) -> a-R (resampler, exact integer model)
 *                          -> a7 at 24 bit -> a11 o a12 [-> endian]
 * Each step calls the row's own function in ohp_oracle.c, so the composition adds no arithmetic.
 */
#include "ohp_oracle.h"
#include "ohp_pipeline.h"

#include <stdlib.h>
#include <string.h>

static int a1_any_size(const uint8_t* src, uint64_t bytes, uint32_t bits, int endian, uint8_t* dst)
{
    /* ConstructPcm is byte-wise independent per subsample; apply it in <= kMaxBytes pieces so that
     * descriptors longer than one DecodedAudio cell (batch mode) stay within the row's contract. */
    const uint32_t bd = bits / 8;
    if (bd == 0) return OHP_ERR_ASSERT;
    const uint32_t piece = OHP_MAX_BYTES - (OHP_MAX_BYTES % bd);
    while (bytes > 0) {
        const uint32_t n = bytes > piece ? piece : (uint32_t)bytes;
        const int err = ohp_construct_pcm(src, n, bits, endian, dst);
        if (err < 0) return err;
        src += n; dst += n; bytes -= n;
    }
    return OHP_OK;
}

int ohp_msg_process(const ohp_msg_desc* d, const uint8_t* src_base, uint8_t* dst_base)
{
    const uint32_t sb = d->src_bits / 8, db = d->dst_bits / 8;
    if (sb < 1 || sb > 4 || db < 1 || db > 4 || d->channels == 0) return OHP_ERR_ASSERT;
    const uint64_t bytes = (uint64_t)d->n_frames * d->channels * sb;
    if (bytes == 0) return OHP_OK;
    if (bytes > 0x7fffffffu) return OHP_ERR_ASSERT;
    uint8_t* buf = (uint8_t*)malloc(bytes);
    uint8_t* buf2 = (uint8_t*)malloc(bytes);
    if (!buf || !buf2) { free(buf); free(buf2); return OHP_ERR_ASSERT; }
    int err = OHP_OK;
    if (d->flags & OHP_FLAG_SILENCE) {
        ohp_playable p;
        memset(&p, 0, sizeof(p));
        p.size_bytes = (uint32_t)bytes; p.bit_depth = d->src_bits; p.channels = d->channels; p.is_silence = 1;
        uint32_t nf = 0, ob = 0;
        err = ohp_playable_read(&p, NULL, buf, (uint32_t)bytes, NULL, 0, &nf, &ob);
    } else {
        err = a1_any_size(src_base + d->src_offset, bytes, d->src_bits, d->src_endian, buf);
        if (err == OHP_OK) err = ohp_apply_attenuation(buf, (uint32_t)bytes, d->src_bits, d->attenuation);
        if (err == OHP_OK && (d->flags & OHP_FLAG_RAMP)) {
            err = ohp_ramp_apply(buf, (uint32_t)bytes, d->src_bits, d->channels, d->ramp_start, d->ramp_end, buf2);
            if (err == OHP_OK) memcpy(buf, buf2, bytes);
        }
    }
    if (err == OHP_OK)
        err = ohp_convert_format(buf, d->n_frames * d->channels, d->src_bits, OHP_ENDIAN_BIG, d->dst_bits,
                                 d->dst_endian, (d->flags & OHP_FLAG_ZERO_LSB32) ? 1 : 0, dst_base + d->dst_offset);
    free(buf); free(buf2);
    return err;
}

int ohp_msg_process_batch(const ohp_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base)
{
    for (size_t i = 0; i < n; i++) {
        const int err = ohp_msg_process(&d[i], src_base, dst_base);
        if (err < 0) return err;
    }
    return OHP_OK;
}

int ohp_src_msg_process(const ohp_src* s, const ohp_src_msg_desc* d, const uint8_t* src_base, uint8_t* dst_base)
{
    const uint32_t sb = d->src_bits / 8, db = d->dst_bits / 8, ch = d->channels;
    if (sb < 1 || sb > 4 || db < 1 || db > 4 || ch == 0) return OHP_ERR_ASSERT;
    if (d->attenuation != OHP_UNITY_ATTENUATION) return OHP_ERR_ASSERT;
    if (d->n_frames == 0) return OHP_OK;
    const uint64_t m_first = d->out_frame0, m_last = d->out_frame0 + d->n_frames - 1;
    const int64_t n_hi = (int64_t)((m_last * s->M) / s->L);
    int64_t n_lo = (int64_t)((m_first * s->M) / s->L) - (int64_t)(s->T - 1);
    if (n_lo < 0) n_lo = 0;
    if (n_lo < (int64_t)d->src_frame0) return OHP_ERR_ASSERT;                     /* history missing */
    if (n_hi >= (int64_t)(d->src_frame0 + d->src_frames)) return OHP_ERR_ASSERT;  /* future missing  */
    const uint64_t frames = (uint64_t)(n_hi - n_lo + 1);
    int32_t* x = (int32_t*)malloc(frames * ch * sizeof(int32_t));
    int32_t* y = (int32_t*)malloc((size_t)d->n_frames * ch * sizeof(int32_t));
    uint8_t* be24 = (uint8_t*)malloc((size_t)d->n_frames * ch * 3);
    uint8_t* ramped = (uint8_t*)malloc((size_t)d->n_frames * ch * 3);
    int err = (x && y && be24 && ramped) ? OHP_OK : OHP_ERR_ASSERT;
    if (err == OHP_OK) {
        const uint8_t* p = src_base + d->src_offset + (uint64_t)(n_lo - (int64_t)d->src_frame0) * ch * sb;
        err = ohp_unpack_s24(p, (uint32_t)(frames * ch), d->src_bits, d->src_endian, x);
    }
    if (err == OHP_OK) err = ohp_src_process_i64(s, x, n_lo, frames, ch, d->out_frame0, d->n_frames, y);
    if (err == OHP_OK) err = ohp_pack_from_s24(y, d->n_frames * ch, 24, OHP_ENDIAN_BIG, be24);
    const uint8_t* stage = be24;
    if (err == OHP_OK && (d->flags & OHP_FLAG_RAMP)) {
        err = ohp_ramp_apply(be24, d->n_frames * ch * 3, 24, ch, d->ramp_start, d->ramp_end, ramped);
        stage = ramped;
    }
    if (err == OHP_OK)
        err = ohp_convert_format(stage, d->n_frames * ch, 24, OHP_ENDIAN_BIG, d->dst_bits, d->dst_endian,
                                 (d->flags & OHP_FLAG_ZERO_LSB32) ? 1 : 0, dst_base + d->dst_offset);
    free(x); free(y); free(be24); free(ramped);
    return err;
}

int ohp_src_msg_process_batch(const ohp_src* s, const ohp_src_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base)
{
    for (size_t i = 0; i < n; i++) {
        const int err = ohp_src_msg_process(s, &d[i], src_base, dst_base);
        if (err < 0) return err;
    }
    return OHP_OK;
}

/* ---- steady-state variant for timing (bench.py's cpu_baseline leg): the same arithmetic as ohp_src_msg_process, with the
 * scratch buffers allocated once per batch instead of four times per message and the resampler's bounds checks made once
 * per message instead of twice per tap.  tests/test_oracle_baseline.py holds it equal, byte for byte, to the plain one. */
typedef struct { int32_t* x; int32_t* y; uint8_t* be24; uint8_t* ramped; size_t x_cap, y_cap; } src_ws;

static int src_process_i64_checked_once(const ohp_src* s, const int32_t* x, int64_t x_first, uint64_t x_frames, uint32_t channels,
                                        uint64_t m0, uint32_t n_out, int32_t* y)
{
    const uint32_t L = s->L, M = s->M, T = s->T;
    const uint64_t t_first = m0 * (uint64_t)M, t_last = (m0 + n_out - 1) * (uint64_t)M;
    const int64_t n_first = (int64_t)(t_first / L), n_last = (int64_t)(t_last / L);
    /* a message whose every window lies inside [x_first, x_first + x_frames) and after the stream start needs no checks */
    if (n_first - (int64_t)(T - 1) < 0 || n_first - (int64_t)(T - 1) < x_first || (uint64_t)(n_last - x_first) >= x_frames)
        return ohp_src_process_i64(s, x, x_first, x_frames, channels, m0, n_out, y);
    uint64_t t = t_first;
    for (uint32_t o = 0; o < n_out; o++, t += M) {
        const int64_t n0 = (int64_t)(t / L);
        const int32_t* c = s->coef_q28 + (size_t)(t % L) * T;
        const int32_t* xp = x + (size_t)(n0 - x_first) * channels;
        for (uint32_t ch = 0; ch < channels; ch++) {
            int64_t acc = 0;
            const int32_t* q = xp + ch;
            for (uint32_t k = 0; k < T; k++, q -= channels) acc += (int64_t)c[k] * (int64_t)*q;
            int64_t v = (acc + ((int64_t)1 << 27)) >> 28;
            if (v > 8388607) v = 8388607;
            if (v < -8388608) v = -8388608;
            y[(size_t)o * channels + ch] = (int32_t)v;
        }
    }
    return OHP_OK;
}

static int src_msg_process_ws(const ohp_src* s, const ohp_src_msg_desc* d, const uint8_t* src_base, uint8_t* dst_base, src_ws* w)
{
    const uint32_t sb = d->src_bits / 8, db = d->dst_bits / 8, ch = d->channels;
    if (sb < 1 || sb > 4 || db < 1 || db > 4 || ch == 0) return OHP_ERR_ASSERT;
    if (d->attenuation != OHP_UNITY_ATTENUATION) return OHP_ERR_ASSERT;
    if (d->n_frames == 0) return OHP_OK;
    const uint64_t m_first = d->out_frame0, m_last = d->out_frame0 + d->n_frames - 1;
    const int64_t n_hi = (int64_t)((m_last * s->M) / s->L);
    int64_t n_lo = (int64_t)((m_first * s->M) / s->L) - (int64_t)(s->T - 1);
    if (n_lo < 0) n_lo = 0;
    if (n_lo < (int64_t)d->src_frame0) return OHP_ERR_ASSERT;
    if (n_hi >= (int64_t)(d->src_frame0 + d->src_frames)) return OHP_ERR_ASSERT;
    const uint64_t frames = (uint64_t)(n_hi - n_lo + 1);
    if (frames * ch > w->x_cap || (size_t)d->n_frames * ch > w->y_cap) return OHP_ERR_ASSERT;
    const uint8_t* p = src_base + d->src_offset + (uint64_t)(n_lo - (int64_t)d->src_frame0) * ch * sb;
    int err = ohp_unpack_s24(p, (uint32_t)(frames * ch), d->src_bits, d->src_endian, w->x);
    if (err == OHP_OK) err = src_process_i64_checked_once(s, w->x, n_lo, frames, ch, d->out_frame0, d->n_frames, w->y);
    if (err == OHP_OK) err = ohp_pack_from_s24(w->y, d->n_frames * ch, 24, OHP_ENDIAN_BIG, w->be24);
    const uint8_t* stage = w->be24;
    if (err == OHP_OK && (d->flags & OHP_FLAG_RAMP)) {
        err = ohp_ramp_apply(w->be24, d->n_frames * ch * 3, 24, ch, d->ramp_start, d->ramp_end, w->ramped);
        stage = w->ramped;
    }
    if (err == OHP_OK)
        err = ohp_convert_format(stage, d->n_frames * ch, 24, OHP_ENDIAN_BIG, d->dst_bits, d->dst_endian,
                                 (d->flags & OHP_FLAG_ZERO_LSB32) ? 1 : 0, dst_base + d->dst_offset);
    return err;
}

int ohp_src_msg_process_batch_steady(const ohp_src* s, const ohp_src_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base)
{
    size_t max_out = 0, max_in = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t o = (size_t)d[i].n_frames * d[i].channels;
        const size_t in = ((size_t)d[i].n_frames * s->M / s->L + s->T + 2) * d[i].channels;
        if (o > max_out) max_out = o;
        if (in > max_in) max_in = in;
    }
    src_ws w;
    w.x = (int32_t*)malloc((max_in + 1) * sizeof(int32_t));
    w.y = (int32_t*)malloc((max_out + 1) * sizeof(int32_t));
    w.be24 = (uint8_t*)malloc(max_out * 3 + 1);
    w.ramped = (uint8_t*)malloc(max_out * 3 + 1);
    w.x_cap = max_in; w.y_cap = max_out;
    int err = (w.x && w.y && w.be24 && w.ramped) ? OHP_OK : OHP_ERR_ASSERT;
    for (size_t i = 0; i < n && err == OHP_OK; i++) err = src_msg_process_ws(s, &d[i], src_base, dst_base, &w);
    free(w.x); free(w.y); free(w.be24); free(w.ramped);
    return err;
}

/* fp64 yardstick for the resampler: unquantised coefficients, result before rounding, one message */
int ohp_src_msg_process_f64(const ohp_src* s, const ohp_src_msg_desc* d, const uint8_t* src_base, double* y)
{
    const uint32_t sb = d->src_bits / 8, ch = d->channels;
    if (sb < 1 || sb > 4 || ch == 0) return OHP_ERR_ASSERT;
    if (d->n_frames == 0) return OHP_OK;
    const uint64_t m_first = d->out_frame0, m_last = d->out_frame0 + d->n_frames - 1;
    const int64_t n_hi = (int64_t)((m_last * s->M) / s->L);
    int64_t n_lo = (int64_t)((m_first * s->M) / s->L) - (int64_t)(s->T - 1);
    if (n_lo < 0) n_lo = 0;
    if (n_lo < (int64_t)d->src_frame0) return OHP_ERR_ASSERT;
    if (n_hi >= (int64_t)(d->src_frame0 + d->src_frames)) return OHP_ERR_ASSERT;
    const uint64_t frames = (uint64_t)(n_hi - n_lo + 1);
    int32_t* x = (int32_t*)malloc(frames * ch * sizeof(int32_t));
    if (!x) return OHP_ERR_ASSERT;
    const uint8_t* p = src_base + d->src_offset + (uint64_t)(n_lo - (int64_t)d->src_frame0) * ch * sb;
    int err = ohp_unpack_s24(p, (uint32_t)(frames * ch), d->src_bits, d->src_endian, x);
    if (err == OHP_OK) err = ohp_src_process_f64(s, x, n_lo, frames, ch, d->out_frame0, d->n_frames, y);
    free(x);
    return err;
}

/* accessors so that ctypes callers need not mirror ohp_src's layout */
ohp_src* ohp_src_new(uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass)
{
    ohp_src* s = (ohp_src*)calloc(1, sizeof(ohp_src));
    if (!s) return NULL;
    if (ohp_src_design(s, rate_in, rate_out, T, beta, f_pass) < 0) { free(s); return NULL; }
    return s;
}
void ohp_src_delete(ohp_src* s) { if (s) { ohp_src_free(s); free(s); } }
uint32_t ohp_src_L(const ohp_src* s) { return s->L; }
uint32_t ohp_src_M(const ohp_src* s) { return s->M; }
uint32_t ohp_src_T(const ohp_src* s) { return s->T; }
const int32_t* ohp_src_coef_q28(const ohp_src* s) { return s->coef_q28; }
const double* ohp_src_coef_f64(const ohp_src* s) { return s->coef_f64; }
int64_t ohp_src_sum_abs_max(const ohp_src* s) { return s->sum_abs_max; }
double ohp_src_f_stop(const ohp_src* s) { return s->f_stop; }
