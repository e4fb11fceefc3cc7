/*
 * ohp_pipeline.h -- message-level oracle entry points (TEST INFRASTRUCTURE ONLY, see ohp_oracle.h).
 * The descriptor structs have the same layout as include/ohgpu.h's so that one numpy record array
 * can drive both the device and the checker; tests/test_capi_loads.py asserts the sizes agree.
 */
#ifndef OHP_PIPELINE_H
#define OHP_PIPELINE_H

#include "ohp_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OHP_FLAG_RAMP       0x01u
#define OHP_FLAG_SILENCE    0x02u
#define OHP_FLAG_ZERO_LSB32 0x04u

typedef struct {
    uint64_t src_offset;
    uint64_t dst_offset;
    uint32_t n_frames;
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint16_t attenuation;
    uint8_t  channels;
    uint8_t  src_bits;
    uint8_t  src_endian;
    uint8_t  dst_bits;
    uint8_t  dst_endian;
    uint8_t  flags;
} ohp_msg_desc;

typedef struct {
    uint64_t src_offset;
    uint64_t src_frame0;
    uint64_t src_frames;
    uint64_t out_frame0;
    uint64_t dst_offset;
    uint32_t n_frames;
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint16_t attenuation;
    uint8_t  channels;
    uint8_t  src_bits;
    uint8_t  src_endian;
    uint8_t  dst_bits;
    uint8_t  dst_endian;
    uint8_t  flags;
    uint8_t  reserved[8];   /* (the device path's src_plane_stride: a planar source is, here, ohp_flac_pack followed by this) */
} ohp_src_msg_desc;

int ohp_msg_process(const ohp_msg_desc* d, const uint8_t* src_base, uint8_t* dst_base);
int ohp_msg_process_batch(const ohp_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base);

int ohp_src_msg_process(const ohp_src* s, const ohp_src_msg_desc* d, const uint8_t* src_base, uint8_t* dst_base);
int ohp_src_msg_process_batch(const ohp_src* s, const ohp_src_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base);
/* the same bytes, for timing: scratch buffers allocated once per batch, window bounds checked once per message */
int ohp_src_msg_process_batch_steady(const ohp_src* s, const ohp_src_msg_desc* d, size_t n, const uint8_t* src_base, uint8_t* dst_base);
int ohp_src_msg_process_f64(const ohp_src* s, const ohp_src_msg_desc* d, const uint8_t* src_base, double* y);

ohp_src*       ohp_src_new(uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass);
void           ohp_src_delete(ohp_src* s);
uint32_t       ohp_src_L(const ohp_src* s);
uint32_t       ohp_src_M(const ohp_src* s);
uint32_t       ohp_src_T(const ohp_src* s);
const int32_t* ohp_src_coef_q28(const ohp_src* s);
const double*  ohp_src_coef_f64(const ohp_src* s);
int64_t        ohp_src_sum_abs_max(const ohp_src* s);
double         ohp_src_f_stop(const ohp_src* s);

#ifdef __cplusplus
}
#endif
#endif
