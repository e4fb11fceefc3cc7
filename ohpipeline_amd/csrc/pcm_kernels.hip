// pcm_kernels.hip -- baseline ("v1") kernels: one workgroup per message, one thread per frame, byte
// loads and stores.  They define the device semantics and serve as the A/B baseline for the tuned
// kernels (ohgpu_set_kernel_variant(ctx, 1) selects them).
#include <hip/hip_runtime.h>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

// unpack -> attenuate -> ramp | silence -> pack for a batch of MsgPlayables
// (MsgPlayablePcm::ReadBlock, Msg.cpp:2753-2786, with the processor's depth conversion fused).
__global__ __launch_bounds__(256) void pcm_msg_kernel_v1(const ohgpu_msg_desc* __restrict__ descs, uint32_t n_msgs,
                                                         const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const uint16_t* __restrict__ ramp_table)
{
    for (uint32_t m = blockIdx.x; m < n_msgs; m += gridDim.x) {
        const ohgpu_msg_desc d = descs[m];
        const uint32_t ch = d.channels, sb = d.src_bits >> 3, db = d.dst_bits >> 3;
        const bool src_le = d.src_endian == OHGPU_ENDIAN_LITTLE && sb > 1;
        const bool dst_le = d.dst_endian == OHGPU_ENDIAN_LITTLE;
        const bool ramp = (d.flags & OHGPU_FLAG_RAMP) != 0;
        const bool silence = (d.flags & OHGPU_FLAG_SILENCE) != 0;
        const bool zero_lsb = (d.flags & OHGPU_FLAG_ZERO_LSB32) != 0;
        const bool atten = d.attenuation != OHGPU_UNITY_ATTENUATION;
        const int32_t total = (int32_t)((uint32_t)d.ramp_start - (uint32_t)d.ramp_end);
        const uint8_t* s = src + d.src_offset;
        uint8_t* o = dst + d.dst_offset;
        for (uint32_t i = threadIdx.x; i < d.n_frames; i += blockDim.x) {
            uint32_t mult = 0;
            if (ramp) mult = ramp_table[ramp_index(d.ramp_start, total, (int32_t)i, (int32_t)d.n_frames)];
            for (uint32_t c = 0; c < ch; c++) {
                const uint64_t sub = (uint64_t)i * ch + c;
                uint32_t w;
                if (silence) {
                    w = silence_word(sub * sb, sb, ch);
                } else {
                    w = load_be_word(s + sub * sb, sb, src_le);
                    if (atten) w = attenuate_word(w, d.attenuation);
                    if (ramp) w = ramp_word(w, mult, sb, ch, c);
                }
                store_word(o + sub * db, w, db, dst_le, zero_lsb);
            }
        }
    }
}

// resample -> ramp -> pack for a batch of OUTPUT messages of rate-converted streams.
// Straightforward form: each thread owns one output frame and walks its T taps from global memory.
__global__ __launch_bounds__(256) void src_msg_kernel_v1(const DevSrcDesc* __restrict__ descs, uint32_t n_msgs,
                                                         const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const uint16_t* __restrict__ ramp_table,
                                                         const double* __restrict__ coef, uint32_t L, uint32_t M, uint32_t T)
{
    for (uint32_t m = blockIdx.x; m < n_msgs; m += gridDim.x) {
        const DevSrcDesc d = descs[m];
        const uint32_t ch = d.channels, sb = d.src_bits >> 3, db = d.dst_bits >> 3;
        const bool src_le = d.src_endian == OHGPU_ENDIAN_LITTLE && sb > 1;
        const bool dst_le = d.dst_endian == OHGPU_ENDIAN_LITTLE;
        const bool ramp = (d.flags & OHGPU_FLAG_RAMP) != 0;
        const bool zero_lsb = (d.flags & OHGPU_FLAG_ZERO_LSB32) != 0;
        const bool planar = (d.flags & OHGPU_FLAG_SRC_PLANAR32) != 0;
        const int32_t total = (int32_t)((uint32_t)d.ramp_start - (uint32_t)d.ramp_end);
        const uint8_t* s = src + d.src_offset;
        uint8_t* o = dst + d.dst_offset;
        for (uint32_t i = threadIdx.x; i < d.n_frames; i += blockDim.x) {
            const uint64_t t = (uint64_t)d.phase0 + (uint64_t)i * M;
            const int64_t r0 = d.in_rel0 + (int64_t)(t / L);
            const uint32_t p = (uint32_t)(t % L);
            const double* cp = coef + (size_t)p * T;
            uint32_t mult = 0;
            if (ramp) mult = ramp_table[ramp_index(d.ramp_start, total, (int32_t)(d.ramp_i0 + i), (int32_t)d.ramp_n)];
            for (uint32_t c = 0; c < ch; c++) {
                double acc = 0.0;
                for (uint32_t k = 0; k < T; k++) {
                    const int64_t r = r0 - (int64_t)k;
                    if (r < 0) break;                              // frames before the stream start are zeros
                    int32_t x;                                      // S24 domain
                    if (planar) {                                   // Flac.cpp:379-417 then Msg.cpp:380-408: the value at its depth, left-justified to 24 bits
                        const int32_t v = *(const int32_t*)(s + ((uint64_t)c * d.plane_frames + (uint64_t)r) * 4);
                        x = ((int32_t)((uint32_t)v << (32 - d.src_bits))) >> 8;   // (only the src_bits low bits count, as in the pack)
                    } else {
                        const uint32_t w = load_be_word(s + ((uint64_t)r * ch + c) * sb, sb, src_le);
                        x = ((int32_t)w) >> 8;
                    }
                    acc = fma(cp[k], (double)x, acc);
                }
                uint32_t w = ((uint32_t)src_round_s24(acc)) << 8;
                if (ramp) w = ramp_word(w, mult, 3, ch, c);
                store_word(o + ((uint64_t)i * ch + c) * db, w, db, dst_le, zero_lsb);
            }
        }
    }
}

// The in-tree processors that change layout (a11, a13, a14 of SURVEY.md 8a): one workgroup per descriptor, one
// thread per frame.  Byte for byte what the cited reference loops write.
__global__ __launch_bounds__(256) void fmt_kernel_v1(const ohgpu_fmt_desc* __restrict__ descs, uint32_t n_descs,
                                                     const uint8_t* __restrict__ src, uint8_t* __restrict__ dst)
{
    for (uint32_t m = blockIdx.x; m < n_descs; m += gridDim.x) {
        const ohgpu_fmt_desc d = descs[m];
        const uint32_t ch = d.channels, sb = d.src_bits >> 3;
        for (uint32_t i = threadIdx.x; i < d.n_frames; i += blockDim.x) {
            if (d.kind == OHGPU_FMT_UNPACK_PLANAR) {            // StarvationRamper.cpp:117-147, 159-186
                const uint8_t* s = src + d.src_offset + (uint64_t)i * ch * sb;
                for (uint32_t c = 0; c < ch; c++) {
                    uint8_t* o = dst + d.dst_offset + c * d.dst_plane_stride + (uint64_t)i * 4;
                    for (uint32_t b = 0; b < 4; b++) o[b] = b < sb ? s[c * sb + b] : (uint8_t)0;
                }
            } else if (d.kind == OHGPU_FMT_SENDER_PACK) {       // Sender.cpp:351-377
                const uint32_t first = ch < 10 ? 0u : 8u;
                const uint32_t db = sb < 3 ? sb : 3, out_ch = ch < 2 ? ch : 2;
                const uint8_t* s = src + d.src_offset + (uint64_t)i * ch * sb + first * sb;
                uint8_t* o = dst + d.dst_offset + (uint64_t)i * out_ch * db;
                for (uint32_t c = 0; c < out_ch; c++)
                    for (uint32_t b = 0; b < db; b++) o[c * db + b] = s[c * sb + b];
            } else {                                            // Flac.cpp:379-417
                const uint32_t db = d.dst_bits >> 3;
                uint8_t* o = dst + d.dst_offset + (uint64_t)i * ch * db;
                for (uint32_t c = 0; c < ch; c++) {
                    const uint32_t v = *(const uint32_t*)(src + d.src_offset + c * d.src_plane_stride + (uint64_t)i * 4);
                    for (uint32_t b = 0; b < db; b++) o[c * db + b] = (uint8_t)(v >> (8 * (db - 1 - b)));
                }
            }
        }
    }
}

static uint32_t grid_for(size_t n)
{
    const size_t cap = 1u << 20;
    return (uint32_t)(n < cap ? n : cap);
}

hipError_t launch_pcm_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    if (b->n == 0) return hipSuccess;
    hipLaunchKernelGGL(pcm_msg_kernel_v1, dim3(grid_for(b->n)), dim3(256), 0, s,
                       (const ohgpu_msg_desc*)b->d_descs, (uint32_t)b->n, src, dst, ctx->d_ramp_table);
    return hipGetLastError();
}

hipError_t launch_fmt_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    (void)ctx;
    if (b->n == 0) return hipSuccess;
    hipLaunchKernelGGL(fmt_kernel_v1, dim3(grid_for(b->n)), dim3(256), 0, s, (const ohgpu_fmt_desc*)b->d_descs, (uint32_t)b->n, src, dst);
    return hipGetLastError();
}

hipError_t launch_src_v1(const ohgpu_ctx* ctx, const void* d_descs, size_t n, const ohgpu_src* flt,
                         const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(src_msg_kernel_v1, dim3(grid_for(n)), dim3(256), 0, s,
                       (const DevSrcDesc*)d_descs, (uint32_t)n, src, dst, ctx->d_ramp_table,
                       flt->d_coef, flt->L, flt->M, flt->T);
    return hipGetLastError();
}

}  // namespace ohgpu
