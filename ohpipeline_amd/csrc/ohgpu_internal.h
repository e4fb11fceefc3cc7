// ohgpu_internal.h -- host-side structures behind the opaque handles of include/ohgpu.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/ohgpu.h"

namespace ohgpu {

// Device-side form of one resampled output message: everything 64-bit that can be precomputed on the
// host has been (48 bytes).  in_rel0 = n0(out_frame0) - src_frame0 (may be negative only at stream start,
// where frames with negative absolute index read as zero); phase0 = (out_frame0 * M) % L.
struct DevSrcDesc {
    uint64_t src_offset;
    uint64_t dst_offset;
    int64_t  in_rel0;
    uint32_t phase0;
    uint32_t n_frames;
    uint16_t ramp_start;
    uint16_t ramp_end;
    uint8_t  channels;
    uint8_t  src_bits;
    uint8_t  src_endian;
    uint8_t  dst_bits;
    uint8_t  dst_endian;
    uint8_t  flags;
    uint8_t  pad[6];
};
static_assert(sizeof(DevSrcDesc) == 48, "DevSrcDesc layout");
static_assert(sizeof(ohgpu_msg_desc) == 32, "ohgpu_msg_desc layout");
static_assert(sizeof(ohgpu_src_msg_desc) == 64, "ohgpu_src_msg_desc layout");

enum BatchKind { kBatchPcm = 1, kBatchSrc = 2 };

}  // namespace ohgpu

struct ohgpu_ctx {
    int          device;
    hipStream_t  stream;          // the context's own stream (used when the caller passes NULL)
    uint16_t*    d_ramp_table;    // 512 x u16 (RampArray.h:7-74)
    int          variant;         // kernel selection, 0 = best
    int          num_cus;
    char         name[128];
};

struct ohgpu_src {
    uint32_t L, M, T;
    double*  d_coef;              // [L][T] exact integer-valued doubles (Q28)
    int32_t* d_coef_q28;          // [L][T] int32
};

struct ohgpu_batch {
    int      kind;
    size_t   n;
    void*    d_descs;             // ohgpu_msg_desc[] or DevSrcDesc[]
    const ohgpu_src* src;         // kBatchSrc only
    uint64_t src_arena_bytes, dst_arena_bytes;
    uint64_t in_frames, out_frames, src_bytes_touched, dst_bytes_written;
    uint32_t max_frames;          // largest n_frames in the batch
    bool     uniform;             // every descriptor has the same format fields
    uint8_t  channels, src_bits, src_endian, dst_bits, dst_endian;
};

namespace ohgpu {

int set_error(int code, const char* fmt, ...);

#define OHGPU_HIP_TRY(expr)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return ::ohgpu::set_error(OHGPU_ERR_DEVICE, "%s failed: %s", #expr,  \
                                                        hipGetErrorString(e_));                     \
    } while (0)

// kernels (pcm_kernels.hip)
hipError_t launch_pcm_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);
hipError_t launch_src_v1(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s);

// host helpers (host_design.cpp)
void build_ramp_table(uint16_t out[512]);
int  design_src(uint32_t rate_in, uint32_t rate_out, uint32_t T, double beta, double f_pass,
                std::vector<int32_t>* coef_q28, uint32_t* L, uint32_t* M);

}  // namespace ohgpu
