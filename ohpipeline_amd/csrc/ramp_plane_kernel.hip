// ramp_plane_kernel.hip -- RampApplicator's per-frame arithmetic for the fused resampler (row a7, Msg.cpp:826-838) ON THE DEVICE.
//
// src_lean_kernel applies a ramp from a PLANE of Q15 multipliers, one uint16 per output frame of a ramped unit (0xffff = the
// frame's message carries no ramp).  Until round 2 the planner filled those planes on the host, one CPU thread evaluating
// ramp value -> table index -> multiplier per frame at batch creation.  Now the planner only lists which message covers which
// stretch of which plane (a RampJob: 32 bytes per ramped message and unit) and this kernel does the per-frame work, with
// pcm_device.h's ramp_index_magic -- the same expression the message path's kernels use: TInt product, C division by
// (iNumSamples - 1) through the host's exact multiplier, (TUint16) cast, (kMax - ramp + 16) >> 5 limited to the last entry --
// and the context's RampArray table.  One workgroup (one wave) per job; launched by ohgpu_src_batch_create on the context's
// stream behind the memset that presets the planes to 0xffff.
#include <hip/hip_runtime.h>

#include "ohgpu_internal.h"
#include "pcm_device.h"

namespace ohgpu {

__global__ __launch_bounds__(64)
void ramp_plane_kernel(const RampJob* __restrict__ jobs, const uint32_t n_jobs, const uint16_t* __restrict__ ramp_table,
                       uint16_t* __restrict__ planes)
{
    const uint32_t jb = blockIdx.x;
    if (jb >= n_jobs) return;
    const RampJob j = jobs[jb];
    const int32_t total = (int32_t)((uint32_t)j.ramp_start - (uint32_t)j.ramp_end);       // iTotalRamp (Msg.cpp:819), negative for an up ramp
    uint16_t* const out = planes + j.plane_entry;
    for (uint32_t k = threadIdx.x; k < j.count; k += 64)
        out[k] = ramp_table[ramp_index_magic(j.ramp_start, total, j.i0 + k, j.n, j.m_n1, j.s_n1)];
}

hipError_t launch_ramp_planes(const ohgpu_ctx* ctx, const void* d_jobs, uint32_t n_jobs, void* d_planes, hipStream_t s)
{
    if (n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(ramp_plane_kernel, dim3(n_jobs), dim3(64), 0, s, (const RampJob*)d_jobs, n_jobs, ctx->d_ramp_table, (uint16_t*)d_planes);
    return hipGetLastError();
}

// The code object is loaded when a kernel of it is first asked for: ask at context creation, so that no batch's creation pays for it.
hipError_t load_ramp_plane_kernel()
{
    hipFuncAttributes attr;
    return hipFuncGetAttributes(&attr, (const void*)ramp_plane_kernel);
}

}  // namespace ohgpu
