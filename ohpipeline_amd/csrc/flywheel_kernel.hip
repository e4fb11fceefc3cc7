// flywheel_kernel.hip -- FlywheelRamper for a batch of starving streams.
//
// The reference runs, per channel of a stream that has run dry, Burg's method (degree 3, 16-bit integer) over the last
// millisecond of audio and then a 3-state integer feedback filter for 20 ms of ramp audio
// (OpenHome/Media/FlywheelRamper.cpp).  Both are serial recurrences with data-dependent 16/32-bit wrap-around, so there is
// no parallelism inside a channel: ONE LANE = ONE CHANNEL OF ONE REQUEST, and a batch of N streams x C channels fills
// N*C/64 waves.  Not an HBM- or MFMA-bound kernel: its bound is the dependent integer chain per output sample
// (three 32x32->64 multiplies, each one issue slot of the quarter-rate integer multiplier), hidden only by having many
// waves.  Burg's working arrays (decimated input, forward and backward prediction errors: 3 x count x int16 per lane)
// live in a global workspace laid out [array][index][lane], so the 64 lanes of a wave touch one contiguous 128-byte run
// per access.  Every expression follows the reference's integer semantics, wrapping where C's implicit conversions wrap.
#include <hip/hip_runtime.h>

#include <vector>

#include "ohgpu_internal.h"

namespace ohgpu {

__device__ __forceinline__ int32_t wrap16(int32_t v) { return (int32_t)(int16_t)(uint16_t)(uint32_t)v; }   // (TInt16) conversion

__device__ __forceinline__ uint32_t decimation_factor(uint32_t rate)       // FlywheelRamper::DecimationFactor, :316-331
{
    return (rate == 192000 || rate == 176400) ? 4u : ((rate == 88200 || rate == 96000) ? 2u : 1u);
}

__global__ __launch_bounds__(64) void flywheel_kernel(const ohgpu_flywheel_desc* __restrict__ descs,
                                                      const FlywheelLane* __restrict__ lanes, const uint32_t n_lanes,
                                                      const uint32_t lanes_padded, const uint32_t max_count,
                                                      int16_t* __restrict__ work, const uint8_t* __restrict__ src,
                                                      uint8_t* __restrict__ dst)
{
    constexpr uint32_t kDegree = 3, kShift = 13;                           // kDegree :14, kBurgScaleShift = 16 - 3 :13
    const uint32_t gl = blockIdx.x * 64 + threadIdx.x;
    if (gl >= n_lanes) return;
    const FlywheelLane ln = lanes[gl];
    const ohgpu_flywheel_desc d = descs[ln.req];
    const uint32_t dec = decimation_factor(d.sample_rate);
    const uint32_t count = d.in_samples / dec;                             // aSamples.Bytes() / (kBytesPerSample * decFactor), :196
    int16_t* const xs = work + gl;                                         // [i] at xs[i * lanes_padded]
    int16_t* const per = work + (size_t)max_count * lanes_padded + gl;
    int16_t* const pef = work + (size_t)2 * max_count * lanes_padded + gl;
    const size_t st = lanes_padded;

    // ---- FlywheelRamper::Initialise, :176-226: decimate, keep the top 16 bits descaled by one bit, remember the last
    // three full-width samples (newest first) as the feedback filter's initial state
    const uint8_t* p = src + d.src_offset + (uint64_t)ln.channel * d.channel_bytes + (d.channel_bytes - (uint64_t)d.in_samples * 4);
    int32_t f0 = 0, f1 = 0, f2 = 0;                                        // iFeedbackSamples[0..2]
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t s = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        p += 4 * dec;
        f2 = f1; f1 = f0; f0 = (int32_t)s;                                 // [count - i - 1] = sample for the last kDegree
        xs[i * st] = (int16_t)((int32_t)(int16_t)(uint16_t)(s >> 16) >> 1);   // sample16 >> kBurgDataDescaleBitCount
        per[i * st] = 0;
        pef[i * st] = 0;
    }

    // ---- FlywheelRamper::BurgsMethod, :246-314
    int32_t out0 = 0, out1 = 0, out2 = 0;                                  // aOutput[0..2] (TInt16 values)
    uint32_t limit1 = count - 1, limit2 = limit1;
    for (uint32_t n = 0; n < kDegree; n++) {
        int32_t sn = 0, sd = 0;
        for (uint32_t j = 0; j < limit1; j++) {
            const int32_t t1 = wrap16((int32_t)xs[(j + n + 1) * st] + pef[j * st]);
            const int32_t t2 = wrap16((int32_t)xs[j * st] + per[j * st]);
            sn = (int32_t)((uint32_t)sn - 2u * (uint32_t)(t1 * t2));
            sd = (int32_t)((uint32_t)sd + (uint32_t)(t1 * t1) + (uint32_t)(t2 * t2));
        }
        limit1--;
        int32_t t3 = 0;
        if (sn != 0) t3 = wrap16((int32_t)(((int64_t)sn * 8192) / (int64_t)sd));   // (TInt16)((sn << 13) / sd), C division
        int32_t h0 = 0, h1 = 0;
        if (n == 0) {
            out0 = t3;
        } else if (n == 1) {                                               // aH[j] = (TInt16)((t3 * aOutput[n-j-1]) >> 13) + aOutput[j]
            out1 = t3;
            h0 = wrap16(wrap16((t3 * out0) >> kShift) + out0);
            out0 = h0;
            limit2--;
        } else {
            out2 = t3;
            h0 = wrap16(wrap16((t3 * out1) >> kShift) + out0);
            h1 = wrap16(wrap16((t3 * out0) >> kShift) + out1);
            out0 = h0; out1 = h1;
            limit2--;
        }
        if (n == kDegree - 1) break;
        for (uint32_t j = 0; j < limit2; j++) {                            // :301-312
            const uint32_t i = j + 1;
            const int32_t pe = (int32_t)((uint32_t)((int32_t)pef[j * st] + xs[(i + n) * st]) * (uint32_t)t3);
            per[j * st] = (int16_t)wrap16((int32_t)per[j * st] + wrap16(pe >> kShift));
            const int32_t fe = (int32_t)((uint32_t)((int32_t)per[i * st] + xs[i * st]) * (uint32_t)t3);
            pef[j * st] = (int16_t)wrap16(wrap16(fe >> kShift) + pef[i * st]);
        }
    }

    // ---- CorrectBurgCoeffs :333-340 with CoeffOverflow :342-372 (format 3: one = 1 << 13), PrepareFeedbackCoeffs :228-235
    {
        const int32_t one = 1 << 13;
        const int32_t total = wrap16(wrap16(wrap16(out0) + out1) + out2);
        int32_t excess = 0;
        if (!(total <= one && total >= -one)) excess = (total & 0x8000) ? wrap16(total + one) : wrap16(total - one);
        if (excess != 0) out0 = wrap16(out0 - excess * 2);
    }
    const int32_t c0 = (int32_t)(0u - ((uint32_t)out0 << 16)), c1 = (int32_t)(0u - ((uint32_t)out1 << 16)),
                  c2 = (int32_t)(0u - ((uint32_t)out2 << 16));            // -(((TInt32)coeff) << 16)

    // ---- FlywheelRamperManager::RenderChannels :83-131 over FeedbackModel(3, 0, 3, 1, 1)::NextSample :449-487
    uint8_t* o = dst + d.dst_offset + (uint64_t)ln.channel * 4;
    const uint64_t frame_bytes = (uint64_t)d.channels * 4;
    uint32_t remaining = d.out_frames;
    int32_t prev = 0;
    while (remaining > 0) {                                                // blocks of <= 1 ms, :52-63
        const uint32_t nb = remaining > d.block_frames ? d.block_frames : remaining;
        remaining -= nb;
        uint32_t hold = 0;
        for (uint32_t j = 0; j < nb; j++) {
            if (hold == 0) {
                uint32_t sum = (uint32_t)(int32_t)(((int64_t)f0 * c0) >> 32);
                sum += (uint32_t)(int32_t)(((int64_t)f1 * c1) >> 32);
                sum += (uint32_t)(int32_t)(((int64_t)f2 * c2) >> 32);
                f2 = f1; f1 = f0;
                f0 = (int32_t)(sum << 3);                                  // sum <<= iCoeffFormat; output shift is 0
                prev = f0;
            }
            o[0] = (uint8_t)((uint32_t)prev >> 24); o[1] = (uint8_t)((uint32_t)prev >> 16);
            o[2] = (uint8_t)((uint32_t)prev >> 8); o[3] = (uint8_t)prev;
            o += frame_bytes;
            if (++hold == dec) hold = 0;
        }
    }
}

void free_flywheel(ohgpu_ctx* ctx, ohgpu_batch* b)
{
    if (b->fly.d_lanes) ctx_dev_free(ctx, b->fly.d_lanes);
    if (b->fly.d_work) ctx_dev_free(ctx, b->fly.d_work);
    b->fly = FlywheelPlan();
}

int plan_flywheel(ohgpu_ctx* ctx, ohgpu_batch* b, const ohgpu_flywheel_desc* descs, size_t n)
{
    (void)ctx;
    b->fly = FlywheelPlan();
    std::vector<FlywheelLane> lanes;
    uint32_t max_count = 0;
    for (size_t i = 0; i < n; i++) {
        const ohgpu_flywheel_desc& d = descs[i];
        const uint32_t dec = (d.sample_rate == 192000 || d.sample_rate == 176400) ? 4 : ((d.sample_rate == 88200 || d.sample_rate == 96000) ? 2 : 1);
        if (d.in_samples / dec > max_count) max_count = d.in_samples / dec;
        for (uint32_t c = 0; c < d.channels; c++) lanes.push_back(FlywheelLane{(uint32_t)i, c});
    }
    if (lanes.empty()) return OHGPU_OK;
    const uint32_t padded = (uint32_t)((lanes.size() + 63) / 64 * 64);
    hipError_t e = ctx_dev_alloc(ctx, &b->fly.d_lanes, lanes.size() * sizeof(FlywheelLane));
    if (e == hipSuccess) e = hipMemcpy(b->fly.d_lanes, lanes.data(), lanes.size() * sizeof(FlywheelLane), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ctx_dev_alloc(ctx, &b->fly.d_work, (size_t)3 * max_count * padded * sizeof(int16_t));
    if (e != hipSuccess) {
        free_flywheel(ctx, b);
        return set_error(e == hipErrorOutOfMemory ? OHGPU_ERR_NOMEM : OHGPU_ERR_DEVICE, "flywheel plan: %s", hipGetErrorString(e));
    }
    b->fly.n_lanes = (uint32_t)lanes.size();
    b->fly.lanes_padded = padded;
    b->fly.max_count = max_count;
    return OHGPU_OK;
}

hipError_t launch_flywheel(const ohgpu_ctx* ctx, const ohgpu_batch* b, const uint8_t* src, uint8_t* dst, hipStream_t s)
{
    (void)ctx;
    if (b->fly.n_lanes == 0) return hipSuccess;
    hipLaunchKernelGGL(flywheel_kernel, dim3(b->fly.lanes_padded / 64), dim3(64), 0, s,
                       (const ohgpu_flywheel_desc*)b->d_descs, (const FlywheelLane*)b->fly.d_lanes, b->fly.n_lanes,
                       b->fly.lanes_padded, b->fly.max_count, (int16_t*)b->fly.d_work, src, dst);
    return hipGetLastError();
}

}  // namespace ohgpu
