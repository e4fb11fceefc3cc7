// Microbenchmark: fp64 FMAC with a DPP row_newbcast coefficient (one VGPR pair holds 16 taps, lane k of each
// 16-lane row) against the same FMAC with an SGPR coefficient.  Checks the broadcast semantics and the issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 dpp_fma.hip -o dpp_fma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FM(k) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(cv), "v"(xv))
__global__ void k_dpp(double* o, const double* c, const double* x, int iters)
{
    double acc = 0.0, cv = c[threadIdx.x & 15], xv = x[threadIdx.x];
    for (int i = 0; i < iters; i++) {
        FM(0); FM(1); FM(2); FM(3); FM(4); FM(5); FM(6); FM(7); FM(8); FM(9); FM(10); FM(11); FM(12); FM(13); FM(14); FM(15);
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void k_sgpr(double* o, const double* c, const double* x, int iters)
{
    double acc = 0.0, xv = x[threadIdx.x];
    double cs[16];
    for (int k = 0; k < 16; k++) cs[k] = c[k];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "s"(cs[k]), "v"(xv));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    const int iters = 4096, blocks = 256 * 8, threads = 256;
    std::vector<double> c(16), x(256);
    for (int i = 0; i < 16; i++) c[i] = i + 1;
    for (int i = 0; i < 256; i++) x[i] = 1.0 + i;
    double *dc, *dx, *dout;
    hipMalloc(&dc, 16 * 8); hipMalloc(&dx, 256 * 8); hipMalloc(&dout, (size_t)blocks * threads * 8);
    hipMemcpy(dc, c.data(), 16 * 8, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), 256 * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            if (which == 0) k_dpp<<<blocks, threads>>>(dout, dc, dx, iters); else k_sgpr<<<blocks, threads>>>(dout, dc, dx, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        std::vector<double> out(256);
        hipMemcpy(out.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; i++) if (out[i] != 136.0 * iters * x[i]) bad++;
        const double fma = (double)blocks * threads * iters * 16;
        printf("%s: %.3f ms, %.2f TFLOP/s fp64, wrong lanes %d (lane 5: %.1f expect %.1f)\n", which == 0 ? "dpp row_newbcast" : "sgpr operand",
               best, 2 * fma / best / 1e9, bad, out[5], 136.0 * iters * x[5]);
    }
    return 0;
}
