// Microbenchmark: issue-to-result latency of v_fmac_f64 (plain and with a DPP row_newbcast operand) on a CDNA4 SIMD.
// ONE wave per SIMD runs `iters` x 32 FMACs spread over NACC independent accumulators; with one accumulator every
// FMAC waits for the one before (cycles per FMAC = latency), with enough accumulators the 4-cycle issue rate shows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC, bool DPP>
__global__ __launch_bounds__(256) void k_lat(double* o, const double* x, int iters)
{
    extern __shared__ char pad[];
    double acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = 0.0;
    const double cv = x[threadIdx.x & 15], xv = x[threadIdx.x & 63];
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[k % NACC]) : "v"(cv), "v"(xv));
            else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[k % NACC]) : "v"(cv), "v"(xv));
        }
    }
    double s = 0;
#pragma unroll
    for (int a = 0; a < NACC; a++) s += acc[a];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    (void)pad;
}

template <int NACC, bool DPP>
static void run(double* dout, double* dx, double ghz)
{
    const int iters = 4000, blocks = 256, threads = 256;      // 4 waves per CU = one per SIMD (the LDS request keeps it to one workgroup per CU)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_lat<NACC, DPP>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_lat<NACC, DPP><<<blocks, threads, 100 * 1024>>>(dout, dx, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%s, %d accumulator(s), 1 wave/SIMD: %.3f ms = %.1f cycles per FMAC at %.1f GHz\n", DPP ? "v_fmac_f64_dpp" : "v_fmac_f64    ", NACC, best,
           best * 1e-3 * ghz * 1e9 / ((double)iters * 32), ghz);
}

int main()
{
    double *dx, *dout;
    std::vector<double> x(64, 1.0);
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 256 * 8);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    const double ghz = 2.3;
    run<1, false>(dout, dx, ghz); run<2, false>(dout, dx, ghz); run<4, false>(dout, dx, ghz);
    run<1, true>(dout, dx, ghz); run<2, true>(dout, dx, ghz); run<4, true>(dout, dx, ghz); run<8, true>(dout, dx, ghz);
    return 0;
}
