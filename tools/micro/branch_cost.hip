// Microbenchmark: what a TAKEN scalar branch costs a wave on a CDNA4 SIMD.  Each iteration issues 16 v_fmac_f64 and
// NB branches (`s_branch` to the next instruction: always taken, nothing skipped).  W waves per SIMD.  If a taken branch
// only cost its issue slot the time would not move; what it adds per branch is the refill of the wave's instruction
// buffer after the redirect.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NB>
__global__ __launch_bounds__(1024) void k_br(double* o, const double* x, int iters)
{
    extern __shared__ char pad[];
    double a0 = 0.0, a1 = 0.0, xv = x[threadIdx.x & 63], c = 1.0 + (threadIdx.x & 1);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(c), "v"(xv));
            asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(c), "v"(xv));
            if (k < NB) asm volatile("s_branch 0");
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    (void)pad;
}

template <int NB>
static double run(int waves_per_simd, double* dout, double* dx, double ghz)
{
    const int iters = 4000, blocks = 256, threads = waves_per_simd * 4 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_br<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_br<NB><<<blocks, threads, 100 * 1024>>>(dout, dx, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e-3 * ghz * 1e9 / iters;      // cycles per iteration (wall), all waves of the SIMD together
}

int main()
{
    double *dx, *dout;
    std::vector<double> x(64, 1.0);
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 1024 * 8);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    const double ghz = 2.2;
    for (int w : {1, 3}) {
        const double c0 = run<0>(w, dout, dx, ghz), c2 = run<2>(w, dout, dx, ghz), c4 = run<4>(w, dout, dx, ghz), c8 = run<8>(w, dout, dx, ghz);
        printf("%d wave(s)/SIMD, 16 fp64 FMA per iteration: 0 / 2 / 4 / 8 taken branches -> %.0f / %.0f / %.0f / %.0f cycles per iteration  (~%.0f cycles per branch)\n",
               w, c0, c2, c4, c8, (c8 - c0) / 8);
    }
    return 0;
}
