// Microbenchmark: does scalar work co-issue with fp64 vector work on a CDNA4 SIMD at low occupancy?
// Each wave runs `iters` x {32 dependent-pair v_fmac_f64 + K scalar adds}; W waves per SIMD.  If the time grows by
// 4 cycles per scalar instruction per wave, the SIMD issues one instruction at a time whatever its type.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ __launch_bounds__(1024) void k_mix(double* o, const double* x, int iters, int* so)
{
    extern __shared__ char pad[];
    double a0 = 0.0, a1 = 0.0, xv = x[threadIdx.x & 63], c = 1.0 + (threadIdx.x & 1);
    int s0 = iters, s1 = 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a0) : "v"(c), "v"(xv));
            asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a1) : "v"(c), "v"(xv));
            if (k < K / 2) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_xor_b32 %0, %0, %1" : "+s"(s1) : "s"(s0) : "scc"); }
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1;
    if (threadIdx.x == 0) so[blockIdx.x] = s0 + s1;
    (void)pad;
}

template <int K>
static void run(int waves_per_simd, double* dout, double* dx, int* dso)
{
    const int iters = 2000, blocks = 256, threads = waves_per_simd * 4 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    hipFuncSetAttribute((const void*)k_mix<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_mix<K><<<blocks, threads, 100 * 1024>>>(dout, dx, iters, dso);     // LDS keeps it to one workgroup per CU
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // cycles per wave-iteration on a SIMD, assuming ~2.0 GHz under fp64 load
    const double cyc = best * 1e-3 * 2.0e9 / ((double)iters * waves_per_simd);
    printf("waves/SIMD %d, %2d scalar + 32 fp64 FMA per iteration: %.3f ms, ~%.0f SIMD cycles per wave-iteration (32 FMA alone = 128)\n",
           waves_per_simd, K, best, cyc);
}

int main()
{
    double *dx, *dout; int* dso;
    std::vector<double> x(64, 1.0);
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 1024 * 8); hipMalloc(&dso, 256 * 4);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3, 4}) {
        run<0>(w, dout, dx, dso); run<8>(w, dout, dx, dso); run<16>(w, dout, dx, dso); run<32>(w, dout, dx, dso);
    }
    return 0;
}
