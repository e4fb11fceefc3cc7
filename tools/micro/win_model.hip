// Microbenchmark: does v_fmac_f64_dpp slow down when its sample operand comes from 32 different registers (the block kernel's
// window) rather than 4?  Each iteration = 32 taps in two statements of 16 (as the kernel issues them), one chain.
// Variants: window of 4 / 32 registers; coefficient register pair fixed.  Reports ns per iteration per SIMD at 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define FM(k, x) "v_fmac_f64_dpp %[a0], %[cv], " x " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void taps16(double& acc, const double cv, const double w15, const double w14, const double w13, const double w12,
    const double w11, const double w10, const double w9, const double w8, const double w7, const double w6, const double w5, const double w4,
    const double w3, const double w2, const double w1, const double w0)
{
    asm volatile(FM(15, "%[w15]") FM(14, "%[w14]") FM(13, "%[w13]") FM(12, "%[w12]") FM(11, "%[w11]") FM(10, "%[w10]") FM(9, "%[w9]") FM(8, "%[w8]")
                 FM(7, "%[w7]") FM(6, "%[w6]") FM(5, "%[w5]") FM(4, "%[w4]") FM(3, "%[w3]") FM(2, "%[w2]") FM(1, "%[w1]") FM(0, "%[w0]")
                 : [a0] "+v"(acc) : [cv] "v"(cv), [w15] "v"(w15), [w14] "v"(w14), [w13] "v"(w13), [w12] "v"(w12), [w11] "v"(w11), [w10] "v"(w10),
                   [w9] "v"(w9), [w8] "v"(w8), [w7] "v"(w7), [w6] "v"(w6), [w5] "v"(w5), [w4] "v"(w4), [w3] "v"(w3), [w2] "v"(w2), [w1] "v"(w1), [w0] "v"(w0));
}
template <int NW>
__global__ __launch_bounds__(768) void k(double* o, const double* x, int iters)
{
    extern __shared__ char pad[];
    double w[32];
#pragma unroll
    for (int i = 0; i < 32; i++) w[i] = x[(threadIdx.x + i) & 63];
    double c0 = x[threadIdx.x & 15], c1 = x[16 + (threadIdx.x & 15)], acc = 0.0, tot = 0.0;
    for (int it = 0; it < iters; it++) {
        asm volatile("v_mov_b64 %0, 0.5" : "=v"(acc));
#define W(i) w[(i) % NW]
        taps16(acc, c1, W(31), W(30), W(29), W(28), W(27), W(26), W(25), W(24), W(23), W(22), W(21), W(20), W(19), W(18), W(17), W(16));
        taps16(acc, c0, W(15), W(14), W(13), W(12), W(11), W(10), W(9), W(8), W(7), W(6), W(5), W(4), W(3), W(2), W(1), W(0));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(tot) : "v"(acc));
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = tot;
    (void)pad;
}
template <int NW> static void run(double* dout, double* dx)
{
    const int iters = 20000, blocks = 256, threads = 768;
    hipFuncSetAttribute((const void*)k<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NW>, dim3(blocks), dim3(threads), 100 * 1024, 0, dout, dx, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("window of %2d registers: %.3f ms = %.1f ns per 34-instruction iteration per SIMD (3 waves per SIMD) = %.1f TFLOP/s fp64\n", NW, best,
           best * 1e6 / (iters * 3.0), 256.0 * 12 * iters * 32 * 128 / best / 1e9);
}
int main()
{
    double *dx, *dout;
    std::vector<double> x(64);
    for (int i = 0; i < 64; i++) x[i] = 1.0 + i * 0.001;
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 768 * 8);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    run<4>(dout, dx); run<32>(dout, dx); run<4>(dout, dx); run<32>(dout, dx);
    return 0;
}
