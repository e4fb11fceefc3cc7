// Microbenchmark: v_mfma_f64_16x16x4_f64 throughput on MI355X, alone and sharing the SIMD with vector work.
// mode 0: every wave issues MFMAs only; mode 1: every wave alternates 1 MFMA with V vector instructions;
// mode 2: even waves MFMA only, odd waves vector fp64 FMA only (do the two pipes overlap across waves?).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(1024) void k(double* o, int iters, int mode)
{
    extern __shared__ char pad[];
    const int wave = threadIdx.x >> 6;
    double a = 1.0 + (threadIdx.x & 3), b = 0.5 + (threadIdx.x & 7);
    d4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    unsigned i0 = threadIdx.x, i1 = 3, i2 = 0x03020100, i3 = 7;
    const bool do_mfma = mode != 2 || (wave & 1) == 0;
    const bool do_valu = mode == 1 || (mode == 2 && (wave & 1) == 1);
    for (int i = 0; i < iters; i++) {
        if (do_mfma) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int q = 0; q < (V > 0 ? V : 1); q += 4) {
#ifdef INT_VALU
                asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(i0) : "v"(i1), "v"(i2));
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(i2));
                asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(i2) : "v"(i3));
                asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i3) : "v"(i0));
#else
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v0) : "v"(a), "v"(b));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v1) : "v"(a), "v"(b));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v2) : "v"(a), "v"(b));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(v3) : "v"(a), "v"(b));
#endif
            }
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = c0.x + c0.y + c1.z + c1.w + v0 + v1 + v2 + v3 + (double)(i0 ^ i1 ^ i2 ^ i3);
    (void)pad;
}

template <int V>
static void run(int waves_per_simd, int mode, double* dout)
{
    const int iters = 4000, blocks = 256, threads = waves_per_simd * 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k<V><<<blocks, threads, 100 * 1024>>>(dout, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double waves = (double)blocks * threads / 64;
    const double mfma_waves = mode == 2 ? waves / 2 : waves;
    const double valu_waves = mode == 0 ? 0 : (mode == 2 ? waves / 2 : waves);
    const double mfma_flop = mfma_waves * iters * 2.0 * (16 * 16 * 4 * 2);
    const double valu_flop = valu_waves * iters * (double)V * 64 * 2;
    printf("waves/SIMD %d mode %d V=%2d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  (cycles per MFMA per SIMD at 2.0 GHz: %.1f)\n",
           waves_per_simd, mode, V, best, mfma_flop / best / 1e9, valu_flop / best / 1e9,
           best * 1e-3 * 2.0e9 / (iters * 2.0 * (mode == 2 ? waves_per_simd / 2.0 : waves_per_simd)));
}

int main()
{
    double* dout; hipMalloc(&dout, 256 * 1024 * 8);
    for (int w : {1, 2, 4}) run<0>(w, 0, dout);
    for (int w : {2, 4}) { run<8>(w, 1, dout); run<16>(w, 1, dout); run<32>(w, 1, dout); }
    for (int w : {2, 4}) { run<16>(w, 2, dout); run<32>(w, 2, dout); }
    return 0;
}
