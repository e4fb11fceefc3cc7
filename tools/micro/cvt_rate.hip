// Microbenchmark: issue cost of the non-FMA fp64 instructions in the block kernel's per-output tail
// (v_floor_f64, v_cvt_i32_f64, v_cvt_f64_i32, v_add_f64) against v_fmac_f64, on a CDNA4 SIMD.
// Each wave runs `iters` x 64 independent instances of one instruction (8 registers round-robin, so latency does not
// bound the rate at 4 waves per SIMD); reported: SIMD cycles per wave-instruction, assuming the clock given below.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { FMA, ADD, FLOOR, CVT_I32_F64, CVT_F64_I32, MOV_B64, MED3, PERM };

template <int OP>
__global__ __launch_bounds__(1024) void k_rate(double* o, const double* x, int iters)
{
    double a[8];
    int b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { a[r] = x[(threadIdx.x + r) & 63]; b[r] = (int)threadIdx.x + r; }
    const double c = 1.0 + (threadIdx.x & 1);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const int r = k & 7;
            if (OP == FMA) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[r]) : "v"(c), "v"(c));
            if (OP == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[r]) : "v"(c));
            if (OP == FLOOR) asm volatile("v_floor_f64 %0, %0" : "+v"(a[r]));
            if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b[r]) : "v"(a[r]));
            if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[r]) : "v"(b[r]));
            if (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(a[r]) : "v"(c));
            if (OP == MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(b[r]) : "v"(b[(r + 1) & 7]), "v"(b[(r + 2) & 7]));
            if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(b[r]) : "v"(b[(r + 1) & 7]), "v"(b[(r + 2) & 7]));
        }
    }
    double s = 0; int t = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) { s += a[r]; t += b[r]; }
    o[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}

template <int OP>
static void run(const char* name, double* dout, double* dx, double ghz)
{
    const int iters = 2000, blocks = 256, waves_per_simd = 4, threads = waves_per_simd * 4 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_rate<OP><<<blocks, threads>>>(dout, dx, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * ghz * 1e9 / ((double)iters * 64 * waves_per_simd);
    printf("%-16s %.3f ms  ~%.1f SIMD cycles per wave-instruction at %.1f GHz\n", name, best, cyc, ghz);
}

int main()
{
    double *dx, *dout;
    std::vector<double> x(64, 1.5);
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 1024 * 8);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    const double ghz = 2.1;
    run<FMA>("v_fmac_f64", dout, dx, ghz);
    run<ADD>("v_add_f64", dout, dx, ghz);
    run<FLOOR>("v_floor_f64", dout, dx, ghz);
    run<CVT_I32_F64>("v_cvt_i32_f64", dout, dx, ghz);
    run<CVT_F64_I32>("v_cvt_f64_i32", dout, dx, ghz);
    run<MOV_B64>("v_mov_b64", dout, dx, ghz);
    run<MED3>("v_med3_i32", dout, dx, ghz);
    run<PERM>("v_perm_b32", dout, dx, ghz);
    return 0;
}
