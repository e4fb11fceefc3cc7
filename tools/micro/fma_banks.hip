// Microbenchmark: does the VGPR bank (register number mod 4) of v_fmac_f64_dpp's operands change its issue rate?
// Every variant runs the same 32 dependent FMACs per trip with hard-coded registers: the accumulator, the DPP
// (coefficient) operand and 16 "window" operands sit in chosen bank pairs ({0,1} = number % 4 == 0, {2,3} = % 4 == 2).
// Build: hipcc --offload-arch=gfx950 -O3 fma_banks.hip -o fma_banks ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define CLOBBERS "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
    "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55", \
    "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","s20","scc","vcc"

// ACC, COEF: first register of the pair; W0: first window pair, WS: stride between window pairs (4 keeps the bank pair, 2 alternates)
template <int ACC, int COEF, int W0, int WS>
__global__ __launch_bounds__(256) void k(double* o, int iters)
{
    uint32_t lo, hi;
    asm volatile(
        ".macro FILL first, stride\n"
        "  .set r, \\first\n"
        "  .rept 16\n"
        "    v_cvt_f64_i32 v[r:r+1], v0\n"
        "    .set r, r + \\stride\n"
        "  .endr\n"
        ".endm\n"
        ".macro TAPS acc, coef, first, stride\n"
        "  .set r, \\first\n"
        "  .set b, 15\n"
        "  .rept 16\n"
        "    v_fmac_f64_dpp v[\\acc:\\acc+1], v[\\coef:\\coef+1], v[r:r+1] row_newbcast:b row_mask:0xf bank_mask:0xf\n"
        "    .set r, r + \\stride\n"
        "    .set b, b - 1\n"
        "  .endr\n"
        ".endm\n"
        "FILL %[w0], %[ws]\n"
        "v_cvt_f64_i32 v[%[coef]:%[coef]+1], v0\n"
        "v_mov_b32 v[%[acc]], 0\n"
        "v_mov_b32 v[%[acc]+1], 0\n"
        "s_mov_b32 s20, %[n]\n"
        "1:\n"
        "TAPS %[acc], %[coef], %[w0], %[ws]\n"
        "TAPS %[acc], %[coef], %[w0], %[ws]\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "v_mov_b32 %[lo], v[%[acc]]\n"
        "v_mov_b32 %[hi], v[%[acc]+1]\n"
        ".purgem FILL\n"
        ".purgem TAPS\n"
        : [lo] "=&v"(lo), [hi] "=&v"(hi) : [n] "s"(iters), [acc] "i"(ACC), [coef] "i"(COEF), [w0] "i"(W0), [ws] "i"(WS) : CLOBBERS);
    o[blockIdx.x * blockDim.x + threadIdx.x] = __hiloint2double((int)hi, (int)lo);
}

template <int ACC, int COEF, int W0, int WS>
static void run(const char* what, double* dout, int blocks)
{
    const int iters = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k<ACC, COEF, W0, WS><<<blocks, 256>>>(dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double fma = (double)blocks * 256 * iters * 32;
    // cycles per wave-FMAC on one SIMD at 2.4 GHz: blocks*4 waves over 1024 SIMDs
    const double per = best * 1e-3 * 2.4e9 / ((double)blocks * 4 / 1024 * iters * 32);
    printf("%-58s %.3f ms  %.1f TFLOP/s  %.2f cycles/FMAC/SIMD (at 2.4 GHz)\n", what, best, 2 * fma / best / 1e9, per);
}

int main()
{
    const int blocks = 256 * 3 * 4;      // three waves per SIMD, four rounds
    double* dout; hipMalloc(&dout, (size_t)blocks * 256 * 8);
    run<88, 8, 20, 4>("acc {0,1} coef {0,1} window {0,1}", dout, blocks);
    run<88, 8, 22, 4>("acc {0,1} coef {0,1} window {2,3}", dout, blocks);
    run<90, 8, 22, 4>("acc {2,3} coef {0,1} window {2,3}", dout, blocks);
    run<90, 8, 20, 4>("acc {2,3} coef {0,1} window {0,1}", dout, blocks);
    run<88, 10, 20, 4>("acc {0,1} coef {2,3} window {0,1}", dout, blocks);
    run<88, 8, 20, 2>("acc {0,1} coef {0,1} window alternating", dout, blocks);
    return 0;
}
