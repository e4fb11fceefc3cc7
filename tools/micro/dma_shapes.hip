// Microbenchmark: what a wave-instruction of LDS-DMA (global_load_lds_dwordx4, 64 lanes x 16 bytes) costs by the SHAPE of what it
// fetches -- the lean resampler's staging (32 rows of a few 16-byte pieces each, rows 7 KB apart, 3.5 instructions per stage of
// 96 bytes per row) against the same bytes as exact 6-piece rows, as whole aligned 128-byte lines, as 8-frame stages (4 pieces
// per row), and as one contiguous run.  11 waves per CU, every CU, nothing but the DMA and its vmcnt(0) per stage: the ceiling
// the staging path has by itself.  Prints per shape: ms, useful GB/s, cycles per stage and wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Shape { const char* name; int rows, pieces, advance, line_align, stride; };

__global__ __launch_bounds__(704) void k(const unsigned char* src, int rows, int pieces, int advance, int line_align, int stride, int stages,
                                          size_t wave_region, unsigned long long* cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x & 63, wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
    const size_t wid = (size_t)blockIdx.x * nw + wave;
    const unsigned char* base = src + wid * wave_region;
    const int total = rows * pieces, iters = (total + 63) / 64;
    unsigned off[8];
    for (int it = 0; it < iters && it < 8; it++) {
        int idx = it * 64 + (int)lane;
        if (idx >= total) idx = 0;
        const int r = idx / pieces, p = idx - r * pieces;
        const unsigned rb = (unsigned)r * (unsigned)stride;
        off[it] = (line_align ? (rb & ~127u) : (rb & ~15u)) + 16u * (unsigned)p;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long b64 = (unsigned long long)(uintptr_t)base;
    unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);                  // (wave-uniform: SGPRs)
    for (int q = 0; q < stages; q++) {
        const unsigned buf = lds0 + (unsigned)(q & 1) * 4096u;
        for (int it = 0; it < iters && it < 8; it++) {
            const unsigned m0v = buf + (unsigned)it * 1024u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off[it]), "s"(sb), "s"(m0v) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sb += (unsigned long long)advance;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[wid] = t1 - t0;
}

int main()
{
    const int blocks = 256, waves = 11, stages = 512;
    const size_t wave_region = 1u << 20;                       // 32 rows x 7 KB + 512 stages x 128 bytes fits
    const size_t bytes = (size_t)blocks * waves * wave_region + (1u << 20);
    unsigned char* d; unsigned long long* c;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&c, sizeof(unsigned long long) * blocks * waves) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, waves * 8192);
    const Shape shapes[] = {
        {"lean staging: 32 rows x 7 pieces, 96 B per stage", 32, 7, 96, 0, 7056},
        {"exact: 32 rows x 6 pieces, 96 B per stage", 32, 6, 96, 0, 7056},
        {"whole lines: 32 rows x 8 pieces, line-aligned, 128 B per stage", 32, 8, 128, 1, 7056},
        {"8-frame stages: 32 rows x 4 pieces, 48 B per stage", 32, 4, 48, 0, 7056},
        {"one-block rows: 32 rows x 7 pieces, rows 882 B apart", 32, 7, 96, 0, 882},
        {"contiguous: 1 row x 256 pieces (4 KiB per stage)", 1, 256, 4096, 1, 0},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* h = new unsigned long long[blocks * waves];
    for (const Shape& s : shapes) {
        float best = 1e9f; double cyc_mean = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), waves * 8192, 0, d, s.rows, s.pieces, s.advance, s.line_align, s.stride, stages, wave_region, c);
            hipEventRecord(e1); hipEventSynchronize(e1);
            if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) {
                best = ms;
                hipMemcpy(h, c, sizeof(unsigned long long) * blocks * waves, hipMemcpyDeviceToHost);
                cyc_mean = 0; for (int i = 0; i < blocks * waves; i++) cyc_mean += (double)h[i];
                cyc_mean /= (double)blocks * waves * stages;
            }
        }
        const double useful = (double)blocks * waves * stages * s.rows * (s.rows == 1 ? 4096.0 : (double)s.advance);
        const int instr = (s.rows * s.pieces + 63) / 64;
        printf("%-66s %7.3f ms  %6.0f GB/s useful  %6.0f cycles per stage and wave (%d DMA instructions: %4.0f each)\n", s.name, best,
               useful / best / 1e6, cyc_mean, instr, cyc_mean / instr);
    }
    return 0;
}
