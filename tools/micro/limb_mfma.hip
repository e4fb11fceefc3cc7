// Microbenchmark + exactness check for the int8-limb formulation of the resampler's taps ("what would move it").
//
// The block kernel's sums are exact integers: y = sum_k c[p][k] * x[k], c Q28 (|c| < 2^29), x 24-bit.  Written in balanced
// base-256 digits (every digit in [-128, 127]) x = d0 + d1 2^8 + d2 2^16, c = e0 + e1 2^8 + e2 2^16 + e3 2^24, the sum is
//      y = sum_{s=0..5} 2^(8s) * S_s,      S_s = sum over (i, j) with i + j = s of  sum_k d_i[k] e_j[k]
// and every S_s is an int8 x int8 dot product -- matrix-pipe work.  A tile = 16 consecutive outputs of 16 (block, channel)
// columns; its 16 windows lie inside one run of 64 input samples (15 * 147/160 + 32 = 46), so with the coefficients laid
// out as a banded 16 x 64 matrix per digit (zero outside each output's 32 taps; the same ten matrices for every block,
// because a block starts at phase 0) a tile is twelve v_mfma_i32_16x16x64_i8 into six accumulators (one per s; at most
// 3 * 64 terms of magnitude <= 2^14 each: no overflow), followed by the recombination in 64-bit integer arithmetic,
// rounding and the clamp.
//
// This program (1) checks that the digits, the MFMA operand layout and the recombination reproduce the exact sum for
// random data, and (2) measures tiles per second per wave with everything a real kernel would do per tile in place:
// seven 16-byte LDS reads (four coefficient digits, three sample digits), twelve MFMAs, the recombination of the lane's
// four outputs, rounding/clamp, and the digit split of the tile's new samples -- but no HBM traffic.
// Build: hipcc --offload-arch=gfx950 -O3 limb_mfma.hip -o limb_mfma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(int32_t x, int& d0, int& d1, int& d2)
{   // balanced digits of a 24-bit value
    d0 = (int8_t)(x & 0xff); x = (x - d0) >> 8;
    d1 = (int8_t)(x & 0xff); x = (x - d1) >> 8;
    d2 = x;                                              // in [-128, 128]; 128 only for x near +2^23 (handled by the range of S24: < 2^23)
}

// One tile: A digits a[0..3] (lane: row = lane % 16, its 16 consecutive k slots of group lane / 16), B digits b[0..2] (lane:
// column = lane % 16, same k slots).  Returns the lane's four outputs (rows 4 * (lane / 16) + v of column lane % 16), rounded
// to S24 and clamped, packed into the low 24 bits of four ints.
__device__ __forceinline__ v4i tile(const v4i (&a)[4], const v4i (&b)[3])
{
    v4i s[6];
#pragma unroll
    for (int k = 0; k < 6; k++) s[k] = v4i{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            s[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[j], b[i], s[i + j], 0, 0, 0);
    v4i y;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        int64_t acc = (int64_t)s[5][v];
#pragma unroll
        for (int k = 4; k >= 0; k--) acc = (acc << 8) + (int64_t)s[k][v];     // Horner over the digit sums
        int64_t r = (acc + ((int64_t)1 << 27)) >> 28;                          // Q28 -> sample, round half up (the kernel's rule)
        r = r > 8388607 ? 8388607 : (r < -8388608 ? -8388608 : r);
        y[v] = (int)r;
    }
    return y;
}

__global__ __launch_bounds__(256) void check_kernel(const int8_t* A, const int8_t* B, int* Y)
{   // A: [4 digits][16 rows][64 k], B: [3 digits][16 cols][64 k], Y: [16 rows][16 cols]
    const int lane = threadIdx.x & 63;
    v4i a[4], b[3];
    for (int j = 0; j < 4; j++) a[j] = *(const v4i*)(A + ((j * 16 + (lane & 15)) * 64 + (lane >> 4) * 16));
    for (int i = 0; i < 3; i++) b[i] = *(const v4i*)(B + ((i * 16 + (lane & 15)) * 64 + (lane >> 4) * 16));
    const v4i y = tile(a, b);
    for (int v = 0; v < 4; v++) Y[(4 * (lane >> 4) + v) * 16 + (lane & 15)] = y[v];
}

// Throughput: every wave runs `tiles` tiles; operands come from LDS (a table of coefficient digit tiles and a plane of sample
// digits per wave), the new samples of a tile (15 per column: 240 per tile, 3.75 per lane -> 4) are split into digits and
// written to the plane, results are summed so that nothing is dropped.
__global__ __launch_bounds__(256) void rate_kernel(int* out, int tiles, const int* seed)
{
    __shared__ __attribute__((aligned(16))) int8_t s_coef[10 * 4 * 16 * 64];        // ten tiles of a block, four digits: 40 KB
    __shared__ __attribute__((aligned(16))) int8_t s_x[4][3 * 16 * 64 + 64];       // per wave: three digit planes of 16 columns x 64 samples
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (int)sizeof(s_coef) / 4; i += 256) ((int*)s_coef)[i] = seed[i & 255] * (i + 1);
    for (int i = lane; i < (int)sizeof(s_x[0]) / 4; i += 64) ((int*)s_x[wave])[i] = seed[(i + 7) & 255];
    __syncthreads();
    int sum = 0;
    int x = seed[lane];
    for (int t = 0; t < tiles; t++) {
        const int tt = t % 10;
        v4i a[4], b[3];
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = *(const v4i*)(s_coef + (((tt * 4 + j) * 16 + (lane & 15)) * 64 + (lane >> 4) * 16));
#pragma unroll
        for (int i = 0; i < 3; i++) b[i] = *(const v4i*)(s_x[wave] + ((i * 16 + (lane & 15)) * 64 + (lane >> 4) * 16));
        const v4i y = tile(a, b);
        sum += y[0] ^ y[1] ^ y[2] ^ y[3];
        // the tile's new samples: four per lane, split into digits and stored to the planes
#ifndef NO_SPLIT
#pragma unroll
        for (int n = 0; n < 4; n++) {
            x = x * 1664525 + 1013904223;
            int d0, d1, d2;
            split3(x >> 8, d0, d1, d2);
            const int at = (lane & 15) * 64 + ((t * 15 + (lane >> 4) * 4 + n) & 63);
            s_x[wave][at] = (int8_t)d0; s_x[wave][16 * 64 + at] = (int8_t)d1; s_x[wave][2 * 16 * 64 + at] = (int8_t)d2;
        }
#endif
    }
    out[blockIdx.x * 256 + tid] = sum;
}

int main()
{
    // ---- exactness ----
    std::vector<int32_t> c(16 * 64), xs(16 * 64);
    srand(1);
    for (auto& v : c) v = (int32_t)((((int64_t)rand() << 16) ^ rand()) % (1 << 29)) * ((rand() & 1) ? 1 : -1);
    for (auto& v : xs) v = (rand() % ((1 << 24) - (1 << 16))) - ((1 << 23) - (1 << 15));   // (the top balanced digit of a sample within 2^15 of +2^23 would be 128: a real kernel uses offset digits, u - 128, and a per-phase constant)
    // banded: output row r uses k in [r, r + 32)
    for (int r = 0; r < 16; r++) for (int k = 0; k < 64; k++) if (k < r || k >= r + 32) c[r * 64 + k] = 0;
    // keep sum |c| < 2^29 per row, as the kernel's filters do (then |y| < 2^24 and the clamp is rarely hit)
    for (int r = 0; r < 16; r++) for (int k = 0; k < 64; k++) c[r * 64 + k] /= 32;
    std::vector<int8_t> A(4 * 16 * 64), B(3 * 16 * 64);
    for (int r = 0; r < 16; r++) for (int k = 0; k < 64; k++) {
        int64_t v = c[r * 64 + k];
        for (int j = 0; j < 4; j++) { int d = (int8_t)(v & 0xff); if (j == 3) d = (int)v; A[(j * 16 + r) * 64 + k] = (int8_t)d; v = (v - d) >> 8; }
        int64_t w = xs[r * 64 + k];
        for (int i = 0; i < 3; i++) { int d = (int8_t)(w & 0xff); if (i == 2) d = (int)w; B[(i * 16 + r) * 64 + k] = (int8_t)d; w = (w - d) >> 8; }
    }
    int8_t *dA, *dB; int* dY;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dY, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    check_kernel<<<1, 64>>>(dA, dB, dY);
    std::vector<int> Y(256);
    hipMemcpy(Y.data(), dY, 1024, hipMemcpyDeviceToHost);
    int bad = 0, digit_overflow = 0;
    for (size_t i = 0; i < A.size(); i++) if (A[i] == -128 && false) digit_overflow++;
    for (int r = 0; r < 16; r++) for (int n = 0; n < 16; n++) {
        int64_t acc = 0;
        for (int k = 0; k < 64; k++) acc += (int64_t)c[r * 64 + k] * xs[n * 64 + k];
        int64_t want = (acc + ((int64_t)1 << 27)) >> 28;
        want = want > 8388607 ? 8388607 : (want < -8388608 ? -8388608 : want);
        if (Y[r * 16 + n] != (int)want) { if (bad < 4) printf("  row %d col %d: got %d want %lld\n", r, n, Y[r * 16 + n], (long long)want); bad++; }
    }
    printf("exactness: %d of 256 outputs differ from the 64-bit dot product\n", bad);

    // ---- rate ----
    int *dout, *dseed;
    std::vector<int> seed(256);
    for (auto& v : seed) v = rand();
    const int blocks = 256 * 8, tiles = 2000;            // 8 workgroups of 4 waves per CU
    hipMalloc(&dout, blocks * 256 * 4); hipMalloc(&dseed, 1024);
    hipMemcpy(dseed, seed.data(), 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        rate_kernel<<<blocks, 256>>>(dout, tiles, dseed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double n_tiles = (double)blocks * 4 * tiles, outputs = n_tiles * 256;
    printf("rate: %.3f ms for %.0f tiles = %.1f G output subsamples/s (the fp64 kernel's launch: 245.8 M output subsamples in 0.475 ms = 517 G/s)\n",
           best, n_tiles, outputs / best / 1e6);
    printf("      = %.1f T limb-MACs/s on the matrix pipe (12 x 16x16x64 per tile), %.2f cycles per output subsample and CU at 2.4 GHz\n",
           n_tiles * 12 * 16384 / best / 1e9, best * 1e-3 * 2.4e9 * 256 / outputs);
    return 0;
}
