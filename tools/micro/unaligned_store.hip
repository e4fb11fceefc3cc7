// Microbenchmark: contiguous 12- and 16-byte-per-lane global stores whose base is 0..3 bytes off dword alignment
// (what a subsample-aligned, not dword-aligned, write-back would do).  Reports GB/s for each misalignment.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int BYTES>
__global__ __launch_bounds__(256) void k(unsigned char* dst, size_t per_block, int mis)
{
    unsigned char* p = dst + blockIdx.x * per_block + mis;
    for (size_t off = threadIdx.x * (size_t)BYTES; off + BYTES <= per_block - 16; off += 256 * (size_t)BYTES) {
        if constexpr (BYTES == 12) {
            u3 v = {(unsigned)off, 1u, 2u};
            asm volatile("global_store_dwordx3 %0, %1, off" : : "v"(p + off), "v"(v) : "memory");
        } else {
            u4 v = {(unsigned)off, 1u, 2u, 3u};
            asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p + off), "v"(v) : "memory");
        }
    }
}

int main()
{
    const size_t per_block = 1 << 20, blocks = 2048;
    unsigned char* d; hipMalloc(&d, per_block * blocks + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bytes : {12, 16}) for (int mis = 0; mis < 4; mis++) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (bytes == 12) k<12><<<blocks, 256>>>(d, per_block, mis); else k<16><<<blocks, 256>>>(d, per_block, mis);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%2d bytes per lane, base off by %d: %.3f ms, %.0f GB/s\n", bytes, mis, best, per_block * blocks / best / 1e6);
    }
    return 0;
}
