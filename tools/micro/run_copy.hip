// Microbenchmark: what the chip gives a kernel that moves the headline workload's bytes -- 0.677 GB in, 0.737 GB out -- the way
// src_mfma_wg_kernel does: a 256-thread workgroup per pass, a pass = ONE contiguous run of 899 16-byte pieces in (16 blocks of 147
// stereo S24 frames plus the filter's history) and ONE contiguous run of 960 pieces out (16 blocks of 160 frames), nothing in
// between.  Swept: how many passes of loads a workgroup keeps in flight (DEPTH), whether the loads go to registers or straight
// into LDS (global_load_lds_dwordx4), non-temporal or plain loads and stores, workgroups per CU, passes dealt round robin or a
// contiguous stretch per workgroup.  Beside it the yardstick of MI355X_MICROARCH.md: a float4 grid-stride copy of the same bytes.
//
//   hipcc --offload-arch=gfx950 -O3 -o run_copy run_copy.hip && ./run_copy
//
// Every load of a pass is issued from ONE asm statement together with the wait for the pass that is DEPTH older and that pass's
// stores: vmcnt counts loads and stores in issue order, so "all but the 8 (DEPTH - 1) youngest" is exactly "pass p has landed".
//
// Measured (round 5, one MI355X, 40 launches per line behind 0.3 s of launches; gpurun_out/r5/run_copy_1.log, figures in DESIGN.md 5.0):
// the float4 copy 0.2516 ms = 5.62 TB/s at its best grid (4 workgroups per CU; 4.2-5.3 at the others); the run-shaped copy at three
// workgroups per CU -- the resampler's occupancy -- 0.262-0.268 ms with one pass of loads in flight, 0.255 with two, 0.251 with three;
// at two workgroups per CU 0.249-0.253 with one; a contiguous stretch per workgroup instead of round robin 0.28-0.30 everywhere; loads
// through LDS-DMA 0.264-0.269 at depth 2-3 (no faster than registers); plain stores instead of non-temporal ones +0.03 ms.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
static constexpr uint32_t kInPieces = 899, kOutPieces = 960;           // per pass
static constexpr uint32_t kInStride = 882 * 16, kOutStride = 960 * 16; // bytes a pass advances in each arena
static constexpr uint32_t kLanes = 240;                                // 240 x 4 = 960 stores, 225 x 4 = 900 loads per pass

__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

#define LD(nt) "global_load_dwordx4 %0, %4, %8" nt "\n\tglobal_load_dwordx4 %1, %5, %8" nt "\n\tglobal_load_dwordx4 %2, %6, %8" nt "\n\tglobal_load_dwordx4 %3, %7, %8" nt "\n\t"
#define ST(nt) "global_store_dwordx4 %4, %0, %8" nt "\n\tglobal_store_dwordx4 %5, %1, %8" nt "\n\tglobal_store_dwordx4 %6, %2, %8" nt "\n\tglobal_store_dwordx4 %7, %3, %8" nt "\n\t"

template <bool NTL>
__device__ __forceinline__ void issue_loads(u32x4 (&r)[4], const uint32_t (&off)[4], uint64_t base)
{
    if (NTL) asm volatile(LD(" nt") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base) : "memory");
    else asm volatile(LD("") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base) : "memory");
}
template <bool NTS, int WAIT>
__device__ __forceinline__ void wait_and_store(u32x4 (&r)[4], const uint32_t (&off)[4], uint64_t base)
{
    if (NTS) asm volatile("s_waitcnt vmcnt(%9)\n\t" ST(" nt") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base), "n"(WAIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%9)\n\t" ST("") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base), "n"(WAIT) : "memory");
}

// pass index of a workgroup's i-th pass: dealt round robin (mode 0) or a contiguous stretch per workgroup (mode 1)
__device__ __forceinline__ uint32_t pass_of(uint32_t i, uint32_t mode, uint32_t per_wg)
{
    return mode ? blockIdx.x * per_wg + i : i * gridDim.x + blockIdx.x;
}

// ---- loads to registers
template <int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void run_copy_regs(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_passes, uint32_t mode)
{
    const uint32_t t = threadIdx.x;
    if (t >= kLanes) return;
    const uint32_t per_wg = (n_passes + gridDim.x - 1) / gridDim.x;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < per_wg; i++) mine += pass_of(i, mode, per_wg) < n_passes ? 1u : 0u;
    if (mine == 0) return;                                       // (a workgroup beyond the last stretch: nothing to fetch, nothing to read ahead of)
    uint32_t ld_off[4], st_off[4];
    for (int k = 0; k < 4; k++) {
        ld_off[k] = t < 225 ? 16u * (t + 225u * k) : 0u;
        st_off[k] = 16u * (t + 240u * k);
    }
    u32x4 r[DEPTH][4];
    for (int d = 0; d < DEPTH; d++) for (int k = 0; k < 4; k++) r[d][k] = (u32x4){0u, 0u, 0u, 0u};
    const uint64_t in0 = (uint64_t)(uintptr_t)in, out0 = (uint64_t)(uintptr_t)out;
#pragma unroll
    for (int d = 0; d < DEPTH; d++)                              // prologue: DEPTH passes of loads in flight (a pass beyond the end re-reads the last)
        issue_loads<NTL>(r[d], ld_off, uniform64(in0 + (uint64_t)pass_of(min((uint32_t)d, mine - 1), mode, per_wg) * kInStride));
    for (uint32_t i = 0; i < mine; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t cur = i + d;
            if (cur < mine) {
                // in flight behind pass `cur`'s loads: (DEPTH - 1) x (4 stores + 4 loads) -- but for the workgroup's first DEPTH - 1 passes,
                // behind which the prologue's loads lie without stores between them: 4 (DEPTH - 1) + 4 cur, waited for as 4 (DEPTH - 1)
                if (cur + 1 < (uint32_t)DEPTH) wait_and_store<NTS, 4 * (DEPTH - 1)>(r[d], st_off, uniform64(out0 + (uint64_t)pass_of(cur, mode, per_wg) * kOutStride));
                else wait_and_store<NTS, 8 * (DEPTH - 1)>(r[d], st_off, uniform64(out0 + (uint64_t)pass_of(cur, mode, per_wg) * kOutStride));
                issue_loads<NTL>(r[d], ld_off, uniform64(in0 + (uint64_t)pass_of(min(cur + DEPTH, mine - 1), mode, per_wg) * kInStride));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- loads straight into LDS (lane t's pieces are t + 240 k of the pass's image, an image is piece-linear), read back by the lane
// that fetched them and stored from registers
template <int DEPTH, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void run_copy_lds(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_passes, uint32_t mode)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t t = threadIdx.x;
    if (t >= kLanes) return;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t per_wg = (n_passes + gridDim.x - 1) / gridDim.x;
    uint32_t mine = 0;
    for (uint32_t i = 0; i < per_wg; i++) mine += pass_of(i, mode, per_wg) < n_passes ? 1u : 0u;
    if (mine == 0) return;                                       // (a workgroup beyond the last stretch: nothing to fetch, nothing to read ahead of)
    uint32_t ld_off[4], st_off[4];
    for (int k = 0; k < 4; k++) {
        ld_off[k] = 16u * min(t + 240u * k, kInPieces);           // (image piece t + 240 k; pieces past the run re-read one)
        st_off[k] = 16u * (t + 240u * k);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    const uint64_t in0 = (uint64_t)(uintptr_t)in, out0 = (uint64_t)(uintptr_t)out;
    auto dma = [&](int d, uint64_t base) {
        const uint32_t buf = lds0 + (uint32_t)d * 16384u + wave * 1024u;
#define DMA1(k, nt) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" nt : : "v"(ld_off[k]), "s"(base), "s"(buf + 3840u * k) : "memory", "m0")
        if (NTL) { DMA1(0, " nt"); DMA1(1, " nt"); DMA1(2, " nt"); DMA1(3, " nt"); }
        else { DMA1(0, ""); DMA1(1, ""); DMA1(2, ""); DMA1(3, ""); }
#undef DMA1
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) dma(d, uniform64(in0 + (uint64_t)pass_of(min((uint32_t)d, mine - 1), mode, per_wg) * kInStride));
    for (uint32_t i = 0; i < mine; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const uint32_t cur = i + d;
            if (cur < mine) {
                const uint32_t a = lds0 + (uint32_t)d * 16384u + 16u * t;
                u32x4 r[4];
                if (cur + 1 < (uint32_t)DEPTH) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * (DEPTH - 1)) : "memory");      // (the first passes: see run_copy_regs)
                asm volatile("s_waitcnt vmcnt(%5)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:3840\n\tds_read_b128 %2, %4 offset:7680\n\tds_read_b128 %3, %4 offset:11520\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a), "n"(8 * (DEPTH - 1)) : "memory");
                const uint64_t ob = uniform64(out0 + (uint64_t)pass_of(cur, mode, per_wg) * kOutStride);
                if (NTS) asm volatile(ST(" nt") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(st_off[0]), "v"(st_off[1]), "v"(st_off[2]), "v"(st_off[3]), "s"(ob) : "memory");
                else asm volatile(ST("") : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(st_off[0]), "v"(st_off[1]), "v"(st_off[2]), "v"(st_off[3]), "s"(ob) : "memory");
                dma(d, uniform64(in0 + (uint64_t)pass_of(min(cur + DEPTH, mine - 1), mode, per_wg) * kInStride));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the yardstick: grid-stride float4 copy
template <bool NT>
__global__ __launch_bounds__(256) void plain_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(&in[i]), &out[i]);
        else out[i] = in[i];
    }
}
// ... and with four independent loads in flight per thread per trip
template <bool NT>
__global__ __launch_bounds__(256) void plain_copy4(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        u32x4 a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(&in[i]); b = __builtin_nontemporal_load(&in[i + stride]); c = __builtin_nontemporal_load(&in[i + 2 * stride]); d = __builtin_nontemporal_load(&in[i + 3 * stride]); }
        else { a = in[i]; b = in[i + stride]; c = in[i + 2 * stride]; d = in[i + 3 * stride]; }
        if (NT) { __builtin_nontemporal_store(a, &out[i]); __builtin_nontemporal_store(b, &out[i + stride]); __builtin_nontemporal_store(c, &out[i + 2 * stride]); __builtin_nontemporal_store(d, &out[i + 3 * stride]); }
        else { out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d; }
    }
    for (; i < n; i += stride) out[i] = in[i];
}

struct Timer {
    hipEvent_t e0, e1;
    Timer() { CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); }
    template <typename F>
    double median_ms(F&& launch, int reps)
    {
        // about 0.3 s of back-to-back launches first: the clock the chip holds under this load, not a cold chip's
        CHECK(hipEventRecord(e0));
        float ms = 0.f;
        int warm = 0;
        do { for (int k = 0; k < 16; k++) launch(); warm += 16; CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1)); } while (ms < 300.f && warm < 4000);
        std::vector<float> v;
        for (int r = 0; r < reps; r++) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            v.push_back(ms);
        }
        CHECK(hipGetLastError());
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
    }
};

int main(int argc, char** argv)
{
    const uint32_t streams = 256, blocks_per_stream = 3000, n_passes = streams * blocks_per_stream / 16;     // 48 000 passes
    const size_t copy_pieces = (size_t)((double)streams * (441000.0 + 480000.0) * 6.0 / 2) / 16;            // the float4 copy moves this many pieces each way
    // every kernel below stays inside [0, arena): pass p reads [p * kInStride, + 900 pieces) and writes [p * kOutStride, + 960 pieces), p < n_passes;
    // the float4 copy reads and writes [0, copy_pieces * 16)
    const size_t arena = std::max({(size_t)(n_passes - 1) * kInStride + 900 * 16, (size_t)(n_passes - 1) * kOutStride + 960 * 16, copy_pieces * 16}) + 65536;
    const size_t in_bytes = arena, out_bytes = arena;
    const double algo = (double)streams * 441000.0 * 6.0 + (double)streams * 480000.0 * 6.0;               // 1.414656 GB: what the resampler's roofline counts
    uint8_t *in = nullptr, *out = nullptr;
    CHECK(hipMalloc((void**)&in, in_bytes));
    CHECK(hipMalloc((void**)&out, out_bytes));
    {   // random-ish contents (the DRAM does not care; the check below does)
        std::vector<uint32_t> h(in_bytes / 4);
        uint32_t x = 12345;
        for (auto& w : h) { x = x * 1664525u + 1013904223u; w = x; }
        CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CHECK(hipMemset(out, 0, out_bytes));
    Timer tm;
    const int reps = argc > 1 ? atoi(argv[1]) : 40;
    printf("# run_copy: %u passes, %.4f GB in + %.4f GB out per launch; TB/s quoted on the resampler's algorithmic %.4f GB\n", n_passes,
           (double)n_passes * kInPieces * 16 / 1e9, (double)n_passes * kOutPieces * 16 / 1e9, algo / 1e9);
    // ---- yardstick
    for (int nt = 0; nt < 2; nt++)
        for (int four = 0; four < 2; four++)
            for (uint32_t per_cu : {4u, 8u, 16u, 32u}) {
                const size_t n = copy_pieces;
                const uint32_t grid = 256 * per_cu;
                const double ms = tm.median_ms([&] {
                    if (four) { if (nt) plain_copy4<true><<<grid, 256>>>((const u32x4*)in, (u32x4*)out, n); else plain_copy4<false><<<grid, 256>>>((const u32x4*)in, (u32x4*)out, n); }
                    else { if (nt) plain_copy<true><<<grid, 256>>>((const u32x4*)in, (u32x4*)out, n); else plain_copy<false><<<grid, 256>>>((const u32x4*)in, (u32x4*)out, n); }
                }, reps);
                printf("float4 copy%s%s  %2u wg/CU                     %.4f ms  %.3f TB/s\n", four ? " x4" : "   ", nt ? " nt" : "   ", per_cu, ms, 2.0 * n * 16 / ms / 1e9);
                fflush(stdout);
            }
    // ---- the workgroup kernel's launch shape
    auto sweep = [&](const char* name, auto kernel, int depth, bool ntl, bool nts, size_t lds) {
        int occ = 0;
        CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, lds));
        for (uint32_t mode = 0; mode < 2; mode++)
            for (int per_cu : {2, 3, 4, 6, 8}) {
                if (per_cu > occ) continue;
                const uint32_t grid = 256u * per_cu;
                const double ms = tm.median_ms([&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, in, out, n_passes, mode); }, reps);
                printf("%-5s depth %d  loads %s stores %s  %d wg/CU (max %d) %s  %.4f ms  %.3f TB/s  (%.0f KB of loads in flight per CU)\n", name, depth, ntl ? "nt" : "  ",
                       nts ? "nt" : "  ", per_cu, occ, mode ? "stretch" : "rr     ", ms, algo / ms / 1e9, per_cu * depth * kInPieces * 16 / 1024.0);
                fflush(stdout);
            }
    };
#define SWEEP_R(D, L, S) sweep("regs", run_copy_regs<D, L, S>, D, L, S, 0)
#define SWEEP_L(D, L, S) sweep("lds", run_copy_lds<D, L, S>, D, L, S, (size_t)D * 16384)
    SWEEP_R(1, true, true); SWEEP_R(2, true, true); SWEEP_R(3, true, true); SWEEP_R(4, true, true);
    SWEEP_R(1, false, true); SWEEP_R(2, false, true); SWEEP_R(3, false, true);
    SWEEP_R(2, true, false); SWEEP_R(2, false, false);
    SWEEP_L(1, true, true); SWEEP_L(2, true, true); SWEEP_L(3, true, true); SWEEP_L(4, true, true);
    SWEEP_L(2, false, true); SWEEP_L(3, false, true);
    // ---- the copies are copies: the LDS path's output against the input, every pass (piece q of a pass's output = piece min(q, 899) of its input)
    {
        std::vector<uint8_t> hi(arena), ho(arena);
        CHECK(hipMemcpy(hi.data(), in, arena, hipMemcpyDeviceToHost));
        for (int depth = 1; depth <= 2; depth++) {
            CHECK(hipMemset(out, 0xEE, out_bytes));
            if (depth == 1) hipLaunchKernelGGL((run_copy_lds<1, true, true>), dim3(768), dim3(256), 16384, 0, in, out, n_passes, 0u);
            else hipLaunchKernelGGL((run_copy_lds<2, true, true>), dim3(768), dim3(256), 32768, 0, in, out, n_passes, 0u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(ho.data(), out, arena, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (uint32_t p = 0; p < n_passes; p++)
                for (uint32_t q = 0; q < kOutPieces; q++) {
                    const uint32_t src_piece = q < kInPieces ? q : kInPieces;
                    if (memcmp(&ho[(size_t)p * kOutStride + 16u * q], &hi[(size_t)p * kInStride + 16u * src_piece], 16) != 0) {
                        if (bad < 8) printf("#   lds depth %d: pass %u piece %u differs (first dword %08x, expected %08x)\n", depth, p, q,
                                            *(const uint32_t*)&ho[(size_t)p * kOutStride + 16u * q], *(const uint32_t*)&hi[(size_t)p * kInStride + 16u * src_piece]);
                        bad++;
                    }
                }
            printf("# check (lds path, depth %d, every pass): %zu of %zu pieces differ\n", depth, bad, (size_t)n_passes * kOutPieces);
        }
    }
    return 0;
}
