#!/usr/bin/env python3
"""Generates body_model.hip: an issue model of the block resampler's per-output body on a CDNA4 SIMD.

One iteration = one output of one wave: 32 v_fmac_f64_dpp plus, by variant, the non-tap work of the real loop (integer
vector instructions, scalar instructions, LDS traffic with counted waits, branches), either as a dependent tail AFTER the
taps (round 1's loop) or interleaved with the NEXT output's taps (software pipelined).  The program prints shader cycles
per iteration per SIMD at 1, 2, 3 (and 4) waves per SIMD and the clock the chip held.
    python3 gen_body_model.py > body_model.hip && hipcc --offload-arch=gfx950 -O3 body_model.hip -o body_model
"""

def fma(acc, cv, xv, k):
    return f"v_fmac_f64_dpp %[{acc}], %[{cv}], %[{xv}] row_newbcast:{k} row_mask:0xf bank_mask:0xf"

TAIL = [   # the previous output's tail + this output's sample unpack, 12 vector instructions (dependent chain)
    "v_cvt_u32_f64 %[y], %[sp]",
    "v_med3_u32 %[y], %[y], %[t0], %[t1]",
    "v_mov_b32_dpp %[got], %[yp] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5",
    "v_mov_b32_dpp %[got], %[y] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xa",
    "v_perm_b32 %[lo], %[got], %[yp], %[sel]",
    "v_perm_b32 %[hi], %[got], %[y], %[sel]",
    "v_add_u32 %[st], %[st], %[t0]",
    "v_mov_b32 %[yp], %[y]",
    "v_add_u32 %[ca], %[ca], %[t1]",
    "v_alignbyte_b32 %[u], %[lo], %[hi], %[t0]",
    "v_bfe_i32 %[u], %[u], 0, 24",
    "v_cvt_f64_i32 %[w0], %[u]",
]
SCALAR = [
    "s_add_u32 %[s0], %[s0], %[s1]", "s_xor_b32 %[s1], %[s1], %[s0]", "s_sub_u32 %[s2], %[s2], %[s1]", "s_cmp_eq_u32 %[s2], %[s3]",
    "s_cselect_b32 %[s2], %[s3], %[s2]", "s_add_u32 %[s3], %[s3], 6", "s_and_b32 %[s3], %[s3], 0xffff", "s_bitcmp1_b32 %[s0], 0",
    "s_add_u32 %[s0], %[s0], 1", "s_sub_u32 %[s1], %[s1], 1",
]


def body(variant):
    """variant: dict(tail='none'|'after'|'inter', scalar=bool, lds=bool, branch=bool, nops=bool, chains=2|4)"""
    ins = []
    chains = variant.get("chains", 2)
    accs = ["a0", "a1", "a2", "a3"][:chains]
    ins.append("v_mov_b64 %[a0], 0.5")
    for a in accs[1:]:
        ins.append(f"v_mov_b64 %[{a}], 0")
    taps = []
    for k in range(32):
        cv = "c1" if k < 16 else "c0"
        taps.append(fma(accs[k % chains], cv, f"w{k % 4}", 15 - (k % 16)))
    tail = list(TAIL) if variant["tail"] != "none" else []
    scal = list(SCALAR) if variant.get("scalar") else []
    if variant.get("lds"):
        ins.append("s_waitcnt lgkmcnt(2)")                 # c1 landed (issued last iteration, followed by c0's read and the sample read)
    if variant["tail"] == "inter":
        # spread tail + scalar instructions between the taps; the sample unpack (last 3 of TAIL) must precede the last tap
        fill = tail[:9]
        unp = tail[9:]
        slots = {}
        pos = 2
        for t in fill:
            slots.setdefault(pos, []).append(t)
            pos += 2
        pos = 3
        for t in scal:
            slots.setdefault(pos, []).append(t)
            pos += 3
        for k, t in enumerate(taps):
            if variant.get("lds") and k == 16:
                ins.append("ds_read_b64 %[c1], %[lc] offset:128")   # next output's high-tap coefficients
                ins.append("s_waitcnt lgkmcnt(2)")                   # c0 + raw sample landed
                ins.append("ds_write2_b32 %[lo_a], %[lo], %[hi] offset1:1")
            if k == 20:
                ins.extend(unp)
            ins.append(t)
            ins.extend(slots.get(k, []))
        if variant.get("lds"):
            ins.append("ds_read_b64 %[c0], %[lc]")
            ins.append("ds_read2_b32 %[raw], %[li] offset1:1")
    else:
        for k, t in enumerate(taps):
            if variant.get("lds") and k == 16:
                ins.append("ds_read_b64 %[c1], %[lc] offset:128")
                ins.append("s_waitcnt lgkmcnt(2)")
            ins.append(t)
        if variant.get("lds"):
            ins.append("ds_read_b64 %[c0], %[lc]")
            ins.append("ds_read2_b32 %[raw], %[li] offset1:1")
    if chains == 4:
        ins.append("v_add_f64 %[a0], %[a0], %[a2]")
        ins.append("v_add_f64 %[a1], %[a1], %[a3]")
    ins.append("v_add_f64 %[sp], %[a0], %[a1]")
    if variant["tail"] == "after":
        for t in tail:
            ins.append(t)
            if variant.get("nops"):
                ins.append("s_nop 0")
        ins.extend(scal)
        if variant.get("lds"):
            ins.append("ds_write2_b32 %[lo_a], %[lo], %[hi] offset1:1")
    if variant.get("branch"):
        ins.append("s_cmp_eq_u32 %[s3], 0x7fffffff")
        ins.append("s_cbranch_scc1 9f")
        ins.append("s_bitcmp1_b32 %[s3], 31")
        ins.append("s_cbranch_scc1 9f")
        ins.append("9:")
    return ins


VARIANTS = [
    ("taps only, 2 chains", dict(tail="none")),
    ("taps only, 4 chains", dict(tail="none", chains=4)),
    ("taps + 12 VALU tail after", dict(tail="after")),
    ("taps + 12 VALU interleaved", dict(tail="inter")),
    ("interleaved + 10 scalar", dict(tail="inter", scalar=True)),
    ("interleaved + scalar + LDS", dict(tail="inter", scalar=True, lds=True)),
    ("interleaved + scalar + LDS + 2 branches", dict(tail="inter", scalar=True, lds=True, branch=True)),
    ("after + scalar + LDS + branches + nops (round 1 shape)", dict(tail="after", scalar=True, lds=True, branch=True, nops=True)),
    ("after + scalar + LDS + branches, no nops", dict(tail="after", scalar=True, lds=True, branch=True)),
]

HEAD = r"""// GENERATED by gen_body_model.py -- do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
"""

KERNEL = r"""
__global__ __launch_bounds__(1024) void k_body_%(idx)d(unsigned long long* out, const double* x, int iters, unsigned* sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, sp = 0.5;
    double c0 = x[lane & 15], c1 = x[16 + (lane & 15)];
    double w0 = x[lane], w1 = x[(lane + 1) & 63], w2 = x[(lane + 2) & 63], w3 = x[(lane + 3) & 63];
    unsigned y = lane, yp = lane * 3, got = 0, lo = 0, hi = 0, sel = 0x07060100, st = 0, t0 = 1, t1 = 0, u = 0, ca = 0;
    unsigned long long raw = lane;
    unsigned s0 = iters, s1 = 1, s2 = 3, s3 = 5;
    const unsigned lds_c = (unsigned)(size_t)(smem) + (lane & 15) * 8;
    const unsigned lds_in = (unsigned)(size_t)(smem) + 4096 + wave * 1024 + lane * 8;
    const unsigned lds_out = (unsigned)(size_t)(smem) + 4096 + 16 * 1024 + wave * 1024 + lane * 8;
    for (unsigned i = threadIdx.x; i < 4096 / 8; i += blockDim.x) ((double*)smem)[i] = 1.0 + i;
    for (unsigned i = threadIdx.x; i < 32 * 1024 / 4; i += blockDim.x) ((unsigned*)(smem + 4096))[i] = i;
    __syncthreads();
    asm volatile("ds_read_b64 %%0, %%3 offset:128\n\tds_read_b64 %%1, %%3\n\tds_read2_b32 %%2, %%4 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(c1), "=v"(c0), "+v"(raw) : "v"(lds_c), "v"(lds_in) : "memory");
    asm volatile("ds_read_b64 %%0, %%3 offset:128\n\tds_read_b64 %%1, %%3\n\tds_read2_b32 %%2, %%4 offset1:1"
                 : "=v"(c1), "=v"(c0), "+v"(raw) : "v"(lds_c), "v"(lds_in) : "memory");
    const unsigned long long tm0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        asm volatile(
%(asm)s
            : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [sp] "+v"(sp), [y] "+v"(y), [yp] "+v"(yp), [got] "+v"(got),
              [lo] "+v"(lo), [hi] "+v"(hi), [st] "+v"(st), [c0] "+v"(c0), [c1] "+v"(c1), [raw] "+v"(raw), [w0] "+v"(w0), [u] "+v"(u), [ca] "+v"(ca),
              [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3)
            : [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [t0] "v"(t0), [t1] "v"(t1), [sel] "v"(sel),
              [lc] "v"(lds_c), [li] "v"(lds_in), [lo_a] "v"(lds_out)
            : "memory", "scc", "vcc");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long tm1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = tm1 - tm0;
        out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = tr1 - tr0;
    }
    if (a0 + a1 + a2 + a3 + sp == 12345.678 || (y ^ yp ^ got ^ lo ^ hi ^ st ^ s0 ^ s1 ^ s2 ^ s3 ^ u ^ ca ^ (unsigned)raw) == 0x12345u) sink[0] = 1;
}
"""

MAIN = r"""
typedef void (*kern_t)(unsigned long long*, const double*, int, unsigned*);
int main()
{
    const int iters = 20000, blocks = 256;
    std::vector<double> x(64);
    for (int i = 0; i < 64; i++) x[i] = 1.0 + i * 0.001;
    double* dx; unsigned long long* dout; unsigned* dsink;
    hipMalloc(&dx, 64 * 8); hipMalloc(&dout, 256 * 16 * 2 * 8); hipMalloc(&dsink, 4);
    hipMemcpy(dx, x.data(), 64 * 8, hipMemcpyHostToDevice);
    kern_t kerns[] = {%(kerns)s};
    const char* names[] = {%(names)s};
    const int nvalu[] = {%(nvalu)s};
    for (size_t v = 0; v < sizeof(kerns) / sizeof(kerns[0]); v++) {
        hipFuncSetAttribute((const void*)kerns[v], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        for (int w = 1; w <= 4; w++) {
            const int threads = w * 4 * 64;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kerns[v], dim3(blocks), dim3(threads), 100 * 1024, 0, dout, dx, iters, dsink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            std::vector<unsigned long long> h(blocks * w * 4 * 2);
            hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i]); clk.push_back((double)h[i] / ((double)h[i + 1] * 10.0)); }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double med = cyc[cyc.size() / 2];
            printf("%%-56s w/SIMD %%d: %%7.1f cyc/iter/wave = %%6.1f cyc/iter/SIMD (%%d vector instr) | %%.3f ms, clock %%.2f GHz\n", names[v], w, med / iters,
                   med / iters / w, nvalu[v], best, clk[clk.size() / 2]);
        }
    }
    return 0;
}
"""


def main():
    out = [HEAD]
    kerns, names, nvalu = [], [], []
    for idx, (name, var) in enumerate(VARIANTS):
        ins = body(var)
        asm = "\n".join('            "%s\\n\\t"' % i for i in ins)
        out.append(KERNEL % dict(idx=idx, asm=asm))
        kerns.append(f"k_body_{idx}")
        names.append('"%s"' % name)
        nvalu.append(str(sum(1 for i in ins if i.startswith("v_"))))
    out.append(MAIN % dict(kerns=", ".join(kerns), names=", ".join(names), nvalu=", ".join(nvalu)))
    print("".join(out))


if __name__ == "__main__":
    main()
