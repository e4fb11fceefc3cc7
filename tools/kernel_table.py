#!/usr/bin/env python3
"""Which resampled layout takes which kernel -- written out from the sources: the lean kernel's instantiation lists
(csrc/src_block_common.h) and the workgroup matrix kernel's admission rule (csrc/src_mfma_wg_kernel.hip: src_mfma_wg_supported, and
the planner's gates in csrc/src_plan.cpp).  For every lean instantiation it says whether the matrix kernel serves the layout too
(then the lean one runs only under ohgpu_set_kernel_variant(4) or for a filter the matrix kernel's tiling does not hold) or whether it
is the layout's only block kernel.  Prints the markdown INTEGRATION.md 2b carries; tests/test_kernel_table.py keeps the two in step.
Usage: python tools/kernel_table.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISTS = [("OHGPU_BLOCK_KERNELS_1", "main"), ("OHGPU_BLOCK_KERNELS_2", "main"), ("OHGPU_BLOCK_KERNELS_3", "main"),
         ("OHGPU_LEAN_PLANAR_KERNELS", "planar"), ("OHGPU_LEAN_HB_KERNELS", "half-band"), ("OHGPU_LEAN_ONLY_KERNELS", "lean-only"),
         ("OHGPU_LEAN_MORE_KERNELS", "lean-only")]


def lean_instantiations():
    text = open(os.path.join(ROOT, "ohpipeline_amd", "csrc", "src_block_common.h")).read()
    out = []
    for macro, kind in LISTS:
        defs = [m.end() for m in re.finditer(r"#define %s\(X\)" % macro, text)]
        body = text[defs[-1]:]                                   # (the last definition: the one behind the diagnostic #else)
        body = body[:body.index("\n#")]
        for m in re.finditer(r"X\((\d+), (\d+), (\d+), (true|false), (\d+), (true|false)\)", body):
            t, ch, sb, sle, db, dle = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true", int(m.group(5)), m.group(6) == "true"
            out.append((t, ch, sb, sle, db, dle, kind))
    return out


def wg_takes(t, ch, sb, db, planar, halfband):
    """src_mfma_wg_supported + the planner's gates, for the filters ohgpu_src_design makes (44.1 -> 48 kHz: 32 taps, blocks of 160
    from 147; 96 -> 48 kHz: the 64-tap half-band decimator)."""
    if db != 3:
        return False
    if planar:
        return ch == 2 and t == 32 and not halfband
    if halfband:
        return t == 64 and ch in (2, 6, 8) and sb == 3
    return t == 32 and ((ch in (2, 6, 8) and sb == 3) or (ch == 2 and sb == 2))


def fmt(sb, le, planar=False):
    if planar:
        return "`TInt32` planes"
    return f"S{8 * sb}{'LE' if le else 'BE'}" if sb > 1 else "S8"


def main():
    rows = []
    for (t, ch, sb, sle, db, dle, kind) in sorted(lean_instantiations(), key=lambda r: (r[0], r[6] != "main", r[1], r[2], not r[3], r[4], r[5])):
        planar = sb == 0
        hb = kind == "half-band"
        also = wg_takes(t, ch, 3 if planar else sb, db, planar, hb)
        flt = "the 2:1 half-band decimator (64 stored taps)" if hb else (f"{t} taps per phase" + (" (any 64-tap table that is not half-band)" if t == 64 else ""))
        need = ("variant 4 only: `src_mfma_wg_kernel` has the layout" if also else "**the layout's only block kernel**")
        if t == 32 and also:
            need = "variant 4, and 32-tap filters whose ratio the matrix kernel's 16-output tiling does not hold (48 -> 44.1 kHz, 32 -> 48 kHz, 88.2 -> 48 kHz ...)"
        rows.append(f"| `src_lean_kernel<{t}, {ch}, {sb}, {'LE' if sle else 'BE'}, {db}, {'LE' if dle else 'BE'}{', HB' if hb else ''}>` | {fmt(sb, sle, planar)} | {ch} | {fmt(db, dle)} | {flt} | {need} |")
    print("| kernel | source | channels | destination | filter | needed for |")
    print("|---|---|---|---|---|---|")
    print("| `src_mfma_wg_kernel<16, 0, PAIRS, HB, ..>` | S24 LE or BE | 2, 6, 8 | S24 LE or BE | 32 taps per phase whose ratio fits the 16-output tiling (blocks of 160 outputs from 145-160 inputs: 44.1 -> 48 kHz), and the 2:1 half-band decimator | the default (variant 0) |")
    print("| `src_mfma_wg_kernel<16, 4, 1, false, ..>` | S16 LE or BE | 2 | S24 LE or BE | 32 taps per phase (as above) | the default |")
    print("| `src_mfma_wg_kernel<16, 1..3, 1, false, ..>` | `TInt32` planes (24-, 16-, 8-bit samples) | 2 | S24 LE or BE | 32 taps per phase (as above) | the default |")
    for r in rows:
        print(r)
    text = open(os.path.join(ROOT, "ohpipeline_amd", "csrc", "src_block_common.h")).read()
    body = text[text.index("#define OHGPU_BLOCK_FALLBACK_KERNELS(X)"):]
    body = body[:body.index("\n#", 1) if "\n#" in body[1:] else len(body)]
    for m in re.finditer(r"X\((\d+), (\d+), (\d+), (true|false), (\d+), (true|false)\)", body.split("\n\n")[0]):
        t, ch, sb, sle, db, dle = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true", int(m.group(5)), m.group(6) == "true"
        print(f"| `src_block_kernel<{t}, {ch}, {sb}, {'LE' if sle else 'BE'}, {db}, {'LE' if dle else 'BE'}>` (round 1's) | {fmt(sb, sle)} | {ch} | {fmt(db, dle)} | {t} taps per phase with a phase whose sum of magnitudes reaches 2^29 (`ohgpu_src_design`'s 48 -> 44.1 kHz and 32 -> 48 kHz filters: 2.02 and 2.40 x 2^28) | the fallback: beyond the lean kernel's exact rounding, under any variant |")
    print("| `src_kernel_v1` (generic) | any | 1-8 | any | any | everything else, block-unaligned stream ends, and any batch created under variant 1 |")


if __name__ == "__main__":
    main()
