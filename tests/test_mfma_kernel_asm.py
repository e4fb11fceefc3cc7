"""Static checks on the matrix-pipe resampler's gfx950 assembly (csrc/src_mfma_wg_kernel.hip; no GPU needed, hipcc cross-compiles).

What DESIGN.md 5.0 says about the workgroup kernel and the launch code relies on, read off the generated code of every
instantiation: three workgroups of four waves per CU need 168 registers or fewer and no scratch in the tiles; every address
stays in its address space (a `flat_` access is a base pointer that went through a register class it should not have); the
output leaves as whole 16-byte pieces, non-temporal; a tile is twelve matrix instructions.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ohpipeline_amd", "csrc")
OUTDIR = os.path.join(ROOT, "ohpipeline_amd", "build")
NAME = re.compile(r"_ZN5ohgpu18src_mfma_wg_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)EE\w+")     # <pair-rows, planar, channel pairs, half-band, src LE, dst LE>


@pytest.fixture(scope="module")
def wg():
    """{mangled name: (rows, planar + 10 * (channel pairs - 1), body lines, metadata text)} for every instantiation the library carries."""
    os.makedirs(OUTDIR, exist_ok=True)
    src = os.path.join(CSRC, "src_mfma_wg_kernel.hip")
    out = os.path.join(OUTDIR, "src_mfma_wg_kernel.test.s")
    deps = [src] + [os.path.join(CSRC, f) for f in ("src_mfma_common.h", "ohgpu_internal.h", "pcm_device.h", "src_block_common.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm",
                        "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out],
                       check=True, capture_output=True, timeout=900)
    text = open(out).read()
    found, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^(" + NAME.pattern + r"):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                found[name] = body
                name = None
    meta = {}
    for entry in re.split(r"\n  - (?=\.)", text[text.index("amdhsa.kernels:"):]):    # one list item per kernel
        m = re.search(r"\.name:\s+(" + NAME.pattern + r")\n", entry)
        if m:
            meta[m.group(1)] = entry
    assert len(found) >= 30 and set(found) == set(meta), (len(found), len(meta))
    return {n: (int(NAME.match(n).group(1)), int(NAME.match(n).group(2)) + 10 * (int(NAME.match(n).group(3)) - 1), found[n], meta[n]) for n in found}


def _field(meta, key):
    return int(re.search(r"\." + key + r":\s+(\d+)", meta).group(1))


def test_three_workgroups_per_cu_fit(wg):
    for name, (rows, planar, body, meta) in wg.items():
        if rows != 16:
            continue
        assert _field(meta, "vgpr_count") <= 168, name                      # 512 / 168 = three waves per SIMD
        assert _field(meta, "max_flat_workgroup_size") == 256, name         # four waves: one per SIMD


def test_no_scratch_but_where_design_says(wg):
    """Packed sources: no spills at all.  The planar instantiations (168 registers, the limit) spill four dwords around the split and
    the edge units' out-of-line loads -- outside the tiles: tolerated, bounded here."""
    for name, (rows, planar, body, meta) in wg.items():
        if rows != 16:
            continue
        if planar in (1, 2, 3):
            assert _field(meta, "private_segment_fixed_size") <= 16, name
            tiles = [i for i, l in enumerate(body) if "v_mfma_i32_16x16x64_i8" in l]
            between = [l for l in body[tiles[0]:tiles[-1]] if re.match(r"^\s*scratch_", l)]
            assert not between, (name, between[:4])
        else:
            assert _field(meta, "private_segment_fixed_size") == 0, name
            assert not any(re.match(r"^\s*scratch_", l) for l in body), name


def test_every_access_keeps_its_address_space(wg):
    for name, (rows, planar, body, meta) in wg.items():
        assert not [l for l in body if re.match(r"^\s*flat_", l)], name


def test_output_leaves_as_whole_pieces_non_temporal(wg):
    for name, (rows, planar, body, meta) in wg.items():
        stores = [l for l in body if re.match(r"^\s*global_store", l)]
        assert stores, name
        assert all(re.match(r"^\s*global_store_dwordx4 .* nt\b", l) for l in stores), (name, stores[:3])


def test_a_tile_is_twelve_matrix_instructions(wg):
    for name, (rows, planar, body, meta) in wg.items():
        n = sum("v_mfma_i32_16x16x64_i8" in l for l in body)
        assert n > 0 and n % 12 == 0, (name, n)
        assert not [l for l in body if "v_mfma" in l and "v_mfma_i32_16x16x64_i8" not in l], name


def test_no_floating_point_in_the_taps(wg):
    """The sums are integers end to end: no fp64 and no conversions anywhere in the kernel."""
    for name, (rows, planar, body, meta) in wg.items():
        assert not [l for l in body if re.match(r"^\s*v_(fma|fmac|mul|add|cvt)_f(64|32)", l)], name


def test_no_operand_read_lands_in_a_register_the_code_has_moved_on_from(wg):
    """A tile's sample operands are requested under the previous tile's matrix instructions and waited for a whole epilogue later
    (issue_planes / take_planes): in between their registers belong to the reads in flight.  The same walk of the LDS queue the
    block kernels are held to (test_block_kernel_asm.py: the round-3 diagnostic fault was exactly such a register, recycled) --
    no instruction names a read's destination before a wait that covers it, in any instantiation."""
    from test_block_kernel_asm import _uncovered_touches
    for name, (rows, planar, body, meta) in wg.items():
        bad = _uncovered_touches(body)
        assert not bad, (name, bad[:3])
