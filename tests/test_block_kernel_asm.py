"""Static checks on the block resampler kernels' gfx950 assembly (no GPU needed; hipcc cross-compiles).

The per-output loops of csrc/src_lean_kernel.hip (round 2, the one the batches run on) and csrc/src_block_kernel.hip
(round 1, kept as the A/B reference) issue their LDS traffic from inline asm and wait for it with hand-counted
`s_waitcnt lgkmcnt(N)`.  That is only sound while
  * nothing else that shares the counter and completes out of order is in flight there (scalar memory),
  * the instructions keep the order the counts assume,
  * and no instruction touches a load's destination register between the load's issue and the wait that covers it --
    the compiler does not know the register is written late, and is free to copy a variable wherever it likes.
These are properties of the generated code, so they are checked on the generated code, for every instantiation.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ohpipeline_amd", "csrc")
OUTDIR = os.path.join(ROOT, "ohpipeline_amd", "build")

PARTS = (1, 2, 3, 4, 5, 6)        # OHGPU_BLOCK_PARTS (csrc/src_block_common.h); part 4 = the half-band instantiations, parts 5 and 6 = the lean-only layouts
SCALAR_MEM = re.compile(r"^\s*(s_load|s_buffer_load|s_memtime|s_memrealtime|s_scratch_load|s_store|s_atomic|s_dcache)")
COUNTED_WAIT = re.compile(r"s_waitcnt lgkmcnt\(([1-9]\d*)\)")


def _compile(stem, prefix):
    """Assembly of every product instantiation of csrc/<stem>.hip, compiled the way the build does: in parts, side by side."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OUTDIR, exist_ok=True)
    src = os.path.join(CSRC, stem + ".hip")
    deps = [src] + [os.path.join(CSRC, f) for f in ("ohgpu_internal.h", "pcm_device.h", "src_block_common.h")]

    from ohpipeline_amd import build as product_build
    own = product_build.SOURCE_FLAGS.get(stem + ".hip", [])
    deps.append(os.path.join(ROOT, "ohpipeline_amd", "build.py"))

    def compile_part(part):
        out = os.path.join(OUTDIR, f"{stem}.test.{part}.s")
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm", *own,
                   f"-DOHGPU_BLOCK_PART={part}", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out]
            subprocess.run(cmd, check=True, capture_output=True, timeout=1500)
        return open(out).read().split("\n")

    with ThreadPoolExecutor(4) as ex:
        texts = list(ex.map(compile_part, PARTS))
    found = {}
    for text in texts:
        name, body = None, []
        for line in text:
            m = re.match(r"^(_ZN5ohgpu\d+" + prefix + r"\w+):", line)
            if m:
                name, body = m.group(1), []
            elif name is not None:
                body.append(line)
                if "s_endpgm" in line:
                    found[name] = body
                    name = None
    assert len(found) >= 15, "expected every instantiation in the assembly"
    return found


@pytest.fixture(scope="module")
def lean():
    return _compile("src_lean_kernel", "src_lean_kernel")




def _halfband(name):
    """The lean kernel's seventh template argument: the half-band 2:1 decimator (T = 64 stored, T / 2 taps meet the window)."""
    m = re.search(r"src_lean_kernelILi\d+ELi\d+ELi\d+ELb\dELi\d+ELb\dELb(\d)E", name)
    return bool(m and m.group(1) == "1")


def _taps_of(name):
    """DPP taps per output: the filter's T, or T / 2 for a half-band instantiation (its centre tap is a plain v_fmac_f64)."""
    t = int(re.search(r"kernelILi(\d+)E", name).group(1))
    return t // 2 if _halfband(name) else t


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_no_scalar_memory_traffic_between_taps(which, request):
    for name, body in request.getfixturevalue(which).items():
        taps = [i for i, l in enumerate(body) if "v_fmac_f64_dpp" in l]
        assert taps, name
        bad = [l for l in body[taps[0]:taps[-1] + 1] if SCALAR_MEM.match(l)]
        assert not bad, (name, bad[:3])


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_no_scratch_and_no_valu_exec_writes(which, request):
    for name, body in request.getfixturevalue(which).items():
        assert not any(re.match(r"^\s*scratch_", l) for l in body), name
        # a VALU write of EXEC needs five wait states before a DPP instruction; the kernels rely on having none
        assert not any(re.match(r"^\s*v_cmpx", l) for l in body), name


def test_lean_kernels_keep_three_waves_per_simd(lean):
    """The 32-tap instantiations are launched with up to twelve waves per workgroup: at most 168 registers (and no spills:
    test_no_scratch_and_no_valu_exec_writes)."""
    seen_halfband = seen_wide = 0
    for part in PARTS:
        text = open(os.path.join(OUTDIR, f"src_lean_kernel.test.{part}.s")).read()
        for m in re.finditer(r"\.name:\s+(_ZN5ohgpu15src_lean_kernelILi(\d+)E\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
            if int(m.group(2)) <= 32 or _halfband(m.group(1)):     # (a half-band kernel's window is 32 frames too: that is its point)
                assert int(m.group(3)) <= 168, (m.group(1), m.group(3))
                seen_halfband += _halfband(m.group(1))
                # six channels and more are launched with up to SIXTEEN waves per workgroup (lean_max_waves): 128 registers --
                # but for the half-band kernels, which carry the delay line and stay at twelve
                ch = int(re.search(r"src_lean_kernelILi\d+ELi(\d+)E", m.group(1)).group(1))
                if ch >= 6 and not _halfband(m.group(1)):
                    assert int(m.group(3)) <= 128, (m.group(1), m.group(3))
                    seen_wide += 1
    assert seen_halfband >= 3 and seen_wide >= 7


def tap_waits(body):
    """Indices of the counted waits that open a tap group: taps follow within a few instructions.  (The lean kernel also waits
    with a count where an advance that emits nothing picks up its sample -- lgkmcnt(1), the look-ahead read just issued stays in
    flight -- and nothing but the unpack follows that one.)"""
    out = []
    for i, l in enumerate(body):
        if COUNTED_WAIT.search(l):
            code = [x for x in body[i + 1:i + 14] if x.strip() and not x.strip().startswith((";", "."))][:8]
            if any("v_fmac_f64_dpp" in x for x in code):
                out.append(i)
    return out


def main_loop(body):
    """From the first output's first counted wait (the last one before the first tap) to the end of the kernel."""
    first_tap = next(i for i, l in enumerate(body) if "v_fmac_f64_dpp" in l)
    start = max(i for i in range(first_tap) if COUNTED_WAIT.search(body[i]))
    return body[start:]


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_every_output_has_its_counted_waits(which, request):
    for name, body in request.getfixturevalue(which).items():
        T = _taps_of(name)
        taps = sum("v_fmac_f64_dpp" in l for l in body)
        assert taps % T == 0
        outputs = taps // T                                   # unrolled output bodies
        loop = main_loop(body)                                # (the unit's set-up has compiler-counted waits of its own)
        counted = [int(COUNTED_WAIT.search(loop[i]).group(1)) for i in tap_waits(loop)]
        # one wait per coefficient register (T / 16 of them) per output body, each leaving younger operations in flight
        assert len(counted) == outputs * (T // 16), (name, len(counted), outputs)
        if which == "lean":                                   # the lean kernel's waits are all lgkmcnt(NCR - 1)
            assert set(counted) == {T // 16 - 1}, (name, set(counted))


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_lds_traffic_keeps_the_order_the_counts_assume(which, request):
    """After every counted wait: [the ring store(s), only after an output's first wait] 16 taps, then the reload of the
    coefficient register those taps used -- and no other LDS instruction in between (the counts in the waits are the
    number of LDS operations issued after the awaited one; a moved instruction would change it silently)."""
    for name, body in request.getfixturevalue(which).items():
        T = _taps_of(name)
        ncr = T // 16
        body = main_loop(body)
        waits = tap_waits(body)
        assert waits
        seen_in_output = 0
        for i in waits:
            # an output's first wait: the old kernel tells it by its count, the lean kernel's waits all look the same
            if which == "block":
                first = int(COUNTED_WAIT.search(body[i]).group(1)) == ncr - 1
            else:
                first = seen_in_output == 0
                seen_in_output = (seen_in_output + 1) % ncr
            taps, stores, k = 0, 0, i + 1
            while taps < 16:
                line = body[k]
                assert not re.match(r"^\s*(s_cbranch|s_branch|s_waitcnt)", line), (name, i, line)
                if "v_fmac_f64_dpp" in line:
                    taps += 1
                elif re.match(r"^\s*ds_write", line):
                    assert first and taps == 0, (name, i, line)                  # the ring store sits in front of the taps
                    stores += 1
                else:
                    assert not re.match(r"^\s*ds_", line), (name, i, line)      # no other LDS traffic among the taps
                k += 1
            assert (stores in (1, 3)) if first else stores == 0, (name, i, stores)
            while not re.match(r"^\s*(ds_|s_waitcnt|s_cbranch|s_branch)", body[k]):
                k += 1
            assert re.match(r"^\s*ds_read_b64", body[k]), (name, i, body[k])    # the reload follows its taps directly


def _regs(tok):
    """Vector registers a token names: 'v12' -> {12}, 'v[8:11]' -> {8..11}."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


LOAD = re.compile(r"^\s*(ds_read\w*|global_load_(?:dword\w*|ushort|ubyte|sbyte|short\w*))\s+(v\[\d+:\d+\]|v\d+)\s*,")
ANY_WAIT = {"ds": re.compile(r"s_waitcnt.*lgkmcnt"), "gl": re.compile(r"s_waitcnt.*vmcnt")}


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_no_instruction_touches_a_load_destination_before_its_wait(which, request):
    """Between a load's issue and the next wait on its counter, no instruction of the straight-line code that follows may
    name the destination registers (the scan stops at a branch or a label: there the next block opens with the wait)."""
    for name, body in request.getfixturevalue(which).items():
        for i, line in enumerate(body):
            m = LOAD.match(line)
            if not m or "lds" in m.group(1):
                continue
            dst = _regs(m.group(2))
            wait = ANY_WAIT["ds" if m.group(1).startswith("ds_") else "gl"]
            for follow in body[i + 1:i + 400]:
                code = follow.split(";")[0].strip()
                if not code or code.startswith("."):
                    if re.match(r"^\.?LBB|^\d+:", code):
                        break
                    continue
                if re.match(r"^(\d+|\.LBB\w+):", code) or re.match(r"^(s_cbranch|s_branch|s_endpgm|s_setpc)", code) or wait.search(code):
                    break
                used = set()
                for tok in re.findall(r"v\[\d+:\d+\]|v\d+", code):
                    used |= _regs(tok)
                assert not (used & dst), (name, line.strip(), code)


# ---- round 4: what the round-3 diagnostic build's memory access fault came down to ----
# A build with the coefficient reloads compiled out (and the taps: nothing consumed the coefficient registers any more) faulted on
# the device.  Its ISA shows why it was an ADDRESS and not a sample that went wrong: the unit's first coefficient reads,
# `ds_read_b64 v[0:1]`, are issued from inline asm and nothing names v[0:1] afterwards -- so the compiler took the pair for dead
# and put a 64-bit global address into it three instructions later (`v_lshl_add_u64 v[0:1], ...`, then `global_load_ubyte`),
# while the LDS read was still on its way: the late write landed in the address.  The product build is safe because every
# asm-issued load's destination is an operand of a later statement that sits behind a wait covering the load; the check below
# holds the generated code to exactly that, by walking the LGKM queue instead of accepting "some wait" as the older test does:
# LDS operations complete in issue order, so `s_waitcnt lgkmcnt(N)` covers a read iff at least N younger LDS operations were issued
# behind it -- and covers nothing counted if a scalar load (out of order on the same counter) is in flight, unless N is 0.
LGKM_OP = re.compile(r"^\s*(ds_\w+|s_load\w*|s_buffer_load\w*)\b")
LGKM_WAIT = re.compile(r"s_waitcnt.*lgkmcnt\((\d+)\)")


def _uncovered_touches(body):
    """(load line, touching line) for every LDS read whose destination is named by an instruction of the straight-line code behind
    it before a wait that covers it."""
    bad = []
    for i, line in enumerate(body):
        m = LOAD.match(line)
        if not m or not m.group(1).startswith("ds_read"):
            continue
        dst = _regs(m.group(2))
        younger, scalar_in_flight = 0, False
        for follow in body[i + 1:i + 600]:
            code = follow.split(";")[0].strip()
            if not code or code.startswith("."):
                if re.match(r"^\.?LBB|^\d+:", code):
                    break
                continue
            if re.match(r"^(\d+|\.LBB\w+):", code) or re.match(r"^(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)", code):
                break
            w = LGKM_WAIT.search(code)
            if w:
                n = int(w.group(1))
                if n == 0 or (n <= younger and not scalar_in_flight):
                    break                                   # covered
                continue
            if re.match(r"^s_waitcnt", code) and "lgkmcnt" not in code:
                continue
            if LGKM_OP.match(code):
                if code.startswith("s_"):
                    scalar_in_flight = True
                younger += 1
            used = set()
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", code):
                used |= _regs(tok)
            # (a later LDS READ into the same registers is the next load's business; a store or ALU instruction naming them is not)
            if used & dst and not (LOAD.match(follow) and _regs(LOAD.match(follow).group(2)) == dst and not (used - dst) & dst):
                bad.append((line.strip(), code))
                break
    return bad


@pytest.mark.parametrize("which", ["lean"])     # (round 1's block kernel is retired from the shipped library: OHGPU_LEGACY builds only)
def test_no_lds_read_lands_in_a_register_the_code_has_moved_on_from(which, request):
    for name, body in request.getfixturevalue(which).items():
        assert not _uncovered_touches(body), (name, _uncovered_touches(body)[:3])


def test_the_check_sees_the_diagnostic_builds_fault():
    """The shape of the round-3 fault, reduced: an asm-issued read nobody names again, its register pair recycled for an address."""
    body = ["\tv_mov_b64_e32 v[0:1], 0", "\tds_read_b64 v[0:1], v28 offset:0x80", "\ts_and_b32 s14, s12, -16",
            "\tv_lshl_add_u64 v[0:1], s[14:15], 0, v[10:11]", "\tglobal_load_ubyte v3, v[0:1], off", "\ts_endpgm"]
    assert _uncovered_touches(body) == [("ds_read_b64 v[0:1], v28 offset:0x80", "v_lshl_add_u64 v[0:1], s[14:15], 0, v[10:11]")]
    # a counted wait with enough younger LDS operations behind the read covers it ...
    ok = ["\tds_read_b64 v[0:1], v28", "\tds_read_b64 v[2:3], v28 offset:8", "\ts_waitcnt lgkmcnt(1)", "\tv_add_f64 v[4:5], v[0:1], v[0:1]", "\ts_endpgm"]
    assert _uncovered_touches(ok) == []
    # ... one that leaves the read itself among the operations still allowed in flight does not, nor does any count above zero
    # while a scalar load shares the counter
    short = ["\tds_read_b64 v[0:1], v28", "\ts_waitcnt lgkmcnt(1)", "\tv_add_f64 v[4:5], v[0:1], v[0:1]", "\ts_endpgm"]
    assert _uncovered_touches(short)
    scalar = ["\tds_read_b64 v[0:1], v28", "\ts_load_dwordx2 s[4:5], s[0:1], 0x0", "\tds_read_b64 v[2:3], v28 offset:8", "\ts_waitcnt lgkmcnt(1)",
              "\tv_add_f64 v[4:5], v[0:1], v[0:1]", "\ts_endpgm"]
    assert _uncovered_touches(scalar)
