"""Static check on the line kernels' gfx950 assembly (no GPU needed; hipcc cross-compiles).

csrc/pcm_line_kernel.hip and csrc/ohm_frame_kernel.hip issue their audio (and header) loads from inline asm and wait for them
with a later `s_waitcnt vmcnt(0)` statement, so that several loads are in flight per lane.  Between issue and wait the
destination register holds stale data and the compiler does not know: if it copies the variable there -- it does so at control-flow
joins and loop entries ("the rule behind the hand-counted waits") -- the copy is stale.  ohm_wide_kernel
shipped such a copy for an afternoon: two of its three depth variants moved the header register right after a conditional
load.  The property is one of the generated code, so it is checked there: from every inline-asm load, along EVERY path of the
kernel's control-flow graph, no instruction may name a destination register before a `vmcnt(0)` wait is reached.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ohpipeline_amd", "csrc")
OUTDIR = os.path.join(ROOT, "ohpipeline_amd", "build")

ASM_LOAD = re.compile(r"^\s*global_load_dword(?:x[234])?\s+(v\[\d+:\d+\]|v\d+)\s*,")
WAIT0 = re.compile(r"^\s*s_waitcnt\b.*vmcnt\(0\)")
LABEL = re.compile(r"^(\.LBB\w+):")
BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)")


def _regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def kernels_of(stem, wanted):
    os.makedirs(OUTDIR, exist_ok=True)
    src = os.path.join(CSRC, stem + ".hip")
    out = os.path.join(OUTDIR, stem + ".test.s")
    deps = [src] + [os.path.join(CSRC, f) for f in ("ohgpu_internal.h", "pcm_device.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        from ohpipeline_amd import build as product_build
        own = product_build.SOURCE_FLAGS.get(stem + ".hip", [])
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm", *own,
               "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True, timeout=1500)
    found, name, body = {}, None, []
    for line in open(out).read().split("\n"):
        m = re.match(r"^(_ZN5ohgpu\d+" + wanted + r"\w*):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                found[name] = body
                name = None
    return found


def stale_reads(body):
    """(load line, offending line) for every inline-asm load whose destination some path names before a vmcnt(0) wait."""
    code, in_asm, from_asm = [], False, []
    for line in body:
        text = line.split(";")[0].rstrip() if not line.lstrip().startswith(";;#") else line.strip()
        if text == ";;#ASMSTART":
            in_asm = True
            continue
        if text == ";;#ASMEND":
            in_asm = False
            continue
        if not text.strip() or text.strip().startswith(";"):
            continue
        if text.lstrip().startswith(".") and not LABEL.match(text.strip()):
            continue
        code.append(text.strip())
        from_asm.append(in_asm)
    label_at = {LABEL.match(c).group(1): i for i, c in enumerate(code) if LABEL.match(c)}
    bad = []
    for i, c in enumerate(code):
        m = ASM_LOAD.match(c)
        if not (m and from_asm[i]):
            continue
        dst = _regs(m.group(1))
        seen, stack = set(), [i + 1]
        while stack:
            k = stack.pop()
            while k < len(code) and k not in seen:
                seen.add(k)
                ins = code[k]
                if LABEL.match(ins):
                    k += 1
                    continue
                if WAIT0.match(ins):
                    break
                if ins.startswith("s_endpgm"):
                    break
                b = BRANCH.match(ins)
                if b:
                    stack.append(label_at[b.group(2)])
                    if b.group(1) == "s_branch":
                        break
                    k += 1
                    continue
                used = set()
                for tok in re.findall(r"v\[\d+:\d+\]|v\d+", ins):
                    used |= _regs(tok)
                if used & dst and not (from_asm[k] and ASM_LOAD.match(ins) and _regs(ASM_LOAD.match(ins).group(1)) == dst):
                    bad.append((c, ins))
                    break
                k += 1
    return bad


@pytest.mark.parametrize("stem,wanted,at_least", [("pcm_line_kernel", "pcm_line_kernel", 10), ("ohm_frame_kernel", "ohm_wide_kernel", 1)])
def test_no_path_names_an_asm_load_destination_before_the_wait(stem, wanted, at_least):
    found = kernels_of(stem, wanted)
    assert len(found) >= at_least, sorted(found)
    n_loads = 0
    for name, body in found.items():
        n_loads += sum(1 for line in body if ASM_LOAD.match(line))
        bad = stale_reads(body)
        assert not bad, (name, bad[:3])
    assert n_loads > 0


def test_the_scan_sees_a_copy_behind_a_branch():
    """The shape that shipped: a conditional load, the join's register copy, then the wait."""
    body = """
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
	;;#ASMSTART
	global_load_dword v35, v34, s[0:1]
	;;#ASMEND
.LBB0_2:
	s_or_b64 exec, exec, s[4:5]
	v_mov_b32_e32 v16, v35
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
""".split("\n")
    assert stale_reads(body) == [("global_load_dword v35, v34, s[0:1]", "v_mov_b32_e32 v16, v35")]
    ok = [line for line in body if "v_mov_b32" not in line]
    assert stale_reads(ok) == []
