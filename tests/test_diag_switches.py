"""Every compile-time diagnostic switch of the kernels is on the register (tools/diag_switches.txt) that says what it does to the
kernel's addressing, and the experiment scripts refuse one that is not: round 4's MF_DIAG_IO_CONTIG faulted on the GPU box in its
first cut because nothing asked that question before the build went there."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ohpipeline_amd", "csrc")
FAMILIES = re.compile(r"\b(MF_(?:DIAG|WG|LOAD|STORE)_[A-Z0-9_]+|OHGPU_DIAG(?:_[A-Z0-9_]+)?|OHGPU_WG_ROWS|OHGPU_PLAN_TIMING|OHGPU_LINE_[A-Z_]*GROUPS_PER_CU|OHGPU_LEGACY_KERNELS)\b")


def listed():
    out = {}
    for line in open(os.path.join(ROOT, "tools", "diag_switches.txt")):
        m = re.match(r"^([A-Z0-9_]+)\s+(unchanged|checked)\b", line)
        if m:
            out[m.group(1)] = m.group(2)
    return out


def test_every_switch_in_the_sources_is_on_the_register():
    reg = listed()
    found = set()
    for name in os.listdir(CSRC):
        if not name.endswith((".hip", ".h", ".cpp")):
            continue
        for line in open(os.path.join(CSRC, name)):
            if re.match(r"\s*#\s*(if|ifdef|ifndef|elif)\b", line):
                found.update(FAMILIES.findall(line))
    assert found, "no switches found: the pattern has rotted"
    assert not (found - set(reg)), f"switches the register does not know: {sorted(found - set(reg))}"


def test_the_guard_refuses_what_is_not_listed():
    guard = os.path.join(ROOT, "tools", "check_diag_flags.sh")
    assert subprocess.run(["bash", guard, "-DMF_DIAG_IO_CONTIG", "-DMF_DIAG_BARRIER_MASK=7", "-O3"]).returncode == 0
    bad = subprocess.run(["bash", guard, "-DMF_DIAG_WALKS_OFF_THE_ARENA"], capture_output=True, text=True)
    assert bad.returncode == 1 and "not listed" in bad.stderr
    for script in ("exp_mfma.sh", "build_variant.sh"):
        assert "check_diag_flags.sh" in open(os.path.join(ROOT, "tools", script)).read()
