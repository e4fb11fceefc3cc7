"""Static checks on the block kernel's gfx950 assembly (no GPU needed; hipcc cross-compiles).

The per-output loop of csrc/src_block_kernel.hip issues its LDS traffic from inline asm and waits for it with
hand-counted `s_waitcnt lgkmcnt(N)`.  That is only sound while nothing else that shares the counter and completes out
of order is in flight there, and the DPP taps need VALU-written EXEC/coefficients to be settled.  These are properties
of the generated code, so they are checked on the generated code, for every instantiation.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ohpipeline_amd", "csrc", "src_block_kernel.hip")
OUT = os.path.join(ROOT, "ohpipeline_amd", "build", "src_block_kernel.test.s")

SCALAR_MEM = re.compile(r"^\s*(s_load|s_buffer_load|s_memtime|s_memrealtime|s_scratch_load|s_store|s_atomic|s_dcache)")


@pytest.fixture(scope="module")
def kernels():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), f) for f in ("ohgpu_internal.h", "pcm_device.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", SRC, "-o", OUT]
        subprocess.run(cmd, check=True, capture_output=True, timeout=1500)
    text = open(OUT).read().split("\n")
    found = {}
    name, body = None, []
    for line in text:
        m = re.match(r"^(_ZN5ohgpu16src_block_kernel\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                # the diagnostic (stamped) instantiation reads the clock, each time with its own full wait: not a product path
                if "Lb1EEEv" not in name:
                    found[name] = body
                name = None
    assert len(found) >= 15, "expected every instantiation in the assembly"
    return found


def test_no_scalar_memory_traffic_between_taps(kernels):
    for name, body in kernels.items():
        taps = [i for i, l in enumerate(body) if "v_fmac_f64_dpp" in l]
        assert taps, name
        bad = [l for l in body[taps[0]:taps[-1] + 1] if SCALAR_MEM.match(l)]
        assert not bad, (name, bad[:3])


def test_no_scratch_and_no_valu_exec_writes(kernels):
    for name, body in kernels.items():
        assert not any(re.match(r"^\s*scratch_", l) for l in body), name
        # a VALU write of EXEC needs five wait states before a DPP instruction; the kernel relies on having none
        assert not any(re.match(r"^\s*v_cmpx", l) for l in body), name


def test_every_output_has_its_counted_waits(kernels):
    for name, body in kernels.items():
        T = int(re.search(r"src_block_kernelILi(\d+)E", name).group(1))
        taps = sum("v_fmac_f64_dpp" in l for l in body)
        assert taps % T == 0
        outputs = taps // T                                   # unrolled output bodies
        counted = sum(1 for l in body if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l))
        # one wait per coefficient register (T / 16 of them) per output body, each leaving younger operations in flight
        assert counted == outputs * (T // 16), (name, counted, outputs)
