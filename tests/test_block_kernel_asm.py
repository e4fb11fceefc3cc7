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
    """Assembly of every product instantiation, compiled the way the build does: in parts, side by side."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), f) for f in ("ohgpu_internal.h", "pcm_device.h")]

    def compile_part(part):
        out = OUT.replace(".test.s", f".test.{part}.s")
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                   f"-DOHGPU_BLOCK_PART={part}", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", SRC, "-o", out]
            subprocess.run(cmd, check=True, capture_output=True, timeout=1500)
        return open(out).read().split("\n")

    with ThreadPoolExecutor(3) as ex:
        texts = list(ex.map(compile_part, (1, 2, 3)))
    found = {}
    for text in texts:
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


def main_loop(body):
    """From the first output's first counted wait (the last one before the first tap) to the end of the kernel."""
    first_tap = next(i for i, l in enumerate(body) if "v_fmac_f64_dpp" in l)
    start = max(i for i in range(first_tap) if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", body[i]))
    return body[start:]


def test_every_output_has_its_counted_waits(kernels):
    for name, body in kernels.items():
        T = int(re.search(r"src_block_kernelILi(\d+)E", name).group(1))
        taps = sum("v_fmac_f64_dpp" in l for l in body)
        assert taps % T == 0
        outputs = taps // T                                   # unrolled output bodies
        loop = main_loop(body)                                # (the unit's set-up has compiler-counted waits of its own)
        counted = sum(1 for l in loop if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l))
        # one wait per coefficient register (T / 16 of them) per output body, each leaving younger operations in flight
        assert counted == outputs * (T // 16), (name, counted, outputs)


def test_lds_traffic_keeps_the_order_the_counts_assume(kernels):
    """After every counted wait: [the ring store(s), only after an output's first wait] 16 taps, then the reload of the
    coefficient register those taps used -- and no other LDS instruction in between (the counts in the waits are the
    number of LDS operations issued after the awaited one; a moved instruction would change it silently)."""
    for name, body in kernels.items():
        T = int(re.search(r"src_block_kernelILi(\d+)E", name).group(1))
        ncr = T // 16
        body = main_loop(body)
        waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l)]
        assert waits
        for i in waits:
            first = int(re.search(r"lgkmcnt\((\d+)\)", body[i]).group(1)) == ncr - 1      # an output's first wait
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
