"""CPU-side checks of the C ABI: the library loads, exports exactly what include/ohgpu.h declares,
its struct layouts match the header, and the host-side tables (ramp multipliers, resampler design)
agree with the golden data / the oracle.  No compute call is made (there is no GPU here)."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from ohpipeline_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ohgpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ohgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/ohgpu.h but not exported by libohgpu.so"
    assert sorted(capi.SYMBOLS) == names, "capi.SYMBOLS and include/ohgpu.h disagree"
    assert L.ohgpu_abi_version() == 1


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ohgpu.h"\n#include "ohp_pipeline.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ohgpu_msg_desc), sizeof(ohgpu_src_msg_desc),'
                   'sizeof(ohp_msg_desc), sizeof(ohp_src_msg_desc), offsetof(ohgpu_msg_desc, flags),'
                   'offsetof(ohgpu_src_msg_desc, n_frames), offsetof(ohgpu_src_msg_desc, flags));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                           str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == out[2] == capi.MSG_DESC.itemsize == O.MSG_DESC.itemsize == 32
    assert out[1] == out[3] == capi.SRC_MSG_DESC.itemsize == O.SRC_MSG_DESC.itemsize == 64
    assert out[4] == capi.MSG_DESC.fields["flags"][1] == 31
    assert out[5] == capi.SRC_MSG_DESC.fields["n_frames"][1] == 40
    assert out[6] == capi.SRC_MSG_DESC.fields["flags"][1] == 55


def test_songcast_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ohgpu.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ohgpu_ohm_stream), sizeof(ohgpu_ohm_fragment),'
                   'sizeof(ohgpu_ohm_frame_desc), offsetof(ohgpu_ohm_stream, codec), offsetof(ohgpu_ohm_fragment, flags),'
                   'offsetof(ohgpu_ohm_frame_desc, first_fragment), offsetof(ohgpu_ohm_frame_desc, flags),'
                   'sizeof(ohgpu_flywheel_desc));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == capi.OHM_STREAM.itemsize == 64
    assert out[1] == capi.OHM_FRAGMENT.itemsize == 24
    assert out[2] == capi.OHM_FRAME_DESC.itemsize == 48
    assert out[3] == capi.OHM_STREAM.fields["codec"][1] == 21
    assert out[4] == capi.OHM_FRAGMENT.fields["flags"][1] == 18
    assert out[5] == capi.OHM_FRAME_DESC.fields["first_fragment"][1] == 36
    assert out[6] == capi.OHM_FRAME_DESC.fields["flags"][1] == 42
    assert out[7] == capi.FLYWHEEL_DESC.itemsize == 48


def test_ramp_table_equals_reference_data():
    """The table the DEVICE uses (generated in host_design.cpp) equals RampArray.h:7-74 entry for entry."""
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "ramp_table_q15.json")))["values"]
    assert capi.ramp_table().tolist() == golden


@pytest.mark.parametrize("rin,rout,T", [(44100, 48000, 32), (96000, 48000, 64), (88200, 48000, 24), (48000, 44100, 32),
                                        (32000, 48000, 16), (192000, 48000, 96)])
def test_src_design_equals_oracle(rin, rout, T):
    """Product-side filter design (C++) and the oracle's (C) produce the same Q28 table."""
    L_, M_, coef = capi.src_design(rin, rout, T, 9.0, 20000.0)
    ref = O.Src(rin, rout, T, 9.0, 20000.0)
    assert (L_, M_) == (ref.L, ref.M)
    assert np.array_equal(coef, ref.coef_q28)
    assert ref.sum_abs_max < (1 << 30)
    for n in (0, 1, 146, 147, 148, 220, 441000):
        assert capi.lib().ohgpu_src_out_frames(L_, M_, n) == ref.out_frames(n)


def test_init_without_gpu_fails_loudly():
    L = capi.lib()
    n = L.ohgpu_device_count()
    if n > 0:
        pytest.skip("a GPU is visible; the no-device path cannot be exercised")
    h = C.c_void_p()
    assert L.ohgpu_init(0, C.byref(h)) == capi.ERR_NO_DEVICE
    assert b"no CPU fallback" in L.ohgpu_last_error()
    with pytest.raises(capi.OhGpuError):
        capi.Context(0)


def test_product_never_touches_the_oracle():
    """The product path may not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "ohpipeline_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                code = re.sub(r"//.*|/\*.*?\*/|#.*|\"\"\".*?\"\"\"", "", text, flags=re.S)
                assert "ohp_" not in code and "oracle_lib" not in code, f"{f} references the oracle"
    needed = subprocess.check_output(["readelf", "-d", capi.LIB_PATH]).decode()
    assert "ohp_oracle" not in needed


def test_the_shipped_library_is_not_a_diagnostic_build():
    """The ablation switches, phase stamps and environment hooks of the kernels exist only under -DOHGPU_DIAG, which
    build.py adds only when OHGPU_EXTRA_FLAGS asks for it (tools/exp_*.sh).  The shipped object must carry none of it: no
    hook names in its data, no getenv among its imports, and every mention of a hook in the sources inside an #ifdef."""
    from ohpipeline_amd import build as product_build
    lib_path = product_build.LIB_PATH
    blob = open(lib_path, "rb").read()
    for needle in (b"OHGPU_DIAG", b"OHGPU_EXP", b"STAMP_FILE"):
        assert needle not in blob, needle
    imports = subprocess.run(["nm", "-D", "--undefined-only", lib_path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in imports
    assert "OHGPU_DIAG" not in " ".join(sum(product_build.SOURCE_FLAGS.values(), []))
    csrc = os.path.join(ROOT, "ohpipeline_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        depth_diag = []                                   # stack of "is this #if level a DIAG guard (or inside one)"
        for no, line in enumerate(open(os.path.join(csrc, name), errors="replace"), 1):
            code = line.split("//")[0]
            s = code.strip()
            if s.startswith(("#ifdef", "#ifndef", "#if ")):
                assert "OHGPU_EXP" not in s, f"{name}:{no}: an experiment switch outside the OHGPU_DIAG family: {s}"
                inside = bool(depth_diag and depth_diag[-1])
                depth_diag.append(inside or ("OHGPU_DIAG" in s and not s.startswith("#ifndef")) or
                                  (s.startswith("#ifndef") and "OHGPU_DIAG" in s and False))
            elif s.startswith("#endif"):
                depth_diag.pop()
            elif s.startswith(("#else", "#elif")):
                pass                                      # (the other arm of a DIAG guard is product code, but it names no hook)
            elif "getenv" in code or re.search(r"OHGPU_(DIAG|EXP)_\w+", code):
                if s.startswith("#define STAMP") or "#ifndef OHGPU_DIAG" in s:
                    continue
                assert depth_diag and depth_diag[-1], f"{name}:{no}: a diagnostic hook outside #ifdef OHGPU_DIAG*: {s}"
