"""Builds libohgpu.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc.  No GPU needed to build."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libohgpu.so")

HIP_SOURCES = ["ohgpu_api.hip", "pcm_kernels.hip", "pcm_line_kernel.hip", "flywheel_kernel.hip", "fmt_line_kernel.hip", "host_design.cpp", "src_plan.cpp", "src_block_kernel.hip"]
HEADERS = ["ohgpu_internal.h", "pcm_device.h", os.path.join(ROOT, "include", "ohgpu.h")]
ARCH = "gfx950"


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def _sources():
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    extra = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith(".hip") and f not in HIP_SOURCES)
    return srcs + extra


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = _sources() + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, save_temps=False):
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           *os.environ.get("OHGPU_EXTRA_FLAGS", "").split(),
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-o", LIB_PATH] + _sources()
    if save_temps:
        tmp = os.path.join(PKG, "build")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=PKG)
    return LIB_PATH


HOST_DIR = os.path.join(PKG, "host")
HOST_LIB_PATH = os.path.join(LIB_DIR, "libohhost.so")


def build_host(force=False, verbose=False):
    """The C++ host adapter (Msg model mirror + control plane) over the C ABI; plain g++, links libohgpu.so."""
    srcs = sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cpp"))
    hdrs = [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "ohgpu.h"))
    if not force and os.path.exists(HOST_LIB_PATH):
        t = os.path.getmtime(HOST_LIB_PATH)
        if all(os.path.getmtime(f) <= t for f in srcs + hdrs) and os.path.getmtime(LIB_PATH) <= t:
            return HOST_LIB_PATH
    build(force=False, verbose=verbose)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-I", os.path.join(ROOT, "include"), "-o", HOST_LIB_PATH] + srcs + \
          ["-L", LIB_DIR, "-lohgpu", "-Wl,-rpath,$ORIGIN", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=PKG)
    return HOST_LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv))
    print(build_host(force="--force" in sys.argv, verbose=True))
