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

HIP_SOURCES = ["ohgpu_api.hip", "pcm_kernels.hip", "pcm_line_kernel.hip", "flywheel_kernel.hip", "fmt_line_kernel.hip", "ohm_frame_kernel.hip", "ramp_plane_kernel.hip", "host_design.cpp", "src_plan.cpp", "src_block_kernel.hip", "src_lean_kernel.hip", "src_mfma_kernel.hip", "src_mfma_wg_kernel.hip"]
# Rounds 1's block kernel and round 4's unit-per-wave matrix kernel are retired from the shipped library (round 5): their device code
# is compiled only with OHGPU_LEGACY=1 in the environment (-DOHGPU_LEGACY_KERNELS: ohgpu_set_kernel_variant 2 and 5 then select them,
# for same-box A/Bs).  What the planner takes from their files -- layout lists, geometry, the matrix kernels' host tables -- is
# compiled always.
LEGACY = os.environ.get("OHGPU_LEGACY", "") not in ("", "0")
PARTED = ("src_block_kernel.hip", "src_lean_kernel.hip") if LEGACY else ("src_lean_kernel.hip",)     # compiled once per part of the instantiation list (csrc/src_block_common.h)
HEADERS = ["ohgpu_internal.h", "pcm_device.h", os.path.join(ROOT, "include", "ohgpu.h")]
# Per-source flags.  The lean kernel's per-frame control flow is wave-uniform (scalar compares); LLVM's structurizer
# rewrites uniform diamonds into flag-and-test chains unless told to leave uniform regions alone (3-4 scalar instructions
# per input frame in a loop the scalar unit co-limits).
SOURCE_FLAGS = {"src_lean_kernel.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions"]}
ARCH = "gfx950"
BLOCK_PARTS = 6                     # OHGPU_BLOCK_PARTS in csrc/src_block_common.h


# The lean kernel's source leans on this toolchain's behaviour in two places: the internal option in SOURCE_FLAGS, and the
# block placement behind __builtin_expect in its advance loop; tests/test_block_kernel_asm.py checks the generated code for
# what the hand-counted waits need, so a different compiler fails those tests rather than silently miscounting -- but say so
# at build time too.
EXPECTED_HIP = "7.2"


def toolchain_note():
    try:
        out = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return None
    first = next((l for l in out.splitlines() if l.startswith("HIP version")), "")
    if EXPECTED_HIP not in first:
        return f"ohpipeline_amd/build.py: built and measured with HIP {EXPECTED_HIP}.x; this is '{first.strip()}' -- run tests/test_block_kernel_asm.py before trusting the block kernels"
    return None


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
    """One object per source (compiled in parallel, kept under build/obj and reused while the source, the headers and the
    flags are unchanged), then one link."""
    if not force and not is_stale():
        return LIB_PATH
    note = toolchain_note()
    if note:
        print(note, file=sys.stderr)
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(PKG, "build", "obj")
    os.makedirs(obj_dir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             *os.environ.get("OHGPU_EXTRA_FLAGS", "").split(), *(["-DOHGPU_LEGACY_KERNELS"] if LEGACY else []),
             "-Wall", "-Wno-inline-asm", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]
    if save_temps:
        flags += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    headers = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    headers += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers if os.path.exists(h))
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]

    def compile_one(job):
        src, part = job
        own = [] if os.environ.get("OHGPU_NO_SOURCE_FLAGS") else SOURCE_FLAGS.get(os.path.basename(src), [])
        obj = os.path.join(obj_dir, f"{os.path.basename(src)}.{tag}{'.f' if own else ''}.{part}.o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc(), *flags, *own, *([f"-DOHGPU_BLOCK_PART={part}"] if part else []), "-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=PKG)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        # the block kernel's instantiations are compiled in BLOCK_PARTS parts (see the end of src_block_kernel.hip)
        jobs = [(src, part) for src in _sources()
                for part in (range(1, BLOCK_PARTS + 1) if os.path.basename(src) in PARTED else (0,))]
        jobs.sort(key=lambda j: 0 if j[1] else 1)                # the long ones first
        objs = list(ex.map(compile_one, jobs))
    with open(os.path.join(obj_dir, "linked.txt"), "w") as f:         # (what THIS library is made of: tools/build_variant.sh swaps objects in this list)
        f.write("\n".join(objs) + "\n")
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link, cwd=PKG)
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
