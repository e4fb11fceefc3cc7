"""Builds and runs tests/cpp/test_host.cpp: the C++ host adapter (Msg model mirror, SampleRateConverter, Songcast sender,
StarvationManager) exercised the way the reference's TestMsg.cpp / TestRamper.cpp / TestStarvationRamper.cpp exercise the
originals; with a GPU every byte is checked against the oracle, including 64 lanes starving in one tick."""
import os
import subprocess

import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "cpp", "build")
EXE = os.path.join(BUILD, "test_host")


def build_test_binary():
    from ohpipeline_amd import build as product_build
    product_build.build()
    product_build.build_host()
    oracle_lib.build()
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_host.cpp")
    lib_dir = os.path.join(ROOT, "ohpipeline_amd", "lib")
    oracle_dir = os.path.join(ROOT, "oracle")
    deps = [src, os.path.join(lib_dir, "libohhost.so"), os.path.join(oracle_dir, "libohp_oracle.so")]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", lib_dir, "-lohhost", "-lohgpu", "-L", oracle_dir, "-lohp_oracle",
                           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{oracle_dir}"])
    return EXE


def run(mode):
    exe = build_test_binary()
    out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        lines = out.stdout.splitlines()
        raise AssertionError("\n".join(sorted(set(lines), key=lines.index)[:60]) + out.stderr[-2000:])
    return out.stdout


def test_host_adapter_control_plane():
    assert "cpu:" in run("cpu") and " 0 failures" in run("cpu")


@pytest.mark.gpu
def test_host_adapter_reads_through_the_gpu():
    out = run("gpu")
    assert " 0 failures" in out, out
