"""The C ABI takes plain device pointers and a HIP stream: check that torch-owned memory and torch's
current stream work through it (torch is plumbing only -- device memory, streams, torch.distributed).

torch bundles its own HIP runtime; a process that uses torch.cuda must import torch BEFORE loading
libohgpu.so so that both share one runtime (same SONAME, first one loaded wins).  The check therefore
runs in a child process with that import order."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, ctypes as C
import numpy as np
import torch                                   # first: its HIP runtime is the one the process keeps
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib as O
import workloads as W
from ohpipeline_amd import capi
assert torch.cuda.is_available()
dev = torch.device("cuda:0")
ctx = capi.Context(0)
frames = 44100
sched = W.ramp_schedule((frames + 219) // 220, 220 * 1280, 50 * O.JIFFIES_PER_MS, 500 * O.JIFFIES_PER_MS)
descs, sb, db = W.pcm_stream_descs(4, frames, 220, 2, 16, O.ENDIAN_LITTLE, 24, O.ENDIAN_BIG, sched)
src = np.concatenate([W.noise_pcm(s, frames, 2, 16, O.ENDIAN_LITTLE) for s in range(4)])
t_src = torch.from_numpy(src).to(dev)
t_dst = torch.zeros(db, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream(dev)
batch = ctx.pcm_batch(descs, src.size, db)
ctx.pcm_run(batch, C.c_void_p(t_src.data_ptr()), C.c_void_p(t_dst.data_ptr()), C.c_void_p(stream.cuda_stream))
stream.synchronize()
got = t_dst.cpu().numpy()
want = np.zeros(db, dtype=np.uint8)
assert O.msg_process_batch(descs, src, want) == 0
assert np.array_equal(got, want), "mismatch"
ctx.batch_destroy(batch)
ctx.close()
print("INTEROP-OK")
"""


def test_torch_tensors_and_stream_through_the_c_abi():
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "INTEROP-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
