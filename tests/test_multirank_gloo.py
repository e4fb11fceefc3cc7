"""N > 1 path on the CPU: two ranks over gloo build their stream shards as bench.py does (rank r owns streams
[r*S, (r+1)*S), no data-path collective), and the union equals the single-process result."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_shard_streams_without_collectives():
    from ohpipeline_amd import build as product_build
    product_build.build()
    product_build.build_host()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res == {"ok": True, "streams": 6, "max": 2.0}
