"""N > 1 path on the CPU: two ranks over gloo build their stream shards as bench.py does -- config 3: rank r owns streams
[r*S, (r+1)*S); config 4: contiguous blocks of the mixed streams balanced by bytes -- with no data-path collective, and the
union equals the single-process result."""
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
    # a stand-in for /sys/bus/pci/devices: GPU r's local_cpulist = every second CPU the test may run on, from the r-th
    import tempfile
    allowed = sorted(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as sysfs:
        for r in range(2):
            os.makedirs(os.path.join(sysfs, f"0000:0{r}:00.0"))
            with open(os.path.join(sysfs, f"0000:0{r}:00.0", "local_cpulist"), "w") as f:
                f.write(",".join(str(c) for c in allowed[r::2]) + "\n")
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OHGPU_TEST_SYSFS=sysfs))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    threads = res.pop("threads_per_rank")
    assert res == {"ok": True, "streams": 12, "max": 2.0, "config3": True, "config4": True, "host_share": True}
    import bench
    budget = min(len(allowed), bench.cgroup_quota_cpus() or len(allowed))
    assert threads == max(1, budget // 2)                          # each of the two ranks: half of what the container grants


def test_host_share_divides_the_quota_and_reads_cpulists():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    one = bench.host_share(1)
    eight = bench.host_share(8)
    assert one["threads"] >= eight["threads"] >= 1 and eight["threads"] == max(1, one["cpu_budget"] // 8)
    assert one["cpus"] is None                                      # no device named: the rank stays where it is


def test_partition_by_bytes_is_contiguous_complete_and_even():
    sys.path.insert(0, ROOT)
    import bench
    for total, world in ((2048, 1), (2048, 2), (2048, 4), (2048, 8), (13, 4), (3, 8)):
        w = [bench.stream_weight(*bench.config4_stream(s), 2.0) for s in range(total)]
        cuts = bench.partition_by_bytes(w, world)
        assert cuts[0] == 0 and cuts[-1] == total and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
        if total >= 4 * world:
            shares = [sum(w[a:b]) for a, b in zip(cuts, cuts[1:])]
            assert max(shares) - min(shares) <= 2 * max(w), (total, world, shares)


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 4])
def test_bench_runs_under_the_launcher_with_two_ranks_on_a_gpu(config):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with two ranks
    sharing the box's one GPU (`local_rank % device_count`): rendezvous, per-rank contexts, the barrier and the max over ranks,
    one JSON line from rank 0 with the whole job's throughput.  The launcher starts before anything touches the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", str(config), "--steps", "2", "--warmup", "1",
           "--seconds", "0.3", "--streams", "24" if config == 4 else "6", "--sustain", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # rank 0 alone reports
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["unit"] == "Msamples/s"
    assert res["scaling"] == ("strong" if config == 4 else "weak")
    if config == 3:                                                      # weak: every rank brought its own 6 streams
        assert abs(res["value"] * res["ms_per_step"] * 1e3 / (2 * 6 * round(0.3 * 44100)) - 1.0) < 0.01   # (both figures are rounded)
    assert res["roofline"]["frac"] > 0 and res["cpu_baseline"] is None    # (the CPU leg runs at N = 1 only)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_when_asked_for_two_gpus():
    """`python bench.py --gpus 2` as the driver types it, no launcher around it: bench.py starts the two ranks itself (a child
    torch.distributed.run, before this process has loaded the C ABI), relays rank 0's line and the child's exit code.  Two
    ranks share the box's one GPU here; every rank checks its first stream against the oracle."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--seconds", "0.3", "--streams", "6", "--sustain", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config"]["sharding"].startswith("streams over 2 rank(s)")
    assert res["check"].startswith("bit-exact vs oracle")


def test_bench_gpus_flag_is_acted_on_before_anything_touches_the_gpu():
    """Static: the launcher branch of bench.main() comes before the first import of the C ABI (an exec or a fork after HIP is
    initialised takes the box down; a child process started earlier does not)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert "launch_ranks(args)" in main
    assert main.index("launch_ranks(args)") < main.index("from ohpipeline_amd import capi")
    body = src[src.index("def launch_ranks("):src.index("def measure(")]
    assert "subprocess.run(" in body and "os.exec" not in body and "capi" not in body
