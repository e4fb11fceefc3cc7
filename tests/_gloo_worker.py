"""Worker for tests/test_multirank_gloo.py (launched by torch.distributed.run, gloo backend, CPU only).
Each rank builds its shard of the bench workloads exactly as bench.py does -- config 3: rank r owns streams [r*S, (r+1)*S);
config 4: a contiguous block of the mixed streams, the blocks balanced by bytes -- runs it through the CPU oracle, and the
ranks check together that the shards are disjoint, complete, and equal to what one process computes for all streams."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
from ohpipeline_amd import capi  # noqa: E402


def digests(groups):
    """{stream id: sha256 of its output bytes}, computed by the CPU oracle."""
    out = {}
    for g in groups:
        ref = O.Src(g.rate_in, bench.RATE_OUT, g.taps, bench.BETA, bench.F_PASS)
        dst = np.zeros(g.dst_bytes, dtype=np.uint8)
        assert ref.process_batch(g.descs.view(O.SRC_MSG_DESC), g.src, dst) == 0
        per = g.dst_bytes // len(g.stream_ids)
        for k, sid in enumerate(g.stream_ids):
            out[sid] = hashlib.sha256(dst[k * per:(k + 1) * per].tobytes()).hexdigest()
    return out


def shard(config, streams, seconds, rank, world):
    args = argparse.Namespace(config=config, streams=streams, seconds=seconds, rate_in=44100, channels=2)
    groups, scaling = bench.build_groups(capi, args, rank, world)
    return digests(groups), scaling, sum(g.algorithmic_bytes for g in groups)


def main():
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {}
    # config 3: weak scaling, 3 streams per rank
    mine, scaling3, _ = shard(3, 3, 0.1, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)       # bench.py's max-over-ranks of the timed region
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        union = {}
        for part in gathered:
            assert not (set(part) & set(union))
            union.update(part)
        whole, _, _ = shard(3, 3 * world, 0.1, 0, 1)
        res["config3"] = bool(union == whole and len(union) == 3 * world and scaling3 == "weak" and t.item() == float(world))
    # config 4: strong scaling, 12 mixed streams in all, contiguous blocks balanced by bytes
    mine, scaling4, my_bytes = shard(4, 12, 0.05, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, my_bytes))
    dist.barrier()
    if rank == 0:
        union, sizes = {}, []
        for part, nbytes in gathered:
            assert not (set(part) & set(union))
            ids = sorted(part)
            assert ids == list(range(ids[0], ids[-1] + 1))          # a contiguous block
            union.update(part)
            sizes.append(nbytes)
        whole, _, all_bytes = shard(4, 12, 0.05, 0, 1)
        biggest = max(bench.stream_weight(*bench.config4_stream(s), 0.05) for s in range(12))
        balanced = max(sizes) - min(sizes) <= 2 * biggest            # as even as whole streams allow
        res["config4"] = bool(union == whole and len(union) == 12 and scaling4 == "strong" and balanced)
    # the host's CPUs are shared by the ranks: every rank takes budget / world threads (planner pool, feeder pools) and, where the
    # topology is readable, the CPUs nearest its GPU (a sysfs tree of the test's making stands in for /sys/bus/pci/devices);
    # rank 0's line carries the SLOWEST rank's plan and decode times (bench.slowest_rank: what bench.measure reports at N > 1)
    allowed = sorted(os.sched_getaffinity(0))
    share = bench.host_share(world, f"0000:0{rank}:00.0", os.environ["OHGPU_TEST_SYSFS"])
    budget = len(allowed)
    quota = bench.cgroup_quota_cpus()
    if quota is not None:
        budget = min(budget, quota)
    share_ok = share["threads"] == max(1, budget // world) and share["local_world"] == world
    near = allowed[rank::world]                                    # what the test's sysfs tree says is near this rank's GPU
    share_ok = share_ok and (share["cpus"] == near or (share["cpus"] is None and near == allowed))
    unreadable = bench.host_share(world, "0000:ff:00.0", os.environ["OHGPU_TEST_SYSFS"])
    share_ok = share_ok and unreadable["cpus"] is None and unreadable["threads"] == share["threads"]
    capi.set_plan_threads(min(16, share["threads"]))                # (what bench.main does with it)
    slow = bench.slowest_rank(dist, 1.5 + rank, 2.5 + 2 * rank, 0.25 * (world - rank))
    oks = [None] * world
    dist.all_gather_object(oks, bool(share_ok))
    if rank == 0:
        res["host_share"] = all(oks) and slow == [1.5 + (world - 1), 2.5 + 2 * (world - 1), 0.25 * world]
        print(json.dumps({"ok": bool(res["config3"] and res["config4"] and res["host_share"]), "streams": len(union), "max": t.item(), **res,
                          "threads_per_rank": share["threads"]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
