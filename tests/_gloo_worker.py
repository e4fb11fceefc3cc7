"""Worker for tests/test_multirank_gloo.py (launched by torch.distributed.run, gloo backend, CPU only).
Each rank builds its shard of the bench workload exactly as bench.py does, runs it through the CPU oracle, and the
ranks check together that shards are disjoint, complete, and equal to what one process computes for all streams."""
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


def run_shard(first_stream, n_streams, in_frames):
    work = bench.build_workload(capi, first_stream, n_streams, in_frames)
    ref = O.Src(bench.RATE_IN, bench.RATE_OUT, bench.TAPS, bench.BETA, bench.F_PASS)
    dst = np.zeros(work["dst_bytes"], dtype=np.uint8)
    assert ref.process_batch(work["descs"].view(O.SRC_MSG_DESC), work["src"], dst) == 0
    per_stream = work["dst_bytes"] // n_streams
    return [hashlib.sha256(dst[s * per_stream:(s + 1) * per_stream].tobytes()).hexdigest() for s in range(n_streams)]


def main():
    dist.init_process_group(backend="gloo", init_method="env://")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_streams, in_frames = 3, 4410
    mine = run_shard(rank * n_streams, n_streams, in_frames)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)       # bench.py's max-over-ranks of the timed region
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        flat = [h for part in gathered for h in part]
        whole = run_shard(0, n_streams * world, in_frames)
        ok = flat == whole and len(set(flat)) == len(flat) and t.item() == float(world)
        print(json.dumps({"ok": bool(ok), "streams": len(flat), "max": t.item()}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
