#!/usr/bin/env python3
"""Host-side cost of planning: ohgpu_src_batch_create / ohgpu_pcm_batch_create on the headline workload's 512 000 messages
(validation, sorting into segments, chunk / work-unit lists, uploads).  Measured on the GPU box: 29 ms and 11 ms."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from ohpipeline_amd import capi
import bench
ctx = capi.Context(0)
work = bench.build_workload(capi, 0, 256, 441000)
h = ctx.src_create(work["L"], work["M"], 32, work["coef"])
t0 = time.perf_counter(); b = ctx.src_batch(h, work["descs"], work["src"].size, work["dst_bytes"]); t1 = time.perf_counter()
print("src_batch_create: %d msgs in %.3f s" % (work["descs"].size, t1 - t0))
ctx.batch_destroy(b)
d = np.zeros(work["descs"].size, dtype=capi.MSG_DESC)
for f in ("src_offset", "dst_offset", "n_frames", "ramp_start", "ramp_end", "attenuation", "channels", "src_bits", "src_endian", "dst_bits", "dst_endian", "flags"):
    d[f] = work["descs"][f]
d["src_offset"] = d["dst_offset"]
t0 = time.perf_counter(); b = ctx.pcm_batch(d, work["dst_bytes"], work["dst_bytes"]); t1 = time.perf_counter()
print("pcm_batch_create: %d msgs in %.3f s" % (d.size, t1 - t0))
