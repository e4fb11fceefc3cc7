#!/usr/bin/env python3
"""bench.py -- headline benchmark: PCM Msamples/s on 256 independent stereo S24 44.1->48 kHz streams,
resample + ramp + format, per GPU (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (one fused resample->ramp->pack launch) over the whole batch:
256 streams x 10 s of audio per GPU, inputs and descriptors already resident in HBM.  Streams shard
across ranks with no collective (weak scaling: every rank owns 256 streams of its own).
Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 256
SECONDS = 10
RATE_IN, RATE_OUT = 44100, 48000
CHANNELS, BITS = 2, 24
TAPS, BETA, F_PASS = 32, 9.0, 20000.0
OUT_FRAMES_PER_MSG = 240            # 5 ms at 48 kHz (CodecController.cpp:792-793 chunking, at the output rate)
JIFFIES_PER_MS = 56448
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
LCG_A, LCG_C, MASK = 1664525, 1013904223, 0xFFFFFFFF


def lcg_block(seed, n):
    out = np.empty(n, dtype=np.uint64)
    out[0] = (seed * LCG_A + LCG_C) & MASK
    have, a_k, c_k = 1, LCG_A, LCG_C
    while have < n:
        take = min(have, n - have)
        out[have:have + take] = (out[:take] * a_k + c_k) & MASK
        c_k = (c_k * a_k + c_k) & MASK
        a_k = (a_k * a_k) & MASK
        have += take
    return out


def noise_s24le(stream_id, n_frames):
    """Seeded LCG full-scale noise: subsample = top 24 bits of each LCG word, packed little endian."""
    x = lcg_block((0x9E3779B9 * (stream_id + 1)) & MASK, n_frames * CHANNELS)
    b = np.empty((x.size, 3), dtype=np.uint8)
    b[:, 0] = (x >> 8) & 0xFF
    b[:, 1] = (x >> 16) & 0xFF
    b[:, 2] = (x >> 24) & 0xFF
    return b.reshape(-1)


def build_workload(capi, first_stream, n_streams, in_frames):
    """Input arena + one descriptor per 5 ms output message, ramp endpoints from the host ramp algebra."""
    from ohpipeline_amd import hostmodel
    L_, M_, coef = capi.src_design(RATE_IN, RATE_OUT, TAPS, BETA, F_PASS)
    out_total = (in_frames * L_ + M_ - 1) // M_
    n_msgs = (out_total + OUT_FRAMES_PER_MSG - 1) // OUT_FRAMES_PER_MSG
    jps_out = 56448000 // RATE_OUT
    first = np.arange(n_msgs, dtype=np.int64) * OUT_FRAMES_PER_MSG
    count = np.minimum(OUT_FRAMES_PER_MSG, out_total - first)
    sched = hostmodel.stream_ramp_schedule([int(c) * jps_out for c in count], 50 * JIFFIES_PER_MS, 500 * JIFFIES_PER_MS)
    sched = np.array(sched, dtype=np.int64)
    fb = CHANNELS * BITS // 8
    descs = np.zeros(n_streams * n_msgs, dtype=capi.SRC_MSG_DESC)
    for s in range(n_streams):
        sl = slice(s * n_msgs, (s + 1) * n_msgs)
        descs["src_offset"][sl] = s * in_frames * fb
        descs["src_frames"][sl] = in_frames
        descs["out_frame0"][sl] = first
        descs["dst_offset"][sl] = s * out_total * fb + first * fb
        descs["n_frames"][sl] = count
        descs["flags"][sl] = sched[:, 0]
        descs["ramp_start"][sl] = sched[:, 1]
        descs["ramp_end"][sl] = sched[:, 2]
    descs["attenuation"] = 256
    descs["channels"], descs["src_bits"], descs["src_endian"] = CHANNELS, BITS, capi.ENDIAN_LITTLE
    descs["dst_bits"], descs["dst_endian"] = BITS, capi.ENDIAN_BIG
    src = np.empty(n_streams * in_frames * fb, dtype=np.uint8)
    for s in range(n_streams):
        src[s * in_frames * fb:(s + 1) * in_frames * fb] = noise_s24le(first_stream + s, in_frames)
    if os.environ.get("OHGPU_BENCH_ZERO_INPUT"):            # (diagnosis only: how much of the time is the clock the data costs)
        src[:] = 0
    return dict(L=L_, M=M_, coef=coef, descs=descs, src=src, out_total=out_total, n_msgs=n_msgs,
                dst_bytes=n_streams * out_total * fb)


def cpu_baseline(work, n_streams, in_frames):
    """Times the CPU oracle (the restatement of the reference path + the resampler model, gcc -O2) on the GPU box's
    host cores: the same descriptors and input, streams statically partitioned over one thread per core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O
    ref = O.Src(RATE_IN, RATE_OUT, TAPS, BETA, F_PASS)
    assert np.array_equal(ref.coef_q28, work["coef"])
    cores = len(os.sched_getaffinity(0))
    threads = max(1, min(cores, n_streams, int(os.environ.get("OHGPU_BENCH_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share
    descs, src = work["descs"], work["src"]
    dst = np.zeros(work["dst_bytes"], dtype=np.uint8)
    n_msgs = work["n_msgs"]
    bounds = np.linspace(0, n_streams, threads + 1).astype(int)
    lib = O.lib()

    def job(t):
        part = np.ascontiguousarray(descs[bounds[t] * n_msgs:bounds[t + 1] * n_msgs])
        return lib.ohp_src_msg_process_batch(ref.h, part.ctypes.data_as(C.c_void_p), part.size,
                                             src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))

    times = []
    with ThreadPoolExecutor(threads) as ex:
        for _ in range(3):                                        # median of three passes: about 20 core-seconds in all
            t0 = time.perf_counter()
            rcs = list(ex.map(job, range(threads)))
            times.append(time.perf_counter() - t0)
            assert all(r == 0 for r in rcs)
    dt = sorted(times)[1]
    # SURVEY.md 8(d) also asks for one thread alone: the first streams of the same step, about a second of work
    n_one = max(1, min(n_streams, 8))
    part = np.ascontiguousarray(descs[:n_one * n_msgs])
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.ohp_src_msg_process_batch(ref.h, part.ctypes.data_as(C.c_void_p), part.size,
                                           src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))
        one.append(time.perf_counter() - t0)
        assert rc == 0
    return dict(value=round(n_streams * in_frames / dt / 1e6, 3), unit="Msamples/s", cores=threads, kind="port",
                single_thread=round(n_one * in_frames / sorted(one)[1] / 1e6, 3),
                sample=f"the whole step, median of 3 passes: {n_streams} streams x {in_frames} frames, {threads} threads "
                       f"(gcc -O2 oracle, {dt:.2f} s per pass, {sum(times) * threads:.0f} core-seconds in all)"), dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=SECONDS, help="audio per stream (default 10 s = the throughput set)")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 tuned, 1 baseline v1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and end_to_end legs (profiling runs: only the timed launches)")
    ap.add_argument("--check", action="store_true", help="compare the GPU output of the last step with the oracle")
    ap.add_argument("--channels", type=int, default=CHANNELS, help="channels per stream (the headline is stereo; 6 and 8 also run the block kernel)")
    ap.add_argument("--rate-in", type=int, default=RATE_IN, help="input rate (the headline is 44100; 96000 with --taps 64 is config 4's other rate)")
    ap.add_argument("--taps", type=int, default=TAPS, help="taps per phase")
    args = ap.parse_args()
    globals()["CHANNELS"] = args.channels
    globals()["RATE_IN"] = args.rate_in
    globals()["TAPS"] = args.taps

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch  # noqa: F401  (torch.distributed is plumbing: barrier + max over ranks)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)

    from ohpipeline_amd import capi
    n_streams = args.streams
    in_frames = int(round(args.seconds * RATE_IN))
    work = build_workload(capi, rank * n_streams, n_streams, in_frames)

    # one rank per GPU; on a box with fewer GPUs than ranks (a rehearsal) the ranks share what there is
    ctx = capi.Context(local_rank % max(capi.device_count(), 1) if world > 1 else 0)
    ctx.set_kernel_variant(args.variant)
    h = ctx.src_create(work["L"], work["M"], TAPS, work["coef"])
    d_src = ctx.upload(work["src"])
    d_dst = ctx.malloc(work["dst_bytes"])
    ctx.memset(d_dst, 0, work["dst_bytes"])
    batch = ctx.src_batch(h, work["descs"], work["src"].size, work["dst_bytes"])
    info = ctx.batch_info(batch)
    plan = ctx.src_plan(batch)
    ctx.sync()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        ctx.src_run(batch, d_src, d_dst)
    barrier()
    ev = [(ctx.event(), ctx.event()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ctx.record(ev[k][0])
        ctx.src_run(batch, d_src, d_dst)
        ctx.record(ev[k][1])
    ctx.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = [ctx.elapsed_ms(a, b) for a, b in ev]
    kernel_avg_ms = float(np.mean(kernel_ms))
    frames_per_step = n_streams * in_frames                       # input frames per rank per step
    bytes_per_in_frame = CHANNELS * BITS / 8 * (1.0 + work["L"] / work["M"])   # 6 + 6*160/147 = 12.531 B
    algorithmic_bytes = frames_per_step * bytes_per_in_frame
    achieved_gbps = algorithmic_bytes / (kernel_avg_ms * 1e-3) / 1e9

    result = None
    if rank == 0:
        total_frames = frames_per_step * world * args.steps
        result = {
            "metric": "PCM Msamples/s, 256-stream 44.1->48k S24 resample+ramp+fmt",
            "value": round(total_frames / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "msubsamples_per_s": round(total_frames * CHANNELS / elapsed / 1e6, 3),   # frames x channels
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {n_streams} independent {'stereo' if CHANNELS == 2 else str(CHANNELS) + '-channel'} S24LE streams per GPU, {RATE_IN / 1000:g}->48 kHz, "
                                   f"{args.seconds:g} s each ({in_frames} frames), 5 ms output messages, "
                                   f"ramp up 50 ms / down 500 ms, S24 BE out",
                       "streams_per_gpu": n_streams, "channels": CHANNELS, "frames_per_stream": in_frames, "taps_per_phase": TAPS,
                       "msgs_per_step": int(info["n_msgs"]), "kernel_variant": args.variant,
                       "block_kernel_out_frames": plan["block_kernel_out_frames"], "generic_pieces": plan["generic_pieces"],
                       "sharding": f"streams x{world} ranks, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbps, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved_gbps / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "fused resample+ramp+pack", "kernel_avg_ms": round(kernel_avg_ms, 4),
                         "algorithmic_bytes_per_launch": int(algorithmic_bytes),
                         "bytes_per_input_frame": round(bytes_per_in_frame, 4),
                         # the pipe that actually bounds this kernel: 2*T*channels*L/M fp64 flop per input frame
                         # on the vector (= matrix) fp64 pipe, 78.6 TFLOP/s dense on MI355X
                         "fp64_tflops": round(frames_per_step * 2.0 * TAPS * CHANNELS * work["L"] / work["M"] / (kernel_avg_ms * 1e-3) / 1e12, 2),
                         "fp64_peak_tflops": 78.6,
                         "fp64_frac": round(frames_per_step * 2.0 * TAPS * CHANNELS * work["L"] / work["M"] / (kernel_avg_ms * 1e-3) / 1e12 / 78.6, 4)},
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                p = json.load(open(pmc))
                if p.get("streams_per_gpu") == n_streams and p.get("frames_per_stream") == in_frames \
                        and p.get("kernel_variant") == args.variant:
                    result["roofline"]["traffic"] = p.get("hbm_bytes_per_launch")
                    result["roofline"]["traffic_source"] = p.get("source")
            except Exception:
                pass
        overlapped_out = None
        if world == 1 and not args.no_cpu:
            try:                                             # (a reported extra: never let it cost the headline line)
                # the same step with the buffers on the host side of the boundary (pinned): H2D of the input,
                # the launch, D2H of the output, wall clock.  Reported beside `value`, never as it.
                h_src = ctx.malloc_host(work["src"].nbytes)
                h_dst = ctx.malloc_host(work["dst_bytes"])
                h_src[:] = work["src"].view(np.uint8).reshape(-1)
                e2e = []
                for _ in range(3):
                    ctx.sync()
                    t1 = time.perf_counter()
                    ctx.copy_h2d(d_src, h_src)
                    ctx.src_run(batch, d_src, d_dst)
                    ctx.copy_d2h(h_dst, d_dst)
                    ctx.sync()
                    e2e.append(time.perf_counter() - t1)
                dt = sorted(e2e)[1]
                result["end_to_end"] = {"value": round(frames_per_step / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
                                        "pcie_gbps": round((work["src"].nbytes + work["dst_bytes"]) / dt / 1e9, 2),
                                        "what": "pinned host input -> H2D -> launch -> D2H -> pinned host output, median of 3"}
                # the same with the streams in groups: one HIP stream uploads, a second waits for each group's upload (event),
                # launches it and downloads its output, so the link carries both directions while the kernel runs
                groups = int(os.environ.get("OHGPU_BENCH_GROUPS", "8"))
                if n_streams % groups != 0:
                    groups = 1
                if groups > 1:
                    import ctypes as C
                    per, fb = n_streams // groups, CHANNELS * BITS // 8
                    n_msgs, sb, db = work["n_msgs"], in_frames * fb, work["out_total"] * fb
                    parts = [ctx.src_batch(h, work["descs"][g * per * n_msgs:(g + 1) * per * n_msgs], work["src"].size, work["dst_bytes"])
                             for g in range(groups)]
                    lanes = [ctx.stream_create(), ctx.stream_create()]
                    up, down = lanes
                    arrived = [ctx.event() for _ in range(groups)]
                    h_dst[:] = 0
                    ctx.memset(d_dst, 0, work["dst_bytes"])
                    ov = []
                    for _ in range(3):
                        ctx.sync()
                        t1 = time.perf_counter()
                        for g in range(groups):
                            s0, s1, o0, o1 = g * per * sb, (g + 1) * per * sb, g * per * db, (g + 1) * per * db
                            ctx.copy_h2d(C.c_void_p(d_src.value + s0), h_src[s0:s1], up)
                            ctx.record(arrived[g], up)
                            ctx.wait_event(down, arrived[g])
                            ctx.src_run(parts[g], d_src, d_dst, down)
                            ctx.copy_d2h(h_dst[o0:o1], C.c_void_p(d_dst.value + o0), down)
                        for st in lanes:
                            ctx.sync(st)
                        ov.append(time.perf_counter() - t1)
                    dt = sorted(ov)[1]
                    result["end_to_end"]["overlapped"] = {"value": round(frames_per_step / dt / 1e6, 3), "ms_per_step": round(dt * 1e3, 3),
                                                          "pcie_gbps": round((work["src"].nbytes + work["dst_bytes"]) / dt / 1e9, 2),
                                                          "what": f"{groups} groups of {per} streams, upload stream + launch/download stream"}
                    overlapped_out = np.array(h_dst)
                    for st in lanes:
                        ctx.stream_destroy(st)
                    for b in parts:
                        ctx.batch_destroy(b)
                ctx.free_host(h_src)
                ctx.free_host(h_dst)
            except Exception as e:
                result["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
                overlapped_out = None
        if world == 1 and not args.no_cpu:
            base, cpu_out = cpu_baseline(work, n_streams, in_frames)
            result["cpu_baseline"] = base
            if args.check:
                got = ctx.download(d_dst, work["dst_bytes"])
                result["check"] = "bit-exact vs oracle" if np.array_equal(got, cpu_out) else "MISMATCH"
                if overlapped_out is not None:
                    result["check_overlapped"] = "bit-exact vs oracle" if np.array_equal(overlapped_out, cpu_out) else "MISMATCH"
        else:
            result["cpu_baseline"] = None
    ctx.batch_destroy(batch)
    ctx.src_destroy(h)
    ctx.free(d_src)
    ctx.free(d_dst)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
