#!/usr/bin/env python3
"""bench.py -- PCM Msamples/s of the resample + ramp + format hot path on MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 5 --warmup 2                  # configs[2], the headline (default)
    python bench.py --config 4                                      # configs[3]: 2048 mixed streams (44.1/96 -> 48 kHz, 2/6/8 channels)
    python bench.py --config 5                                      # configs[4]: FLAC frames -> pack -> resample -> ramp -> S24
    python bench.py --gpus N                                        # N > 1 without a launcher: bench.py starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                      # ... or is started as one of them (WORLD_SIZE set)

A "step" is one pass of the hot path over the whole batch a rank owns: every launch the batch needs (one fused
resample->ramp->pack launch per filter/layout group; config 5 also the FLAC pack), inputs and descriptors already resident in
HBM.  Streams shard across ranks with no collective.  Rank 0 prints ONE JSON line with `roofline` (HIP events on every
launch of the step, on the launch stream), `cpu_baseline` (the CPU oracle timed on this host's cores, and the full-size
bit-exact check of the GPU's output against it) and `cadence` (one 5 ms message per stream per call, the live regime).
The default line (config 3, one GPU) also carries `configs`: BASELINE configs[3] and configs[4] run in the same process at
their full size (`--no-extra-configs` skips them), each with its own ms_per_step, roofline fraction and bit-exact check.
`plan_ms` is the wall time of ohgpu_src_batch_create for the step's batches (the plan is made once and reused by every launch;
median of three creations, `plan_ms_first` the process's first).
The CPU oracle (oracle/, tests/oracle_lib.py) is used only for the `cpu_baseline` leg, never on the measured path.  A run
whose check fails prints the line with `value` null and exits 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RATE_OUT = 48000
BITS = 24
BETA, F_PASS = 9.0, 20000.0
OUT_FRAMES_PER_MSG = 240            # 5 ms at 48 kHz (CodecController.cpp:792-793 chunking, at the output rate)
JIFFIES_PER_MS = 56448
HOST_THREADS = None                 # a rank's share of the host's CPUs (main(): quota / ranks on this host); None = no cap
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


def noise_s24le(stream_id, n_subsamples):
    """Seeded LCG full-scale noise: subsample = top 24 bits of each LCG word, packed little endian."""
    x = lcg_block((0x9E3779B9 * (stream_id + 1)) & MASK, n_subsamples)
    b = np.empty((x.size, 3), dtype=np.uint8)
    b[:, 0] = (x >> 8) & 0xFF
    b[:, 1] = (x >> 16) & 0xFF
    b[:, 2] = (x >> 24) & 0xFF
    return b.reshape(-1)


def noise_s16le(stream_id, n_subsamples):
    """... the top 16 bits of the same words ("top 24 (or 16) bits")."""
    x = lcg_block((0x9E3779B9 * (stream_id + 1)) & MASK, n_subsamples)
    b = np.empty((x.size, 2), dtype=np.uint8)
    b[:, 0] = (x >> 16) & 0xFF
    b[:, 1] = (x >> 24) & 0xFF
    return b.reshape(-1)


def taps_for(rate_in):
    return 64 if rate_in >= 2 * RATE_OUT else 32        # 96 -> 48 kHz: twice the prototype length per phase


class Group:
    """Streams that share a filter and a layout: one source arena, one destination arena, one batch, one launch per step."""

    def __init__(self, capi, rate_in, channels, stream_ids, in_frames, src_bits=BITS, src_endian=None, planar=False,
                 dst_bits=BITS, dst_endian=None):
        self.rate_in, self.channels, self.stream_ids, self.in_frames = rate_in, channels, list(stream_ids), in_frames
        self.planar = planar                                  # config 5: the source is the FLAC decoder's TInt32 planes (OHGPU_FLAG_SRC_PLANAR32)
        self.taps = taps_for(rate_in)
        self.L, self.M, self.coef = capi.src_design(rate_in, RATE_OUT, self.taps, BETA, F_PASS)
        self.out_total = (in_frames * self.L + self.M - 1) // self.M
        self.n_msgs = (self.out_total + OUT_FRAMES_PER_MSG - 1) // OUT_FRAMES_PER_MSG
        self.src_bits = src_bits
        self.src_endian = capi.ENDIAN_LITTLE if src_endian is None else src_endian
        self.dst_bits = dst_bits                              # (the bench's own groups write S24 big endian; tests vary it)
        self.dst_endian = capi.ENDIAN_BIG if dst_endian is None else dst_endian
        self.fb_src, self.fb_dst = (channels * 4 if planar else channels * src_bits // 8), channels * dst_bits // 8
        n = len(self.stream_ids)
        self.src_bytes, self.dst_bytes = n * in_frames * self.fb_src, n * self.out_total * self.fb_dst
        self.descs = self._descs(capi)
        self.src = None                                       # filled by the caller (noise, or config 5's decoded planes)
        self.d_src_external = None
        # what the CPU oracle is given: the same descriptors and source -- for a planar group the oracle's COMPOSITION, the planes
        # packed as CodecFlac::CallbackWrite packs them (filled by bench_flac) and descriptors that read that packed audio
        self.oracle_src = None
        self.oracle_descs = self.descs
        if planar:
            od = self.descs.copy()
            per = in_frames * channels * src_bits // 8
            for k in range(n):
                od["src_offset"][k * self.n_msgs:(k + 1) * self.n_msgs] = k * per
            od["flags"] &= ~np.uint8(capi.FLAG_SRC_PLANAR32)
            od["src_plane_stride"] = 0
            self.oracle_descs = od
        # algorithmic bytes of a step: every input frame read once, every output frame written once
        self.algorithmic_bytes = n * in_frames * self.fb_src + n * self.out_total * self.fb_dst
        self.flops = 2.0 * self.taps * channels * n * self.out_total

    def _descs(self, capi):
        from ohpipeline_amd import hostmodel
        jps_out = 56448000 // RATE_OUT
        first = np.arange(self.n_msgs, dtype=np.int64) * OUT_FRAMES_PER_MSG
        count = np.minimum(OUT_FRAMES_PER_MSG, self.out_total - first)
        # Ramper's schedule (Ramper.cpp:114-134 through the host ramp algebra): up over the first 50 ms, down over the last 500 ms
        sched = np.array(hostmodel.stream_ramp_schedule([int(c) * jps_out for c in count], 50 * JIFFIES_PER_MS, 500 * JIFFIES_PER_MS), dtype=np.int64)
        n = len(self.stream_ids)
        d = np.zeros(n * self.n_msgs, dtype=capi.SRC_MSG_DESC)
        for s in range(n):
            sl = slice(s * self.n_msgs, (s + 1) * self.n_msgs)
            d["src_offset"][sl] = s * self.in_frames * self.fb_src
            d["src_frames"][sl] = self.in_frames
            d["out_frame0"][sl] = first
            d["dst_offset"][sl] = s * self.out_total * self.fb_dst + first * self.fb_dst
            d["n_frames"][sl] = count
            d["flags"][sl] = sched[:, 0]
            d["ramp_start"][sl] = sched[:, 1]
            d["ramp_end"][sl] = sched[:, 2]
        d["attenuation"] = 256
        if self.planar:
            d["flags"] |= capi.FLAG_SRC_PLANAR32
            d["src_plane_stride"] = self.in_frames * 4
        d["channels"], d["src_bits"], d["src_endian"] = self.channels, self.src_bits, self.src_endian
        d["dst_bits"], d["dst_endian"] = self.dst_bits, self.dst_endian
        return d

    def fill_noise(self):
        from concurrent.futures import ThreadPoolExecutor
        self.src = np.empty(self.src_bytes, dtype=np.uint8)
        per = self.in_frames * self.fb_src

        def one(k):
            make = noise_s16le if self.fb_src == 2 * self.channels else noise_s24le
            self.src[k * per:(k + 1) * per] = make(self.stream_ids[k], self.in_frames * self.channels)
        with ThreadPoolExecutor(max(1, min(32, HOST_THREADS or len(os.sched_getaffinity(0))))) as ex:     # (numpy releases the GIL in these passes; a rank's share of the host: main())
            list(ex.map(one, range(len(self.stream_ids))))

    def attach(self, ctx):
        self.h = ctx.src_create(self.L, self.M, self.taps, self.coef)
        self.d_src = self.d_src_external if self.d_src_external is not None else ctx.upload(self.src)
        self.d_dst = ctx.malloc(self.dst_bytes)
        ctx.memset(self.d_dst, 0, self.dst_bytes)
        ctx.sync()
        # validation, plan, upload (synchronous): created three times, the median reported -- and the first beside it, which also pays
        # for what a process pays once (the planner's thread pool starting, the allocator's first pages)
        times = []
        for k in range(3):
            if k:
                ctx.batch_destroy(self.batch)
            t0 = time.perf_counter()
            self.batch = ctx.src_batch(self.h, self.descs, self.src_bytes, self.dst_bytes)
            times.append((time.perf_counter() - t0) * 1e3)
        self.plan_ms_first = times[0]
        self.plan_ms = sorted(times)[1]
        self.info, self.plan = ctx.batch_info(self.batch), ctx.src_plan(self.batch)
        self.plan_ms_repeat = self._repeat_ms(ctx)

    def _repeat_ms(self, ctx):
        """What the NEXT period of the same streams costs a caller that keeps its batch: the group's messages one whole period (a
        whole number of blocks) further on -- a batch of its own here, because the bench's own starts its streams, which a period in
        the middle of them does not -- then ohgpu_src_batch_advance by a period and ohgpu_src_batch_set_ramps with every message's
        endpoints: wall clock, median of three.  (Other arena positions cost nothing: they are the base pointers of the launch.)"""
        if self.planar:
            return None
        try:
            L_blk, M_blk = ctx.src_batch_block(self.batch)
            if self.out_total % L_blk or self.in_frames % M_blk or self.out_total // L_blk != self.in_frames // M_blk:
                return None
            blocks, hist = self.out_total // L_blk, M_blk
            d = self.descs.copy()
            d["out_frame0"] += self.out_total
            d["src_frame0"] = self.in_frames - hist                  # the buffer: a block of history, then the period
            d["src_frames"] = self.in_frames + hist
            n = len(self.stream_ids)
            for s in range(n):
                d["src_offset"][s * self.n_msgs:(s + 1) * self.n_msgs] = s * (self.in_frames + hist) * self.fb_src
            b2 = ctx.src_batch(self.h, d, n * (self.in_frames + hist) * self.fb_src, self.dst_bytes)
            starts, ends = np.ascontiguousarray(d["ramp_start"]), np.ascontiguousarray(d["ramp_end"])
            times = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.src_batch_advance(b2, blocks)
                ctx.src_batch_set_ramps(b2, starts, ends)
                times.append((time.perf_counter() - t0) * 1e3)
            ctx.batch_destroy(b2)
            return sorted(times)[1]
        except Exception:
            return None

    def detach(self, ctx):
        ctx.batch_destroy(self.batch)
        ctx.src_destroy(self.h)
        if self.d_src_external is None:
            ctx.free(self.d_src)
        ctx.free(self.d_dst)


def partition_by_bytes(weights, world):
    """Contiguous blocks of streams, balanced by bytes: rank r owns streams [cut[r], cut[r+1])."""
    w = np.asarray(weights, dtype=np.float64)
    c = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(c, c[-1] * r / world, side="left")))
    cuts.append(len(w))
    for r in range(1, world + 1):                  # every rank at least one stream when there are enough
        cuts[r] = max(cuts[r], cuts[r - 1] + (1 if len(w) >= world else 0))
    cuts[-1] = len(w)
    return cuts


def config4_stream(sid):
    """BASELINE configs[3]: the mix is a function of the stream id -- rate 44.1 / 96 kHz alternating, channels 2 / 6 / 8 in turn
    (the reference admits more than two channels for raw PCM only, Codec/CodecController.cpp:727-730)."""
    return (44100 if sid % 2 == 0 else 96000), (2, 6, 8)[sid % 3]


def stream_weight(rate_in, channels, seconds):
    fin = int(round(seconds * rate_in))
    return fin * channels * 3 + int(round(seconds * RATE_OUT)) * channels * 3


def config4_share(total_streams, seconds, rank, world):
    """The streams of `rank`: a contiguous block, the blocks balanced by the bytes a stream moves."""
    w = [stream_weight(*config4_stream(s), seconds) for s in range(total_streams)]
    cuts = partition_by_bytes(w, world)
    return range(cuts[rank], cuts[rank + 1]), w


def build_groups(capi, args, rank, world):
    """The rank's share of the workload as a list of Groups."""
    if args.config == 3:
        ids = range(rank * args.streams, (rank + 1) * args.streams)          # weak scaling: every rank owns `streams` of its own
        g = Group(capi, args.rate_in, args.channels, ids, int(round(args.seconds * args.rate_in)), src_bits=getattr(args, "src_bits", BITS))
        g.fill_noise()
        return [g], "weak"
    mine, _ = config4_share(args.streams, args.seconds, rank, world)         # strong scaling: 2048 streams over the ranks, by bytes
    groups = []
    for rate in (44100, 96000):
        for ch in (2, 6, 8):
            ids = [s for s in mine if config4_stream(s) == (rate, ch)]
            if ids:
                g = Group(capi, rate, ch, ids, int(round(args.seconds * rate)))
                g.fill_noise()
                groups.append(g)
    return groups, "strong"


def pin_and_call(cpu, fn, *a):
    try:
        os.sched_setaffinity(0, {cpu})             # (the calling thread only)
    except OSError:
        pass
    return fn(*a)


def physical_cpus(cpus):
    """One logical CPU per physical core among `cpus` (the kernel's thread_siblings_list): SURVEY.md 8d(ii) asks for one pinned
    worker per PHYSICAL core."""
    seen, out = set(), []
    for c in cpus:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_baseline(groups, got_by_group, light=False):
    """Times the CPU oracle (restatement of the reference path + the resampler's integer model, gcc -O2) on this host: the same
    descriptors and input, streams statically partitioned over pinned threads -- ONE PER PHYSICAL CORE the process may run on
     -- scratch buffers allocated once per job (ohp_src_msg_process_batch_steady); median of three passes
    (`light`: one pass, no side figures).  Beside it: the same step on 16 threads (a one-GPU box's CPU share, what rounds 1-2
    reported) and on one thread.  Its output is the bit-exact check of the GPU's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O
    allowed = sorted(os.sched_getaffinity(0))
    phys = physical_cpus(allowed)
    if os.environ.get("OHGPU_BENCH_CPU_THREADS"):
        phys = phys[:max(1, int(os.environ["OHGPU_BENCH_CPU_THREADS"]))]
    # The container's CPU quota (cgroup v2 cpu.max = "<quota us> <period us>", or "max ..."): more runnable threads than that get no
    # more CPU time -- on the pool's boxes 128 physical cores are visible and 16 CPUs' worth of time is granted, and 128 pinned
    # threads measured 15 cores busy (cpu_seconds / wall_seconds), 1.2x the 16-thread figure.  The baseline therefore runs one thread
    # per physical core UP TO the quota, and says so.
    cpu_max, quota_cpus = None, None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            cpu_max = f.read().strip()
        q, per = cpu_max.split()
        if q != "max":
            quota_cpus = max(1, int(-(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    phys_all = len(phys)
    if quota_cpus is not None and quota_cpus < len(phys):
        phys = phys[:quota_cpus]
    lib = O.lib()
    outs = [np.zeros(g.dst_bytes, dtype=np.uint8) for g in groups]
    frames = sum(len(g.stream_ids) * g.in_frames for g in groups)
    refs = []
    for g in groups:
        ref = O.Src(g.rate_in, RATE_OUT, g.taps, BETA, F_PASS)
        assert np.array_equal(ref.coef_q28, g.coef)
        refs.append(ref)

    def timed(cpus, passes):
        threads = len(cpus)
        jobs = []
        for g, ref, dst in zip(groups, refs, outs):
            n = len(g.stream_ids)
            bounds = np.linspace(0, n, min(2 * threads, n) + 1).astype(int)
            for t in range(len(bounds) - 1):
                if bounds[t + 1] > bounds[t]:
                    part = np.ascontiguousarray(g.oracle_descs[bounds[t] * g.n_msgs:bounds[t + 1] * g.n_msgs])
                    jobs.append((ref, part, g.src if g.oracle_src is None else g.oracle_src, dst,
                                 (bounds[t + 1] - bounds[t]) * g.in_frames * g.channels * g.taps))
        jobs.sort(key=lambda j: -j[4])                                   # longest first over the pinned threads

        def run(job):
            ref, part, src, dst, _ = job
            return lib.ohp_src_msg_process_batch_steady(ref.h, part.ctypes.data_as(C.c_void_p), part.size,
                                                        src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p))
        # every worker thread gets a CPU of its own, once, when it starts (pinning per job let two live jobs share a core while
        # others idled); the pool is started -- all threads pinned -- before the first timed pass
        import threading
        slot = iter(range(threads))
        slot_lock = threading.Lock()
        started = threading.Barrier(threads + 1)

        def pin_worker():
            with slot_lock:
                k = next(slot)
            try:
                os.sched_setaffinity(0, {cpus[k]})         # (the calling thread only)
            except OSError:
                pass

        times, cpu_user, cpu_sys = [], [], []
        with ThreadPoolExecutor(threads, initializer=pin_worker) as ex:
            warm = [ex.submit(started.wait) for _ in range(threads)]        # (every worker exists and is pinned once these return)
            started.wait()
            for w in warm:
                w.result()
            for _ in range(passes):
                c0 = os.times()
                t0 = time.perf_counter()
                rcs = list(ex.map(run, jobs))
                times.append(time.perf_counter() - t0)
                c1 = os.times()
                cpu_user.append(c1.user - c0.user)
                cpu_sys.append(c1.system - c0.system)
                assert all(r == 0 for r in rcs)
        mid = sorted(range(len(times)), key=lambda i: times[i])[len(times) // 2]
        return times[mid], len(jobs), cpu_user[mid], cpu_sys[mid]

    passes = 1 if light else 3
    dt, n_jobs, cpu_u, cpu_s = timed(phys, passes)
    ok = all(np.array_equal(a, b) for a, b in zip(got_by_group, outs))
    base = dict(value=round(frames / dt / 1e6, 3), unit="Msamples/s", cores=len(phys), kind="port",
                host_cores_online=os.cpu_count(), host_cores_allowed=len(allowed), host_physical_cores_allowed=phys_all,
                cgroup_cpu_max=cpu_max, cgroup_quota_cpus=quota_cpus,
                # what the pass consumed, by the kernel's accounting for the whole process (os.times): user time is the oracle's
                # arithmetic, system time the page faults of its per-job scratch and of the output's first touch
                cpu_seconds=round(cpu_u + cpu_s, 3), cpu_seconds_user=round(cpu_u, 3), cpu_seconds_system=round(cpu_s, 3),
                wall_seconds=round(dt, 4),
                sample=f"the whole step, median of {passes} pass(es): {frames} input frames in {n_jobs} jobs, one thread per physical core up to the "
                       f"container's CPU quota, each pinned to its core when it starts ({len(phys)} of {phys_all} physical cores; gcc -O2 oracle, "
                       f"scratch allocated once per job)")
    if not light:
        if len(phys) > 16:                                  # (no quota: beside the all-cores figure, a one-GPU box's usual CPU share)
            dt16, _, u16, s16 = timed(phys[:16], 3)
            base["threads_16"] = round(frames / dt16 / 1e6, 3)
            base["threads_16_cpu_seconds"] = round(u16 + s16, 3)
        # one thread alone: the first streams of the first group, about a second of work
        g0 = groups[0]
        n_one = max(1, min(len(g0.stream_ids), 8))
        one_part = np.ascontiguousarray(g0.oracle_descs[:n_one * g0.n_msgs])
        g0_src = g0.src if g0.oracle_src is None else g0.oracle_src
        one = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = pin_and_call(phys[0], lib.ohp_src_msg_process_batch_steady, refs[0].h, one_part.ctypes.data_as(C.c_void_p), one_part.size,
                              g0_src.ctypes.data_as(C.c_void_p), outs[0].ctypes.data_as(C.c_void_p))
            one.append(time.perf_counter() - t0)
            assert rc == 0
        try:
            os.sched_setaffinity(0, set(allowed))
        except OSError:
            pass
        base["single_thread"] = round(n_one * g0.in_frames / sorted(one)[1] / 1e6, 3)
        # the threads' arithmetic is independent: what the many-thread figure falls short of threads x one thread is time not spent in
        # it (the CPU seconds say how much of the wall time the cores were busy at all)
        base["parallel_efficiency"] = round(base["value"] / (base["single_thread"] * len(phys)), 4)
        base["cores_busy"] = round((cpu_u + cpu_s) / (dt * len(phys)), 4)
    return base, ("bit-exact vs oracle" if ok else "MISMATCH")


def rank_check(groups, got_first):
    """Multi-rank runs: every rank checks the first stream of each of its groups against the oracle (a cheap check; the whole
    step is checked at N = 1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import oracle_lib as O
    for g, got in zip(groups, got_first):
        ref = O.Src(g.rate_in, RATE_OUT, g.taps, BETA, F_PASS)
        part = np.ascontiguousarray(g.oracle_descs[:g.n_msgs])
        src = g.src if g.oracle_src is None else g.oracle_src
        want = np.zeros(g.out_total * g.fb_dst, dtype=np.uint8)
        rc = O.lib().ohp_src_msg_process_batch_steady(ref.h, part.ctypes.data_as(C.c_void_p), part.size, src.ctypes.data_as(C.c_void_p),
                                                      want.ctypes.data_as(C.c_void_p))
        if rc != 0 or not np.array_equal(got, want):
            return False
    return True


def cadence(ctx, capi, g, calls=200):
    """The live regime (AnimatorBasic.h:30: one pull per 5 ms): ONE 5 ms message per stream per call, for the group's streams.
    `host_buffers` = ohgpu_src_process_host: descriptors validated, input uploaded, launch, output downloaded, synchronised --
    what a driver thread pays per period with host buffers; `resident_launch` = a batch created once, launched and
    synchronised per call."""
    n = len(g.stream_ids)
    k = g.n_msgs // 2                                                       # a message in the middle of the stream (no ramp)
    d = np.ascontiguousarray(g.descs[k::g.n_msgs][:n]).copy()
    m0, nf = int(d["out_frame0"][0]), int(d["n_frames"][0])
    n_lo = max(0, (m0 * g.M) // g.L - (g.taps - 1))                         # the message's window of input, per stream
    n_hi = ((m0 + nf - 1) * g.M) // g.L
    frames = n_hi - n_lo + 1
    src = np.empty(n * frames * g.fb_src, dtype=np.uint8)
    for s in range(n):
        a = s * g.in_frames * g.fb_src + n_lo * g.fb_src
        src[s * frames * g.fb_src:(s + 1) * frames * g.fb_src] = g.src[a:a + frames * g.fb_src]
    d["src_offset"] = np.arange(n, dtype=np.uint64) * (frames * g.fb_src)
    d["src_frame0"], d["src_frames"] = n_lo, frames
    d["dst_offset"] = np.arange(n, dtype=np.uint64) * (nf * g.fb_dst)
    dst_bytes = n * nf * g.fb_dst
    dst = np.zeros(dst_bytes, dtype=np.uint8)
    host = []
    for _ in range(calls):
        t0 = time.perf_counter()
        ctx.src_process_host(g.h, d, src, dst)
        host.append((time.perf_counter() - t0) * 1e6)
    d_src, d_dst = ctx.upload(src), ctx.malloc(dst_bytes)
    b = ctx.src_batch(g.h, d, src.size, dst_bytes)
    ctx.sync()
    res = []
    for _ in range(calls):
        t0 = time.perf_counter()
        ctx.src_run(b, d_src, d_dst)
        ctx.sync()
        res.append((time.perf_counter() - t0) * 1e6)
    ctx.batch_destroy(b)
    ctx.free(d_src)
    ctx.free(d_dst)

    def q(v, p):
        return round(float(np.percentile(v, p)), 1)
    out = {"what": f"{n} streams x one 5 ms message per call ({nf} output frames each), {calls} calls, wall clock per call",
           "period_us": 5000, "host_buffers_us": {"median": q(host, 50), "p99": q(host, 99)},
           "resident_launch_us": {"median": q(res, 50), "p99": q(res, 99)}}
    out["adapter"] = cadence_adapter(g, calls)
    return out


def cadence_adapter(g, ticks=200):
    """The same regime through the C++ host adapter (libohhost.so), as the reference's driver thread would meet it: per stream a
    SampleRateConverter element whose output becomes a playable (PreDriver.cpp:115-133), per 5 ms tick every stream is fed its
    next 5 ms of input and ALL the playables are read with ONE PlayableBatch::Run (MsgPlayable::Read per message on the driver
    thread in the reference: Msg.cpp:2646-2653, AnimatorBasic.cpp:77-142) -- message objects, window packing, one C-ABI call per
    filter, the callbacks' copy-out included.  Two lanes' whole output is checked against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from ohpipeline_amd import hostmodel
    n = len(g.stream_ids)
    if g.planar or g.src_bits not in (16, 24, 32):
        return None
    in_per_tick = g.rate_in // 200
    ticks = min(ticks, g.in_frames // in_per_tick)
    lane_stride = g.in_frames * g.fb_src
    out_stride = 4096
    src = g.src.view(np.uint8).reshape(-1)
    out = np.zeros(n * out_stride, dtype=np.uint8)
    t_us, kept = [], {0: [], n - 1: []}
    with hostmodel.LiveDriver(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch_device_count()), n, g.rate_in, RATE_OUT, g.channels,
                              g.src_bits, g.src_endian == O.ENDIAN_LITTLE, 24) as live:
        for k in range(ticks):
            t0 = time.perf_counter()
            nbytes = live.tick(src[k * in_per_tick * g.fb_src:], lane_stride, in_per_tick, out, out_stride)
            t_us.append((time.perf_counter() - t0) * 1e6)
            for lane in kept:
                kept[lane].append(out[lane * out_stride:lane * out_stride + int(nbytes[lane])].copy())
        st = live.stats()
    ok = True
    ref = O.Src(g.rate_in, RATE_OUT, g.taps, BETA, F_PASS)
    for lane, parts in kept.items():
        got = np.concatenate(parts)
        d = np.zeros(1, dtype=O.SRC_MSG_DESC)
        frames_in = ticks * in_per_tick
        d["src_offset"], d["src_frames"], d["n_frames"] = lane * lane_stride, frames_in, ref.out_frames(frames_in)
        d["attenuation"], d["channels"], d["src_bits"], d["src_endian"] = 256, g.channels, g.src_bits, g.src_endian
        d["dst_bits"], d["dst_endian"] = 24, O.ENDIAN_BIG
        want = np.zeros(int(d["n_frames"][0]) * g.channels * 3, dtype=np.uint8)
        ok = ok and ref.process_batch(d, src, want) == 0 and got.size == want.size and np.array_equal(got, want)

    def q(v, p):
        return round(float(np.percentile(v, p)), 1)
    steady = t_us[10:]
    return {"what": f"{n} SampleRateConverter lanes behind one driver thread, one PlayableBatch::Run per 5 ms tick, {ticks} ticks, through libohhost.so",
            "adapter_us": {"median": q(steady, 50), "p99": q(steady, 99)}, "resampler_calls_per_tick": round(st["src_calls"] / ticks, 3),
            "filters": st["filters"], "h2d_bytes_per_tick": int(st["h2d_bytes"] / ticks), "d2h_bytes_per_tick": int(st["d2h_bytes"] / ticks),
            "device_allocations": st["device_allocs"], "check": "bit-exact vs oracle (2 lanes)" if ok else "MISMATCH"}


def torch_device_count():
    import torch
    return torch.cuda.device_count()


def end_to_end(ctx, capi, g):
    """the same step with the buffers on the host side of the boundary (pinned): H2D of the input, the launch,
    D2H of the output, wall clock -- and the same with the streams in groups on two HIP streams, so that the link carries both
    directions while the kernel runs.  Reported beside `value`, never as it."""
    import ctypes as C
    n_streams = len(g.stream_ids)
    h_src = ctx.malloc_host(g.src.nbytes)
    h_dst = ctx.malloc_host(g.dst_bytes)
    h_src[:] = g.src.view(np.uint8).reshape(-1)
    e2e = []
    for _ in range(3):
        ctx.sync()
        t1 = time.perf_counter()
        ctx.copy_h2d(g.d_src, h_src)
        ctx.src_run(g.batch, g.d_src, g.d_dst)
        ctx.copy_d2h(h_dst, g.d_dst)
        ctx.sync()
        e2e.append(time.perf_counter() - t1)
    dt = sorted(e2e)[1]
    frames = n_streams * g.in_frames
    out = {"value": round(frames / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
           "pcie_gbps": round((g.src.nbytes + g.dst_bytes) / dt / 1e9, 2),
           "what": "pinned host input -> H2D -> launch -> D2H -> pinned host output, median of 3"}
    groups = int(os.environ.get("OHGPU_BENCH_GROUPS", "8"))
    if n_streams % groups == 0 and groups > 1:
        per = n_streams // groups
        sb, db = g.in_frames * g.fb_src, g.out_total * g.fb_dst
        parts = [ctx.src_batch(g.h, g.descs[k * per * g.n_msgs:(k + 1) * per * g.n_msgs], g.src_bytes, g.dst_bytes) for k in range(groups)]
        up, down = ctx.stream_create(), ctx.stream_create()
        arrived = [ctx.event() for _ in range(groups)]
        ov = []
        for _ in range(3):
            ctx.sync()
            t1 = time.perf_counter()
            for k in range(groups):
                s0, s1, o0, o1 = k * per * sb, (k + 1) * per * sb, k * per * db, (k + 1) * per * db
                ctx.copy_h2d(C.c_void_p(g.d_src.value + s0), h_src[s0:s1], up)
                ctx.record(arrived[k], up)
                ctx.wait_event(down, arrived[k])
                ctx.src_run(parts[k], g.d_src, g.d_dst, down)
                ctx.copy_d2h(h_dst[o0:o1], C.c_void_p(g.d_dst.value + o0), down)
            ctx.sync(up)
            ctx.sync(down)
            ov.append(time.perf_counter() - t1)
        dt = sorted(ov)[1]
        out["overlapped"] = {"value": round(frames / dt / 1e6, 3), "ms_per_step": round(dt * 1e3, 3),
                             "pcie_gbps": round((g.src.nbytes + g.dst_bytes) / dt / 1e9, 2),
                             "what": f"{groups} groups of {per} streams, upload stream + launch/download stream",
                             "check": "bit-exact vs the resident run" if np.array_equal(np.array(h_dst), ctx.download(g.d_dst, g.dst_bytes)) else "MISMATCH"}
        for st in (up, down):
            ctx.stream_destroy(st)
        for b in parts:
            ctx.batch_destroy(b)
    ctx.free_host(h_src)
    ctx.free_host(h_dst)
    return out


def source_fingerprint():
    """sha256 over the sources of the kernel the headline times (what a stored profile must have been taken from)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("src_mfma_wg_kernel.hip", "src_mfma_kernel.hip", "src_mfma_common.h", "src_lean_kernel.hip", "src_block_common.h", "src_plan.cpp", "ohgpu_internal.h"):
        with open(os.path.join(ROOT, "ohpipeline_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cgroup_quota_cpus():
    """The container's CPU quota in CPUs (cgroup v2 cpu.max, v1 cfs quota), or None when there is none."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return max(1, int(round(q / per))) if q > 0 and per > 0 else None
    except Exception:
        return None


def parse_cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def host_share(local_world, pci_bus_id=None, sysfs="/sys/bus/pci/devices"):
    """What one of `local_world` ranks on this host may use of it: `threads` = the CPUs the process may keep busy (its affinity mask,
    capped by the container's CPU quota) divided among the ranks -- planner threads, noise / decoder pools -- and `cpus` = the CPUs
    nearest the rank's GPU (the PCI device's local_cpulist) among those it may run on, or None when the topology cannot be read or
    leaves nothing (then the rank stays where it is)."""
    allowed = set(os.sched_getaffinity(0))
    budget = len(allowed)
    quota = cgroup_quota_cpus()
    if quota is not None:
        budget = min(budget, quota)
    threads = max(1, budget // max(1, local_world))
    cpus = None
    if pci_bus_id:
        try:
            near = parse_cpulist(open(os.path.join(sysfs, pci_bus_id, "local_cpulist")).read()) & allowed
            if near and near != allowed:
                cpus = sorted(near)
        except Exception:
            cpus = None
    return {"threads": threads, "cpus": cpus, "cpu_budget": budget, "quota_cpus": quota, "local_world": local_world}


def slowest_rank(dist, *figures):
    """Host-side figures of a multi-rank job (plan_ms, plan_ms_first, config 5's decode seconds): the job's is the slowest rank's."""
    import torch
    t = torch.tensor([float(f) for f in figures], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: start the N ranks as a CHILD process -- before this
    process has imported the C ABI or touched HIP, and never by exec -- relay rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1])
    sys.exit(proc.returncode if proc.returncode != 0 or lines else 1)


def measure(capi, ctx, args, rank, world, dist, light=False):
    """One workload (args.config) on this rank's share: sustain, warm-up, `steps` timed steps between barriers, the result line
    as a dict (rank 0; None elsewhere) and whether every check passed.  `light` (the extra configs of the default line): no
    cadence / end_to_end legs, one CPU pass."""
    flac = None
    if args.config == 5:
        import bench_flac
        groups, scaling, flac = bench_flac.build(capi, ctx, args, rank, world, Group)
    else:
        groups, scaling = build_groups(capi, args, rank, world)
    for g in groups:
        g.attach(ctx)
    ctx.sync()

    def step(events=None):
        if flac is not None:
            flac.run(ctx, events[-1] if events is not None else None)
        for i, g in enumerate(groups):
            # (the timed steps' events ride on the launches themselves -- ohgpu_src_batch_run_timed: the dispatch's own timestamps; two
            # event records around every launch cost back-to-back launches 5 us apiece, 1.7 % of the headline's step)
            ctx.src_run(g.batch, g.d_src, g.d_dst, events=events[i] if events is not None else None)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # steady state first: about `sustain` seconds of launches (untimed), so that the timed steps see the clock the chip holds
    # (by the clock, in bursts: the first step of a process is no measure of the others -- under a profiler it is ten times as long)
    step()
    ctx.sync()
    t0 = time.perf_counter()
    launched = 0
    while time.perf_counter() - t0 < args.sustain and launched < 200000:
        for _ in range(16):
            step()
        ctx.sync()
        launched += 16
    for _ in range(args.warmup):
        step()
    barrier()
    n_ev = len(groups) + (1 if flac is not None else 0)
    ev = [[(ctx.event(), ctx.event()) for _ in range(n_ev)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    ctx.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    frames_all = float(sum(len(g.stream_ids) * g.in_frames for g in groups))
    subs_all = float(sum(len(g.stream_ids) * g.in_frames * g.channels for g in groups))
    checks_ok = True
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([frames_all, subs_all], dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        frames_all, subs_all = float(tot[0]), float(tot[1])
        # a launch waits for the slowest rank's plan, and config 5's step for the slowest rank's decode: the maxima are the job's
        plan_ms_max, plan_ms_first_max, decode_s_max = slowest_rank(dist, float(sum(g.plan_ms for g in groups)), float(sum(g.plan_ms_first for g in groups)),
                                                                    flac.decode_s if flac is not None else 0.0)
        if not args.no_cpu:
            first = [ctx.download(g.d_dst, g.out_total * g.fb_dst) for g in groups]
            okt = torch.tensor([1.0 if rank_check(groups, first) else 0.0], dtype=torch.float64)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            checks_ok = bool(okt.item() > 0.5)

    # per-launch kernel time (HIP events on the launch stream), this rank; and the shader clock the chip holds right behind the
    # timed launches (a 0.2 ms probe kernel, queued before anything lets the chip idle): a slow box shows in the clock or, the clock
    # being the same, in the memory system -- either way not in the build
    try:
        clock_mhz = round(ctx.shader_clock_mhz(), 1)
    except Exception:
        clock_mhz = None
    per_step = np.array([[ctx.elapsed_ms(ev[k][i][0], ev[k][i][1]) for i in range(n_ev)] for k in range(args.steps)], dtype=np.float64)
    grp_ms = [float(per_step[:, i].mean()) for i in range(n_ev)]
    step_kernel_ms = per_step[:, :len(groups)].sum(axis=1)             # per timed step: its launches' kernel time
    for row in ev:
        for pair in row:
            for e in pair:
                ctx.event_destroy(e)
    kernel_ms = float(sum(grp_ms[:len(groups)]))
    alg_bytes = float(sum(g.algorithmic_bytes for g in groups))
    flops = float(sum(g.flops for g in groups))
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    plan_ms = float(sum(g.plan_ms for g in groups))
    plan_ms_first = float(sum(g.plan_ms_first for g in groups))
    if dist is not None:
        plan_ms, plan_ms_first = plan_ms_max, plan_ms_first_max       # (the slowest rank's)

    result = None
    if rank == 0:
        names = {3: "configs[2]", 4: "configs[3]", 5: "configs[4]"}
        head = groups[int(np.argmax([g.algorithmic_bytes for g in groups]))]
        if args.config == 3:
            what = (f"{args.streams} independent {'stereo' if args.channels == 2 else str(args.channels) + '-channel'} S{getattr(args, 'src_bits', BITS)}LE streams per GPU, "
                    f"{args.rate_in / 1000:g}->48 kHz")
        elif args.config == 4:
            what = (f"{args.streams} S24LE streams in all, by stream id 44.1 / 96 kHz x 2 / 6 / 8 channels, ->48 kHz, "
                    f"contiguous blocks of streams per rank balanced by bytes")
        else:
            what = (f"{args.streams} stereo FLAC streams per GPU (16- and 24-bit, level 5), frames decoded on the host by the reference's libFLAC, "
                    f"the decoder's planar TInt32 output read by the resampler itself (CodecFlac::CallbackWrite's pack, Flac.cpp:379-417, "
                    f"fused into its load) -> 44.1->48 kHz")
        result = {
            "metric": "PCM Msamples/s, 256-stream 44.1->48k S24 resample+ramp+fmt" if args.config == 3 else
                      ("PCM Msamples/s, 2048-stream mixed 44.1/96->48k 2/6/8-channel S24 resample+ramp+fmt" if args.config == 4 else
                       "PCM Msamples/s, 256-stream FLAC frames -> pack -> 44.1->48k resample+ramp+fmt (one pass)"),
            "value": round(frames_all * args.steps / elapsed / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "plan_ms": round(plan_ms, 3),                    # ohgpu_src_batch_create for the step's batches (median of three creations): once, reused by every launch
            "plan_ms_first": round(plan_ms_first, 3),        # ... and the process's first creation of them
            # ... and what the same batch costs for the next period of the same streams (ohgpu_src_batch_advance + _set_ramps)
            "plan_ms_repeat": (round(float(sum(g.plan_ms_repeat for g in groups)), 3) if all(g.plan_ms_repeat is not None for g in groups) else None),
            "msubsamples_per_s": round(subs_all * args.steps / elapsed / 1e6, 3),   # frames x channels
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            # exact integer sums either way: 24-bit samples x Q28 coefficients as int8 digits accumulated in int32 (matrix-pipe
            # kernels), or as integer-valued doubles (lean kernel)
            "dtype": "int32" if all("mfma" in ctx.src_kernel_name(g.batch) for g in groups) else "f64",
            "data": "synthetic",
            "config": {"workload": f"{names[args.config]}: {what}, {args.seconds:g} s each, 5 ms output messages, ramp up 50 ms / down 500 ms, S24 BE out",
                       "kernel_variant": args.variant, "sustain_s": args.sustain,
                       "streams_per_gpu": len(set(s for g in groups for s in g.stream_ids)),
                       "groups": [{"rate_in": g.rate_in, "channels": g.channels, "src_bits": g.src_bits, "streams": len(g.stream_ids),
                                   "taps_per_phase": g.taps, "frames_per_stream": g.in_frames, "msgs": int(g.info["n_msgs"]),
                                   "kernel_ms": round(grp_ms[i], 4), "gbps": round(g.algorithmic_bytes / (grp_ms[i] * 1e-3) / 1e9, 1),
                                   "plan_ms": round(g.plan_ms, 3),
                                   "block_kernel_out_frames": g.plan["block_kernel_out_frames"], "generic_pieces": g.plan["generic_pieces"]}
                                  for i, g in enumerate(groups)],
                       "sharding": f"streams over {world} rank(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "kernel": "fused resample+ramp+pack: " + ", ".join(sorted(set(ctx.src_kernel_name(g.batch) for g in groups)))
                                   + f" (kernel variant {args.variant}), every launch of the step",
                         "kernel_avg_ms": round(kernel_ms, 4),
                         "kernel_min_ms": round(float(step_kernel_ms.min()), 4), "kernel_median_ms": round(float(np.median(step_kernel_ms)), 4),
                         "shader_clock_mhz": clock_mhz,
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "bytes_per_input_frame": round(head.fb_src + head.fb_dst * head.L / head.M, 4),
                         # the taps as arithmetic: 2*T*channels multiply-adds' worth of flop per output frame -- on the
                         # fp64 vector pipe (78.6 TFLOP/s dense on MI355X) in the lean kernel; the matrix-pipe kernels do the same sums
                         # exactly in int8 digits (twelve 16x16x64 int8 MFMAs per 256 output subsamples, half of each band matrix zeros)
                         "fp64_tflops": round(flops / (kernel_ms * 1e-3) / 1e12, 2), "fp64_peak_tflops": 78.6,
                         "fp64_frac": round(flops / (kernel_ms * 1e-3) / 1e12 / 78.6, 4)},
        }
        if flac is not None:
            result["config"]["flac"] = flac.report(grp_ms[-1])
            if dist is not None:
                result["config"]["flac"]["host_decode"]["seconds_max_over_ranks"] = round(decode_s_max, 3)
        share = getattr(args, "host_share", None)
        if share is not None:
            result["host"] = {"plan_threads": min(16, share["threads"]) if world > 1 else "library default (<= 16, <= the CPU quota)",
                              "feeder_threads": share["threads"] if world > 1 else None, "ranks_on_host": share["local_world"],
                              "cpu_budget": share["cpu_budget"], "cgroup_quota_cpus": share["quota_cpus"],
                              "pinned_to_cpus_near_gpu": (f"{len(share['cpus'])} CPUs" if (world > 1 and share["cpus"]) else None),
                              "plan_ms": "max over ranks" if world > 1 else "this rank"}
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc) and args.config == 3 and world == 1:
            # `traffic` is a counter measurement of THIS workload taken in separate --pmc passes (tools/profile_bench.sh), not of
            # this run: it is quoted only while the kernel's sources are the ones that profile was taken from
            try:
                p = json.load(open(pmc))
                if p.get("streams_per_gpu") == args.streams and p.get("frames_per_stream") == groups[0].in_frames \
                        and p.get("kernel_variant") == args.variant and p.get("source_sha16") == source_fingerprint():
                    result["roofline"]["traffic"] = p.get("hbm_bytes_per_launch")
                    result["roofline"]["traffic_source"] = "stored profile, not this run: " + str(p.get("source"))
            except Exception:
                pass
        if world == 1 and not args.no_cpu:
            got = [ctx.download(g.d_dst, g.dst_bytes) for g in groups]
            if not light:
                try:                                         # (reported extras: never let them cost the headline line)
                    if args.config == 3:
                        result["end_to_end"] = end_to_end(ctx, capi, groups[0])
                        result["end_to_end"]["plan_ms"] = round(plan_ms, 3)
                        result["end_to_end"]["ms_per_step_with_plan"] = round(result["end_to_end"]["ms_per_step"] + plan_ms, 3)
                    result["cadence"] = None if head.planar else cadence(ctx, capi, head)    # (the live regime is measured on the packed layouts)
                except Exception as e:
                    result["extras_error"] = f"{type(e).__name__}: {e}"
            base, check = cpu_baseline(groups, got, light=light)
            result["cpu_baseline"] = base
            result["check"] = check
            checks_ok = checks_ok and check == "bit-exact vs oracle"
            if flac is not None:
                result["check_flac_decode"] = flac.check(ctx)
                checks_ok = checks_ok and result["check_flac_decode"].startswith("lossless")
        else:
            result["cpu_baseline"] = None
            if world > 1 and not args.no_cpu:
                result["check"] = ("bit-exact vs oracle: first stream of every group on every rank" if checks_ok else "MISMATCH")
    for g in groups:
        g.detach(ctx)
    if flac is not None:
        flac.close(ctx)
    return result, checks_ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5), help="BASELINE.json configs[config-1]: 3 = the headline")
    ap.add_argument("--seconds", type=float, default=None, help="audio per stream (default: 10 s)")
    ap.add_argument("--streams", type=int, default=None, help="config 3/5: streams per GPU (256); config 4: streams in all (2048)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 tuned, 1 generic v1, 4 round 2's lean kernel; 2 and 5 -- round 1's block kernel, round 4's unit-per-wave matrix kernel -- in a legacy build only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / check, end_to_end and cadence legs (profiling runs)")
    ap.add_argument("--no-extra-configs", action="store_true", help="the default line without its `configs` object (configs[3] and configs[4] at full size)")
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of back-to-back launches before the warm-up and the timed steps, so that they see the clock the chip holds under this load")
    ap.add_argument("--channels", type=int, default=2, help="config 3: channels per stream (the headline is stereo)")
    ap.add_argument("--rate-in", type=int, default=44100, help="config 3: input rate (the headline is 44100)")
    ap.add_argument("--src-bits", type=int, default=BITS, choices=(16, 24), help="config 3: source depth (the headline is 24; 16 = CD audio into the same S24 output)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                                     # (does not return; nothing of HIP or the C ABI has been loaded yet)
    explicit = args.seconds is not None or args.streams is not None
    if args.seconds is None:
        args.seconds = 10.0                                    # (every configuration: 10 s of audio per stream)
    if args.streams is None:
        args.streams = 2048 if args.config == 4 else 256

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch  # noqa: F401  (torch.distributed is plumbing: barrier + max over ranks)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)

    from ohpipeline_amd import capi
    ctx = capi.Context(local_rank % max(capi.device_count(), 1) if world > 1 else 0)   # fewer GPUs than ranks (a rehearsal): shared
    ctx.set_kernel_variant(args.variant)
    # One rank per GPU, several ranks per host: each takes its share of the CPUs the container grants -- the planner's pool
    # (ohgpu_set_plan_threads), the noise generator's and the FLAC decoder's -- and stays near its GPU where the topology says where
    # that is.  Eight ranks that each start sixteen planner threads and thirty-two feeders on a sixteen-CPU grant would only queue.
    global HOST_THREADS
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    try:
        bus = ctx.pci_bus_id()
    except Exception:
        bus = None
    share = host_share(local_world, bus)
    if world > 1:
        if share["cpus"]:
            try:
                os.sched_setaffinity(0, share["cpus"])
            except OSError:
                share["cpus"] = None
        HOST_THREADS = share["threads"]
        capi.set_plan_threads(min(16, share["threads"]))
    args.host_share = share
    result, ok = measure(capi, ctx, args, rank, world, dist)
    # The default line (config 3, one GPU, sizes untouched) also carries BASELINE configs[3] and configs[4], at their full size
    # (2048 streams x 10 s; 256 FLAC streams x 10 s), same process, each with its own sustain phase, steps and whole-step check.
    if rank == 0 and world == 1 and args.config == 3 and not explicit and not args.no_extra_configs and not args.no_cpu \
            and args.channels == 2 and args.rate_in == 44100 and args.src_bits == BITS:
        result["configs"] = {}
        for cfg in (4, 5):
            t0 = time.perf_counter()
            try:
                sub = argparse.Namespace(**vars(args))
                sub.config, sub.streams, sub.sustain = cfg, (2048 if cfg == 4 else 256), 0.5
                r, sub_ok = measure(capi, ctx, sub, rank, world, dist, light=True)
                ok = ok and sub_ok
                entry = {"metric": r["metric"], "workload": r["config"]["workload"], "value": r["value"], "ms_per_step": r["ms_per_step"],
                         "kernel_avg_ms": r["roofline"]["kernel_avg_ms"], "plan_ms": r["plan_ms"], "plan_ms_first": r["plan_ms_first"], "plan_ms_repeat": r.get("plan_ms_repeat"), "frac": r["roofline"]["frac"],
                         "fp64_frac": r["roofline"]["fp64_frac"], "achieved_gbps": r["roofline"]["achieved"], "scaling": r["scaling"],
                         "check": r.get("check"), "cpu_baseline": {k: r["cpu_baseline"][k] for k in ("value", "cores")},
                         "groups": [{k: g[k] for k in ("rate_in", "channels", "src_bits", "streams", "taps_per_phase", "kernel_ms", "gbps", "generic_pieces")}
                                    for g in r["config"]["groups"]]}
                if "check_flac_decode" in r:
                    entry["check_flac_decode"] = r["check_flac_decode"]
                    entry["host_decode"] = r["config"]["flac"]["host_decode"]
            except Exception as e:                             # (an extra config never costs the headline line; a failed CHECK does)
                entry = {"error": f"{type(e).__name__}: {e}"}
            entry["wall_s"] = round(time.perf_counter() - t0, 1)
            result["configs"][str(cfg)] = entry
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not ok:
            result["value"] = None                             # a fast result that differs from the oracle's is not a result
        print(json.dumps(result))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
