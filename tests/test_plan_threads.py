"""The resampler's host planner cuts a batch's streams into work units on several threads (ohpipeline_amd/csrc/src_plan.cpp,
ohgpu_api.hip: the per-message checks, the ordering test, the segments).  The plan must not depend on how many: this compares
`ohgpu_src_plan_digest` -- the unit list, ramp jobs and generic-kernel pieces hashed on the host, no device -- across thread
counts, message orders and kernel variants.  CPU only."""
import numpy as np
import pytest

import oracle_lib as O
import workloads as W
from ohpipeline_amd import capi


def headline_like(n_streams, seconds, channels=2, rate_in=44100):
    ref = O.Src(rate_in, 48000)
    in_frames = int(seconds * rate_in)
    out_total = (in_frames * ref.L + ref.M - 1) // ref.M
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * 56448, 500 * 56448)
    d, sb, db, _, _ = W.src_stream_descs(n_streams, in_frames, ref.L, ref.M, 240, channels, 24, O.ENDIAN_LITTLE, 24, O.ENDIAN_BIG,
                                         schedule=sched, dtype=capi.SRC_MSG_DESC)
    return ref, d, sb, db


@pytest.fixture(autouse=True)
def _no_cap():
    yield
    capi.set_plan_threads(0)


def test_the_plan_does_not_depend_on_the_thread_count():
    ref, d, sb, db = headline_like(96, 4.0)                 # ~77 000 messages: enough for several ranges of every parallel pass
    assert d.size > 70000
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    assert one["units"] > 0 and one["kernel"] == 3 and one["generic_pieces"] == 0
    for threads in (2, 3, 5, 16, 0):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db) == one, threads


def test_messages_in_any_order_make_the_same_plan():
    ref, d, sb, db = headline_like(40, 3.0)
    capi.set_plan_threads(1)
    want = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    rng = np.random.default_rng(11)
    shuffled = d[rng.permutation(d.size)]
    for threads in (1, 4, 0):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, shuffled, sb, db) == want, threads


@pytest.mark.parametrize("variant,kernel", [(0, 3), (5, 1), (4, 1), (3, 1), (2, 1)])      # (3 = the workgroup matrix kernel, 1 = the lean kernel: the two the library ships)
def test_every_variants_plan_is_thread_independent(variant, kernel):
    ref, d, sb, db = headline_like(64, 3.0)
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, variant)
    assert one["kernel"] == kernel
    capi.set_plan_threads(7)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, variant) == one


def test_gaps_and_empty_messages_cut_the_segments_where_one_thread_cuts_them():
    ref, d, sb, db = headline_like(48, 3.0)
    d = d.copy()
    # a hole in every third stream (a message dropped) and an empty message in every fifth
    n_msgs = d.size // 48
    keep = np.ones(d.size, dtype=bool)
    for s in range(0, 48, 3):
        keep[s * n_msgs + n_msgs // 2] = False
    for s in range(1, 48, 5):
        d["n_frames"][s * n_msgs + n_msgs // 3] = 0
    d = d[keep]
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    assert one["generic_pieces"] > 0
    for threads in (2, 6, 16):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db) == one, threads


def test_a_bad_descriptor_is_reported_by_its_index_whatever_thread_finds_it():
    ref, d, sb, db = headline_like(64, 3.0)
    d = d.copy()
    bad = d.size * 3 // 4 + 5
    d["channels"][bad] = 9
    d["channels"][bad + 1000] = 0
    for threads in (1, 8):
        capi.set_plan_threads(threads)
        with pytest.raises(capi.OhGpuError) as e:
            capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
        assert f"src desc {bad}:" in str(e.value)


@pytest.mark.parametrize("channels, unit_rows", [(6, 30), (8, 32)])
def test_wide_plans_for_the_workgroup_kernel_are_thread_independent_and_cut_long_units(channels, unit_rows):
    """Six and eight channels: the workgroup kernel's units are 30 / 32 rows long (src_plan.cpp), a third to a quarter of the lean
    kernel's count, whatever the thread count; the lean kernel's plan (variant 4) keeps its own rows."""
    ref, d, sb, db = headline_like(24, 3.0, channels=channels)
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    assert one["kernel"] == 3 and one["generic_pieces"] == 0
    blocks = 24 * ((3 * 48000) // 160)                     # whole 160-output blocks of the batch
    assert abs(one["units"] - blocks / unit_rows) <= 24 * 2 + 1, (one, blocks)
    for threads in (3, 8, 0):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db) == one, threads
    lean = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, 4)
    assert lean["kernel"] == 1 and lean["units"] < blocks / (64 // channels) + 24 * 2 + 1


@pytest.mark.parametrize("channels", [2, 6, 8])
def test_the_half_band_plans_that_ship_are_thread_independent(channels):
    """96 -> 48 kHz (config 4's second half): with the filter's coefficients the digest plans as ohgpu_src_create's filter would be
    planned for -- the half-band tables, the workgroup kernel's 128-output blocks, 30 / 32-row units for six and eight channels --
    and not, as without them, for a plain 64-tap filter (round 4's digest never planned these: the advisor's finding)."""
    ref = O.Src(96000, 48000, 64)
    in_frames = 3 * 96000
    out_total = (in_frames * ref.L + ref.M - 1) // ref.M
    n_msgs = (out_total + 239) // 240
    sched = W.ramp_schedule(n_msgs, 240 * 1176, 50 * 56448, 500 * 56448)
    d, sb, db, _, _ = W.src_stream_descs(24, in_frames, ref.L, ref.M, 240, channels, 24, O.ENDIAN_LITTLE, 24, O.ENDIAN_BIG,
                                         schedule=sched, dtype=capi.SRC_MSG_DESC)
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, coef_q28=ref.coef_q28)
    assert one["kernel"] == 3, one                               # src_mfma_wg_kernel, half-band form
    blocks = 24 * (out_total // 128)
    rows = {2: 32, 6: 30, 8: 32}[channels]
    assert abs(one["units"] - blocks / rows) <= 24 * 2 + 1, (one, blocks)
    plain = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)   # the same messages, no coefficients: another filter, another plan
    assert plain["kernel"] != 3 and plain["digest"] != one["digest"]
    for threads in (2, 5, 16, 0):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, coef_q28=ref.coef_q28) == one, threads
    # a 64-tap table that is NOT half-band keeps the plain kernels (and says so)
    broken = np.array(ref.coef_q28, dtype=np.int32)
    broken[5] = 777
    capi.set_plan_threads(1)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, coef_q28=broken)["kernel"] != 3


def test_the_long_unit_schedule_follows_the_device_size():
    """Variant 4's lean plan cuts one long unit per wave: the number of waves is the CU count's, which the digest now takes."""
    ref, d, sb, db = headline_like(96, 4.0)
    capi.set_plan_threads(1)
    big = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, 4, coef_q28=ref.coef_q28, num_cus=256)
    small = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, 4, coef_q28=ref.coef_q28, num_cus=64)
    assert big["digest"] != small["digest"]
    capi.set_plan_threads(6)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, 4, coef_q28=ref.coef_q28, num_cus=64) == small


def test_a_period_a_whole_number_of_blocks_later_has_the_same_plan():
    """ohgpu_src_batch_advance's ground: nothing of a plan names an absolute position -- a unit is where its rows lie in the arenas, a
    ramp job where its frames lie in its message, a generic-kernel piece its window relative to the buffer -- so the messages of the
    next period (every out_frame0 a whole number of blocks on, every src_frame0 the same number of input blocks on, offsets as
    they were) plan to the very same arrays.  Not so for a period that starts its streams (block 0's rows read zeros in front)."""
    ref, d, sb, db = headline_like(48, 2.0)
    L_blk, M_blk = 160, 147
    assert ref.L == 160 and ref.M == 147

    def shifted(blocks, history_blocks=1):
        # every stream's period: the same messages `blocks` blocks on; the buffer holds one block of history in front of what it held
        s = d.copy()
        s["out_frame0"] += blocks * L_blk
        s["src_frame0"] += (blocks - history_blocks) * M_blk
        s["src_frames"] += history_blocks * M_blk
        per = int(s["src_frames"][0]) * 6
        n_msgs = d.size // 48
        for k in range(48):
            s["src_offset"][k * n_msgs:(k + 1) * n_msgs] = k * per
        return s, 48 * per

    capi.set_plan_threads(1)
    a, sba = shifted(10)
    b_, sbb = shifted(27)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, a, sba, db, coef_q28=ref.coef_q28)
    assert one["units"] > 0 and one["kernel"] == 3
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, b_, sbb, db, coef_q28=ref.coef_q28) == one
    capi.set_plan_threads(5)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, b_, sbb, db, coef_q28=ref.coef_q28) == one
    # ... other ramp endpoints change the jobs' endpoints and nothing else: the same units, the same pieces
    c = b_.copy()
    c["ramp_start"] = np.where(c["flags"] & 1, 9000, c["ramp_start"])
    other = capi.src_plan_digest(ref.L, ref.M, ref.T, c, sbb, db, coef_q28=ref.coef_q28)
    assert other["units"] == one["units"] and other["generic_pieces"] == one["generic_pieces"] and other["digest"] != one["digest"]
    # the streams' own first period is another plan (kWorkFirst)
    capi.set_plan_threads(1)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db, coef_q28=ref.coef_q28)["digest"] != one["digest"]


def test_the_pool_survives_many_short_jobs_from_two_callers():
    """The pool's threads claim a job's ranges as they wake (csrc/ohgpu_api.hip PlanPool): a helper that wakes after its job has
    finished must not touch it, and two callers take turns.  Many short plans from two threads at once, every digest the same."""
    import threading
    ref, d, sb, db = headline_like(128, 6.0)                # ~154 000 messages: several threads' worth for every pass
    assert d.size > 140000
    capi.set_plan_threads(1)
    want = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    capi.set_plan_threads(16)
    bad = []

    def caller():
        for _ in range(60):
            if capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db) != want:
                bad.append(1)

    threads = [threading.Thread(target=caller) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad


DEFECTS = [
    ("ramp_start", lambda d, k: d["ramp_start"].__setitem__(k, 0x4001)),
    ("ramped message too long", lambda d, k: (d["flags"].__setitem__(k, d["flags"][k] | capi.FLAG_RAMP), d["n_frames"].__setitem__(k, 131072))),
    ("attenuation", lambda d, k: d["attenuation"].__setitem__(k, 255)),
    ("flag bits", lambda d, k: d["flags"].__setitem__(k, 0x80)),
    ("plane stride without the flag", lambda d, k: d["src_plane_stride"].__setitem__(k, 4096)),
    ("frame index out of range", lambda d, k: d["out_frame0"].__setitem__(k, (1 << 48) + 1)),
    ("history missing", lambda d, k: d["src_frame0"].__setitem__(k, int(d["out_frame0"][k]) * 147 // 160 + 1)),
    ("window too short", lambda d, k: d["src_frames"].__setitem__(k, 8)),
    ("source beyond the arena", lambda d, k: d["src_offset"].__setitem__(k, (1 << 40))),
    ("destination beyond the arena", lambda d, k: d["dst_offset"].__setitem__(k, (1 << 40))),
    ("bit depth", lambda d, k: d["src_bits"].__setitem__(k, 12)),
    ("endian", lambda d, k: d["dst_endian"].__setitem__(k, 7)),
]


@pytest.mark.parametrize("what, spoil", DEFECTS, ids=[w for w, _ in DEFECTS])
def test_every_kind_of_bad_descriptor_is_the_same_error_on_both_routes(what, spoil):
    """A batch of 4096 messages or more is checked by the planner's own pass, the usual message in line (SrcQuickCheck) and anything
    else by src_check_range; a smaller one by src_check_range alone, ahead of the planner.  The same defect must be the same
    error -- code and text -- on both routes, named by the message's index, on one thread and on several."""
    ref, d, sb, db = headline_like(64, 3.0)                 # ~38 000 messages: the fused route
    n_msgs = d.size // 64
    small = d[:3 * n_msgs].copy()[:3000]                    # under 4096: the two-pass route (whole messages of the first streams)
    k_small = 1234
    spoil(small, k_small)
    with pytest.raises(capi.OhGpuError) as e_small:
        capi.src_plan_digest(ref.L, ref.M, ref.T, small, sb, db)
    text_small = str(e_small.value)
    assert f"src desc {k_small}:" in text_small
    for threads in (1, 6):
        capi.set_plan_threads(threads)
        for k_big in (k_small, d.size * 2 // 3 + 17):
            big = d.copy()
            spoil(big, k_big)
            with pytest.raises(capi.OhGpuError) as e_big:
                capi.src_plan_digest(ref.L, ref.M, ref.T, big, sb, db)
            assert e_big.value.code == e_small.value.code, (what, threads)
            if k_big == k_small:
                assert str(e_big.value) == text_small, (what, threads)
            else:
                assert f"src desc {k_big}:" in str(e_big.value), (what, threads, str(e_big.value))


def test_a_message_of_another_layout_or_out_of_order_sends_the_batch_the_two_pass_way_with_the_same_plan():
    """Not errors: a message whose layout differs (the batch is then planned per layout) and messages out of the planner's order (it
    sorts).  The fused route must notice both and hand over; the digests are what the two-pass route makes of the same batch."""
    ref, d, sb, db = headline_like(64, 3.0)
    n_msgs = d.size // 64
    capi.set_plan_threads(1)
    want = capi.src_plan_digest(ref.L, ref.M, ref.T, d, sb, db)
    # two streams' messages swapped wholesale: the same plan as in order
    swapped = d.copy()
    swapped[5 * n_msgs:6 * n_msgs], swapped[40 * n_msgs:41 * n_msgs] = d[40 * n_msgs:41 * n_msgs].copy(), d[5 * n_msgs:6 * n_msgs].copy()
    for threads in (1, 5):
        capi.set_plan_threads(threads)
        assert capi.src_plan_digest(ref.L, ref.M, ref.T, swapped, sb, db) == want
    # one stream little-endian out: two layouts, no error, the same on any thread count
    mixed = d.copy()
    mixed["dst_endian"][20 * n_msgs:21 * n_msgs] = capi.ENDIAN_LITTLE
    capi.set_plan_threads(1)
    one = capi.src_plan_digest(ref.L, ref.M, ref.T, mixed, sb, db)
    capi.set_plan_threads(7)
    assert capi.src_plan_digest(ref.L, ref.M, ref.T, mixed, sb, db) == one
