"""The host adapter's files against the reference files whose behaviour they answer to: normalised line overlap
(tools/overlap.py: comments and whitespace stripped, lines of fewer than 8 characters ignored) must stay below 20 % for the
files that are this repository's own design.  Round 1 shipped two that were 54 % and 61 % the reference's lines; they are
gone.  Runs where the reference tree exists (the build container), skipped elsewhere."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/OpenHome"
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")

OWN_DESIGN = [
    ("ohpipeline_amd/host/StarvationManager.cpp", ["Media/Pipeline/StarvationRamper.cpp"]),
    ("ohpipeline_amd/host/StarvationManager.h", ["Media/Pipeline/StarvationRamper.h"]),
    ("ohpipeline_amd/host/StarvationRamper.h", ["Media/Pipeline/StarvationRamper.h"]),
    ("ohpipeline_amd/host/Sender.cpp", ["Av/Songcast/Sender.cpp", "Av/Songcast/OhmSender.cpp"]),
    ("ohpipeline_amd/host/SampleRateConverter.cpp", ["Media/Pipeline/Ramper.cpp", "Media/Pipeline/PreDriver.cpp", "Media/Pipeline/StarvationRamper.cpp"]),
    ("ohpipeline_amd/host/DecodedAudioAggregator.cpp", ["Media/Pipeline/DecodedAudioAggregator.cpp"]),
    ("ohpipeline_amd/host/FlywheelRamper.cpp", ["Media/FlywheelRamper.cpp"]),
    ("tests/cpp/test_host.cpp", ["Media/Tests/TestStarvationRamper.cpp", "Media/Pipeline/Ramper.cpp", "Media/Pipeline/PreDriver.cpp"]),
]


# The files that MIRROR a reference interface so that the adapter can stand where the reference classes stand (same class
# names, method signatures and member names -- INTEGRATION.md 2): their declarations and signature lines cannot differ, their
# bodies can and must.  Each has a cap a little above what its signatures alone cost, so that a body copied from the reference
# shows: round 2's Msg.cpp held five of the reference's functions line for line (24 %), now restated (cap 15 %).
MIRRORS = [
    ("ohpipeline_amd/host/Msg.cpp", ["Media/Pipeline/Msg.cpp", "Media/Pipeline/Msg.h"], 0.15),
    ("ohpipeline_amd/host/Msg.h", ["Media/Pipeline/Msg.cpp", "Media/Pipeline/Msg.h"], 0.40),
    ("ohpipeline_amd/host/Ramp.cpp", ["Media/Pipeline/Msg.cpp", "Media/Pipeline/Msg.h"], 0.42),
    ("ohpipeline_amd/host/Ramp.h", ["Media/Pipeline/Msg.cpp", "Media/Pipeline/Msg.h"], 0.65),
    ("ohpipeline_amd/host/DecodedAudioAggregator.cpp", ["Media/Pipeline/DecodedAudioAggregator.cpp", "Media/Codec/CodecController.cpp"], 0.30),
]


def overlap_share(mine, theirs):
    import overlap
    own = overlap.significant(os.path.join(ROOT, mine))
    ref = set()
    for t in theirs:
        ref.update(overlap.significant(os.path.join(REF, t)))
    return sum(1 for l in own if l in ref) / max(1, len(own))


@pytest.mark.parametrize("mine,theirs", OWN_DESIGN, ids=[m for m, _ in OWN_DESIGN])
def test_overlap_with_the_reference_stays_low(mine, theirs):
    share = overlap_share(mine, theirs)
    assert share < 0.20, f"{mine}: {100 * share:.1f} % of its significant lines are in {theirs}"


@pytest.mark.parametrize("mine,theirs,cap", MIRRORS, ids=[m for m, _, _ in MIRRORS])
def test_interface_mirrors_share_signatures_not_bodies(mine, theirs, cap):
    share = overlap_share(mine, theirs)
    assert share < cap, f"{mine}: {100 * share:.1f} % of its significant lines are in {theirs} (cap {100 * cap:.0f} %)"


def test_the_deleted_mirrors_stay_deleted():
    for gone in ("Elements.cpp", "Elements.h", "RampGenerator.cpp", "RampGenerator.h", "StarvationRamper.cpp"):
        assert not os.path.exists(os.path.join(ROOT, "ohpipeline_amd", "host", gone)), gone
