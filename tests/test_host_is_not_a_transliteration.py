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


@pytest.mark.parametrize("mine,theirs", OWN_DESIGN, ids=[m for m, _ in OWN_DESIGN])
def test_overlap_with_the_reference_stays_low(mine, theirs):
    import overlap
    own = overlap.significant(os.path.join(ROOT, mine))
    ref = set()
    for t in theirs:
        ref.update(overlap.significant(os.path.join(REF, t)))
    share = sum(1 for l in own if l in ref) / max(1, len(own))
    assert share < 0.20, f"{mine}: {100 * share:.1f} % of its significant lines are in {theirs}"


def test_the_deleted_mirrors_stay_deleted():
    for gone in ("Elements.cpp", "Elements.h", "RampGenerator.cpp", "RampGenerator.h", "StarvationRamper.cpp"):
        assert not os.path.exists(os.path.join(ROOT, "ohpipeline_amd", "host", gone)), gone
