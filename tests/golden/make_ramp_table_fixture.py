#!/usr/bin/env python3
"""Extracts the 512 Q15 multipliers of the reference's RampArray.h into a JSON data fixture.

Run in the build container only (it reads /root/reference as text; nothing is imported or executed):
    python3 tests/golden/make_ramp_table_fixture.py
Writes tests/golden/ramp_table_q15.json = {"source": ..., "values": [512 ints]}.
"""
import json
import os
import re

SRC = "/root/reference/OpenHome/Media/Pipeline/RampArray.h"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ramp_table_q15.json")

text = open(SRC).read()
body = text[text.index("kRampArray[]"):text.index("};")]
values = [int(v, 16) for v in re.findall(r"0x([0-9A-Fa-f]{4})", body)]
assert len(values) == 512, len(values)
json.dump({"source": "OpenHome/Media/Pipeline/RampArray.h:7-74 (kRampArray, Q15)", "values": values},
          open(OUT, "w"), separators=(",", ":"))
print("wrote", OUT, len(values))
