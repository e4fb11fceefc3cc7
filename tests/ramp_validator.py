"""RampValidator's continuity rule (OpenHome/Media/Pipeline/RampValidator.cpp:91-124) restated over descriptors.

The reference threads a RampValidator between pipeline elements; it warns when a ramp does not start where the stream's
last one ended, or starts anywhere but an end of the scale.  Here the same rule runs over the (flags, ramp_start, ramp_end)
of every message a stream's descriptors carry, in stream order -- test infrastructure: it looks at INPUTS of the device
path (what the host model emitted), never at audio.
"""
RAMP_MIN, RAMP_MAX = 0, 1 << 14          # Ramp::kMin / kMax (Msg.h)
FLAG_RAMP = 1                            # OHGPU_FLAG_RAMP (include/ohgpu.h)


def discontinuities(messages, draining_at=()):
    """messages: iterable of (ramp_enabled, start, end) in stream order.  Returns the warnings RampValidator::ProcessAudio would
    print, as (index, text).  draining_at: indices in front of which a drain passed (its one permitted jump to an end)."""
    out = []
    ramping, last = False, None
    draining = set(draining_at)
    for i, (enabled, start, end) in enumerate(messages):
        start, end = int(start), int(end)
        direction = "up" if end > start else ("down" if end < start else None)     # (Ramp::Direction(); equal ends: a held level)
        if ramping:
            if start != last and not (i in draining and start in (RAMP_MIN, RAMP_MAX)):
                out.append((i, f"discontinuity in ramp: expected {last:#x}, got {start:#x}"))      # :100-104
            last = end                                                                                # :105
        elif enabled:
            ramping = True                                                                            # :109
            if direction == "up" and start != RAMP_MIN:
                out.append((i, f"ramp up started at {start:#x}"))                                     # :110-114
            elif direction == "down" and start != RAMP_MAX:
                out.append((i, f"ramp down started at {start:#x}"))                                   # :115-119
            last = end
        else:
            continue
        # ResetIfRampComplete (:80-89): a ramp that reached its end of the scale is over
        if (direction == "up" and last == RAMP_MAX) or (direction == "down" and last == RAMP_MIN):
            ramping, last = False, None
    return out


def check_descriptors(descs, stream_key=("src_offset",)):
    """Every stream of a descriptor array (numpy structured, SRC_MSG_DESC or MSG_DESC): messages grouped by stream_key in
    array order.  Returns {stream: warnings} for the streams that have any."""
    import numpy as np
    keys = np.stack([descs[k].astype(np.int64) for k in stream_key], axis=1)
    bad = {}
    order = np.lexsort(keys.T[::-1]) if len(descs) else []
    start = 0
    ks = keys[order] if len(descs) else keys
    for i in range(1, len(descs) + 1):
        if i == len(descs) or (ks[i] != ks[start]).any():
            idx = np.sort(order[start:i])                       # array order within the stream
            msgs = [(bool(descs["flags"][j] & FLAG_RAMP), descs["ramp_start"][j], descs["ramp_end"][j]) for j in idx]
            w = discontinuities(msgs)
            if w:
                bad[tuple(int(v) for v in ks[start])] = w
            start = i
    return bad
