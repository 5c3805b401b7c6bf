"""Splitting ONE long stream into independently processed segments whose stacked outputs, cut and
joined, equal the unsplit result sample for sample ("contiguous window ranges of each
station-stream; ranks own disjoint output sample ranges except a halo").

The window grid is ``i * step`` from the start of the stream (plus one tail window flush with the end,
SURVEY §8a A2), and output sample t stacks every window w with ``s_w + blind_l <= t < s_w + T - blind_r``
(A6/A7).  A segment [lo, hi) with ``lo`` on the grid therefore reproduces the global output exactly for

    t >= lo - step + T - blind_r      (no window that starts before ``lo`` reaches t; any t if lo == 0)
    t <  hi - T + blind_l             (every grid window that reaches t lies inside the segment, and the
                                       segment's own off-grid tail window does not; any t if hi == N)

``plan_segments`` places cuts c_0 = 0 < c_1 < ... < c_P = N and gives segment r the smallest such range
around its owned output range [c_r, c_{r+1}).  Used by ``WaveformModel`` to spread a day-long stream over
its device contexts and by ``distributed.classify_stream_sharded`` to spread it over GPUs.
"""
from __future__ import annotations


def plan_segments(n_samples: int, in_samples: int, overlap: int, blinding, parts: int):
    """-> list of dicts {lo, hi, keep_lo, keep_hi} (sample indices into the stream), at most ``parts`` of them
    (fewer when the stream is too short for every part to own at least two windows' worth of output)."""
    T, step = int(in_samples), int(in_samples) - int(overlap)
    bl, br = int(blinding[0]), int(blinding[1])
    N = int(n_samples)
    if step <= 0 or bl + br >= T:
        raise ValueError("bad overlap / blinding")
    parts = max(1, min(int(parts), N // (4 * T)))
    if N < T or parts <= 1:
        return [dict(lo=0, hi=N, keep_lo=0, keep_hi=N)]
    cuts = [0] + [(r * N) // parts for r in range(1, parts)] + [N]
    segs = []
    for r in range(parts):
        c0, c1 = cuts[r], cuts[r + 1]
        lo = 0 if r == 0 else max(0, (c0 - T + br + step) // step * step)
        hi = N if r == parts - 1 else min(N, c1 + T - bl)
        segs.append(dict(lo=lo, hi=hi, keep_lo=c0, keep_hi=c1))
    return segs


def check_plan(segs, n_samples, in_samples, overlap, blinding):
    """The exactness conditions of the module docstring (used by the tests and as a cheap runtime guard)."""
    T, step = in_samples, in_samples - overlap
    bl, br = blinding
    pos = 0
    for s in segs:
        assert s["keep_lo"] == pos and s["keep_hi"] > s["keep_lo"], s
        assert s["lo"] % step == 0 and s["hi"] - s["lo"] >= T, s
        assert s["lo"] == 0 or s["keep_lo"] >= s["lo"] - step + T - br, s
        assert s["hi"] == n_samples or s["keep_hi"] <= s["hi"] - T + bl, s
        pos = s["keep_hi"]
    assert pos == n_samples
    return True
