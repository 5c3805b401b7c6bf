"""Synthetic miniSEED inputs shared by the CPU and GPU ingestion tests (made with the oracle's encoder)."""
import numpy as np

from oracle import mseed as OM

T0 = 1_600_000_000_123_400  # a BTIME-representable start (multiple of 100 us)


def seismogram(n, rng, scale=300, spikes=True):
    """Random-walk counts with occasional large steps so every Steim word kind occurs."""
    d = rng.integers(-scale, scale + 1, n)
    small = rng.random(n) < 0.5
    d = np.where(small, rng.integers(-7, 8, n), d)
    if spikes:
        big = rng.random(n) < 0.01
        d = np.where(big, rng.integers(-(1 << 27), 1 << 27, n), d)
    return np.clip(np.cumsum(d), -(1 << 30), (1 << 30) - 1).astype(np.int32)


def three_component(n, rng, start_us=T0, rate=100.0, net="XX", sta="VOLC", loc="", band="HH", **kw):
    return [dict(network=net, station=sta, location=loc, channel=band + c, start_us=start_us, rate=rate,
                 data=seismogram(n, rng, **kw)) for c in "ZNE"]


def file_bytes(traces, **kw):
    return OM.write_mseed(traces, **kw)
