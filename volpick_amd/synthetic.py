"""Seeded synthetic 3-component waveforms.

Used by tests, ``bench.py`` and ``tools/make_golden.py``; no reference data is
available offline, so every workload in this repo is generated here.
"""
from __future__ import annotations

import numpy as np

SAMPLING_RATE = 100.0


def synthetic_stream_array(n_samples: int, seed: int, n_events: int | None = None, dtype=np.float32):
    """(3, n_samples) ZNE array: white noise sigma 0.05 plus impulsive events.

    Each event: P onset = 8 Hz burst on Z, S onset 6-9 s later = 4 Hz burst on
    N/E, both with an exp(-t/1.5 s) envelope and amplitude U(0.5, 3).
    Returns (data, p_samples, s_samples).
    """
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((3, n_samples)) * 0.05
    if n_events is None:
        n_events = max(1, n_samples // 10_000)
    t = np.arange(0, 1500) / SAMPLING_RATE
    env = np.exp(-t / 1.5)
    p_on = np.sort(rng.integers(500, max(501, n_samples - 2500), size=n_events))
    s_on = p_on + rng.integers(600, 900, size=n_events)
    for p, s in zip(p_on, s_on):
        a = rng.uniform(0.5, 3.0)
        pw = a * env * np.sin(2 * np.pi * 8.0 * t)
        sw = 1.5 * a * env * np.sin(2 * np.pi * 4.0 * t)
        lp = min(len(t), n_samples - p)
        ls = min(len(t), n_samples - s)
        if lp > 0:
            x[0, p : p + lp] += pw[:lp]
            x[1, p : p + lp] += 0.3 * pw[:lp]
            x[2, p : p + lp] += 0.3 * pw[:lp]
        if ls > 0:
            x[1, s : s + ls] += sw[:ls]
            x[2, s : s + ls] += 0.8 * sw[:ls]
            x[0, s : s + ls] += 0.2 * sw[:ls]
    return x.astype(dtype), p_on, s_on


def synthetic_windows(batch: int, in_samples: int, seed: int, dtype=np.float32):
    """(batch, 3, in_samples) independent windows, each with one event at a random position."""
    rng = np.random.default_rng(seed)
    out = np.empty((batch, 3, in_samples), dtype=dtype)
    for b in range(batch):
        w, _, _ = synthetic_stream_array(in_samples, int(rng.integers(1 << 31)), n_events=1, dtype=dtype)
        out[b] = w * rng.uniform(0.1, 1000.0)  # arbitrary physical scale; the path normalises
    return out


def eqt_stream_length(n_windows: int, in_samples=6000, overlap=5500) -> int:
    """Stream length that yields exactly ``n_windows`` windows."""
    return in_samples + (in_samples - overlap) * (n_windows - 1)
