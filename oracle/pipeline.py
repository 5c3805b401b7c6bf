"""Numpy restatement of the SeisBench ``WaveformModel.annotate/classify`` array
pipeline around the model forward pass, plus ObsPy's ``trigger_onset``.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

Stages (call stack §3.1):
  A2 cut_windows          starts = arange(0, N-in+1, in-overlap) + one tail window
  A3 annotate_batch_pre   demean, peak/std normalise (+6-sample taper, EQT)
  A4/A5 model forward     oracle.models
  A6 blind                NaN the first/last ``blinding`` samples of every window
  A7 reassemble           (L, n_out, coverage) NaN buffer, nanmean / nanmax, NaN trim
  A8 trigger_onset/picks  on where data>thr1, off = last sample of the run >thr2;
                          peak = on + argmax   (in-repo twin: volpick/model/eval_taks0.py:46-56)
"""
from __future__ import annotations

import warnings
from collections import deque

import numpy as np
import torch

from . import constants as C


# ----------------------------------------------------------------------- A2
def window_starts(n_samples: int, in_samples: int, overlap: int) -> np.ndarray:
    """Window start indices; empty if the block is shorter than one window."""
    if not 0 <= overlap < in_samples:
        raise ValueError("overlap must be in [0, in_samples)")
    starts = np.arange(0, n_samples - in_samples + 1, in_samples - overlap)
    if len(starts) == 0:
        return starts.astype(np.int64)
    if starts[-1] + in_samples < n_samples:  # one more window flush with the end
        starts = np.concatenate([starts, [n_samples - in_samples]])
    return starts.astype(np.int64)


# ----------------------------------------------------------------------- A3
def batch_pre(model, batch: torch.Tensor) -> torch.Tensor:
    """``annotate_batch_pre`` of PhaseNet / EQTransformer on a (B,C,T) float32 tensor."""
    batch = batch - batch.mean(dim=-1, keepdim=True)
    per_comp = model.name == "PhaseNet" or getattr(model, "norm_amp_per_comp", False)
    norm = model.norm
    if model.name == "EQTransformer" and per_comp:
        norm = "peak"  # EQTransformer.annotate_batch_pre: norm_amp_per_comp always uses the per-channel peak
    if norm == "peak":
        amp = batch.abs().amax(dim=-1 if per_comp else (-2, -1), keepdim=True)
    elif norm == "std":
        amp = batch.std(dim=-1 if per_comp else (-2, -1), keepdim=True)  # unbiased, as torch.std
    else:
        raise ValueError(model.norm)
    batch = batch / (amp + C.NORM_EPS)
    if model.name == "EQTransformer":
        n = C.EQT_TAPER_SAMPLES
        tap = 0.5 * (1 + torch.cos(torch.linspace(np.pi, 2 * np.pi, n)))
        batch = batch.clone()
        batch[:, :, :n] *= tap
        batch[:, :, -n:] *= tap.flip(0)
    return batch


# ----------------------------------------------------------------- A4-A6
def predict_windows(model, data: np.ndarray, starts: np.ndarray, blinding, batch_size=256) -> np.ndarray:
    """Forward all windows -> (n_win, T, n_out) float32 with NaN-blinded edges."""
    T = model.in_samples
    out = []
    with torch.no_grad():
        for b0 in range(0, len(starts), batch_size):
            ss = starts[b0 : b0 + batch_size]
            x = np.stack([data[:, s : s + T] for s in ss]).astype(np.float32)
            y = model(batch_pre(model, torch.from_numpy(x)))
            if isinstance(y, tuple):
                y = torch.stack(y, dim=-1)  # EQT: (B,T,3) = (det,P,S)
            else:
                y = y.transpose(-1, -2)  # PhaseNet: (B,T,3)
            y = y.numpy().copy()
            pre, post = blinding
            if pre > 0:
                y[:, :pre] = np.nan
            if post > 0:
                y[:, -post:] = np.nan
            out.append(y)
    return np.concatenate(out, axis=0)


# ----------------------------------------------------------------------- A7
def reassemble(preds: np.ndarray, starts: np.ndarray, in_samples: int, overlap: int, stacking="avg"):
    """NaN-buffer overlap stacking exactly as the reference pipeline lays it out."""
    coverage = int(np.ceil(in_samples / (in_samples - overlap) + 1))
    length = int(np.max(starts) + in_samples)
    merge = np.full((length, preds.shape[2], coverage), np.nan, dtype=preds.dtype)
    for i, (p, s) in enumerate(zip(preds, starts)):
        merge[s : s + p.shape[0], :, i % coverage] = p
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)
        if stacking == "avg":
            return np.nanmean(merge, axis=-1)
        if stacking == "max":
            return np.nanmax(merge, axis=-1)
    raise ValueError(f"Stacking method {stacking} unknown.")


def trim_nan(x: np.ndarray):
    """Drop leading / trailing NaN; returns (trimmed, n_front, n_back)."""
    isn = np.isnan(x)
    if isn.all():
        return x[:0], len(x), 0
    f = int(np.argmax(~isn))
    b = int(np.argmax(~isn[::-1]))
    return x[f : len(x) - b], f, b


def annotate_array(model, data: np.ndarray, overlap=None, blinding=None, stacking="avg", batch_size=256):
    """A2-A7 on one (3,N) block.  Returns list of (label, offset_samples, trace) per output."""
    d = C.PN_DEFAULTS if model.name == "PhaseNet" else C.EQT_DEFAULTS
    overlap = d["overlap"] if overlap is None else overlap
    blinding = d["blinding"] if blinding is None else blinding
    starts = window_starts(data.shape[1], model.in_samples, overlap)
    if len(starts) == 0:
        return []
    preds = predict_windows(model, data, starts, blinding, batch_size)
    stacked = reassemble(preds, starts, model.in_samples, overlap, stacking)
    out = []
    for i, label in enumerate(model.labels):
        tr, f, _ = trim_nan(stacked[:, i])
        out.append((label, f, tr))
    return out


# ----------------------------------------------------------------------- A8
def trigger_onset(charfct, thres1, thres2, max_len=9e99, max_len_delete=False):
    """ObsPy 1.4 ``obspy.signal.trigger.trigger_onset`` (pure-python form)."""
    charfct = np.asarray(charfct)
    ind1 = np.where(charfct > thres1)[0]
    if len(ind1) == 0:
        return np.empty((0, 2), dtype=np.int64)
    ind2 = np.where(charfct > thres2)[0]
    on = deque([ind1[0]])
    of = deque([-1])
    ind2_ = np.empty_like(ind2, dtype=bool)
    ind2_[:-1] = np.diff(ind2) > 1
    ind2_[-1] = True  # last run end is missed by diff
    of.extend(ind2[ind2_].tolist())
    on.extend(ind1[np.where(np.diff(ind1) > 1)[0] + 1].tolist())
    if max_len_delete:
        of.extend([1e99])
        on.extend([on[-1]])
    else:
        of.extend([ind2[-1]])
    pick = []
    while on[-1] > of[0]:
        while on[0] <= of[0]:
            on.popleft()
        while of[0] < on[0]:
            of.popleft()
        if of[0] - on[0] > max_len:
            if max_len_delete:
                on.popleft()
                continue
            of.appendleft(on[0] + max_len)
        pick.append([on[0], of[0]])
    return np.array(pick, dtype=np.int64).reshape(-1, 2)


def picks_from_trace(data: np.ndarray, thr_on: float, thr_off: float | None = None):
    """[(on, off, peak_index, peak_value)] per trigger.

    ``classify`` uses thr_off = thr_on for picks and thr_on / 2 for detections;
    the reference's own evaluation uses thr_on / 2 for picks
    (volpick/model/eval_taks0.py:46-56).
    """
    thr_off = thr_on if thr_off is None else thr_off
    res = []
    for s0, s1 in trigger_onset(data, thr_on, thr_off):
        seg = data[s0 : s1 + 1]
        res.append((int(s0), int(s1), int(s0 + np.argmax(seg)), float(np.max(seg))))
    return res


def classify_array(model, data, thresholds=None, **kw):
    """A2-A8 on one (3,N) block -> {"picks": [(phase, on, off, peak, value)], "detections": [...]}.

    Sample indices are relative to the block start.
    """
    d = C.PN_DEFAULTS if model.name == "PhaseNet" else C.EQT_DEFAULTS
    thresholds = dict(thresholds or {})
    ann = annotate_array(model, data, **kw)
    picks, dets = [], []
    for label, off, tr in ann:
        if label == "N":
            continue
        if label == "Detection":
            thr = thresholds.get("detection", model.default_args.get("detection_threshold", d["detection_threshold"]))
            for on, of, pk, v in picks_from_trace(tr, thr, thr / 2):
                dets.append((on + off, of + off, v))
            continue
        thr = thresholds.get(label, model.default_args.get(f"{label}_threshold", d["threshold"]))
        for on, of, pk, v in picks_from_trace(tr, thr, thr):
            picks.append((label, on + off, of + off, pk + off, v))
    picks.sort(key=lambda p: (p[1], p[0]))
    return {"picks": picks, "detections": dets, "annotations": ann}


# ----------------------------------------------------------------------- A1
def stream_to_array(traces, component_order="ZNE", sampling_rate=C.SAMPLING_RATE):
    """(t0, (C,N) float array) from [(component_letter, start_time_s, data)].

    In-repo twin: volpick/data/convert.py:26-70 (zero-fill to the common
    start/end).  Unlike that converter the annotate pipeline does not demean
    the whole block here; windows are demeaned in ``batch_pre``.
    """
    t0 = min(t for _, t, _ in traces)
    t1 = max(t + (len(x) - 1) / sampling_rate for _, t, x in traces)
    n = int(round((t1 - t0) * sampling_rate)) + 1
    data = np.zeros((len(component_order), n), dtype=np.float64)
    for comp, t, x in traces:
        if comp not in component_order:
            continue
        c = component_order.index(comp)
        s = int(round((t - t0) * sampling_rate))
        l = min(len(x), n - s)
        data[c, s : s + l] = x[:l]
    return t0, data
