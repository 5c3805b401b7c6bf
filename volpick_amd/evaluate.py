"""The reference's own window-level evaluation protocol on the MI355X path.

Mirrors ``evaluate()`` of volpick/model/eval_taks0.py:20-200 for pre-cut, already normalised
windows: forward pass, slice ``window_borders``, ``trigger_onset(prob, thr, thr / 2)`` and per
trigger ``max`` / ``argmax`` (``get_picks_from_prob``, eval_taks0.py:46-56).  The per-sample
Python loop of the reference (eval_taks0.py:96-142) runs as one kernel launch per phase
(``vp_pick_windows``).  Dataset handling, TP/FP/FN counting and CSV bookkeeping stay out of scope.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def evaluate_windows(model, X, window_borders=None, threshold=0.3, batch_size=1024, max_picks=64):
    """X: (N, 3, T) float32, normalised as the reference's generator does
    (``Normalize(demean_axis=-1, amp_norm_axis=-1, amp_norm_type=model.norm)``, eval_taks0.py:458-469).
    window_borders: (N, 2) int [start, end) local sample ranges, or None for whole windows.
    threshold: float, or [P_threshold, S_threshold].

    Returns ``[p_picks, p_scores, s_picks, s_scores]``: four lists with one ndarray per window
    (pick samples are local to the window border start, ascending), the first four entries of the
    reference's ``merged_predictions`` before the per-trace ``start_sample`` offset is added.
    """
    torch = __import__("torch")
    lib = _lib.load()
    X = np.ascontiguousarray(X, dtype=np.float32)
    n = X.shape[0]
    p_thr, s_thr = (threshold if isinstance(threshold, (list, tuple)) else (threshold, threshold))
    if model.name == "PhaseNet":
        rows = {"P": model.labels.index("P"), "S": model.labels.index("S")}
    else:  # EQTransformer returns (detection, P, S)
        rows = {"P": 1, "S": 2}
    lo = hi = None
    if window_borders is not None:
        wb = np.asarray(window_borders, dtype=np.int32).reshape(n, 2)
        lo, hi = np.ascontiguousarray(wb[:, 0]), np.ascontiguousarray(wb[:, 1])
    out = {ph: ([], []) for ph in rows}
    h = model._ensure_handle()
    for b0 in range(0, n, batch_size):
        xb = torch.from_numpy(X[b0 : b0 + batch_size]).to(model.device)
        y = model._forward_raw(xb)  # (B, 3, T) on the device
        B = y.shape[0]
        for ph, thr in (("P", p_thr), ("S", s_thr)):
            count = np.zeros(B, np.int32)
            peak = np.zeros((B, max_picks), np.int32)
            value = np.zeros((B, max_picks), np.float32)
            plo = lo[b0 : b0 + B].ctypes.data_as(C.c_void_p) if lo is not None else None
            phi = hi[b0 : b0 + B].ctypes.data_as(C.c_void_p) if hi is not None else None
            _lib.check(lib.vp_pick_windows(h, C.c_void_p(y.data_ptr()), _lib.VP_MEM_DEVICE, B, 3, rows[ph], plo, phi,
                                           float(thr), float(thr) / 2.0, max_picks,
                                           count.ctypes.data_as(C.c_void_p), peak.ctypes.data_as(C.c_void_p),
                                           value.ctypes.data_as(C.c_void_p)), "vp_pick_windows")
            if (count > max_picks).any():
                raise RuntimeError(f"more than max_picks={max_picks} triggers in a window; raise max_picks")
            for i in range(B):
                k = count[i]
                order = np.argsort(peak[i, :k], kind="stable")
                out[ph][0].append(peak[i, :k][order].astype(np.int64))
                out[ph][1].append(value[i, :k][order])
    return [out["P"][0], out["P"][1], out["S"][0], out["S"][1]]
