"""GPU: HIP path vs the committed golden fixtures, plus size-independent properties at the
BASELINE.json batch size (256 windows per pass)."""
from pathlib import Path

import numpy as np
import pytest
import torch

import volpick_amd as va
from oracle import pipeline as OP
from volpick_amd.synthetic import synthetic_stream_array

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
TOL = 1e-4


@pytest.fixture(scope="module", params=["phasenet", "eqtransformer"])
def pair(request):
    cls = va.PhaseNet if request.param == "phasenet" else va.EQTransformer
    return request.param, cls.from_pretrained("volpick").cuda(), np.load(GOLD / f"{request.param}_volpick.npz")


def test_forward_matches_golden(pair):
    name, model, z = pair
    y = model._forward_raw(z["windows_pre"])
    assert np.abs(y - z["forward"]).max() < TOL
    y2 = model._forward_raw(z["windows"], preprocess=True)  # device-side annotate_batch_pre
    assert np.abs(y2 - z["forward"]).max() < TOL


def test_annotate_and_picks_match_golden(pair):
    name, model, z = pair
    T = model.in_samples
    st = va.Stream([va.Trace(z["stream"][i], dict(network="XX", station="GOLD", channel=f"HH{c}", sampling_rate=100.0))
                    for i, c in enumerate("ZNE")])
    kw = dict(overlap=int(z["overlap"]), blinding=tuple(int(b) for b in z["blinding"]))
    ann = model.annotate(st, **kw)
    assert [tr.stats.channel for tr in ann] == [f"{model.name}_{lab}" for lab in model.labels]
    for tr, off in zip(ann, z["ann_offsets"]):
        lab = tr.stats.channel.split("_")[1]
        assert tr.stats.starttime.timestamp == pytest.approx(off / 100.0)
        assert np.abs(tr.data - z[f"ann_{lab}"]).max() < TOL
    picks = model.classify(st, **kw).picks
    assert len(picks) == len(z["picks"])
    for p, row in zip(picks, z["picks"]):
        assert "PS".index(p.phase) == int(row[0])
        assert abs(p.peak_time.timestamp * 100 - row[3]) <= 1 and abs(p.peak_value - row[4]) < TOL


def test_full_batch_properties(pair):
    """256 windows per pass (BASELINE configs[1]/[2] sizes): stacking on the device equals
    numpy stacking of the device's own window predictions; avg <= max; deterministic."""
    name, model, _ = pair
    T = model.in_samples
    overlap, blinding = (1500, (0, 0)) if name == "phasenet" else (5500, (500, 500))
    n = T + (T - overlap) * 255 + 37  # 256 regular windows + a tail window
    data, _, _ = synthetic_stream_array(n, seed=31)
    args = model._argdict(dict(overlap=overlap, blinding=blinding, stacking="avg"))
    out, fv, lv, nw = model._annotate_block(data, args)
    out = out.cpu().numpy()
    assert nw == 257 and fv == blinding[0] and lv == n - 1 - blinding[1]
    out2, *_ = model._annotate_block(data, args)
    assert np.array_equal(out, out2.cpu().numpy(), equal_nan=True)  # bitwise reproducible
    starts = OP.window_starts(n, T, overlap)
    x = np.stack([data[:, s:s + T] for s in starts])
    preds = model._forward_raw(x, preprocess=True).transpose(0, 2, 1).copy()  # (W, T, 3)
    if blinding[0]:
        preds[:, :blinding[0]] = np.nan
        preds[:, -blinding[1]:] = np.nan
    want = OP.reassemble(preds, starts, T, overlap, "avg").T
    assert np.array_equal(np.isnan(want), np.isnan(out))
    assert np.nanmax(np.abs(want - out)) < 2e-6
    mx, *_ = model._annotate_block(data, dict(args, stacking="max"))
    mx = mx.cpu().numpy()
    ok = ~np.isnan(out)
    assert (mx[ok] >= out[ok] - 1e-7).all()
    if name == "phasenet":
        assert np.abs(out[:, fv:lv + 1].sum(0) - 1).max() < 1e-5  # averaged softmaxes still sum to 1
    else:
        assert (out[ok] > 0).all() and (out[ok] < 1).all()
