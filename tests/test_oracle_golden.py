"""CPU: the oracle reproduces the committed (self-)golden vectors, and its pipeline pieces
obey the reference's rules (window cut, NaN stacking, trigger_onset)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pipeline as OP
from oracle.models import load_pretrained
from volpick_amd.synthetic import synthetic_stream_array

GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("model", ["phasenet", "eqtransformer"])
def test_oracle_reproduces_golden(model):
    z = np.load(GOLD / f"{model}_volpick.npz")
    net = load_pretrained(model)
    xn = OP.batch_pre(net, torch.from_numpy(z["windows"]))
    assert np.abs(xn.numpy() - z["windows_pre"]).max() < 1e-6
    with torch.no_grad():
        y = net(xn)
    y = torch.stack(y, 1).numpy() if isinstance(y, tuple) else y.numpy()
    assert np.abs(y - z["forward"]).max() < 2e-5
    res = OP.classify_array(net, z["stream"], overlap=int(z["overlap"]), blinding=tuple(z["blinding"]))
    for (lab, off, tr), o in zip(res["annotations"], z["ann_offsets"]):
        assert off == o
        assert np.nanmax(np.abs(tr - z[f"ann_{lab}"])) < 2e-5
    assert len(res["picks"]) == len(z["picks"])
    for (ph, on, off, pk, v), row in zip(res["picks"], z["picks"]):
        assert "PS".index(ph) == int(row[0]) and abs(pk - row[3]) <= 1
    # plausibility: most synthetic arrivals are picked within 0.15 s (alignment is not grossly off)
    picked = {ph: [pk for p, _, _, pk, _ in res["picks"] if p == ph] for ph in "PS"}
    hits = total = 0
    for ph, truth in (("P", z["true_p"]), ("S", z["true_s"])):
        for t in truth:
            if 300 < t < len(z["stream"][0]) - 300:
                total += 1
                hits += bool(picked[ph]) and min(abs(t - pk) for pk in picked[ph]) <= 15
    assert hits >= 0.6 * total, (hits, total, picked)


def test_phasenet_structure():
    net = load_pretrained("phasenet")
    assert net.labels == "PSN" and net.norm == "peak" and net.component_order == "ZNE"
    assert net.default_args == {"P_threshold": 0.39, "S_threshold": 0.34}
    y = net(torch.randn(2, 3, 3001))
    assert y.shape == (2, 3, 3001) and torch.allclose(y.sum(1), torch.ones(2, 3001), atol=1e-5)


def test_eqt_structure():
    net = load_pretrained("eqtransformer")
    assert net.labels == ["Detection", "P", "S"]
    assert abs(net.default_args["detection_threshold"] - 0.10141666) < 1e-9
    assert net.encoder.paddings == [0, 0, 0, 0, 1, 0, 0] and net.decoder_d.crops == [2]
    with torch.no_grad():
        assert net.bottleneck(torch.randn(1, 3, 6000)).shape == (1, 16, 47)
        outs = net(torch.randn(1, 3, 6000))
    assert len(outs) == 3 and all(o.shape == (1, 6000) and (o > 0).all() and (o < 1).all() for o in outs)


def test_window_starts_rule():
    assert OP.window_starts(6890, 6000, 1000).tolist() == [0, 890]  # demo.ipynb stream: 2 windows
    assert OP.window_starts(6000, 6000, 1000).tolist() == [0]
    assert OP.window_starts(5999, 6000, 1000).tolist() == []
    assert len(OP.window_starts(60_000, 3001, 1500)) == 39  # 10 min PhaseNet
    assert len(OP.window_starts(8_640_000, 6000, 5500)) == 17_269  # 24 h EQT
    s = OP.window_starts(10_000, 3001, 1500)
    assert s[-1] == 10_000 - 3001 and (np.diff(s[:-1]) == 1501).all()
    with pytest.raises(ValueError):
        OP.window_starts(100, 10, 10)


def test_reassemble_nan_semantics():
    T, ov = 10, 5
    starts = OP.window_starts(22, T, ov)
    preds = np.stack([np.full((T, 1), float(i + 1), np.float32) for i in range(len(starts))])
    preds[:, :2] = np.nan
    preds[:, -1:] = np.nan
    avg = OP.reassemble(preds, starts, T, ov, "avg")[:, 0]
    mx = OP.reassemble(preds, starts, T, ov, "max")[:, 0]
    assert np.isnan(avg[:2]).all() and np.isnan(avg[-1])
    assert avg[2] == 1 and avg[7] == 1.5 and mx[7] == 2
    tr, f, b = OP.trim_nan(avg)
    assert (f, b) == (2, 1) and not np.isnan(tr[[0, -1]]).any()


def test_trigger_onset_known_answers():
    z = np.load(GOLD / "trigger_cases.npz")
    x = z["x"]
    assert OP.trigger_onset(x, .3, .3).tolist() == [[2, 3], [7, 8], [11, 14]] == z["t_03_03"].tolist()
    assert OP.trigger_onset(x, .3, .15).tolist() == [[2, 4], [7, 8], [11, 14]]
    assert OP.trigger_onset(x, .5, .1).tolist() == [[3, 4], [7, 8], [11, 14]]
    assert OP.trigger_onset(np.zeros(5), .3, .3).shape == (0, 2)
    assert OP.picks_from_trace(x, .3) == [(2, 3, 3, pytest.approx(.6)), (7, 8, 8, pytest.approx(.8)),
                                          (11, 14, 11, pytest.approx(.9))]


@pytest.mark.parametrize("model_name,overlap,blinding,stacking,parts", [
    ("phasenet", 1500, (0, 0), "avg", 3), ("phasenet", 2500, (250, 100), "max", 4),
    ("eqtransformer", 5500, (500, 500), "avg", 2)])
def test_segmented_annotate_equals_unsplit(model_name, overlap, blinding, stacking, parts):
    """volpick_amd.segments: a long stream cut into independently annotated segments gives, after the cut and
    join, exactly the unsplit stacked output (NaN pattern included) -- checked on the CPU oracle."""
    from volpick_amd.segments import check_plan, plan_segments

    net = load_pretrained(model_name)
    T = net.in_samples
    N = (4 * parts + 1) * T + 1234
    data, _, _ = synthetic_stream_array(N, seed=77, n_events=5)

    def stacked(block):
        starts = OP.window_starts(block.shape[1], T, overlap)
        preds = OP.predict_windows(net, block, starts, blinding, 64)
        out = OP.reassemble(preds, starts, T, overlap, stacking)  # (length, n_out)
        full = np.full((block.shape[1], out.shape[1]), np.nan, dtype=out.dtype)
        full[: out.shape[0]] = out
        return full

    want = stacked(data)
    segs = plan_segments(N, T, overlap, blinding, parts)
    assert len(segs) == parts and check_plan(segs, N, T, overlap, blinding)
    got = np.full_like(want, np.nan)
    for s in segs:
        part = stacked(data[:, s["lo"]:s["hi"]])
        got[s["keep_lo"]:s["keep_hi"]] = part[s["keep_lo"] - s["lo"]:s["keep_hi"] - s["lo"]]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    # the oracle itself is not bit-reproducible across batch compositions / nanmean slot order (1 ulp); the HIP path is,
    # and tests/test_gpu_async.py asserts equality there
    assert np.abs(got[~np.isnan(got)] - want[~np.isnan(want)]).max() < 5e-7
    # short streams and degenerate requests fall back to one segment
    assert len(plan_segments(3 * T, T, overlap, blinding, 8)) == 1
    assert plan_segments(T - 1, T, overlap, blinding, 4) == [dict(lo=0, hi=T - 1, keep_lo=0, keep_hi=T - 1)]
