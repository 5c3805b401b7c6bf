"""NaN / Inf in the input: torch (the reference) carries a non-finite sample into EVERY output sample of its window
(demeaning spreads it over the channel, the first conv over all channels; oracle checked below), and SeisBench's
nanmean / nanmax stacking then ignores that window wherever another one covers the sample.  The HIP path must show
the same windows as NaN — its kernels lose a NaN at the first ReLU (v_max is maxNum) and restore it explicitly."""
import numpy as np
import pytest
import torch

from oracle import pipeline as OP
from oracle.models import load_pretrained
from volpick_amd import EQTransformer, PhaseNet
from volpick_amd.synthetic import synthetic_stream_array, synthetic_windows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cls,name,flags", [(PhaseNet, "phasenet", (0,)), (PhaseNet, "phasenet", (0, 0, 0, 0, 0, 2)),
                                            (PhaseNet, "phasenet", (1, 0)), (EQTransformer, "eqtransformer", (0,))],
                         ids=["phasenet", "phasenet-three-launches", "phasenet-layer-plan", "eqtransformer"])
def test_nonfinite_windows_come_out_as_nan(cls, name, flags):
    model = cls.from_pretrained("volpick")
    model._plan_flags = flags
    model.cuda()
    T = model.in_samples
    x = synthetic_windows(6, T, seed=31)
    clean = np.asarray(model._forward_raw(x, preprocess=True))
    bad = x.copy()
    bad[1, 1, T // 2] = np.nan
    bad[3, 2, T - 1] = np.inf
    bad[4, 0, 0] = -np.inf
    got = np.asarray(model._forward_raw(bad, preprocess=True))
    for w in (1, 3, 4):
        assert np.isnan(got[w]).all(), w
    for w in (0, 2, 5):
        assert np.array_equal(got[w], clean[w]), w
    # the oracle agrees on which windows are lost
    net = load_pretrained(name)
    with torch.no_grad():
        y = net(OP.batch_pre(net, torch.from_numpy(bad)))
    y = (torch.stack(y, 1) if isinstance(y, (tuple, list)) else y).numpy()
    assert [bool(np.isnan(y[w]).all()) for w in range(6)] == [False, True, False, True, True, False]
    # model(x) on already normalised windows (no preprocessing): a NaN in x poisons its window as well
    xn = OP.batch_pre(net, torch.from_numpy(x))
    xn[2, 0, 17] = float("nan")
    out = model(xn)
    out = (torch.stack(list(out), 1) if isinstance(out, (tuple, list)) else out).numpy()
    assert np.isnan(out[2]).all() and not np.isnan(out[[0, 1, 3, 4, 5]]).any()


def test_nonfinite_windows_do_not_leak_into_their_neighbours_in_a_wrapped_grid():
    """300 EQTransformer windows: the persistent fused kernels run several rows / tiles per workgroup, their LDS images
    are reused from tile to tile without clearing, and the bf16-piece stages read one zero-weight tap beyond the
    filter -- whatever a non-finite window leaves behind must not reach a kept sample of the next one (0 x NaN)."""
    model = EQTransformer.from_pretrained("volpick").cuda()
    T = model.in_samples
    base = synthetic_windows(7, T, seed=77)
    x = base[np.arange(300) % 7].copy()
    clean = np.asarray(model._forward_raw(x, preprocess=True))
    bad = x.copy()
    lost = [0, 1, 85, 86, 150, 255, 256, 299]
    for k, w in enumerate(lost):
        bad[w, k % 3, (k * 977) % T] = [np.nan, np.inf, -np.inf][k % 3]
    bad[150] = 3e38  # finite input whose activations overflow on the way
    got = np.asarray(model._forward_raw(bad, preprocess=True))
    keep = np.setdiff1d(np.arange(300), lost)
    assert np.array_equal(got[keep], clean[keep])
    for w in lost:
        if w != 150:
            assert np.isnan(got[w]).all(), w


def test_annotate_ignores_poisoned_windows_like_the_oracle():
    model = PhaseNet.from_pretrained("volpick").cuda()
    net = load_pretrained("phasenet")
    data, _, _ = synthetic_stream_array(30_000, seed=77, n_events=3)
    data[1, 9_000] = np.nan  # inside windows 4 and 5 of the 1501-sample grid
    for stacking in ("avg", "max"):
        want = OP.annotate_array(net, data, overlap=1500, blinding=(0, 0), stacking=stacking)
        args = model._argdict(dict(overlap=1500, blinding=(0, 0), stacking=stacking))
        out, fv, lv, nw = model._annotate_block(data, args)
        out = out.cpu().numpy()
        for i, (label, off, tr) in enumerate(want):
            assert off == fv and len(tr) == lv - fv + 1
            got = out[i, fv:lv + 1]
            assert np.array_equal(np.isnan(got), np.isnan(tr)), (stacking, label)
            assert np.nanmax(np.abs(got - tr)) < 1e-4
        assert np.isnan(out[0, fv:lv + 1]).any()  # samples covered by poisoned windows only
