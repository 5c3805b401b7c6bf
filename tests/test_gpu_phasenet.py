"""GPU parity tests for the PhaseNet path: HIP (through the C ABI) vs the CPU oracle.
Tolerance: 1e-4 absolute on probabilities (BASELINE.json north_star), fp32."""
import numpy as np
import pytest
import torch

from oracle import pipeline as OP
from oracle.models import load_pretrained
from tests.gpu_util import debug_tensors
from volpick_amd import PhaseNet
from volpick_amd.synthetic import synthetic_stream_array, synthetic_windows

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def oracle():
    return load_pretrained("phasenet")


@pytest.fixture(scope="module")
def model():
    return PhaseNet.from_pretrained("volpick").cuda()


def _oracle_acts(net, x):
    acts = {}
    hs = [m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach())) for n, m in
          net.named_modules() if n]
    with torch.no_grad():
        y = net(x)
    for h in hs:
        h.remove()
    return y, acts


@pytest.mark.parametrize("flags", [(0, 1), (0, 1, 0, 0, 0, 1), (1, 0)], ids=["fused+dumps", "fused-mfma+dumps", "layerwise"])
def test_layers_match_oracle(oracle, flags):
    model = PhaseNet.from_pretrained("volpick")
    model._plan_flags = flags
    model.cuda()
    x = synthetic_windows(3, 3001, seed=5)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    y = model(xn).cpu().numpy()
    yo, acts = _oracle_acts(oracle, xn)
    t = debug_tensors(model, 3)
    relu = lambda a: torch.relu(a).numpy()
    exp = {"input": xn.numpy(), "inc": relu(acts["in_bn"])}
    for i in range(5):
        exp[f"down{i}.same"] = relu(acts[f"down_branch.{i}.1"])
        if i < 4:
            exp[f"down{i}.down"] = relu(acts[f"down_branch.{i}.3"])
    for j in range(4):
        full = relu(acts[f"up_branch.{j}.1"])[:, :, 1:-2]
        L = exp[f"down{3 - j}.same"].shape[2]
        off = (full.shape[2] - L) // 2
        exp[f"up{j}.convT"] = full[:, :, off:off + L]
        if j < 3:
            exp[f"up{j}.same"] = relu(acts[f"up_branch.{j}.3"])
    report = []
    for name, want in exp.items():
        got = t[name]
        err = float(np.abs(got - want).max())
        report.append((name, err, float(np.abs(want).max())))
    print("\n".join(f"{n:14s} max|diff| {e:.3e}  (max|ref| {m:.3e})" for n, e, m in report))
    for n, e, m in report:
        assert e <= 2e-4 * max(1.0, m), (n, e)
    assert np.abs(y - yo.numpy()).max() < TOL


def test_fused_equals_layerwise_bitwise(oracle):
    """Same packed weights, same k-order of every MFMA chain: the all-MFMA fused plan (plan_flags[5] = 1) and the
    layer plan must agree exactly; the default plan runs the level-0 layers on the VALU (another summation
    order) and agrees to rounding."""
    x = synthetic_windows(9, 3001, seed=77)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    fused = PhaseNet.from_pretrained("volpick")
    fused._plan_flags = (0, 0, 0, 0, 0, 1)
    fused.cuda()
    layer = PhaseNet.from_pretrained("volpick")
    layer._plan_flags = (1, 0)
    layer.cuda()
    want = layer(xn).numpy()
    assert np.array_equal(fused(xn).numpy(), want)
    default = PhaseNet.from_pretrained("volpick").cuda()  # the whole network in one launch; its five deepest layers run on
    got = default(xn).numpy()                              # the bf16 matrix cores with exact three-piece operands
    assert np.abs(got - want).max() < 5e-6
    one32 = PhaseNet.from_pretrained("volpick")  # the same launch with every core layer on the fp32 MFMA
    one32._plan_flags = (0, 0, 0, 0, 0, 3)
    one32.cuda()
    got32 = one32(xn).numpy()
    assert np.abs(got32 - want).max() < 5e-6 and np.abs(got32 - got).max() < 5e-6
    three = PhaseNet.from_pretrained("volpick")  # three launches, level-0 stride-1 convs on the VALU
    three._plan_flags = (0, 0, 0, 0, 0, 2)
    three.cuda()
    assert np.abs(three(xn).numpy() - want).max() < 5e-6


@pytest.mark.parametrize("B", [1, 7, 256])
def test_up_path_on_the_bf16_matrix_cores_agrees_with_the_fp32_mfma_forms(oracle, B):
    """Default plan: every core layer runs on the bf16 matrix cores with exact three-piece operands (up1.same and up2.same in two
    K halves); plan_flags[5] = 3 keeps them all on the fp32 MFMA (the reference form; the intermediate forms 4 .. 7 were removed
    in round 6): the two agree to fp32 rounding, each within the regression bar of the oracle, and an input with a non-finite
    window poisons that window only."""
    x = synthetic_windows(B, 3001, seed=300 + B)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    with torch.no_grad():
        want = oracle(xn).numpy()
    outs = []
    for flags in ((0,), (0, 0, 0, 0, 0, 3)):
        m = PhaseNet.from_pretrained("volpick")
        m._plan_flags = flags
        m.cuda()
        outs.append(m(xn).numpy())
        assert np.abs(outs[-1] - want).max() < 3e-5, flags
    assert np.abs(outs[0] - outs[1]).max() < 1e-5
    if B > 1:
        bad = xn.clone()
        bad[B // 2, 1, 1500] = float("nan")
        got = PhaseNet.from_pretrained("volpick").cuda()(bad).numpy()
        assert np.isnan(got[B // 2]).all()
        keep = [b for b in range(B) if b != B // 2]
        assert np.array_equal(got[keep], outs[0][keep])


@pytest.mark.parametrize("B", [1, 9, 256])
def test_level_0_down_path_on_the_matrix_cores_agrees_with_the_vector_alu_form(oracle, B):
    """Default plan (round 5, pn_window_kernel D0T): inc and down0.same run time-tiled on the bf16 matrix cores with exact
    three-piece operands (x pieces from registers into the rows down0.same fills later, inc's output as a ring of pieces, six
    tiles of 512 samples), up3.convT / up3.same / head in twelve tiles of 256; plan_flags[5] = 8 keeps the packed-FMA / fp32-MFMA
    forms of round 4 for all of level 0.  The two agree to fp32 rounding, each within the regression bar of the oracle -- on windows whose energy
    sits at the two ENDS (the tiles that meet the zero padding), with a DC offset, on the device front end as well; a
    non-finite window poisons only itself."""
    x = synthetic_windows(B, 3001, seed=8800 + B)
    x[0, :, :40] *= 50.0
    x[-1, :, -40:] *= 50.0
    x[B // 2, 1] += 777.0
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    with torch.no_grad():
        want = oracle(xn).numpy()
    outs, raws = [], []
    for flags in ((0,), (0, 0, 0, 0, 0, 8)):  # default (level 0 down AND up on the matrix cores) | neither
        m = PhaseNet.from_pretrained("volpick")
        m._plan_flags = flags
        m.cuda()
        outs.append(m(xn).numpy())  # through the input tensor
        raws.append(m._forward_raw(torch.from_numpy(x).cuda(), preprocess=True).cpu().numpy())  # window cut + normalisation in the kernel
        assert np.abs(outs[-1] - want).max() < 3e-5 and np.abs(raws[-1] - want).max() < 3e-5, flags
        m._release()
    assert np.abs(outs[0] - outs[1]).max() < 1e-5 and np.abs(raws[0] - raws[1]).max() < 1e-5
    if B > 1:
        bad = xn.clone()
        bad[1, 2, 7] = float("inf")
        got = PhaseNet.from_pretrained("volpick").cuda()(bad).numpy()
        assert np.isnan(got[1]).all()
        keep = [b for b in range(B) if b != 1]
        assert np.array_equal(got[keep], outs[0][keep])


@pytest.mark.parametrize("B", [1, 5, 256, 300])
def test_forward_parity(model, oracle, B):
    x = synthetic_windows(B, 3001, seed=100 + B)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    with torch.no_grad():
        want = oracle(xn).numpy()
    got_host = model(xn).numpy()  # host tensor in -> host tensor out
    got_dev = model(xn.cuda()).cpu().numpy()
    assert np.abs(got_host - want).max() < TOL
    assert np.array_equal(got_host, got_dev)
    assert np.abs(got_host.sum(1) - 1).max() < 1e-5  # softmax over channels


def test_preprocess_matches_annotate_batch_pre(oracle):
    x = synthetic_windows(7, 3001, seed=9)
    x[2] += 1234.5  # DC offset
    want = OP.batch_pre(oracle, torch.from_numpy(x)).numpy()
    model = PhaseNet.from_pretrained("volpick")
    model._plan_flags = (0, 0, 0, 0, 0, 2)  # a plan that materialises the input tensor (the default cuts and
    model.cuda()                            # normalises its window inside the forward kernel)
    model._forward_raw(x, preprocess=True)
    got = debug_tensors(model, 7)["input"]
    assert np.abs(got - want).max() < 2e-5


def test_in_kernel_preprocessing_is_bitwise_gather_normalize(model):
    """The default plan runs annotate_batch_pre inside its forward kernel with the arithmetic and reduction order of
    gather_normalize_kernel: same bits as the plan that goes through the input tensor (plan_flags[6] = 1)."""
    data, _, _ = synthetic_stream_array(40_000, seed=77, n_events=4)
    data[1] += 321.0
    via_tensor = PhaseNet.from_pretrained("volpick")
    via_tensor._plan_flags = (0, 0, 0, 0, 0, 0, 1)
    via_tensor.cuda()
    args = model._argdict(dict(overlap=1500, blinding=(0, 0), stacking="avg"))
    a = model._annotate_block(data, args)[0].cpu().numpy()
    b = via_tensor._annotate_block(data, args)[0].cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True)
    x = synthetic_windows(5, 3001, seed=3)
    assert np.array_equal(model._forward_raw(x, preprocess=True), via_tensor._forward_raw(x, preprocess=True))


@pytest.mark.parametrize("overlap,blinding,stacking", [(1500, (0, 0), "avg"), (2500, (500, 500), "avg"),
                                                       (2000, (250, 100), "max"), (0, (0, 0), "avg")])
def test_annotate_parity(model, oracle, overlap, blinding, stacking):
    data, _, _ = synthetic_stream_array(60_000, seed=1001, n_events=6)
    want = OP.annotate_array(oracle, data, overlap=overlap, blinding=blinding, stacking=stacking)
    args = model._argdict(dict(overlap=overlap, blinding=blinding, stacking=stacking))
    out, fv, lv, nw = model._annotate_block(data, args)
    out = out.cpu().numpy()
    assert nw == len(OP.window_starts(60_000, 3001, overlap))
    for i, (label, off, tr) in enumerate(want):
        assert off == fv and len(tr) == lv - fv + 1
        got = out[i, fv:lv + 1]
        assert np.array_equal(np.isnan(got), np.isnan(tr))
        assert np.nanmax(np.abs(got - tr)) < TOL
    assert np.isnan(out[:, :fv]).all() and np.isnan(out[:, lv + 1:]).all()


def test_classify_picks_match_oracle(model, oracle):
    from volpick_amd import Stream, Trace, UTCDateTime

    data, p_on, s_on = synthetic_stream_array(60_000, seed=1001, n_events=6)
    t0 = UTCDateTime("2020-01-01T00:00:00")
    st = Stream([Trace(data[i], dict(network="XX", station="SYN", location="", channel=f"HH{c}", starttime=t0,
                                     sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    res = model.classify(st, batch_size=256)
    want = OP.classify_array(oracle, data)
    assert len(res.picks) == len(want["picks"]) >= 10
    for p, (ph, on, off, pk, v) in zip(res.picks, want["picks"]):
        assert p.phase == ph and p.trace_id == "XX.SYN."
        assert abs((p.peak_time - t0) * 100 - pk) <= 1
        assert abs((p.start_time - t0) * 100 - on) <= 1 and abs((p.end_time - t0) * 100 - off) <= 1
        assert abs(p.peak_value - v) < TOL
    ann = model.annotate(st)
    assert [tr.stats.channel for tr in ann] == ["PhaseNet_P", "PhaseNet_S", "PhaseNet_N"]


def test_short_and_edge_inputs(model):
    from volpick_amd import Stream, Trace

    short = np.zeros((3, 2000), np.float32)
    st = Stream([Trace(short[i], dict(station="A", channel=f"HH{c}", sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    with pytest.warns(UserWarning):
        assert len(model.annotate(st)) == 0
    assert len(model.classify(Stream()).picks) == 0
    exact, _, _ = synthetic_stream_array(3001, seed=3, n_events=1)  # exactly one window, no tail
    args = model._argdict({})
    out, fv, lv, nw = model._annotate_block(exact, args)
    assert (nw, fv, lv) == (1, 0, 3000)
    with pytest.raises(ValueError):
        model.annotate(st, overlap=3001)


def test_classify_resamples_other_rates(model):
    """A 200 Hz and a 50 Hz copy of a 100 Hz stream go through SeisBench's resampling rule on the way in; the picks
    land on the same arrivals (the signal content sits well below 25 Hz)."""
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.resample import resample_fourier

    data, p_on, s_on = synthetic_stream_array(60_000, seed=1001, n_events=6)
    t0 = UTCDateTime("2020-01-01T00:00:00")

    def stream(arr, rate):
        return Stream([Trace(arr[i], dict(network="XX", station="SYN", location="", channel=f"HH{c}", starttime=t0,
                                          sampling_rate=rate)) for i, c in enumerate("ZNE")])

    ref = model.classify(stream(data, 100.0)).picks
    assert len(ref) >= 6
    fast = np.stack([resample_fourier(data[i].astype(np.float64), 100.0, 200.0, window=None) for i in range(3)])
    got = model.classify(stream(fast, 200.0)).picks  # 200 -> 100 Hz: low-pass + decimation
    assert len(got) == len(ref)
    for a, b in zip(sorted(ref, key=lambda p: p.peak_time), sorted(got, key=lambda p: p.peak_time)):
        assert a.phase == b.phase and abs(a.peak_time - b.peak_time) <= 0.05 and abs(a.peak_value - b.peak_value) < 0.1
    slow = data[:, ::2]  # 50 Hz (aliased a little: the synthetic bursts reach 8 Hz, the noise is white)
    got50 = model.classify(stream(slow, 50.0)).picks  # 50 -> 100 Hz: Fourier method
    p_ref = sorted(p.peak_time - t0 for p in ref if p.phase == "P")
    p_got = sorted(p.peak_time - t0 for p in got50 if p.phase == "P")
    assert len(p_got) >= len(p_ref) - 1 and all(min(abs(t - r) for r in p_ref) < 0.3 for t in p_got)
