"""GPU parity tests for the EQTransformer path: HIP (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline as OP
from oracle.models import load_pretrained
from tests.gpu_util import debug_tensors
from volpick_amd import EQTransformer
from volpick_amd.synthetic import synthetic_stream_array, synthetic_windows

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def oracle():
    return load_pretrained("eqtransformer")


@pytest.fixture(scope="module")
def model():
    return EQTransformer.from_pretrained("volpick").cuda()


def _oracle_stages(net, x):
    """Every intermediate the device plan materialises, keyed by its tensor name."""
    out = {}
    up2 = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    with torch.no_grad():
        h = x
        for s, (conv, pad) in enumerate(zip(net.encoder.convs, net.encoder.paddings)):
            y = torch.relu(conv(h))
            if pad:
                y = F.pad(y, (0, pad), "constant", -1e10)
            h = F.max_pool1d(y, 2)
            out[f"encoder.{s}"] = h
        h = net.res_cnn_stack(h)
        out["res.xa"] = h  # block 6 writes the ping buffer "res.xa"
        for s, blk in enumerate(net.bi_lstm_stack.members):
            h = blk(h)
            out[f"bilstm.{s}"] = h
        h = net.transformer_d0(h)
        out["transformer_d0"] = h
        h = net.transformer_d(h)
        out["transformer_d"] = h
        dec_in = [h]
        for lstm, att in zip(net.pick_lstms, net.pick_attentions):
            px = lstm(h.permute(2, 0, 1))[0].permute(1, 2, 0)
            px, _ = att(px)
            dec_in.append(px)
        out["decoder.in"] = torch.cat(dec_in, 0)
        decs = [net.decoder_d, net.pick_decoders[0], net.pick_decoders[1]]
        hs = list(dec_in)
        for s in range(7):
            ys = []
            for d, dec in enumerate(decs):
                u = up2(hs[d])
                if s in dec.crops:
                    u = u[:, :, :-1]
                ys.append(torch.relu(dec.convs[s](u)))
            hs = ys
            if s < 6:  # decoder.6 never leaves the chip: the sigmoid heads are fused into its epilogue
                out[f"decoder.{s}"] = torch.cat(ys, 0)
        final = net(x)
    return {k: v.numpy() for k, v in out.items()}, [f.numpy() for f in final]


# vp_config.plan_flags[7]: bit 0 = decoder.4 / .5 / .6+heads as three launches, bit 1 = decoder.0 .. .3 as five launches,
# bit 2 = encoder.0 .. .2 as three launches, bit 3 = encoder.3 .. .6 as four launches; bit 4 = ResCNN on the fp32 MFMA,
# bit 5 = every stage of the fused decoder.0 .. .3 on the fp32 MFMA, bit 6 = the fused decoder tail, bit 7 = the fused encoder
# 3-6 kernel, bit 8 = stages 1 and 2 of the fused encoder 0-2 kernel on the fp32 MFMA (the default runs encoder stages 1-6, decoder stages 1-6, the heads and the ResCNN on the
# bf16 matrix cores with exact three-piece operands: fp32-accurate, different rounding)
UNFUSED = (0, 0, 0, 0, 0, 0, 0, 15)
FP32_MFMA = 32 | 64 | 128 | 256


@pytest.fixture(scope="module")
def model_fp32_mfma():
    m = EQTransformer.from_pretrained("volpick")
    m._plan_flags = (0, 0, 0, 0, 0, 0, 0, FP32_MFMA)
    return m.cuda()


def test_layers_match_oracle(oracle):
    """Every tensor the layer plan materialises, stage by stage.  The default plan never writes decoder.0 - .2 and
    .4 / .5 nor encoder.0 / .1 (they live in LDS inside the fused kernels), so this runs the plan that keeps those
    launches; the next test ties the fused kernels to it bit for bit."""
    model = EQTransformer.from_pretrained("volpick")
    model._plan_flags = UNFUSED
    model.cuda()
    B = 3
    x = synthetic_windows(B, 6000, seed=21)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    det, p, s = model(xn)
    want, final = _oracle_stages(oracle, xn)
    t = debug_tensors(model, 3 * B)
    report = []
    for name, w in want.items():
        got = t[name][: w.shape[0]]
        report.append((name, float(np.abs(got - w).max()), float(np.abs(w).max())))
    print("\n" + "\n".join(f"{n:16s} max|diff| {e:.3e}  (max|ref| {m:.3e})" for n, e, m in report))
    for n, e, m in report:
        assert e <= 3e-4 * max(1.0, m), (n, e)
    for got, w in zip((det, p, s), final):
        assert np.abs(got.cpu().numpy() - w).max() < TOL


@pytest.mark.parametrize("keep", [1, 2, 4, 8, 15])
@pytest.mark.parametrize("B", [1, 2, 5, 86, 256, 300])
def test_fused_decoder_kernels_are_bitwise_the_layer_launches(model_fp32_mfma, B, keep):
    """eqt_front_kernel (encoder stages 0-2 per 250-sample time tile, halos recomputed, MaxPool in registers / across
    neighbouring lanes), eqt_enc36_kernel (encoder stages 3-6, one window per workgroup), eqt_dec03_kernel (decoder stages 0-3, one row per workgroup, intermediates in LDS, the cropped edge of stage 2
    beside it) and eqt_tail_kernel (stages 4-6 + heads per 2000-sample time tile, halos recomputed, heads as a Toeplitz
    product on the matrix cores) use the same packed fragments and the same K order as the conv_mfma_kernel launches
    they replace: identical bits, for every tile of every row (left edge, interior, right edge of the signal), for
    batch sizes that leave the persistent grids partly filled, exactly filled, and wrapped several times.  `keep`
    (vp_config.plan_flags[7]) un-fuses the tail, the decoder stages 0-3, the encoder front, the encoder stages 3-6, or all of them.
    The fused side is the plan with eqt_dec03_kernel's stages all on the fp32 MFMA (the next test covers the default)."""
    model = model_fp32_mfma
    other = EQTransformer.from_pretrained("volpick")
    other._plan_flags = (0, 0, 0, 0, 0, 0, 0, keep | FP32_MFMA)
    other.cuda()
    x = synthetic_windows(7, 6000, seed=640 + B)[np.arange(B) % 7] * np.linspace(0.5, 2.0, B, dtype=np.float32)[:, None, None]
    xd = torch.from_numpy(x).cuda()
    got = model._forward_raw(xd, preprocess=True)
    want = other._forward_raw(xd, preprocess=True)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    assert (got > 0).all() and (got < 1).all()


@pytest.mark.parametrize("B", [1, 2, 5, 86, 256, 300])
def test_bf16_piece_kernels_agree_with_the_fp32_mfma_forms(model, model_fp32_mfma, B):
    """The default eqt_enc36_b3_kernel runs its four stages, eqt_dec03_kernel its stages 1-3 as six bf16 MFMAs per K-step over exact three-piece operands
    (conv_b3.h), eqt_tail3_kernel its three stages and the heads: what they drop is below the rounding of one fp32 product, so the two forms differ by fp32 rounding
    only -- for every row of batches that leave the persistent grid partly filled, filled, and wrapped (the fp32 image
    of stage 3 is rebuilt in the place of the stage-0 / stage-1 images every row)."""
    x = synthetic_windows(7, 6000, seed=640 + B)[np.arange(B) % 7] * np.linspace(0.5, 2.0, B, dtype=np.float32)[:, None, None]
    xd = torch.from_numpy(x).cuda()
    got = model._forward_raw(xd, preprocess=True)
    want = model_fp32_mfma._forward_raw(xd, preprocess=True)
    # (2e-6 while only the decoder differed; rounding differences of the encoder pass through the BiLSTM / attention
    # stages on their way out -- the distance of either form to the oracle is 6.5e-6, tools/err_check.py)
    assert (got - want).abs().max().item() < 1e-5
    assert (got > 0).all() and (got < 1).all()
    again = model._forward_raw(xd, preprocess=True)
    assert torch.equal(got.view(torch.int32), again.view(torch.int32))


@pytest.mark.parametrize("B", [1, 4, 256, 270])
def test_forward_parity(model, oracle, B):
    x = synthetic_windows(B, 6000, seed=200 + B)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    with torch.no_grad():
        want = oracle(xn)
    got = model(xn.cuda())
    assert isinstance(got, tuple) and len(got) == 3
    for g, w in zip(got, want):
        g = g.cpu().numpy()
        assert g.shape == (B, 6000)
        assert np.abs(g - w.numpy()).max() < TOL
        assert (g > 0).all() and (g < 1).all()


@pytest.mark.parametrize("B", [1, 2, 3, 5, 256, 257, 258])
def test_middle_kernel_with_several_windows_per_workgroup_is_bitwise_the_one_window_form(model, B):
    """Default: eqt_mid4_kernel -- 1024 threads, FOUR windows per workgroup in teams of four waves (odd teams' roles rotated by
    two waves), a batch's last workgroup computing its last window up to four times; plan_flags[2] = 3: eqt_mid_kernel<2>, two
    windows in teams of eight waves (the default of rounds 3-5); plan_flags[2] = 2: one window per 512-thread workgroup.
    Same arithmetic and the same order of every sum per window: every output bit-identical."""
    x = torch.from_numpy(synthetic_windows(B, 6000, seed=500 + B)).cuda()
    one = EQTransformer.from_pretrained("volpick")
    one._plan_flags = (0, 0, 2)
    one.cuda()
    two = EQTransformer.from_pretrained("volpick")
    two._plan_flags = (0, 0, 3)
    two.cuda()
    want = one._forward_raw(x, preprocess=True)
    assert torch.equal(model._forward_raw(x, preprocess=True), want)
    assert torch.equal(two._forward_raw(x, preprocess=True), want)
    one._release(), two._release()


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 7, 256, 257, 258])
def test_rescnn_kernels_with_several_windows_per_workgroup_are_bitwise_the_one_window_form(model, B):
    """Default: eqt_res3t_kernel -- eight waves per THREE windows, wave = (16 output channels, half of the nine n-tiles: 5 + 4),
    the operand requested in parts inside the K loops, the residual rows in registers, a batch's last workgroup computing its last
    window up to three times; plan_flags[7] bit 13: eqt_res3_kernel<2> (waves 0-3 one window, waves 4-7 the next: the default of
    rounds 5-6); bit 9: one window per 256-thread workgroup.  Same products in the same order into every accumulator, same epilogue arithmetic: bit-identical.  (Bit 12, eight
    waves per window with K split over wave pairs, was removed in round 6.)"""
    x = torch.from_numpy(synthetic_windows(B, 6000, seed=600 + B)).cuda()
    want = None
    for bit in (512, 8192):
        m = EQTransformer.from_pretrained("volpick")
        m._plan_flags = (0, 0, 0, 0, 0, 0, 0, bit)
        m.cuda()
        got = m._forward_raw(x, preprocess=True)
        m._release()
        if want is None:
            want = got
        assert torch.equal(got, want), bit
    assert torch.equal(model._forward_raw(x, preprocess=True), want)


def test_six_launch_plan_matches_fused_middle_kernel(model, oracle):
    """plan_flags[2] = 1 keeps the BiLSTM / transformer / pick-branch launches that eqt_mid_kernel replaces (scalar FMA
    chains instead of matrix-core tiles, same algorithm): both plans within the oracle tolerance, and of each other."""
    B = 5
    x = synthetic_windows(B, 6000, seed=77)
    xn = OP.batch_pre(oracle, torch.from_numpy(x))
    with torch.no_grad():
        want = oracle(xn)
    six = EQTransformer.from_pretrained("volpick")
    six._plan_flags = (0, 0, 1)
    six.cuda()
    got6 = six(xn.cuda())
    got1 = model(xn.cuda())
    for g6, g1, w in zip(got6, got1, want):
        assert np.abs(g6.cpu().numpy() - w.numpy()).max() < TOL
        assert np.abs(g6.cpu().numpy() - g1.cpu().numpy()).max() < 2e-5


@pytest.mark.parametrize("flags", [(0,), (0, 0, 1)])
def test_attention_beyond_the_exp_product_guard(oracle, flags):
    """The score loop works on exp(2q) * exp(2k) while |q|, |k| <= 30 and falls back to tanh(q + k) per window beyond
    (eqt_kernels.hip attn_scores).  Projection weights scaled by 40 push q and k past the guard in one transformer block
    and one pick branch: same answer as the oracle with the same weights, in the fused and in the six-launch plan."""
    import copy

    big = copy.deepcopy(oracle)
    names = ["transformer_d.attention.Wx", "transformer_d.attention.Wt", "pick_attentions.1.Wx", "pick_attentions.1.Wt"]
    params = dict(big.named_parameters())
    with torch.no_grad():
        for n in names:
            params[n].mul_(40.0)
    m = EQTransformer.from_pretrained("volpick")
    sd = m.state_dict()
    for n in names:
        sd[n] = sd[n] * 40.0
    m.load_state_dict(sd)
    m._plan_flags = flags
    m.cuda()
    B = 3
    x = synthetic_windows(B, 6000, seed=91)
    xn = OP.batch_pre(big, torch.from_numpy(x))
    with torch.no_grad():
        want = big(xn)
        h = big.transformer_d0(big.bi_lstm_stack(big.res_cnn_stack(big.encoder(xn))))
        att = big.transformer_d.attention
        q = torch.matmul(h.permute(0, 2, 1), att.Wt)
        assert float(q.abs().max()) > 30.0, "the test must leave the guarded range"
    got = m(xn.cuda())
    for g, w in zip(got, want):
        assert np.isfinite(g.cpu().numpy()).all()
        assert np.abs(g.cpu().numpy() - w.numpy()).max() < TOL


@pytest.mark.parametrize("per_comp", [False, True])
def test_preprocess_matches_annotate_batch_pre(oracle, per_comp):
    m = EQTransformer.from_pretrained("volpick")
    m.norm_amp_per_comp = per_comp
    m.cuda()
    oracle.norm_amp_per_comp = per_comp
    try:
        x = synthetic_windows(5, 6000, seed=9)
        x[1] -= 77.0
        want = OP.batch_pre(oracle, torch.from_numpy(x)).numpy()
        m._forward_raw(x, preprocess=True)
        got = debug_tensors(m, 5)["input"]
        assert np.abs(got - want).max() < 2e-5
    finally:
        oracle.norm_amp_per_comp = False


@pytest.mark.parametrize("norm,per_comp", [("peak", False), ("peak", True), ("std", False)])
def test_in_kernel_preprocessing_is_bitwise_gather_normalize(norm, per_comp):
    """plan_flags[6] = 2 runs annotate_batch_pre inside eqt_front_kernel (window cut from the raw stream, statistics in
    the reduction order of gather_normalize_kernel, normalisation + taper while a tile is parked): same bits as the default
    plan, which goes through gather_normalize_kernel and the input tensor -- on a stream with a tail window, on dense
    windows, with non-finite windows (the tail kernel writes their NaN in both plans)."""
    def make(flags):
        m = EQTransformer.from_pretrained("volpick")
        m.norm, m.norm_amp_per_comp, m._plan_flags = norm, per_comp, flags
        return m.cuda()

    fused, via_tensor = make((0, 0, 0, 0, 0, 0, 2)), make((0, 0))
    data, _, _ = synthetic_stream_array(6000 + 500 * 21 + 137, seed=78, n_events=3)
    data[1] += 321.0
    args = fused._argdict(dict(overlap=5500, blinding=(500, 500), stacking="avg"))
    a = fused._annotate_block(data, args)[0].cpu().numpy()
    b = via_tensor._annotate_block(data, args)[0].cpu().numpy()
    assert np.array_equal(a, b, equal_nan=True)
    x = synthetic_windows(7, 6000, seed=3)
    x[2, 1, 4000] = np.inf
    x[5, 0, 17] = np.nan
    ya, yb = fused._forward_raw(x, preprocess=True), via_tensor._forward_raw(x, preprocess=True)
    assert np.isnan(ya[2]).all() and np.isnan(ya[5]).all() and np.isfinite(ya[[0, 1, 3, 4, 6]]).all()
    assert np.array_equal(ya, yb, equal_nan=True)
    fused._release(), via_tensor._release()


@pytest.mark.parametrize("blinding", [(500, 500), (1000, 1000), (250, 777), (0, 0), (496, 1), (17, 0), (2999, 2999)])
def test_decoder_tail_skips_only_blinded_tiles(model, blinding):
    """annotate / classify blind the first and last samples of every window; eqt_tail3_kernel computes only the time tiles that
    hold kept samples (four tiles of 1256 instead of five of 1200 for blinding (500, 500)).  The stacked rows are bit for bit
    those of the plan that computes every tile (plan_flags[7] bit 10), for avg and max stacking, with a tail window."""
    full = EQTransformer.from_pretrained("volpick")
    full._plan_flags = (0, 0, 0, 0, 0, 0, 0, 1024)
    full.cuda()
    data, _, _ = synthetic_stream_array(6000 + 500 * 17 + 233, seed=431, n_events=4)
    for stacking in ("avg", "max"):
        args = model._argdict(dict(overlap=5500, blinding=blinding, stacking=stacking))
        a, fva, lva, nwa = model._annotate_block(data, args)
        b, fvb, lvb, nwb = full._annotate_block(data, args)
        assert (fva, lva, nwa) == (fvb, lvb, nwb) and nwa == 19
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)
    x = synthetic_windows(3, 6000, seed=12)  # model(x): the whole row, unchanged
    assert np.array_equal(model._forward_raw(x, preprocess=True), full._forward_raw(x, preprocess=True))
    full._release()


@pytest.mark.parametrize("overlap,blinding,stacking", [(5500, (500, 500), "avg"), (1800, (500, 500), "avg"),
                                                       (3000, (1000, 1000), "max")])
def test_annotate_parity(model, oracle, overlap, blinding, stacking):
    data, _, _ = synthetic_stream_array(36_000, seed=1003, n_events=4)
    want = OP.annotate_array(oracle, data, overlap=overlap, blinding=blinding, stacking=stacking)
    args = model._argdict(dict(overlap=overlap, blinding=blinding, stacking=stacking))
    out, fv, lv, nw = model._annotate_block(data, args)
    out = out.cpu().numpy()
    assert nw == len(OP.window_starts(36_000, 6000, overlap))
    for i, (label, off, tr) in enumerate(want):
        assert off == fv and len(tr) == lv - fv + 1, (label, off, fv, len(tr), lv)
        got = out[i, fv:lv + 1]
        assert np.array_equal(np.isnan(got), np.isnan(tr))
        assert np.nanmax(np.abs(got - tr)) < TOL


def test_classify_matches_oracle_and_demo_shape(model, oracle):
    """Single-component input (as Final_models/demo.ipynb:242,397-398) must work: missing
    components are zero-filled; picks/detections match the oracle."""
    from volpick_amd import Stream, Trace, UTCDateTime

    data, _, _ = synthetic_stream_array(6890, seed=77, n_events=1)
    t0 = UTCDateTime("2005-05-31T21:04:52.110000Z")
    st = Stream([Trace(data[0], dict(network="NC", station="MMT", location="", channel="EHZ", starttime=t0,
                                     sampling_rate=100.0))])
    res = model.classify(st, overlap=1000, blinding=[500, 500], P_threshold=0.15, S_threshold=0.15)
    zdata = np.zeros_like(data)
    zdata[0] = data[0]
    want = OP.classify_array(oracle, zdata, thresholds={"P": 0.15, "S": 0.15}, overlap=1000, blinding=(500, 500))
    assert len(res.picks) == len(want["picks"])
    for p, (ph, on, off, pk, v) in zip(res.picks, want["picks"]):
        assert p.phase == ph and p.trace_id == "NC.MMT."
        assert abs((p.peak_time - t0) * 100 - pk) <= 1
        assert abs(p.peak_value - v) < TOL
    assert len(res.detections) == len(want["detections"])
    assert str(res.picks).startswith(f"PickList with {len(res.picks)} entries:")
