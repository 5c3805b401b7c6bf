"""Round 4: the advisor's findings as tests.
 * eqt_mid_kernel<2> pairs two windows in one workgroup; which form a window's attention scores take (E_q E_k or plain tanh)
   must not depend on its partner (team_vote_or);
 * what vp_step_issued_work reports must not depend on which launch ran last (Step::issued_for_range is pure);
 * a vp_profile_* call after the stream it last saw has been freed must not read freed memory (pre_is_replayable)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.models import load_pretrained
from volpick_amd import EQTransformer, _lib
from volpick_amd.synthetic import synthetic_stream_array, synthetic_windows

pytestmark = pytest.mark.gpu


def _scaled_attention(scale, flags=()):
    m = EQTransformer.from_pretrained("volpick")
    sd = m.state_dict()
    for key in ("transformer_d0.attention.Wt", "transformer_d0.attention.Wx"):  # the first attention layer only: its input does not move
        sd[key] = sd[key] * np.float32(scale)
    m.load_state_dict(sd)
    m._plan_flags = flags
    return m.cuda(), sd


def test_a_windows_attention_form_does_not_depend_on_its_workgroup_partner():
    """Attention projections scaled until SOME windows of the batch leave the |q|, |k| <= 30 guard of the E_q E_k form and
    others stay inside (checked on the oracle's activations): pairs (2i, 2i + 1) of eqt_mid_kernel<2> (plan_flags[2] = 3) and the
    quadruples of eqt_mid4_kernel (the default) are then mixed.  Every window must come out bit-identical to the
    one-window-per-workgroup plan (plan_flags[2] = 2), whatever it shares a workgroup with."""
    B = 64
    x = synthetic_windows(B, 6000, seed=4242)
    x[1::4] *= 0.02  # quiet windows between loud ones: a spread of activation sizes at the attention input
    oracle = load_pretrained("eqtransformer")
    seen = {}
    hooks = [getattr(oracle, name).attention.register_forward_hook(lambda mod, inp, out, name=name: seen.__setitem__(name, inp[0].detach()))
             for name in ("transformer_d0",)]
    import oracle.pipeline as OP

    with torch.no_grad():
        oracle(OP.batch_pre(oracle, torch.from_numpy(x)))
    for hk in hooks:
        hk.remove()
    # per window: max |x Wt|, |x Wx + bh| of the first attention layer, per unit of scale
    att, xin = oracle.transformer_d0.attention, seen["transformer_d0"]
    xin = xin if xin.shape[-1] == att.Wt.shape[0] else xin.transpose(1, 2)  # (B, T, 16)
    with torch.no_grad():
        top = lambda t: t.abs().amax(dim=(1, 2)).numpy()
        scale = 30.0 / np.median(np.maximum(top(xin @ att.Wt), top(xin @ att.Wx)))  # about half of the windows beyond the guard
        worst = np.maximum(top(xin @ (att.Wt * scale)), top(xin @ (att.Wx * scale) + att.bh))
    big, small = worst > 30.0 * 1.02, worst < 30.0 * 0.98
    mixed_pairs = int(sum((big[2 * i] and small[2 * i + 1]) or (small[2 * i] and big[2 * i + 1]) for i in range(B // 2)))
    assert mixed_pairs >= 4, (mixed_pairs, np.sort(worst))
    mixed_quads = int(sum(big[4 * i:4 * i + 4].any() and small[4 * i:4 * i + 4].any() for i in range(B // 4)))
    assert mixed_quads >= 4, (mixed_quads, np.sort(worst))
    one, _ = _scaled_attention(scale, (0, 0, 2))
    xt = torch.from_numpy(x).cuda()
    b = one._forward_raw(xt, preprocess=True)
    for flags in ((), (0, 0, 3)):
        many, _ = _scaled_attention(scale, flags)
        a = many._forward_raw(xt, preprocess=True)
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), flags
        # ... and whatever a window is paired with: the same windows shifted by one (every window gets other partners)
        c = many._forward_raw(torch.roll(xt, 1, 0).contiguous(), preprocess=True)
        assert torch.equal(torch.roll(c, -1, 0), a), flags
        many._release()
    one._release()


def _issued(model, index, rng=None):
    lib, w = _lib.load(), _lib.VpIssuedWork()
    if rng is None:
        _lib.check(lib.vp_step_issued_work(model._handle, index, C.byref(w)))
    else:
        _lib.check(lib.vp_step_issued_work_for_range(model._handle, index, rng[0], rng[1], C.byref(w)))
    return (w.mfma_f32_flop, w.mfma_bf16_flop, w.valu_flop)


def test_issued_work_of_the_decoder_tail_is_a_pure_function_of_the_kept_range():
    m = EQTransformer.from_pretrained("volpick").cuda()
    lib = _lib.load()
    n = lib.vp_step_count(m._handle)
    name = C.c_char_p()
    lib.vp_step_info(m._handle, n - 1, C.byref(name), None)
    assert name.value.decode().startswith("fused.tail")
    whole, blinded = _issued(m, n - 1, (0, 0)), _issued(m, n - 1, (500, 5500))
    assert whole[1] > blinded[1] > 0 and whole[1] / blinded[1] == pytest.approx(611.9 / 513.1, rel=2e-3)  # DESIGN.md section 4 (four 1256-sample tiles; rounds 3-5: 541.5 with 1264-sample tiles)
    data, _, _ = synthetic_stream_array(6000 + 500 * 9, seed=3, n_events=2)
    x = synthetic_windows(3, 6000, seed=5)
    for _ in range(2):  # whichever launch ran last, the answers for a GIVEN range stay the same
        m._annotate_block(data, m._argdict(dict(overlap=5500, blinding=(500, 500), stacking="avg")))
        assert _issued(m, n - 1) == blinded  # the range the latest preprocessing batch kept = what vp_profile_* would time
        assert (_issued(m, n - 1, (0, 0)), _issued(m, n - 1, (500, 5500))) == (whole, blinded)
        m._forward_raw(x, preprocess=True)
        assert _issued(m, n - 1) == whole
        assert (_issued(m, n - 1, (0, 0)), _issued(m, n - 1, (500, 5500))) == (whole, blinded)
    for i in range(n - 1):  # launches without a range-dependent tiling: both calls agree
        assert _issued(m, i) == _issued(m, i, (500, 5500))
    m._release()


@pytest.mark.parametrize("cls_name", ["PhaseNet", "EQTransformer"])
def test_profile_after_the_source_stream_is_gone_reads_no_freed_memory(cls_name):
    """vp_classify on a caller-owned device stream, the stream freed (really freed: hipFree, not torch's cache), then the
    profile calls: they must notice and time the launches on the handle's own input tensor instead."""
    import volpick_amd as va

    cls = getattr(va, cls_name)
    m = cls.from_pretrained("volpick").cuda()
    lib, T = _lib.load(), cls.in_samples
    # the HIP runtime this process already runs on (torch's bundled copy): never a second one
    hip = C.CDLL(next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln))
    n = T + (T - 1500) * 7
    data, _, _ = synthetic_stream_array(n, seed=9)
    src = C.c_void_p()
    assert hip.hipMalloc(C.byref(src), C.c_size_t(3 * n * 4)) == 0
    assert hip.hipMemcpy(src, data.ctypes.data_as(C.c_void_p), C.c_size_t(3 * n * 4), 1) == 0
    out = torch.empty((3, n), dtype=torch.float32, device="cuda")
    fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.vp_annotate(m._handle, src, _lib.VP_MEM_DEVICE, n, 1500, 0, 0, _lib.VP_STACK_AVG, 8, C.c_void_p(out.data_ptr()),
                               _lib.VP_MEM_DEVICE, C.byref(fv), C.byref(lv), C.byref(nw)))
    assert nw.value == 8
    steps = lib.vp_step_count(m._handle)
    ms = (C.c_float * steps)()
    _lib.check(lib.vp_profile_steps(m._handle, 8, 2, ms, steps))  # the stream is alive: replayed
    torch.cuda.synchronize()
    assert hip.hipFree(src) == 0
    _lib.check(lib.vp_profile_steps(m._handle, 8, 2, ms, steps))
    one = C.c_float()
    _lib.check(lib.vp_profile_step_in_pipeline(m._handle, 8, 3, 0, C.byref(one)))
    _lib.check(lib.vp_profile_one_step(m._handle, 8, 3, 0, C.byref(one)))
    torch.cuda.synchronize()
    assert all(v > 0 for v in ms) and one.value > 0
    x = synthetic_windows(2, T, seed=1)
    assert torch.isfinite(m._forward_raw(torch.from_numpy(x).cuda(), preprocess=True)).all()  # the handle is still good
    m._release()


def test_first_come_first_served_forward_launches_change_no_result():
    """PhaseNet's device contexts queue their forward launches first come, first served (csrc/api.hip ForwardGate: launch n + 2
    waits for the end of launch n); plan_flags[3] = 64 switches the gate off.  Same numbers either way: a day-long block over
    all contexts (the stacked rows, bitwise) and many station blocks pipelined over the contexts (the picks)."""
    import volpick_amd as va

    gated = va.PhaseNet.from_pretrained("volpick").cuda()
    free = va.PhaseNet.from_pretrained("volpick")
    free._plan_flags = (0, 0, 0, 64)
    free.cuda()
    data, _, _ = synthetic_stream_array(3001 + 1501 * 2300, seed=21, n_events=120)
    args = gated._argdict(dict(overlap=1500, blinding=(0, 0), stacking="avg", batch_size=256))
    assert gated._is_long(data.shape[1], args)
    a, fva, lva, nwa = gated._annotate_segments(data, args)
    b, fvb, lvb, nwb = free._annotate_segments(data, args)
    assert (fva, lva, nwa) == (fvb, lvb, nwb) and nwa == 2301
    assert torch.equal(a, b)
    t0 = va.UTCDateTime("2021-01-01T00:00:00")
    st = va.Stream()
    for k in range(9):
        d, _, _ = synthetic_stream_array(60_000, seed=300 + k, n_events=5)
        for i, c in enumerate("ZNE"):
            st.append(va.Trace(d[i], dict(network="XX", station=f"S{k:02d}", location="", channel=f"HH{c}", starttime=t0, sampling_rate=100.0)))
    key = lambda res: [(p.trace_id, p.phase, p.peak_time.timestamp, p.peak_value) for p in res.picks]
    pa, pb = key(gated.classify(st)), key(free.classify(st))
    assert len(pa) > 20 and pa == pb
    gated._release(), free._release()
