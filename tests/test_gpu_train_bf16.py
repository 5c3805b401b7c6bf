"""GPU parity tests of the bf16-storage PhaseNet training step (BASELINE configs[4]: "bf16"):
`vp_train_create_dtype(..., VP_TRAIN_BF16, ...)` keeps x, z, a, gz, ga of every layer in memory as bfloat16 and
accumulates in fp32 (include/volpick_hip.h).

End to end the step is compared with torch autograd on the oracle module carrying the SAME rounding points
(oracle/bf16_emulation.py) and, loosely, with plain fp32 autograd.  With the released weights and a batch of six windows
the network is chaotic under storage rounding (one bf16 ulp in a thin BatchNorm channel flips ReLU gates downstream:
fp32 autograd and its own bf16 emulation differ by up to 0.8 of a tensor's max in single elements), so the sharp
check is LAYER BY LAYER: every kernel's output is recomputed by torch from the inputs that kernel actually read (the
trainer's own stored tensors) -- conv forward, BatchNorm + ReLU forward, BatchNorm backward, the input-gradient conv,
the weight gradient, the head -- and must agree to one bf16 ulp (stored tensors) or to fp32 summation accuracy
(weight gradients, loss).  No error can hide behind another layer's."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import constants as OC
from oracle.bf16_emulation import bf16_storage, round_bf16
from oracle.models import load_pretrained
from tests.test_gpu_train import make_batch, ref_loss, torch_step
from volpick_amd import PhaseNet
from volpick_amd.train import PhaseNetLit, PhaseNetTrainer

pytestmark = pytest.mark.gpu

ULP = 2.0 ** -8  # one unit in the last place of a bfloat16 significand (8 bits), relative


def close_bf16(got, want, what, abs_frac=2e-5, knife_edge=0.0):
    """got (read back from bf16 rows) against want (fp32, torch): one ulp of the value plus a floor relative to the tensor's
    max (sums of hundreds of products in a different order, then rounded: a value next to a rounding boundary may land on
    either side).  knife_edge: the fraction of elements (at least one when > 0) that may miss it altogether -- BatchNorm's
    backward takes its ReLU gate from fma(z, sc, sh) > 0; an element whose activation is zero to the last bit of the
    statistics gets the gate from whichever side its own sc / sh fall on (seen: 1 of 3.1 M at B = 512)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = ULP * np.abs(want) + abs_frac * np.abs(want).max() + 1e-30
    bad = np.abs(got - want) > tol
    allowed = int(max(1, knife_edge * bad.size)) if knife_edge > 0 else 0
    assert int(bad.sum()) <= allowed, (what, int(bad.sum()), float((np.abs(got - want) / tol).max()))
    return bad  # the elements let through under `knife_edge`: the caller proves that each of them IS one (assert_knife_edges)


def assert_knife_edges(bad, name, bn, z_stored, ga, gz_got, crop, abs_frac):
    """Every element of gz that close_bf16 let through under `knife_edge` must be a knife edge and nothing else:
    (1) its BatchNorm output u = fma(z, sc, sh) is zero to within two bf16 ulps of |z sc| + |sh| (sc, sh: the channel's scale and
        shift from the batch statistics of the stored z), i.e. the ReLU gate hangs on the last bits of the statistics;
    (2) its gz is torch's value for the OTHER gate state (the backward recomputed with exactly these elements' gates flipped),
        to the same tolerance every other element meets.
    An indexing bug that corrupts one element per tensor passes neither."""
    idx = np.argwhere(bad)
    z_in = torch.from_numpy(np.ascontiguousarray(z_stored)).clone().requires_grad_(True)
    u = bn(z_in)  # training mode: batch statistics of the stored z
    with torch.no_grad():
        mean = z_in.mean(dim=(0, 2))
        var = z_in.var(dim=(0, 2), unbiased=False)
        sc = bn.weight / torch.sqrt(var + bn.eps)
        sh = bn.bias - mean * sc
    for b, c, tpos in idx:
        zz, ss, hh = float(z_in[b, c, tpos].detach()), float(sc[c]), float(sh[c])
        assert abs(float(u[b, c, tpos].detach())) <= 2 * ULP * (abs(zz * ss) + abs(hh)) + 1e-30, \
            (name, "let through, but not a knife edge", (int(b), int(c), int(tpos)), float(u[b, c, tpos].detach()), zz * ss, hh)
    mask = (u.detach() > 0).to(torch.float32)
    for b, c, tpos in idx:
        mask[b, c, tpos] = 1.0 - mask[b, c, tpos]
    a_alt = u * mask
    if crop is not None:
        lo, La = crop
        a_alt = a_alt[:, :, lo: lo + La]
    (gz_alt,) = torch.autograd.grad(a_alt, z_in, ga)  # (not .backward(): the module's parameter gradients stay as the caller left them)
    want = gz_alt.numpy().astype(np.float64)
    got = np.asarray(gz_got, np.float64)
    tol = ULP * np.abs(want) + abs_frac * np.abs(want).max() + 1e-30
    for b, c, tpos in idx:
        assert abs(got[b, c, tpos] - want[b, c, tpos]) <= tol[b, c, tpos], \
            (name, "not the other gate's value either", (int(b), int(c), int(tpos)), got[b, c, tpos], want[b, c, tpos])


def assert_differences_sit_on_rounding_boundaries(conv, x_in, diff, got, what):
    """Where the kernel's stored bf16 conv output differs from torch's rounded one, it must still be THE nearest-even bf16 rounding
    of a value that two fp32 summation orders of the same exact products can produce: |got - z32| <= half a bf16 spacing at that
    magnitude + 2^-23 (sum |w||x| + |b|), z32 = torch's unrounded fp32 sum.  For a sum of ordinary size that means the unrounded
    value sits on a rounding boundary (the two sides land one ulp apart); for a sum that nearly cancels the slack of the summation
    order is all there is.  A wrong tap, a shifted sample or a corrupted element is neither."""
    with torch.no_grad():
        z32 = conv(x_in).numpy().astype(np.float64)
        absconv = torch.nn.functional.conv1d(x_in.abs(), conv.weight.abs(), None if conv.bias is None else conv.bias.abs(),
                                             padding=conv.padding).numpy().astype(np.float64)
    v, gv = z32[diff], np.asarray(got, np.float64)[diff]
    big = np.maximum(np.abs(v), np.abs(gv)).astype(np.float32)
    lo = (big.view(np.uint32) & np.uint32(0xFFFF0000))
    spacing = ((lo + np.uint32(0x10000)).view(np.float32).astype(np.float64) - lo.view(np.float32).astype(np.float64))
    slack = 2.0 ** -23 * absconv[diff] + 1e-30
    off = np.abs(gv - v)
    assert np.all(off <= spacing / 2 + slack), (what, int((off > spacing / 2 + slack).sum()), float((off / (spacing / 2 + slack)).max()))
    # ... and most of them ARE boundary cases: the unrounded value within the slack of the midpoint of two bf16 neighbours
    f = v.astype(np.float32)
    flo = (f.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32).astype(np.float64)
    mid = flo + np.sign(v) * spacing / 2
    assert np.mean(np.abs(v - mid) <= slack + spacing * 2.0 ** -10) > 0.9 or len(v) < 20, what


def is_bf16(a):
    return np.array_equal(round_bf16(torch.from_numpy(np.ascontiguousarray(a))).numpy(), a)


@pytest.fixture(scope="module")
def step():
    B = 6
    x, y = make_batch(B, 7)
    tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=8, dtype="bf16")
    assert tr.dtype == "bf16" and tr._lib.vp_train_dtype(tr._h) == 1
    loss = tr.step(x, y, lr=0.0, update=False)
    return B, x, y, tr, loss, tr.tensors(B), tr.gradients(), tr.predictions(B)


def test_rows_hold_bfloat16_values(step):
    B, x, y, tr, loss, t, g, pred = step
    for name, a in t.items():
        assert is_bf16(a), name
    assert np.array_equal(t["x"], round_bf16(torch.from_numpy(x)).numpy())  # nearest-even on the way in
    assert not is_bf16(g["inc.weight"]) and not is_bf16(pred)  # gradients and predictions are fp32


def test_end_to_end_against_autograd_with_the_same_rounding_points(step):
    B, x, y, tr, loss, t, g, pred = step
    with bf16_storage(load_pretrained("phasenet")) as net:
        want_loss, grads, z, gz, want_pred = torch_step(net, x, y)
    loss32, *_ = torch_step(load_pretrained("phasenet"), x, y)
    assert abs(loss - want_loss) < 1e-3 * want_loss, (loss, want_loss)
    assert abs(loss - loss32) < 1e-2 * loss32, (loss, loss32)  # what the storage format itself costs: 2e-3 here
    # the first conv sees identical inputs: the same bf16 values, up to the few sums that land on a rounding boundary (the
    # bf16-MFMA form adds its exact products in another order than torch's conv: one bf16 ulp there, nothing else)
    diff = t["inc.z"] != z["inc"]
    assert diff.mean() < 0.02, diff.mean()
    assert np.all(np.abs(t["inc.z"] - z["inc"])[diff] <= 2.0 ** -7 * np.abs(z["inc"])[diff] + 1e-30)
    assert_differences_sit_on_rounding_boundaries(load_pretrained("phasenet").inc, round_bf16(torch.from_numpy(x)), diff, t["inc.z"], "inc.z")
    d = np.abs(pred - want_pred)
    assert np.median(d) < 1e-4 and np.percentile(d, 99) < 5e-3, (np.median(d), np.percentile(d, 99))
    for name in z:  # bulk agreement; single elements diverge (module docstring)
        e = np.abs(t[name + ".z"] - z[name]) / np.abs(z[name]).max()
        assert np.median(e) < 2e-3 and np.percentile(e, 99) < 3e-2, (name, np.median(e), np.percentile(e, 99))
        e = np.abs(t[name + ".gz"] - gz[name]) / np.abs(gz[name]).max()
        assert np.median(e) < 5e-3 and np.percentile(e, 99) < 1e-1, (name + ".gz", np.median(e), np.percentile(e, 99))


def _layers(net):
    """(name, conv module, bn module, kind, input tensor names, where its input gradient goes)"""
    out = [("inc", net.inc, net.in_bn, "same", ["x"], None)]
    prev = "inc"
    for i, (same, bn1, down, bn2) in enumerate(net.down_branch):
        n = f"down{i}.same"
        out.append((n, same, bn1, "same", [prev + ".a"], prev + ".ga"))
        prev = n
        if down is not None:
            d = f"down{i}.down"
            out.append((d, down, bn2, ("down", i), [n + ".a"], f"down{i}.gskip"))
            prev = d
    for j, (up, bn1, same, bn2) in enumerate(net.up_branch):
        u = f"up{j}.convT"
        out.append((u, up, bn1, "convT", [prev + ".a"], prev + ".ga"))
        s = f"up{j}.same"
        out.append((s, same, bn2, "same", [f"down{3 - j}.same.a", u + ".a"], f"up{j}.gcat"))
        prev = s
    return out


def test_every_kernel_against_torch_on_the_inputs_it_read(step):
    check_every_kernel(*step)


def check_every_kernel(B, x, y, tr, loss, t, g, pred, net=None):
    """Every launch of one step against torch on the inputs that launch read (the trainer's stored tensors `t`, its
    gradients `g`): shared with tests/test_gpu_train_large_batch.py (B = 128 / 512, the forms the bench times)."""
    net = (net if net is not None else load_pretrained("phasenet")).train()
    T = {k: torch.from_numpy(v) for k, v in t.items()}
    pname = {id(p): k for k, p in net.named_parameters()}
    # gradient wrt a of every layer = what its consumers wrote (skips: up path + strided conv below)
    ga = {}
    for i in range(4):
        ch = 8 * 2**i
        ga[f"down{i}.same"] = T[f"up{3 - i}.gcat"][:, :ch] + T[f"down{i}.gskip"]
        ga[f"up{3 - i}.convT"] = T[f"up{3 - i}.gcat"][:, ch:]
    for n in ("inc", "down0.down", "down1.down", "down2.down", "down3.down", "down4.same", "up0.same", "up1.same", "up2.same",
              "up3.same"):
        ga[n] = T[n + ".ga"]
    for name, conv, bn, kind, srcs, grad_dst in _layers(net):
        xin = torch.cat([T[s] for s in srcs], dim=1).clone().requires_grad_(True)
        for p in list(conv.parameters()) + list(bn.parameters()):
            p.grad = None
        # ---- conv forward on the stored input -----------------------------------------------------------------
        h = xin
        if isinstance(kind, tuple) and kind[1] > 0:
            h = F.pad(h, OC.PN_DOWN_PAD[kind[1]], "constant", 0.0)
        z_ref = conv(h)
        close_bf16(t[name + ".z"], z_ref.detach().numpy(), name + ".z")
        # ---- BatchNorm (batch statistics) + ReLU on the stored z --------------------------------------------------
        z_in = T[name + ".z"].clone().requires_grad_(True)
        a_ref = torch.relu(bn(z_in))
        crop = None
        if kind == "convT":
            a_ref = a_ref[:, :, OC.PN_UP_CROP[0]: a_ref.shape[-1] - OC.PN_UP_CROP[1]]
            La = t[name + ".a"].shape[-1]
            off = (a_ref.shape[-1] - La) // 2
            a_ref = a_ref[:, :, off: off + La]
            crop = (OC.PN_UP_CROP[0] + off, La)
        close_bf16(t[name + ".a"], a_ref.detach().numpy(), name + ".a", abs_frac=1e-4)
        # ---- BatchNorm backward from the stored ga: gz, d gamma, d beta -------------------------------------------
        # (gates from the STORED a, as the kernel takes them: rounding never turns a positive value into zero)
        a_ref.backward(ga[name])
        let_through = close_bf16(t[name + ".gz"], z_in.grad.numpy(), name + ".gz", abs_frac=2e-4, knife_edge=2e-6)
        if let_through.any():  # (the same bn module: its running statistics move, nothing the checks below read)
            assert_knife_edges(let_through, name, bn, t[name + ".z"], ga[name], t[name + ".gz"], crop, abs_frac=2e-4)
        for p in bn.parameters():
            want = p.grad.numpy()
            assert np.abs(g[pname[id(p)]] - want).max() < 2e-4 * np.abs(want).max() + 1e-7, (name, pname[id(p)])
        # ---- input gradient and weight gradient from the stored gz --------------------------------------------------
        z_ref.backward(T[name + ".gz"])
        if grad_dst is not None:
            close_bf16(t[grad_dst], xin.grad.numpy(), grad_dst, abs_frac=1e-4)
        want = conv.weight.grad.numpy()
        assert np.abs(g[pname[id(conv.weight)]] - want).max() < 1e-4 * np.abs(want).max(), (name, "weight gradient")
    # ---- head: 1x1 conv + softmax + vector cross entropy and its backward ---------------------------------------------
    a17 = T["up3.same.a"].clone().requires_grad_(True)
    for p in net.out.parameters():
        p.grad = None
    p_ref = torch.softmax(net.out(a17), dim=1)
    l_ref = ref_loss(p_ref, torch.from_numpy(y))
    l_ref.backward()
    assert abs(loss - l_ref.item()) < 2e-6 * l_ref.item()
    assert np.abs(pred - p_ref.detach().numpy()).max() < 2e-5
    close_bf16(t["up3.same.ga"], a17.grad.numpy(), "up3.same.ga", abs_frac=1e-4)
    # the head's weight / bias gradients are sums over B x 3001 terms that nearly cancel (the three classes' bias gradients add up
    # to zero): the kernel finishes them in fp64, torch's fp32 sums are 1.2e-4 off at B = 512 -- so the reference is taken in fp64
    import copy

    out64 = copy.deepcopy(net.out).double()
    for p in out64.parameters():
        p.grad = None
    ref_loss(torch.softmax(out64(T["up3.same.a"].double()), dim=1), torch.from_numpy(y).double()).backward()
    for k in ("weight", "bias"):
        want = getattr(out64, k).grad.numpy()
        assert np.abs(g["out." + k].reshape(want.shape) - want).max() < 1e-4 * np.abs(want).max(), "out." + k


def test_large_batch_takes_the_vector_batchnorm_path_for_every_crop():
    """At B = 40 the 32-channel layers (188 / 191 samples, ConvTranspose crop 1) leave the one-workgroup-per-channel
    BatchNorm kernels for the vectorised ones: forward activations against torch on the stored z, layer by layer."""
    B = 40
    x, y = make_batch(B, 11)
    tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=B, dtype="bf16")
    tr.step(x, y, lr=0.0, update=False)
    t = tr.tensors(B)
    net = load_pretrained("phasenet").train()
    for name, conv, bn, kind, srcs, grad_dst in _layers(net):
        a_ref = torch.relu(bn(torch.from_numpy(t[name + ".z"])))
        if kind == "convT":
            a_ref = a_ref[:, :, OC.PN_UP_CROP[0]: a_ref.shape[-1] - OC.PN_UP_CROP[1]]
            La = t[name + ".a"].shape[-1]
            off = (a_ref.shape[-1] - La) // 2
            a_ref = a_ref[:, :, off: off + La]
        close_bf16(t[name + ".a"], a_ref.detach().numpy(), name + ".a", abs_frac=1e-4)


def test_loss_curve_from_a_random_initialisation_tracks_fp32_torch():
    """40 Adam steps with the reference's warm-up from the same random weights over the same batches: the bf16-storage
    step follows the fp32 torch curve (fp32 master weights and Adam state: only the activations are rounded)."""
    from oracle.models import PhaseNet as TorchPhaseNet

    torch.manual_seed(1)
    net = TorchPhaseNet(phases="PSN", norm="peak").train()
    model = PhaseNet(phases="PSN", norm="peak")
    model.load_state_dict({k: v.detach().numpy() for k, v in net.state_dict().items()})
    lit = PhaseNetLit(lr=1e-3, max_batch=8, model=model, precision="bf16-mixed")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    batches = [make_batch(8, 900 + i) for i in range(4)]
    ours, theirs = [], []
    for k in range(40):
        xb, yb = batches[k % 4]
        ours.append(lit.training_step({"X": xb, "y": yb}, k))
        for pg in opt.param_groups:
            pg["lr"] = lit.learning_rate(k)
        l, *_ = torch_step(net, xb, yb, opt=opt)
        theirs.append(l)
    ours, theirs = np.array(ours), np.array(theirs)
    assert lit.trainer_state.dtype == "bf16"
    assert abs(ours[0] - theirs[0]) < 2e-3 * theirs[0]
    assert np.abs(ours - theirs).max() < 2e-2 * theirs.max(), (ours[-5:], theirs[-5:])
    assert ours[-1] < 0.97 * ours[0]


def test_dtype_argument_errors():
    with pytest.raises(ValueError, match="dtype"):
        PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=4, dtype="fp8")
