"""CPU checks of the host side of libvolpick_hip: canonical weight order, BN folding,
polyphase A-matrix construction and MFMA fragment packing, validated layer by layer
against the torch oracle through a numpy emulation of the kernel's index arithmetic."""
import numpy as np
import pytest
import torch

from oracle.models import load_pretrained, WEIGHTS_DIR
from tests.emulator import flat_weights, plan_conv, emulate_conv
from volpick_amd import _lib


def _oracle_activations(net, x):
    acts = {}

    def hook(name):
        def f(mod, inp, out):
            acts[name] = out.detach()
        return f

    hs = [m.register_forward_hook(hook(n)) for n, m in net.named_modules() if n]
    with torch.no_grad():
        y = net(x)
    for h in hs:
        h.remove()
    return y, acts


@pytest.mark.parametrize("name", ["volpick", "volpick_95train"])
def test_phasenet_layers_match_oracle(lib, name):
    net = load_pretrained("phasenet", name)
    z = np.load(WEIGHTS_DIR / "phasenet" / f"{name}.npz")
    w = flat_weights(_lib.VP_MODEL_PHASENET, z)
    assert w.size == lib.vp_weight_count(_lib.VP_MODEL_PHASENET) == 269675
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 3, 3001)).astype(np.float32)
    y, acts = _oracle_activations(net, torch.from_numpy(x))

    relu = lambda t: torch.relu(t)[0].numpy()
    # expected outputs per conv layer (post BN + ReLU) and their inputs, in launch order
    exp = {"inc": relu(acts["in_bn"])}
    for i in range(5):
        exp[f"down{i}.same"] = relu(acts[f"down_branch.{i}.1"])
        if i < 4:
            exp[f"down{i}.down"] = relu(acts[f"down_branch.{i}.3"])
    skips = [exp[f"down{i}.same"] for i in range(4)]
    for j in range(4):
        full = relu(acts[f"up_branch.{j}.1"])
        skip = skips[3 - j]
        cropped = full[:, 1:-2]
        off = (cropped.shape[1] - skip.shape[1]) // 2
        exp[f"up{j}.convT"] = cropped[:, off:off + skip.shape[1]]
        exp[f"up{j}.same"] = relu(acts[f"up_branch.{j}.3"])
    exp["up3.same+out"] = exp.pop("up3.same")

    src = {"inc": x[0]}
    prev = "inc"
    order = []
    i = 0
    while True:
        L = plan_conv(_lib.VP_MODEL_PHASENET, w, i)
        if L is None:
            break
        order.append(L)
        i += 1
    assert [L["name"] for L in order] == [
        "inc", "down0.same", "down0.down", "down1.same", "down1.down", "down2.same", "down2.down", "down3.same",
        "down3.down", "down4.same", "up0.convT", "up0.same", "up1.convT", "up1.same", "up2.convT", "up2.same",
        "up3.convT", "up3.same+out"]
    inputs = {
        "inc": x[0], "down0.same": exp["inc"], "down0.down": exp["down0.same"], "down1.same": exp["down0.down"],
        "down1.down": exp["down1.same"], "down2.same": exp["down1.down"], "down2.down": exp["down2.same"],
        "down3.same": exp["down2.down"], "down3.down": exp["down3.same"], "down4.same": exp["down3.down"],
        "up0.convT": exp["down4.same"], "up0.same": np.concatenate([exp["down3.same"], exp["up0.convT"]]),
        "up1.convT": exp["up0.same"], "up1.same": np.concatenate([exp["down2.same"], exp["up1.convT"]]),
        "up2.convT": exp["up1.same"], "up2.same": np.concatenate([exp["down1.same"], exp["up2.convT"]]),
        "up3.convT": exp["up2.same"], "up3.same+out": np.concatenate([exp["down0.same"], exp["up3.convT"]]),
    }
    for L in order:
        got = emulate_conv(L, inputs[L["name"]])
        want = exp[L["name"]]
        assert got.shape == want.shape, L["name"]
        err = np.abs(got - want).max()
        assert err < 2e-4 * max(1.0, np.abs(want).max()), (L["name"], err)


def test_weight_table_matches_npz(lib):
    for kind, model in [(_lib.VP_MODEL_PHASENET, "phasenet"), (_lib.VP_MODEL_EQTRANSFORMER, "eqtransformer")]:
        z = np.load(WEIGHTS_DIR / model / "volpick.npz")
        names = [k for k in z.files if not k.endswith("num_batches_tracked")]
        assert lib.vp_param_count(kind) == len(names)
        for i, k in enumerate(names):
            assert lib.vp_param_name(kind, i).decode() == k
            assert lib.vp_param_size(kind, i) == z[k].size


def test_eqt_conv_layers_match_oracle(lib):
    import torch.nn.functional as F

    net = load_pretrained("eqtransformer")
    z = np.load(WEIGHTS_DIR / "eqtransformer" / "volpick.npz")
    w = flat_weights(_lib.VP_MODEL_EQTRANSFORMER, z)
    assert w.size == lib.vp_weight_count(_lib.VP_MODEL_EQTRANSFORMER) == 378823
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((1, 3, 6000)).astype(np.float32))
    layers = {}
    i = 0
    while True:
        L = plan_conv(_lib.VP_MODEL_EQTRANSFORMER, w, i)
        if L is None:
            break
        layers[L["name"]] = L
        i += 1
    assert len(layers) == 7 + 14 + 7

    def check(name, src, want, skip_last=0, **kw):
        got = emulate_conv(layers[name], src, **kw)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        if skip_last:
            got, want = got[:, :-skip_last], want[:, :-skip_last]
        err = np.abs(got - want).max()
        assert err < 2e-4 * max(1.0, np.abs(want).max()), (name, err)

    with torch.no_grad():
        # encoder: conv + relu (pre-pool) per stage
        h = x
        for s, (conv, pad) in enumerate(zip(net.encoder.convs, net.encoder.paddings)):
            y = torch.relu(conv(h))
            check(f"encoder.{s}", h[0].numpy(), y[0].numpy())
            if pad:
                y = F.pad(y, (0, pad), "constant", -1e10)
            h = F.max_pool1d(y, 2)
        # ResCNN
        for s, blk in enumerate(net.res_cnn_stack.members):
            a = torch.relu(blk.norm1(h))
            pad = (0, 1) if blk.right_pad else (0, 0)
            m = torch.relu(blk.norm2(blk.conv1(F.pad(a, pad))))
            check(f"res{s}.conv1", a[0].numpy(), m[0].numpy())
            c2 = blk.conv2(F.pad(m, pad))
            check(f"res{s}.conv2", m[0].numpy(), c2[0].numpy())
            h = h + c2
        bott = net.bottleneck(x)
        # decoders: three weight sets per stage, Upsample(2) folded into the conv (input = the NOT-upsampled rows).
        # Stage 2 crops the upsampled row by one sample: the folded conv is exact up to its last two outputs, which
        # decoder2_edge_kernel recomputes on the GPU (tests/test_gpu_eqt.py covers the result).
        decs = [net.decoder_d, net.pick_decoders[0], net.pick_decoders[1]]
        for d, dec in enumerate(decs):
            h = bott
            for s, conv in enumerate(dec.convs):
                u = F.interpolate(h, scale_factor=2, mode="nearest")
                if s in dec.crops:
                    u = u[:, :, :-1]
                y = torch.relu(conv(u))
                check(f"decoder.{s}", h[0].numpy(), y[0].numpy(), set_index=d, sets=3, skip_last=2 if s == 2 else 0)
                h = y
