"""GPU parity tests of the PhaseNet training step: vp_train_step against torch
autograd / torch.optim.Adam on the CPU oracle module in train() mode, same weights and batch."""
import numpy as np
import pytest
import torch

from oracle import pipeline as OP
from oracle.models import load_pretrained
from volpick_amd import PhaseNet
from volpick_amd.synthetic import synthetic_windows
from volpick_amd.train import PhaseNetLit, PhaseNetTrainer, gaussian_labels, vector_cross_entropy

pytestmark = pytest.mark.gpu


def ref_loss(y_pred, y_true, eps=1e-5):  # volpick/model/models.py:34-51
    h = y_true * torch.log(y_pred + eps)
    return -h.mean(-1).sum(-1).mean()


def make_batch(B, seed):
    rng = np.random.default_rng(seed)
    x = synthetic_windows(B, 3001, seed=seed)
    net = load_pretrained("phasenet")
    xn = OP.batch_pre(net, torch.from_numpy(x)).numpy()
    p = rng.integers(300, 1500, B).astype(float)
    s = p + rng.integers(200, 1200, B)
    s[::3] = np.nan  # some windows without an S pick
    return xn.astype(np.float32), gaussian_labels(p, s)


def torch_step(net, x, y, lr=None, opt=None):
    """One torch training step; returns loss, grads, intermediate z (conv outputs) and their grads."""
    net.train()
    zs, names = {}, {}
    convs = [("inc", net.inc)]
    for i, (same, bn1, down, bn2) in enumerate(net.down_branch):
        convs.append((f"down{i}.same", same))
        if down is not None:
            convs.append((f"down{i}.down", down))
    for j, (up, bn1, same, bn2) in enumerate(net.up_branch):
        convs.append((f"up{j}.convT", up))
        convs.append((f"up{j}.same", same))
    hooks = []
    for name, mod in convs:
        def hook(m, inp, out, name=name):
            out.retain_grad()
            zs[name] = out
        hooks.append(mod.register_forward_hook(hook))
    net.zero_grad()
    pred = net(torch.from_numpy(x))
    loss = ref_loss(pred, torch.from_numpy(y))
    loss.backward()
    for h in hooks:
        h.remove()
    grads = {k: v.grad.detach().numpy().copy() for k, v in net.named_parameters()}
    z = {k: v.detach().numpy().copy() for k, v in zs.items()}
    gz = {k: v.grad.detach().numpy().copy() for k, v in zs.items()}
    if opt is not None:
        opt.step()
    return float(loss.detach()), grads, z, gz, pred.detach().numpy()


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def setup():
    B = 6
    x, y = make_batch(B, 7)
    model = PhaseNet.from_pretrained("volpick")
    tr = PhaseNetTrainer(model, max_batch=8)
    net = load_pretrained("phasenet")
    return B, x, y, tr, net


def test_loss_activations_and_gradients_match_autograd(setup):
    B, x, y, tr, net = setup
    loss = tr.step(x, y, lr=0.0, update=False)
    want_loss, grads, z, gz, pred = torch_step(net, x, y)
    assert abs(loss - want_loss) < 2e-6 * max(1.0, abs(want_loss)), (loss, want_loss)
    assert abs(vector_cross_entropy(pred.astype(np.float64), y.astype(np.float64)) - want_loss) < 1e-6
    assert np.abs(tr.predictions(B) - pred).max() < 2e-5
    t = tr.tensors(B)
    report = []
    for name in z:
        report.append((name + ".z", rel(t[name + ".z"], z[name])))
        report.append((name + ".gz", rel(t[name + ".gz"], gz[name])))
    g = tr.gradients()
    for k, w in grads.items():
        report.append(("grad " + k, rel(g[k], w)))
    print("\n" + "\n".join(f"{n:40s} rel err {e:.2e}" for n, e in report))
    for n, e in report:
        if n == "grad inc.bias":  # mathematically zero (BatchNorm removes the mean): both sides hold rounding noise
            assert np.abs(g["inc.bias"]).max() < 1e-4 and np.abs(grads["inc.bias"]).max() < 1e-4
            continue
        assert e < 5e-4, (n, e)


def test_adam_steps_and_running_statistics_match_torch(setup):
    B, x, y, _, _ = setup
    model = PhaseNet.from_pretrained("volpick")
    tr = PhaseNetTrainer(model, max_batch=8)
    net = load_pretrained("phasenet")
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x2, y2 = make_batch(B, 8)
    for step, (xb, yb, lr) in enumerate([(x, y, 1e-3), (x2, y2, 5e-4), (x, y2, 1e-3)]):
        for pg in opt.param_groups:
            pg["lr"] = lr
        want_loss, *_ = torch_step(net, xb, yb, opt=opt)
        loss = tr.step(xb, yb, lr=lr)
        assert abs(loss - want_loss) < 1e-4 * max(1.0, abs(want_loss)), (step, loss, want_loss)
    w = tr.weights()
    sd = net.state_dict()
    # Adam's step m / (sqrt(v) + eps) is scale-free: an element whose gradient sits at the fp32 noise floor moves
    # by ~lr in a direction the noise decides, on either side.  So: the bulk of every tensor must agree to 2 % of
    # one step, no element may differ by more than a fraction of the distance travelled, and the moments (linear
    # in the gradients) are compared tightly below.
    travelled = 1e-3 + 5e-4 + 1e-3
    worst, n_off, n_all = 0.0, 0, 0
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue  # compared through export() below
        err = np.abs(w[k] - v.numpy())
        if k == "inc.bias":  # its gradient is pure rounding noise (BatchNorm cancels the bias: nothing else sees it)
            assert err.max() < 2 * travelled * 1.05
            continue
        if k == "in_bn.running_mean":  # the batch mean of inc's output carries inc.bias, momentum 0.1
            assert err.max() < 0.3 * np.abs(w["inc.bias"] - sd["inc.bias"].numpy()).max() + 5e-5
            continue
        worst = max(worst, float(err.max()))
        n_off += int((err > 2e-5).sum())
        n_all += err.size
        assert err.max() < 0.25 * travelled, (k, float(err.max()))
    print(f"max |w - w_torch| after 3 Adam steps: {worst:.2e}; elements off by more than 2e-5: {n_off} of {n_all}")
    assert n_off < 0.002 * n_all
    m, v = tr.adam_state()
    st = opt.state_dict()["state"]
    for i, (k, p) in enumerate(net.named_parameters()):
        if k == "inc.bias":
            continue
        # steps 2-3 see weights that already differ at the noise level, and with six windows per batch the BatchNorm
        # statistics amplify that (tests/test_gpu_train_bf16.py): 1.0e-2 was observed for single tensors
        assert rel(m[k], st[i]["exp_avg"].numpy()) < 2e-2, k
        assert rel(v[k], st[i]["exp_avg_sq"].numpy()) < 2e-2, k
    # the trained weights drive the inference path
    out = tr.export()
    assert np.array_equal(out.state_dict()["in_bn.running_mean"], w["in_bn.running_mean"])
    assert int(out.state_dict()["in_bn.num_batches_tracked"]) == int(sd["in_bn.num_batches_tracked"])
    net2 = load_pretrained("phasenet")
    net2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in out.state_dict().items()}, strict=True)
    net2.eval()
    with torch.no_grad():
        want = net2(torch.from_numpy(x)).numpy()
    got = out.cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


def test_training_reduces_the_loss_and_lit_schedule():
    lit = PhaseNetLit(lr=1e-3, max_batch=16, model=PhaseNet.from_pretrained("volpick"))
    assert lit.learning_rate(0) == 1e-3 and abs(lit.learning_rate(1) - 1e-3 * 1 / 500) < 1e-12
    assert lit.learning_rate(499) == pytest.approx(1e-3 * 499 / 500) and lit.learning_rate(500) == 1e-3
    x, y = make_batch(16, 21)
    batch = {"X": torch.from_numpy(x).cuda(), "y": torch.from_numpy(y).cuda()}  # device-resident batch
    first = lit.training_step(batch, 0)
    for i in range(1, 40):
        last = lit.training_step(batch, i)
    assert np.isfinite(first) and np.isfinite(last) and last < 0.7 * first, (first, last)
    val = lit.validation_step({"X": x, "y": y})
    assert np.isfinite(val)


def test_ema_tracks_the_weights():
    tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=8)
    with pytest.raises(Exception, match="EMA is off"):
        tr.ema_weights()
    tr.enable_ema(0.9)
    x, y = make_batch(8, 5)
    want = {k: v.astype(np.float64) for k, v in tr.weights().items()}
    for _ in range(3):
        tr.step(x, y, 1e-3)
        w = tr.weights()
        for k in want:
            want[k] = w[k] if "running_" in k else 0.9 * want[k] + 0.1 * w[k]
    got = tr.ema_weights()
    for k in want:
        assert np.abs(got[k] - want[k]).max() < 1e-6, k
    assert np.abs(got["inc.weight"] - w["inc.weight"]).max() > 1e-5  # it lags behind the live weights


def test_argument_errors():
    tr = PhaseNetTrainer(PhaseNet.from_pretrained("volpick"), max_batch=4)
    x, y = make_batch(4, 3)
    with pytest.raises(ValueError):
        tr.step(x[:, :, :3000], y[:, :, :3000], 1e-3)
    with pytest.raises(Exception, match="batch"):
        tr.step(np.concatenate([x, x]), np.concatenate([y, y]), 1e-3)  # larger than max_batch
    with pytest.raises(Exception, match="batch"):
        tr.step(x[:1], y[:1], 1e-3)  # BatchNorm statistics need more than one window


def test_loss_curve_tracks_torch_from_a_random_initialisation():
    """40 Adam steps (with the reference's warm-up) from the SAME random weights over the same batches: the HIP
    step and torch autograd + torch.optim.Adam (CPU) follow the same loss curve."""
    from oracle.models import PhaseNet as TorchPhaseNet

    torch.manual_seed(1)
    net = TorchPhaseNet(phases="PSN", norm="peak").train()
    model = PhaseNet(phases="PSN", norm="peak")
    model.load_state_dict({k: v.detach().numpy() for k, v in net.state_dict().items()})
    lit = PhaseNetLit(lr=1e-3, max_batch=8, model=model)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    batches = [make_batch(8, 900 + i) for i in range(4)]
    ours, theirs = [], []
    for k in range(40):
        x, y = batches[k % 4]
        ours.append(lit.training_step({"X": x, "y": y}, k))
        for pg in opt.param_groups:
            pg["lr"] = lit.learning_rate(k)
        loss, *_ = torch_step(net, x, y, opt=opt)
        theirs.append(loss)
    ours, theirs = np.array(ours), np.array(theirs)
    assert abs(ours[0] - theirs[0]) < 1e-6 * theirs[0]
    assert np.abs(ours - theirs).max() < 2e-3 * theirs.max(), (ours[-5:], theirs[-5:])
    assert ours[-1] < 0.97 * ours[0]
