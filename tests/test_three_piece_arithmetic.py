"""The arithmetic the bf16-piece kernels rest on (volpick_amd/csrc/conv_b3.h), checked in numpy / torch on the host:

* an fp32 number IS the sum of three bfloat16 pieces  hi = rne(x), mid = rne(x - hi), lo = x - hi - mid;
* bf16 x bf16 products are exact in fp32;
* of the nine piece products of w x, the six with i + j <= 2 -- accumulated in fp32 as the kernels do, smallest first
  -- reproduce a K = 32 dot product to fp32 rounding (what is dropped is of the order of 2^-24 |w||x|);
* the packing of two taps of a 16-channel layer into one K = 32 step (B3Steps) is a re-indexing of the same sum.

CPU only: the GPU parity tests (tests/test_gpu_eqt.py, test_gpu_phasenet.py) compare the kernels themselves."""
import numpy as np
import torch


def rne_bf16(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def split3(x):
    x = np.asarray(x, np.float32)
    hi = rne_bf16(x)
    r1 = (x - hi).astype(np.float32)
    mid = rne_bf16(r1)
    lo = (r1 - mid).astype(np.float32)
    return hi, mid, lo


def samples(n, seed):
    rng = np.random.default_rng(seed)
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-20, 8, n)
    x = np.ldexp(mant, expo).astype(np.float32)
    x[::97] = 0.0
    x[1::101] = np.float32(1.0) + np.float32(2.0 ** -23)  # all 24 significant bits in use
    return x


def test_three_bfloat16_pieces_carry_an_fp32_number_exactly():
    x = samples(200_000, 1)
    hi, mid, lo = split3(x)
    assert np.array_equal(rne_bf16(lo), lo), "the third piece needs no rounding: it is a bfloat16 already"
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_bf16_products_are_exact_in_fp32():
    a, b = rne_bf16(samples(100_000, 2)), rne_bf16(samples(100_000, 3))
    assert np.array_equal((a * b).astype(np.float64), a.astype(np.float64) * b.astype(np.float64))


def six_product_dot(w, x):
    """One output of a K-step as the kernels compute it: for every (w piece, x piece) pair with i + j <= 2, smallest
    products first, a K = 32 dot product accumulated in fp32 (the MFMA's accumulator), chained over the six."""
    wp, xp = split3(w), split3(x)
    acc = np.zeros(w.shape[:-1], np.float32)
    for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
        # products exact in fp32 (previous test); the 32-term sum of one instruction in float64, rounded once
        acc = (acc.astype(np.float64) + (wp[i].astype(np.float64) * xp[j].astype(np.float64)).sum(-1)).astype(np.float32)
    return acc


def test_six_products_reproduce_the_fp32_dot_product():
    rng = np.random.default_rng(4)
    w = (rng.standard_normal((20_000, 32)) * 0.2).astype(np.float32)
    x = np.maximum(rng.standard_normal((20_000, 32)), 0).astype(np.float32)  # ReLU outputs, as in the network
    exact = (w.astype(np.float64) * x.astype(np.float64)).sum(-1)
    scale = (np.abs(w).astype(np.float64) * np.abs(x).astype(np.float64)).sum(-1)
    got = six_product_dot(w, x).astype(np.float64)
    fp32 = (w * x).astype(np.float32)  # an fp32 FMA chain in the reference's order
    chain = np.zeros(len(w), np.float32)
    for k in range(32):
        chain = (chain + fp32[:, k]).astype(np.float32)
    err_six, err_chain = np.abs(got - exact) / scale, np.abs(chain.astype(np.float64) - exact) / scale
    # the dropped products are bounded by 3 * 2^-25 of sum |w||x| (2^-8 * 2^-17 twice, 2^-17 * 2^-17), the six fp32
    # roundings of the chain by 6 * 2^-24 of the running sum
    assert err_six.max() < 8 * 2.0 ** -24
    assert err_six.mean() < 1.5 * err_chain.mean() + 2.0 ** -26, (err_six.mean(), err_chain.mean())


def test_two_taps_per_k_step_is_the_same_sum():
    """16-channel layers put (tap, channel) pairs of two taps into one K = 32 step, the filter padded with a zero tap."""
    rng = np.random.default_rng(5)
    taps, cin, T = 5, 16, 64
    w = rng.standard_normal((taps, cin)).astype(np.float32)
    x = rng.standard_normal((cin, T + taps)).astype(np.float32)
    want = np.array([sum(float(np.dot(w[k].astype(np.float64), x[:, t + k].astype(np.float64))) for k in range(taps)) for t in range(T)])
    steps = (taps + 1) // 2
    wpad = np.zeros((2 * steps, cin), np.float32)
    wpad[:taps] = w
    got = np.zeros(T)
    for t in range(T):
        for s in range(steps):  # lane group g supplies tap 2 s + g / 2, channels 8 (g % 2) .. + 7
            kvec_w = np.concatenate([wpad[2 * s + g // 2, 8 * (g % 2):8 * (g % 2) + 8] for g in range(4)])
            kvec_x = np.concatenate([x[8 * (g % 2):8 * (g % 2) + 8, t + 2 * s + g // 2] for g in range(4)])
            got[t] += float(np.dot(kvec_w.astype(np.float64), kvec_x.astype(np.float64)))
    assert np.allclose(got, want, rtol=0, atol=1e-12)
