"""Torch-CPU restatement of the two SeisBench models the volpick path drives.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

The reference calls ``sbm.PhaseNet(**kw)`` / ``sbm.EQTransformer(**kw)``
(volpick/model/models.py:141,520) and ``model(x)`` on (B,3,T) float32 tensors
(volpick/model/eval_taks0.py:69,86).  The stock torch CPU ops used here
(conv1d, conv_transpose1d, batch_norm, LSTM, matmul, softmax, sigmoid) are the
ones SeisBench dispatches, so on CPU this file *is* the reference arithmetic up
to the unpinned constants in oracle/constants.py.  Parameter names equal the
state-dict keys of Final_models/**/*.pt.v1 so the
released weights load with ``strict=True``.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import constants as C

WEIGHTS_DIR = Path(__file__).resolve().parents[1] / "volpick_amd" / "weights"


# --------------------------------------------------------------------------
# PhaseNet  (SeisBench PhaseNet)
# --------------------------------------------------------------------------
class PhaseNet(nn.Module):
    name = "PhaseNet"
    in_samples = C.PN_IN_SAMPLES

    def __init__(self, phases="PSN", norm="peak", component_order="ZNE"):
        super().__init__()
        self.labels = phases
        self.norm = norm
        self.component_order = component_order
        k, s, root, depth = C.PN_KERNEL, C.PN_STRIDE, 8, 5
        self.inc = nn.Conv1d(3, root, k, padding="same")
        self.in_bn = nn.BatchNorm1d(root, eps=C.BN_EPS)
        self.down_branch = nn.ModuleList()
        self.up_branch = nn.ModuleList()
        last = root
        for i in range(depth):
            f = root * 2**i
            same = nn.Conv1d(last, f, k, padding="same", bias=False)
            bn1 = nn.BatchNorm1d(f, eps=C.BN_EPS)
            last = f
            if i == depth - 1:
                down, bn2 = None, None
            else:
                # level 0 pads inside the conv, levels 1-3 are padded by hand in forward()
                down = nn.Conv1d(f, f, k, s, padding=(k // 2 if i == 0 else 0), bias=False)
                bn2 = nn.BatchNorm1d(f, eps=C.BN_EPS)
            self.down_branch.append(nn.ModuleList([same, bn1, down, bn2]))
        for i in range(depth - 1):
            f = root * 2 ** (depth - 2 - i)
            up = nn.ConvTranspose1d(last, f, k, s, bias=False)
            bn1 = nn.BatchNorm1d(f, eps=C.BN_EPS)
            same = nn.Conv1d(2 * f, f, k, padding="same", bias=False)
            bn2 = nn.BatchNorm1d(f, eps=C.BN_EPS)
            last = f
            self.up_branch.append(nn.ModuleList([up, bn1, same, bn2]))
        self.out = nn.Conv1d(last, 3, 1, padding="same")

    def forward(self, x, logits=False):
        x = torch.relu(self.in_bn(self.inc(x)))
        skips = []
        for i, (same, bn1, down, bn2) in enumerate(self.down_branch):
            x = torch.relu(bn1(same(x)))
            if down is not None:
                skips.append(x)
                if i > 0:
                    x = F.pad(x, C.PN_DOWN_PAD[i], "constant", 0.0)
                x = torch.relu(bn2(down(x)))
        for (up, bn1, same, bn2), skip in zip(self.up_branch, skips[::-1]):
            x = torch.relu(bn1(up(x)))
            x = x[:, :, C.PN_UP_CROP[0] : x.shape[-1] - C.PN_UP_CROP[1]]
            off = (x.shape[-1] - skip.shape[-1]) // 2
            x = torch.cat([skip, x[:, :, off : off + skip.shape[-1]]], dim=1)
            x = torch.relu(bn2(same(x)))
        x = self.out(x)
        return x if logits else torch.softmax(x, dim=1)


# --------------------------------------------------------------------------
# EQTransformer  (SeisBench EQTransformer)
# --------------------------------------------------------------------------
class _Encoder(nn.Module):
    def __init__(self, cin, filters, kernels, in_samples):
        super().__init__()
        self.convs = nn.ModuleList()
        self.paddings = []
        n = in_samples
        for ci, co, k in zip([cin] + list(filters[:-1]), filters, kernels):
            self.convs.append(nn.Conv1d(ci, co, k, padding=k // 2))
            self.paddings.append(n % 2)  # TF "same" max-pool on odd lengths
            n = (n + n % 2) // 2

    def forward(self, x):
        for conv, pad in zip(self.convs, self.paddings):
            x = torch.relu(conv(x))
            if pad:
                x = F.pad(x, (0, pad), "constant", C.EQT_POOL_PAD_VALUE)
            x = F.max_pool1d(x, 2)
        return x


class _Decoder(nn.Module):
    def __init__(self, cin, filters, kernels, out_samples):
        super().__init__()
        self.crops = []
        n = out_samples
        for i in range(len(filters)):
            pad = n % 2
            n = (n + pad) // 2
            if pad:
                self.crops.append(len(filters) - 1 - i)
        self.convs = nn.ModuleList(
            nn.Conv1d(ci, co, k, padding=k // 2) for ci, co, k in zip([cin] + list(filters[:-1]), filters, kernels)
        )

    def forward(self, x):
        for i, conv in enumerate(self.convs):
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if i in self.crops:
                x = x[:, :, :-1]
            x = torch.relu(conv(x))
        return x


class _ResCNNBlock(nn.Module):
    def __init__(self, filters, ker):
        super().__init__()
        self.right_pad = ker == 2  # TF "same" for an even kernel pads on the right only
        pad = 1 if ker == 3 else 0
        self.norm1 = nn.BatchNorm1d(filters, eps=C.BN_EPS)
        self.conv1 = nn.Conv1d(filters, filters, ker, padding=pad)
        self.norm2 = nn.BatchNorm1d(filters, eps=C.BN_EPS)
        self.conv2 = nn.Conv1d(filters, filters, ker, padding=pad)

    def forward(self, x):
        y = torch.relu(self.norm1(x))
        if self.right_pad:
            y = F.pad(y, (0, 1))
        y = self.conv1(y)
        y = torch.relu(self.norm2(y))
        if self.right_pad:
            y = F.pad(y, (0, 1))
        y = self.conv2(y)
        return x + y


class _Members(nn.Module):
    def __init__(self, members):
        super().__init__()
        self.members = nn.ModuleList(members)

    def forward(self, x):
        for m in self.members:
            x = m(x)
        return x


class _BiLSTMBlock(nn.Module):
    def __init__(self, input_size, hidden):
        super().__init__()
        self.lstm = nn.LSTM(input_size, hidden, bidirectional=True)
        self.conv = nn.Conv1d(2 * hidden, hidden, 1)
        self.norm = nn.BatchNorm1d(hidden, eps=C.BN_EPS)

    def forward(self, x):
        x = self.lstm(x.permute(2, 0, 1))[0]  # (T, B, 2H)
        x = x.permute(1, 2, 0)
        return self.norm(self.conv(x))


class _SeqSelfAttention(nn.Module):
    """Additive (Bahdanau) self attention; optional band of ``attention_width``."""

    def __init__(self, input_size=16, units=32, attention_width=None, eps=C.EQT_ATTENTION_EPS):
        super().__init__()
        self.attention_width = attention_width
        self.eps = eps
        self.Wx = nn.Parameter(torch.zeros(input_size, units))
        self.Wt = nn.Parameter(torch.zeros(input_size, units))
        self.bh = nn.Parameter(torch.zeros(units))
        self.Wa = nn.Parameter(torch.zeros(units, 1))
        self.ba = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        x = x.permute(0, 2, 1)  # (B, T, C)
        q = torch.matmul(x, self.Wt).unsqueeze(2)  # (B, T, 1, U)
        k = torch.matmul(x, self.Wx).unsqueeze(1)  # (B, 1, T, U)
        h = torch.tanh(q + k + self.bh)
        e = (torch.matmul(h, self.Wa) + self.ba).squeeze(-1)  # (B, T, T)
        e = torch.exp(e - e.max(dim=-1, keepdim=True).values)  # max over the FULL row, before the band mask
        if self.attention_width is not None:
            t = torch.arange(e.shape[1])
            lower = t - self.attention_width // 2
            upper = lower + self.attention_width
            idx = t.unsqueeze(1)
            mask = torch.logical_and(lower <= idx, idx < upper)
            e = torch.where(mask, e, torch.zeros_like(e))
        a = e / (e.sum(dim=-1, keepdim=True) + self.eps)
        v = torch.matmul(a, x)
        return v.permute(0, 2, 1), a


class _LayerNorm(nn.Module):
    """Normalises over the channel axis per (batch, time) with eps under the sqrt."""

    def __init__(self, filters, eps=C.EQT_LAYERNORM_EPS):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(filters, 1))
        self.beta = nn.Parameter(torch.zeros(filters, 1))
        self.eps = eps

    def forward(self, x):
        mean = x.mean(dim=1, keepdim=True)
        var = ((x - mean) ** 2).mean(dim=1, keepdim=True) + self.eps
        return (x - mean) / torch.sqrt(var) * self.gamma + self.beta


class _FeedForward(nn.Module):
    def __init__(self, io_size, hidden=128):
        super().__init__()
        self.lin1 = nn.Linear(io_size, hidden)
        self.lin2 = nn.Linear(hidden, io_size)

    def forward(self, x):
        x = x.permute(0, 2, 1)
        x = self.lin2(torch.relu(self.lin1(x)))
        return x.permute(0, 2, 1)


class _Transformer(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.attention = _SeqSelfAttention(input_size)
        self.norm1 = _LayerNorm(input_size)
        self.ff = _FeedForward(input_size)
        self.norm2 = _LayerNorm(input_size)

    def forward(self, x):
        y, _ = self.attention(x)
        y = self.norm1(x + y)
        return self.norm2(y + self.ff(y))


class EQTransformer(nn.Module):
    name = "EQTransformer"
    in_samples = C.EQT_IN_SAMPLES

    def __init__(self, phases="PS", norm="peak", component_order="ZNE", norm_amp_per_comp=False):
        super().__init__()
        self.labels = ["Detection"] + list(phases)
        self.phases = phases
        self.norm = norm
        self.norm_amp_per_comp = norm_amp_per_comp
        self.component_order = component_order
        f, k = list(C.EQT_FILTERS), list(C.EQT_KERNELS)
        self.encoder = _Encoder(3, f, k, self.in_samples)
        self.res_cnn_stack = _Members([_ResCNNBlock(f[-1], kk) for kk in C.EQT_RES_KERNELS])
        self.bi_lstm_stack = _Members(
            [_BiLSTMBlock(f[-1] if i == 0 else 16, 16) for i in range(C.EQT_LSTM_BLOCKS)]
        )
        self.transformer_d0 = _Transformer(16)
        self.transformer_d = _Transformer(16)
        self.decoder_d = _Decoder(16, f[::-1], k[::-1], self.in_samples)
        self.conv_d = nn.Conv1d(f[0], 1, 11, padding=5)
        n = len(phases)
        self.pick_lstms = nn.ModuleList(nn.LSTM(16, 16) for _ in range(n))
        self.pick_attentions = nn.ModuleList(
            _SeqSelfAttention(16, attention_width=C.PICK_ATTENTION_WIDTH) for _ in range(n)
        )
        self.pick_decoders = nn.ModuleList(_Decoder(16, f[::-1], k[::-1], self.in_samples) for _ in range(n))
        self.pick_convs = nn.ModuleList(nn.Conv1d(f[0], 1, 11, padding=5) for _ in range(n))

    def bottleneck(self, x):
        x = self.encoder(x)
        x = self.res_cnn_stack(x)
        x = self.bi_lstm_stack(x)
        x = self.transformer_d0(x)
        return self.transformer_d(x)

    def forward(self, x):
        x = self.bottleneck(x)
        outs = [torch.sigmoid(self.conv_d(self.decoder_d(x))).squeeze(1)]
        for lstm, att, dec, conv in zip(self.pick_lstms, self.pick_attentions, self.pick_decoders, self.pick_convs):
            px = lstm(x.permute(2, 0, 1))[0].permute(1, 2, 0)
            px, _ = att(px)
            outs.append(torch.sigmoid(conv(dec(px))).squeeze(1))
        return tuple(outs)  # (detection, P, S)


# --------------------------------------------------------------------------
def load_pretrained(model: str, name: str = "volpick"):
    """Build the restated model and load a converted weight set with strict keys.

    Mirrors ``sbm.<Model>.from_pretrained(name)`` (README.md:46-47): model_args
    from the JSON go to the constructor, default_args are attached.
    """
    model = model.lower()
    d = WEIGHTS_DIR / model
    meta = json.loads((d / f"{name}.json").read_text())
    cls = {"phasenet": PhaseNet, "eqtransformer": EQTransformer}[model]
    net = cls(**meta.get("model_args", {}))
    with np.load(d / f"{name}.npz") as z:
        sd = {k: torch.from_numpy(z[k].copy()) for k in z.files}
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.default_args = meta.get("default_args", {})
    net.weights_docstring = meta.get("docstring", "")
    return net
