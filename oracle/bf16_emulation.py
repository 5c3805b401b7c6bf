"""TEST INFRASTRUCTURE (see oracle/__init__.py): torch emulation of the bf16-STORAGE training mode
(`vp_train_create_dtype(..., VP_TRAIN_BF16, ...)`, include/volpick_hip.h).

That mode keeps every activation and gradient tensor of the PhaseNet training step (the input x, every conv output z,
every a = relu(bn(z)), and the gradients gz, ga) in memory as bfloat16, rounded to nearest-even when stored, while all
arithmetic accumulates in fp32.  Against fp32 autograd the results then differ by the storage rounding itself
(2^-9 per stored value, amplified wherever BatchNorm divides by a small batch deviation), which says little about
whether the kernels are right.  Here the SAME rounding points are put into the oracle module -- forward hooks round x,
z and bn(z) (relu commutes with the rounding), the matching backward rounds the gradients flowing into them -- so that
the HIP step can be compared with torch autograd at a tolerance set by summation order, not by the storage format.

The reference itself trains in fp32 (volpick/model/train.py: no precision argument to the Lightning trainer); bf16 is
BASELINE.json configs[4]'s dtype.  Nothing under volpick_amd/ imports this module.
"""
import contextlib

import torch
from torch import nn


def round_bf16(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest-even bfloat16 -> fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundBoth(torch.autograd.Function):
    """y = round_bf16(x) on the way forward, grad_x = round_bf16(grad_y) on the way back: a tensor AND its gradient
    rest in memory as bfloat16."""

    @staticmethod
    def forward(ctx, x):
        return round_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return round_bf16(g)


class _RoundForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return round_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


@contextlib.contextmanager
def bf16_storage(net: nn.Module):
    """Within the context, `net` (oracle.models.PhaseNet) computes with bf16-stored activations and gradients."""
    handles = [net.register_forward_pre_hook(lambda m, args: (_RoundForward.apply(args[0]),) + tuple(args[1:]))]
    for mod in net.modules():
        if mod is net.out:
            continue  # the 1x1 head + softmax + loss run in fp32 registers; nothing is stored in between
        if isinstance(mod, (nn.Conv1d, nn.ConvTranspose1d, nn.BatchNorm1d)):
            handles.append(mod.register_forward_hook(lambda m, inp, out: _RoundBoth.apply(out)))
    try:
        yield net
    finally:
        for h in handles:
            h.remove()
