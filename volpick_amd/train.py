"""PhaseNet training step on the GPU (BASELINE config 5).

Host-side mirror of the pieces of the reference's Lightning module that define the arithmetic of
one optimisation step (/root/reference volpick/model/models.py):

* ``vector_cross_entropy`` (:34-51),
* ``PhaseNetLit.shared_step`` / ``training_step`` (:160-169): ``loss(model(batch["X"]), batch["y"])``,
* ``configure_optimizers`` (:177-185): ``torch.optim.Adam(lr)``,
* ``optimizer_step`` (:168-175): linear learning-rate warm-up over the first 500 steps.

Data loading, augmentation, logging and checkpointing (Lightning / SeisBench generators) are not
part of this package.  Forward (training-mode BatchNorm), loss, backward and Adam all run in
``libvolpick_hip.so`` (``vp_train_*``); there is no CPU fallback.
"""
from __future__ import annotations

import collections
import ctypes as C

import numpy as np

from . import _lib
from .models import PhaseNet


def vector_cross_entropy(y_pred, y_true, eps=1e-5):
    """models.py:34-51 on numpy arrays (B, C, T): mean over samples, sum over classes, mean over batch."""
    h = y_true * np.log(y_pred + eps)
    h = h.mean(-1).sum(-1) if y_pred.ndim == 3 else h.sum(-1)
    return -float(h.mean())


def gaussian_labels(p_samples, s_samples, n_samples=3001, sigma=20.0):
    """Soft labels in ``phases = "PSN"`` order: Gaussians of width sigma at the P and S picks,
    noise = 1 - P - S (SeisBench ProbabilisticLabeller as PhaseNetLit configures it, models.py:254-260).
    NaN / negative picks leave the phase row zero."""
    B = len(p_samples)
    t = np.arange(n_samples, dtype=np.float64)
    y = np.zeros((B, 3, n_samples), dtype=np.float64)
    for row, picks in ((0, p_samples), (1, s_samples)):
        for b, s in enumerate(picks):
            if s is not None and np.isfinite(s) and s >= 0:
                y[b, row] = np.exp(-((t - float(s)) ** 2) / (2.0 * sigma ** 2))
    y[:, 2] = np.clip(1.0 - y[:, 0] - y[:, 1], 0.0, 1.0)
    return y.astype(np.float32)


class PhaseNetTrainer:
    """One device-resident PhaseNet with its gradients and Adam state."""

    DTYPES = {"fp32": _lib.VP_TRAIN_FP32, "float32": _lib.VP_TRAIN_FP32, "32": _lib.VP_TRAIN_FP32, "32-true": _lib.VP_TRAIN_FP32,
              "bf16": _lib.VP_TRAIN_BF16, "bfloat16": _lib.VP_TRAIN_BF16, "bf16-mixed": _lib.VP_TRAIN_BF16}

    def __init__(self, model: PhaseNet, max_batch=512, device=0, betas=(0.9, 0.999), eps=1e-8, bn_momentum=0.1,
                 loss_eps=1e-5, dtype="fp32"):
        """``dtype="bf16"``: activation and gradient tensors stored as bfloat16 with fp32 accumulation, fp32 master
        weights, statistics and Adam state (Lightning's ``precision="bf16-mixed"`` in spirit; BASELINE config 5);
        ``"fp32"`` (default) is what the reference's trainer runs."""
        if str(dtype) not in self.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(self.DTYPES)}, got {dtype!r}")
        self.dtype = "bf16" if self.DTYPES[str(dtype)] == _lib.VP_TRAIN_BF16 else "fp32"
        if not isinstance(model, PhaseNet):
            raise TypeError("the training step is implemented for PhaseNet")
        if model._weights is None:
            raise RuntimeError("model has no weights (use from_pretrained / load / load_state_dict)")
        self._lib = _lib.load()
        self.model = model
        self.max_batch = int(max_batch)
        self.in_samples = model.in_samples
        self.n_params = int(model._weights.size)
        self._h = C.c_void_p()
        w = np.ascontiguousarray(model._weights, dtype=np.float32)
        _lib.check(self._lib.vp_train_create_dtype(int(device), _lib.VP_MODEL_PHASENET, w.ctypes.data_as(C.c_void_p), w.size,
                                                   self.max_batch, self.DTYPES[str(dtype)], C.byref(self._h)), "vp_train_create")
        _lib.check(self._lib.vp_train_set_hyper(self._h, betas[0], betas[1], eps, bn_momentum, loss_eps))
        self.global_step = 0
        self._in_flight = collections.deque()  # (number of a queued step, its x, its y): device inputs kept alive

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_in_flight", None):
                self._lib.vp_train_synchronize(self._h)
                self._in_flight.clear()
            self._lib.vp_train_destroy(self._h)
            self._h = None
            self._last_inputs = None

    __del__ = close

    @staticmethod
    def _arg(a):
        """(pointer, mem) of a numpy array or a CUDA torch tensor, fp32 contiguous."""
        if hasattr(a, "data_ptr"):
            if a.dtype != _torch().float32 or not a.is_contiguous():
                a = a.float().contiguous()
            return a, C.c_void_p(a.data_ptr()), (_lib.VP_MEM_DEVICE if a.is_cuda else _lib.VP_MEM_HOST)
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST

    def step(self, x, y, lr, update=True, want_loss=True, inputs_unchanged=False):
        """Forward + loss + backward (+ Adam when ``update``).  x, y: (B, 3, 3001).

        ``inputs_unchanged=True`` is the caller's PROMISE that x and y are the very tensors of the previous step and that
        nothing has written to them since, by any route (a benchmark re-running one batch): the trainer's stream then does not
        wait for the producer stream again.  It is never inferred: torch's ``Tensor._version`` does not see writes through
        ``x.data``, raw-pointer kernels (this library's own device writers, DLPack consumers) or collective outputs, and a
        loader that refills a persistent buffer that way would be read half-filled."""
        if tuple(x.shape) != tuple(y.shape) or x.ndim != 3 or x.shape[1] != 3 or x.shape[2] != self.in_samples:
            raise ValueError(f"expected x and y of shape (B, 3, {self.in_samples}), got {tuple(x.shape)} / {tuple(y.shape)}")
        xk, xp, xm = self._arg(x)
        yk, yp, ym = self._arg(y)
        self._drop_consumed()
        if xm != ym:
            raise ValueError("x and y must both be host arrays or both be device tensors")
        if xm == _lib.VP_MEM_DEVICE:
            # the trainer runs on its own stream: x / y (or their fp32 copies made above) must be complete first.  An
            # event recorded on torch's stream that the trainer's stream waits for (hipStreamWaitEvent): the host does
            # not block, so a data loader filling the next batch on torch's stream keeps running
            # (The wait is a marker in the trainer's queue in front of the step's first launch, ~25 us of the step; skipped only
            # on the caller's explicit promise -- and then only for the very OBJECTS of the previous step, held here so that
            # their identity cannot be recycled -- or when the producer has already finished.)
            torch = _torch()
            producer = torch.cuda.current_stream(xk.device)
            last = getattr(self, "_last_inputs", None)
            same = bool(inputs_unchanged) and last is not None and last[0] is xk and last[1] is yk and last[2] == producer.cuda_stream
            self._last_inputs = (xk, yk, producer.cuda_stream)
            if not same:
                ev = torch.cuda.Event()
                ev.record(producer)
                if not ev.query():  # (already complete -- a batch prepared well ahead: nothing to wait for either)
                    if getattr(self, "_ext_stream", None) is None:
                        self._ext_stream = torch.cuda.ExternalStream(int(self._lib.vp_train_stream(self._h)), device=xk.device)
                    self._ext_stream.wait_event(ev)
        loss = C.c_double(float("nan"))
        _lib.check(self._lib.vp_train_step(self._h, xp, yp, xm, int(x.shape[0]), float(lr), int(bool(update)),
                                           C.byref(loss) if want_loss else None), "vp_train_step")
        if xm == _lib.VP_MEM_DEVICE:
            # The step is (or may be) still queued on the trainer's non-blocking stream and reads x / y there:
            #  * this object keeps x / y (or the fp32 copies `_arg` made) alive until the library reports the step's last
            #    read of them complete (`_drop_consumed`), so the caching allocator cannot hand their memory out while the step
            #    reads it, whatever the caller does with its own references.  (Not Tensor.record_stream: the allocator would
            #    then record events on the trainer's stream when the tensors die -- possibly after vp_train_destroy has
            #    destroyed that stream.)
            #  * torch's current stream waits -- on the device, the host does not block -- for the event behind the step's
            #    last read of x / y, so `x.copy_(next_batch)` or any other refill enqueued there cannot overtake the step.
            self._in_flight.append((int(self._lib.vp_train_steps_enqueued(self._h)) - 1, xk, yk))
            cur = torch.cuda.current_stream(xk.device)
            _lib.check(self._lib.vp_train_wait_inputs_consumed(self._h, C.c_void_p(cur.cuda_stream)), "vp_train_wait_inputs_consumed")
        self.forward_count = getattr(self, "forward_count", 0) + 1  # every step moves the BatchNorm running statistics
        if update:
            self.global_step += 1
        return loss.value if want_loss else None

    def _drop_consumed(self):
        """Device inputs of the steps the trainer's stream has read to the end may go (no event of ours enters that stream:
        the library keeps one per step behind the last reader of x / y and answers which of them have completed)."""
        if self._in_flight:
            upto = int(self._lib.vp_train_inputs_consumed_upto(self._h))
            while self._in_flight and self._in_flight[0][0] <= upto:
                self._in_flight.popleft()

    def synchronize(self):
        _lib.check(self._lib.vp_train_synchronize(self._h))
        self._in_flight.clear()

    def _read(self, which):
        out = np.empty(self.n_params, dtype=np.float32)
        _lib.check(self._lib.vp_train_read(self._h, which, out.ctypes.data_as(C.c_void_p), out.size), "vp_train_read")
        return out

    def _named(self, blob):
        lib, out, off = self._lib, {}, 0
        shapes = {k: s for k, s, is_param in self.model._state_layout if is_param}
        for i in range(lib.vp_param_count(_lib.VP_MODEL_PHASENET)):
            key = lib.vp_param_name(_lib.VP_MODEL_PHASENET, i).decode()
            n = lib.vp_param_size(_lib.VP_MODEL_PHASENET, i)
            out[key] = blob[off:off + n].reshape(shapes[key])
            off += n
        return out

    def weights(self):
        return self._named(self._read(0))

    def gradients(self):
        return self._named(self._read(1))

    def enable_ema(self, decay=0.999):
        """Exponential moving average of the weights after every update (train.py:153-176 of the reference)."""
        _lib.check(self._lib.vp_train_set_ema(self._h, float(decay)), "vp_train_set_ema")

    def ema_weights(self):
        return self._named(self._read(4))

    def adam_state(self):
        return self._named(self._read(2)), self._named(self._read(3))

    def predictions(self, B):
        out = np.empty((B, 3, self.in_samples), dtype=np.float32)
        _lib.check(self._lib.vp_train_predictions(self._h, out.ctypes.data_as(C.c_void_p), B))
        return out

    def tensors(self, B):
        """Every z / a / gz / ga tensor of the last step as (B, C, L) arrays, by name (parity tests)."""
        lib, out = self._lib, {}
        for i in range(lib.vp_train_tensor_count(self._h)):
            name, c, l = C.c_char_p(), C.c_int(), C.c_int()
            _lib.check(lib.vp_train_tensor_info(self._h, i, C.byref(name), C.byref(c), C.byref(l)))
            a = np.empty((B, c.value, l.value), dtype=np.float32)
            _lib.check(lib.vp_train_tensor_read(self._h, i, B, a.ctypes.data_as(C.c_void_p)))
            out[name.value.decode()] = a
        return out

    def export(self, ema=False):
        """Copy the trained weights (and BatchNorm running statistics) back into ``self.model``;
        ``ema=True`` exports the EMA weights instead (``enable_ema`` must have been called)."""
        sd = self.model.state_dict()
        sd.update(self.ema_weights() if ema else self.weights())
        for k in sd:
            if k.endswith("num_batches_tracked"):
                sd[k] = np.asarray(sd[k] + self.global_step - getattr(self, "_exported_at", 0))
        self._exported_at = self.global_step
        self.model.load_state_dict(sd)
        return self.model


def _torch():
    import torch

    return torch


class PhaseNetLit:
    """The arithmetic of the reference's ``PhaseNetLit`` (models.py:108-185): Adam at ``lr`` with the
    500-step linear warm-up of ``optimizer_step``.  ``training_step(batch)`` takes
    ``{"X": (B, 3, 3001), "y": (B, 3, 3001)}`` and performs forward, loss, backward AND the
    optimiser step (Lightning's loop does the last two around the reference's ``training_step``)."""

    WARMUP_STEPS = 500

    def __init__(self, lr=1e-2, sigma=20, max_batch=512, model=None, device=0, precision="32", **model_kwargs):
        self.lr = float(lr)
        self.precision = str(precision)  # "32" (the reference's) or "bf16-mixed" (PhaseNetTrainer dtype="bf16")
        self.sigma = sigma
        self.model = model if model is not None else PhaseNet(**model_kwargs)
        self._trainer = None
        self._ema = False
        self._max_batch, self._device = max_batch, device

    def _ensure(self):
        if self._trainer is None:
            self._trainer = PhaseNetTrainer(self.model, max_batch=self._max_batch, device=self._device, dtype=self.precision)
        return self._trainer

    def learning_rate(self, step):
        """lr used by optimiser step number ``step`` (0-based).  Step 0 runs at ``lr`` (the optimiser's initial
        value).  The reference's ``optimizer_step`` hook (models.py:177-185) runs with ``trainer.global_step`` = the
        steps completed BEFORE the current one (Lightning counts the step after the hook returns), so after step k it
        sets lr * (k + 1) / 500 for step k + 1: step s >= 1 uses lr * s / 500, and the full lr is reached at step 500."""
        if step == 0 or step >= self.WARMUP_STEPS:
            return self.lr
        return self.lr * step / float(self.WARMUP_STEPS)

    def training_step(self, batch, batch_idx=None):
        tr = self._ensure()
        return tr.step(batch["X"], batch["y"], self.learning_rate(tr.global_step), update=True)

    def validation_step(self, batch, batch_idx=None):
        """Loss in eval mode (running statistics), as Lightning's validation loop computes it: the current weights
        go through the inference path (``vp_forward``).  The device plan is rebuilt only when an optimiser step
        has happened since the last validation batch.  With EMA enabled (``enable_ema``) the EMA weights are the
        ones validated, as the reference's EMA callback does with ``validate_original_weights=False``
        (volpick/model/ema.py)."""
        tr = self._ensure()
        # what the exported model depends on: the weights (global_step), the BatchNorm running statistics (every
        # step(), update or not, moves them) and which weights are validated (EMA on / off)
        key = (tr.global_step, getattr(tr, "forward_count", 0), self._ema)
        if getattr(self, "_validated_at", None) != key:
            model = tr.export(ema=self._ema)
            self._validated_at = key
        else:
            model = self.model
        if model._device_index is None:
            model.cuda(self._device)
        p = model(batch["X"])
        p = p.cpu().numpy() if hasattr(p, "cpu") else np.asarray(p)
        y = batch["y"]
        y = y.cpu().numpy() if hasattr(y, "cpu") else np.asarray(y)
        return vector_cross_entropy(p.astype(np.float64), y.astype(np.float64))

    def enable_ema(self, decay=0.999):
        self._ensure().enable_ema(decay)
        self._ema = True

    @property
    def trainer_state(self):
        return self._ensure()
