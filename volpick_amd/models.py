"""Picker API of the volpick path on MI355X: ``PhaseNet`` / ``EQTransformer`` with the
``seisbench.models`` surface volpick users call.

Reference usage this mirrors (names, argument meaning, error behaviour):
  README.md:46-66            from_pretrained("volpick"); classify(stream, batch_size=256, overlap=5500,
                             blinding=(500,500), stacking="avg", parallelism=None, P_threshold=..,
                             S_threshold=.., copy=True).picks
  Final_models/demo.ipynb    list_pretrained() (:60,92), .weights_docstring (:121), .device/.cuda()
                             (:224-227), annotate(stream, overlap=, blinding=) -> traces
                             "<Model>_<label>" (:300-327), classify(...).picks (:397-413)
  volpick/model/eval_taks0.py:58-89   model.eval(); model(x) on (B,3,T) float32 tensors;
                             model.name / model.labels / model.device
  model_training/test.ipynb:231-240   .labels .norm .component_order .default_args

Every numerical stage runs in libvolpick_hip.so (HIP, gfx950) through ctypes; torch is only
used for device buffers and tensors handed back to the caller.  There is no CPU path: using a
model without a GPU / without the built library raises ``VolpickHipError``.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time
import warnings
from pathlib import Path

import numpy as np

from . import _lib
from ._lib import VolpickHipError
from .picks import ClassifyOutput, Detection, DetectionList, Pick, PickList
from .stream import Stream, Trace, UTCDateTime

WEIGHTS_DIR = Path(__file__).resolve().parent / "weights"


def _torch():
    import torch

    return torch


def OrderedDictTensors(arrays, torch):
    from collections import OrderedDict

    return OrderedDict((k, torch.from_numpy(np.array(v, order="C"))) for k, v in arrays.items())


def _no_triggers():
    return (np.empty(0, np.int32), np.empty(0, np.int64), np.empty(0, np.int64), np.empty(0, np.int64), np.empty(0, np.float32))


def _records_from_columns(cols, tids, t0s, labels, sr):
    """Trigger columns of all station blocks -> (PickList, DetectionList), sorted as the records sort ((start_time,
    trace_id, phase) / (start_time, trace_id), stable), the records deferred.  Times: t0 + k / sr in integer microseconds,
    rounded as stream.UTCDateTime.__add__ rounds (float64 k / sr * 1e6, half to even)."""
    if not cols:
        return PickList(), DetectionList()
    grp = np.concatenate([np.full(len(c[1]), c[0], np.int64) for c in cols])
    spec, on, off, peak, val = (np.concatenate([c[i] for c in cols]) for i in range(1, 6))
    t0 = np.asarray(t0s, np.int64)[grp]
    us = lambda k: t0 + np.rint(k / float(sr) * 1e6).astype(np.int64)
    start, end, top = us(on), us(off), us(peak)
    order = {t: i for i, t in enumerate(sorted(set(tids)))}  # equal ids (two blocks of one station) compare equal
    tid_rank = np.asarray([order[t] for t in tids], np.int64)[grp]
    lab_rank = np.asarray([sorted(set(labels)).index(l) for l in labels], np.int64)[spec] if labels else spec
    is_det = np.asarray([l == "Detection" for l in labels], bool)[spec]
    U = UTCDateTime._from_us

    def build(mask, keys, make_one):
        idx = np.flatnonzero(mask)
        idx = idx[np.lexsort(tuple(k[idx] for k in keys))]
        g, s_, a, b, c, v = grp[idx].tolist(), spec[idx].tolist(), start[idx].tolist(), end[idx].tolist(), top[idx].tolist(), val[idx].tolist()
        return len(idx), (lambda: [make_one(tids[g[i]], a[i], b[i], c[i], v[i], labels[s_[i]]) for i in range(len(g))])

    n_p, make_p = build(~is_det, (lab_rank, tid_rank, start), lambda tid, a, b, c, v, ph: Pick(tid, U(a), U(b), U(c), v, ph))
    n_d, make_d = build(is_det, (tid_rank, start), lambda tid, a, b, c, v, ph: Detection(tid, U(a), U(b), v))
    return PickList._deferred(n_p, make_p), DetectionList._deferred(n_d, make_d)


class WaveformModel:
    """Common host logic: weight registry, device handle, stream handling, annotate/classify."""

    name = "WaveformModel"
    _kind = -1
    _weights_subdir = ""
    in_samples = 0
    default_contexts = 3
    sampling_rate = 100.0
    # class-level annotate defaults (SeisBench ``_annotate_args``), overridden by the JSON's default_args
    _annotate_args = {
        "batch_size": 256,
        "overlap": 0,
        "stacking": "avg",
        "blinding": (0, 0),
        "*_threshold": 0.3,
    }
    _known_args = {"batch_size", "overlap", "stacking", "blinding", "parallelism", "copy", "strict",
                   "flexible_horizontal_components", "stride"}

    def __init__(self, component_order="ZNE", norm="peak", **kwargs):
        self.component_order = component_order
        self.norm = norm
        self.default_args = {}
        self.weights_docstring = None
        self._weights_metadata = None
        self._weights_version = None
        self._weights = None  # flat fp32 blob in the library's canonical order
        self._handle = None
        self._extra_handles = []
        self.batch_across_blocks = True  # classify(): windows of several device-resident blocks share the forward batches
        self._max_windows_per_call = 32768  # bounds the prediction buffer of one multi-block call (2.4 GB for EQT)
        # device contexts classify() pipelines station blocks over (tools/ctx_sweep.sh: PhaseNet 3, EQTransformer 4 with
        # >= 5 hardware queues -- volpick_amd/__init__.py; more contexts than that only add queueing)
        self.n_contexts = self.default_contexts
        self._seg_per_context = 2  # segments of a long block per device context (<= VP_MAX_INFLIGHT submit slots)
        self._device_index = None
        self._max_batch = 256
        self._plan_flags = (0, 0)  # vp_config.plan_flags[0:2]: (layer-by-layer plan, dump fused intermediates)
        self._timing = None        # dict: classify() records its phases there, SERIALISED by device synchronisations (bench.py `api`)
        self._training = False
        if norm not in ("peak", "std"):
            raise ValueError("norm must be 'peak' or 'std'")

    # ------------------------------------------------------------------ weights
    @classmethod
    def list_pretrained(cls, details=False, remote=False):
        d = WEIGHTS_DIR / cls._weights_subdir
        names = sorted(p.stem for p in d.glob("*.json"))
        if details:
            return {n: json.loads((d / f"{n}.json").read_text()).get("docstring", "") for n in names}
        return names

    @classmethod
    def from_pretrained(cls, name, version_str="latest", update=False, force=False, wait_for_file=False):
        d = WEIGHTS_DIR / cls._weights_subdir
        meta_path, npz_path = d / f"{name}.json", d / f"{name}.npz"
        if not meta_path.exists() or not npz_path.exists():
            raise ValueError(f"No pretrained {cls.__name__} weights '{name}'. Available: {cls.list_pretrained()}")
        meta = json.loads(meta_path.read_text())
        with np.load(npz_path) as z:
            tensors = {k: z[k] for k in z.files}
        return cls._from_state(meta, tensors)

    @classmethod
    def load(cls, path, version_str=None):
        """Load SeisBench-format ``<path>.json[.vN]`` + ``<path>.pt[.vN]`` (torch state dict)."""
        path = str(path)
        suffix = f".v{version_str}" if version_str else ""
        meta = json.loads(Path(path + ".json" + suffix).read_text())
        sd = _torch().load(path + ".pt" + suffix, map_location="cpu", weights_only=True)
        return cls._from_state(meta, {k: v.numpy() for k, v in sd.items()})

    @classmethod
    def _from_state(cls, meta, tensors):
        model = cls(**meta.get("model_args", {}))
        model.load_state_dict(tensors)
        model.default_args = dict(meta.get("default_args", {}))
        model.weights_docstring = meta.get("docstring")
        model._weights_version = meta.get("version")
        model._weights_metadata = meta
        return model

    def load_state_dict(self, tensors, strict=True):
        lib = _lib.load()
        parts = []
        expected = set()
        for i in range(lib.vp_param_count(self._kind)):
            key = lib.vp_param_name(self._kind, i).decode()
            expected.add(key)
            if key not in tensors:
                raise KeyError(f"missing key in state dict: {key}")
            a = np.asarray(getattr(tensors[key], "numpy", lambda: tensors[key])(), dtype=np.float32).ravel()
            if a.size != lib.vp_param_size(self._kind, i):
                raise ValueError(f"size mismatch for {key}: {a.size} vs {lib.vp_param_size(self._kind, i)}")
            parts.append(a)
        extra = [k for k in tensors if k not in expected and not k.endswith("num_batches_tracked")]
        if strict and extra:
            raise KeyError(f"unexpected keys in state dict: {extra[:5]}")
        self._weights = np.ascontiguousarray(np.concatenate(parts))
        # key order / shapes / non-float buffers of the source dict, so state_dict() and save() round-trip it
        self._state_layout = [(k, tuple(np.shape(tensors[k])), k in expected) for k in tensors if k not in extra]
        self._state_buffers = {k: np.array(getattr(tensors[k], "numpy", lambda k=k: tensors[k])())
                               for k, _, is_param in self._state_layout if not is_param}
        self._release()

    def state_dict(self):
        """``OrderedDict`` name -> numpy array in the source dict's key order (torch layout and names)."""
        from collections import OrderedDict

        if self._weights is None:
            raise RuntimeError("no weights loaded")
        lib = _lib.load()
        offsets, off = {}, 0
        for i in range(lib.vp_param_count(self._kind)):
            n = lib.vp_param_size(self._kind, i)
            offsets[lib.vp_param_name(self._kind, i).decode()] = (off, n)
            off += n
        out = OrderedDict()
        for key, shape, is_param in self._state_layout:
            if is_param:
                o, n = offsets[key]
                out[key] = self._weights[o:o + n].reshape(shape).copy()
            else:
                out[key] = self._state_buffers[key].copy()
        return out

    def get_model_args(self):
        return {"component_order": self.component_order, "norm": self.norm}

    def save(self, path, weights_docstring="", version_str=None):
        """Write ``<path>.json[.vN]`` + ``<path>.pt[.vN]`` in the SeisBench model-zoo format
        (the files ``model_training/tune.ipynb`` cell 7 ``export_model`` produces and
        ``Final_models/*/*.{json,pt}.v1`` hold), so weights round-trip through ``load``."""
        torch = _torch()
        path = str(path)
        suffix = f".v{version_str}" if version_str else ""
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        meta = {
            "docstring": weights_docstring or self.weights_docstring or "",
            "model_args": self.get_model_args(),
            "seisbench_requirement": (self._weights_metadata or {}).get("seisbench_requirement", "0.4.0"),
            "version": str(version_str) if version_str else (self._weights_version or "1"),
            "default_args": dict(self.default_args),
        }
        Path(path + ".json" + suffix).write_text(json.dumps(meta, indent=4))
        torch.save(OrderedDictTensors(self.state_dict(), torch), path + ".pt" + suffix)

    # ------------------------------------------------------------------ device
    @property
    def device(self):
        torch = _torch()
        return torch.device("cpu") if self._device_index is None else torch.device("cuda", self._device_index)

    def cuda(self, device=None):
        torch = _torch()
        idx = torch.cuda.current_device() if device is None else torch.device(device if not isinstance(device, int) else f"cuda:{device}").index
        if idx is None:
            idx = torch.cuda.current_device()
        if idx != self._device_index:
            self._release()
            self._device_index = idx
        self._ensure_handle()
        return self

    def to(self, device):
        dev = _torch().device(device)
        if dev.type == "cpu":
            return self.cpu()
        return self.cuda(dev)

    def cpu(self):
        self._release()
        self._device_index = None
        return self

    def eval(self):
        self._training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("model objects run the inference path; the PhaseNet training step lives in "
                                      "volpick_amd.train (PhaseNetTrainer / PhaseNetLit)")
        return self.eval()

    def _config(self):
        lib = _lib.load()
        cfg = _lib.VpConfig()
        _lib.check(lib.vp_default_config(self._kind, C.byref(cfg)))
        cfg.norm = _lib.VP_NORM_PEAK if self.norm == "peak" else _lib.VP_NORM_STD
        cfg.max_batch = int(self._max_batch)
        flags = list(self._plan_flags)
        if os.environ.get("VOLPICK_PLAN_FLAGS"):  # A/B timing of plan variants through bench.py (debug): the entries the model
            env = [int(v) for v in os.environ["VOLPICK_PLAN_FLAGS"].split(",")]  # itself left at 0 come from the environment
            flags += [0] * (len(env) - len(flags))
            flags = [f if f != 0 or i >= len(env) else env[i] for i, f in enumerate(flags)]
        for i, v in enumerate(flags):
            cfg.plan_flags[i] = int(v)
        return cfg

    def _ensure_handle(self, weights_device_ptr=None):
        if self._handle is not None:
            return self._handle
        torch = _torch()
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise VolpickHipError("no HIP device visible: the volpick path runs on MI355X only (no CPU fallback)")
        if self._weights is None:
            raise VolpickHipError("model has no weights; use from_pretrained() or load_state_dict()")
        if self._device_index is None:
            self._device_index = int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())) % torch.cuda.device_count()
        cfg = self._config()
        h = C.c_void_p()
        if weights_device_ptr is not None:
            ptr, mem = C.c_void_p(weights_device_ptr), _lib.VP_MEM_DEVICE
        else:
            ptr, mem = self._weights.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST
        _lib.check(lib.vp_create(self._device_index, self._kind, ptr, self._weights.size, mem, C.byref(cfg),
                                 C.byref(h)), "vp_create")
        self._handle = h
        return h

    def _context(self, k):
        """k-th device context of this model (own HIP stream + workspace).  Context 0 is ``_handle``;
        further ones are created on first use.  ``classify`` round-robins station blocks over two
        contexts so that one block's latency-bound stages overlap the other's MFMA-bound ones."""
        self._ensure_handle()
        if k == 0:
            return self._handle
        while len(self._extra_handles) < k:
            lib = _lib.load()
            cfg = self._config()
            h = C.c_void_p()
            _lib.check(lib.vp_create(self._device_index, self._kind, self._weights.ctypes.data_as(C.c_void_p),
                                     self._weights.size, _lib.VP_MEM_HOST, C.byref(cfg), C.byref(h)), "vp_create")
            self._extra_handles.append(h)
        return self._extra_handles[k - 1]

    def _release(self):
        for h in getattr(self, "_extra_handles", []):
            _lib.load().vp_destroy(h)
        self._extra_handles = []
        if getattr(self, "_handle", None) is not None:
            try:
                _lib.load().vp_destroy(self._handle)
            finally:
                self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ model(x)
    def _forward_raw(self, x, preprocess=False):
        """(B,3,T) float32 torch tensor / ndarray -> (B,n_out,T) tensor on the input's device."""
        torch = _torch()
        lib = _lib.load()
        h = self._ensure_handle()
        as_numpy = isinstance(x, np.ndarray)
        if as_numpy:
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        if x.ndim != 3 or x.shape[1] != 3 or x.shape[2] != self.in_samples:
            raise ValueError(f"expected input of shape (B, 3, {self.in_samples}), got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        on_dev = x.is_cuda
        if on_dev and x.device.index != self._device_index:
            raise ValueError(f"input on {x.device} but model on {self.device}")
        y = torch.empty((B, 3, self.in_samples), dtype=torch.float32, device=x.device)
        if B == 0:
            return y
        if on_dev:
            torch.cuda.current_stream(x.device).synchronize()
        mem = _lib.VP_MEM_DEVICE if on_dev else _lib.VP_MEM_HOST
        _lib.check(lib.vp_forward(h, C.c_void_p(x.data_ptr()), mem, B, int(preprocess), C.c_void_p(y.data_ptr()), mem),
                   "vp_forward")
        return y.numpy() if as_numpy else y

    def __call__(self, x, logits=False):
        if logits:
            raise NotImplementedError("logits=True is not part of the inference path")
        return self._format_output(self._forward_raw(x))

    forward = __call__

    def _format_output(self, y):
        return y

    # ------------------------------------------------------------------ annotate / classify
    def _argdict(self, kwargs):
        for k in kwargs:
            if k not in self._known_args and not k.endswith("_threshold"):
                warnings.warn(f"Unknown argument '{k}' will be ignored.")
        args = dict(kwargs)
        for k in ("overlap", "blinding", "stacking", "batch_size"):
            if k not in args:
                args[k] = self.default_args.get(k, self._annotate_args[k])
        if isinstance(args["overlap"], float) and 0 <= args["overlap"] < 1:  # fraction of the window
            args["overlap"] = int(self.in_samples * args["overlap"])
        args["overlap"] = int(args["overlap"])
        if not 0 <= args["overlap"] < self.in_samples:
            raise ValueError(f"overlap must be in [0, {self.in_samples}), got {args['overlap']}")
        b = tuple(int(v) for v in args["blinding"])
        if len(b) != 2 or min(b) < 0 or b[0] + b[1] >= self.in_samples:
            raise ValueError(f"invalid blinding {args['blinding']}")
        args["blinding"] = b
        if args["stacking"] not in ("avg", "max"):
            raise ValueError(f"Stacking method {args['stacking']} unknown. Known methods are 'avg' and 'max'.")
        return args

    def _threshold(self, args, label):
        key = f"{label}_threshold"
        if key in args:
            return float(args[key])
        if key in self.default_args:
            return float(self.default_args[key])
        return float(self._annotate_args.get(key, self._annotate_args["*_threshold"]))

    def _annotate_block(self, data, args):
        """(3,N) float32 ndarray -> (device tensor (n_out,N) with NaN, first_valid, last_valid, n_windows)."""
        torch = _torch()
        lib = _lib.load()
        h = self._ensure_handle()
        dev = torch.device("cuda", self._device_index)
        n = data.shape[1]
        if torch.is_tensor(data):  # assembled on the device by _group_stream
            x = data.to(dev, torch.float32).contiguous()
        elif isinstance(data, _Rows):
            x = data.upload(torch, dev)[0]
        else:
            x = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
        out = torch.empty((3, n), dtype=torch.float32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()
        stacking = _lib.VP_STACK_AVG if args["stacking"] == "avg" else _lib.VP_STACK_MAX
        batch = max(1, min(int(args["batch_size"]), self._max_batch))
        _lib.check(lib.vp_annotate(h, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE, n, args["overlap"],
                                   args["blinding"][0], args["blinding"][1], stacking, batch,
                                   C.c_void_p(out.data_ptr()), _lib.VP_MEM_DEVICE, C.byref(fv), C.byref(lv),
                                   C.byref(nw)), "vp_annotate")
        return out, fv.value, lv.value, nw.value

    # ---- one long block spread over the device contexts ---------------------------------------------
    def _is_long(self, n_samples, args):
        """Worth splitting: every context gets at least two full forward batches."""
        step = self.in_samples - int(args["overlap"])
        batch = max(1, min(int(args["batch_size"]), self._max_batch))
        return self.n_contexts > 1 and (n_samples - self.in_samples) // step + 1 >= 2 * batch * self.n_contexts

    def _annotate_segments(self, data, args):
        """Like ``_annotate_block`` for a long (3,N) block: ``segments.plan_segments`` cuts it into one segment
        per device context; segment r+1 is uploaded while segment r computes, the stacked outputs are cut and
        joined on the device -- sample for sample the unsplit result (volpick_amd/segments.py)."""
        job = self._segments_submit(data, args, 0)
        return job if isinstance(job, tuple) else self._segments_collect(job)

    def _segments_submit(self, data, args, slot_base, seg_per_context=None):
        """First half of ``_annotate_segments``: upload and enqueue every segment (submit slots ``slot_base`` .. on each
        context) -> a job for ``_segments_collect``; a block too short for more than one segment comes back finished, as
        ``_annotate_block``'s tuple.  ``classify()`` uploads the NEXT station's segments between the two halves."""
        from .segments import plan_segments

        torch = _torch()
        lib = _lib.load()
        dev = torch.device("cuda", self._device_index)
        n = data.shape[1]
        # _seg_per_context segments per device context (one submit slot each): the first upload, which nothing overlaps, is that
        # much shorter
        nc = self.n_contexts
        segs = plan_segments(n, self.in_samples, args["overlap"], args["blinding"], nc * (seg_per_context or self._seg_per_context))
        if len(segs) == 1:
            return self._annotate_block(data, args)
        stacking = _lib.VP_STACK_AVG if args["stacking"] == "avg" else _lib.VP_STACK_MAX
        batch = max(1, min(int(args["batch_size"]), self._max_batch))
        tm = self._timing

        def upload(sg):
            lo, hi = sg["lo"], sg["hi"]
            if torch.is_tensor(data):
                x = data[:, lo:hi].to(dev, torch.float32).contiguous()
            elif isinstance(data, _Rows):
                x = _Rows([p[lo:hi] for p in data.parts]).upload(torch, dev)[0]
            else:
                x = torch.from_numpy(np.ascontiguousarray(data[:, lo:hi], dtype=np.float32)).to(dev)
            y = torch.empty((3, hi - lo), dtype=torch.float32, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            return x, y

        def submit(r, sg, x, y):
            _lib.check(lib.vp_classify_submit(self._context(r % nc), slot_base + r // nc, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE,
                                              sg["hi"] - sg["lo"], args["overlap"], args["blinding"][0], args["blinding"][1],
                                              stacking, batch, None, 0, C.c_void_p(y.data_ptr()), _lib.VP_MEM_DEVICE, 0),
                       "vp_classify_submit")

        jobs = []
        t0 = time.perf_counter()
        if tm is None:  # segment r + 1 is uploaded while segment r computes
            for r, sg in enumerate(segs):
                x, y = upload(sg)
                submit(r, sg, x, y)
                jobs.append((x, y))
        else:  # profiled: all uploads, then all compute (the two phases separated; their sum exceeds the pipelined wall time)
            jobs = [upload(sg) for sg in segs]
            tm["h2d_ms"] = tm.get("h2d_ms", 0.0) + (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            for r, (sg, (x, y)) in enumerate(zip(segs, jobs)):
                submit(r, sg, x, y)
        return dict(segs=segs, jobs=jobs, slot_base=slot_base, n=n, t0=t0, overlap=args["overlap"])

    def _segments_collect(self, job):
        """Second half: wait for the segments, cut and join their stacked outputs on the device."""
        torch = _torch()
        lib = _lib.load()
        dev = torch.device("cuda", self._device_index)
        nc, n, segs, tm = self.n_contexts, job["n"], job["segs"], self._timing
        out = torch.empty((3, n), dtype=torch.float32, device=dev)
        fv = lv = -1
        found = C.c_int()
        for r, (sg, (x, y)) in enumerate(zip(segs, job["jobs"])):
            f, l, w = C.c_int64(), C.c_int64(), C.c_int64()
            _lib.check(lib.vp_classify_collect(self._context(r % nc), job["slot_base"] + r // nc, C.byref(f), C.byref(l), C.byref(w), None, None, None,
                                               None, None, 0, C.byref(found)), "vp_classify_collect")
            out[:, sg["keep_lo"]:sg["keep_hi"]] = y[:, sg["keep_lo"] - sg["lo"]:sg["keep_hi"] - sg["lo"]]
            if r == 0:
                fv = f.value
            lv = l.value + sg["lo"]
        n_windows = int(lib.vp_window_starts(n, self.in_samples, job["overlap"], None, 0))
        torch.cuda.current_stream(dev).synchronize()
        if tm is not None:
            tm["gpu_ms"] = tm.get("gpu_ms", 0.0) + (time.perf_counter() - job["t0"]) * 1e3
            tm["windows"] = tm.get("windows", 0) + n_windows
        return out, fv, lv, n_windows

    def _pick_rows(self, dev_out, specs, cap=8192, columns=False):
        """Trigger scan of the rows of a device (n_out, N) array -> [(spec_index, on, off, peak, value)]
        (``columns=True``: the same as five arrays)."""
        lib = _lib.load()
        h = self._ensure_handle()
        n = dev_out.shape[1]
        if not specs:
            return _no_triggers() if columns else []
        dev_out = dev_out.contiguous()  # (a column slice of the stacked rows is a strided view)
        self._torch_sync(dev_out)
        c_specs = (_lib.VpTriggerSpec * len(specs))(*[_lib.VpTriggerSpec(r, on, off) for r, _, on, off in specs])
        I64 = C.POINTER(C.c_int64)
        while True:  # every row in one launch, one synchronisation, one result copy
            on, off, peak = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
            val, spec_of, found = np.empty(cap, np.float32), np.empty(cap, np.int32), C.c_int()
            _lib.check(lib.vp_pick_rows(h, C.c_void_p(dev_out.data_ptr()), n, c_specs, len(specs), on.ctypes.data_as(I64),
                                        off.ctypes.data_as(I64), peak.ctypes.data_as(I64),
                                        val.ctypes.data_as(C.POINTER(C.c_float)), spec_of.ctypes.data_as(C.POINTER(C.c_int32)),
                                        cap, C.byref(found)), "vp_pick_rows")
            if found.value <= cap:
                break
            cap = found.value
        m = found.value  # grouped by spec, sorted by onset inside each group
        if columns:
            return spec_of[:m], on[:m], off[:m], peak[:m], val[:m]
        return list(zip(spec_of[:m].tolist(), on[:m].tolist(), off[:m].tolist(), peak[:m].tolist(), val[:m].tolist()))

    @staticmethod
    def _torch_sync(t):
        """The library scans on its own stream: what torch has queued for this tensor must be done first."""
        _torch().cuda.current_stream(t.device).synchronize()

    def _trigger_specs(self, args):
        """[(row, label, thr_on, thr_off)]: picks use thr/thr, detections thr/(thr/2); 'N' is skipped."""
        specs = []
        for i, label in enumerate(self.labels):
            if label == "N":
                continue
            if label == "Detection":
                thr = self._threshold(args, "detection")
                specs.append((i, label, thr, thr / 2))
            else:
                thr = self._threshold(args, label)
                specs.append((i, label, thr, thr))
        return specs

    def _submit_block(self, ctx, data, args, specs, cap):
        """Enqueue one (3,N) block on device context ``ctx`` (no host synchronisation)."""
        torch = _torch()
        lib = _lib.load()
        h = self._context(ctx)
        dev = torch.device("cuda", self._device_index)
        n = data.shape[1]
        if torch.is_tensor(data):  # assembled on the device by _group_stream
            x = data.to(dev, torch.float32).contiguous()
        elif isinstance(data, _Rows):
            x = data.upload(torch, dev)[0]
        else:
            x = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
        torch.cuda.current_stream(dev).synchronize()
        c_specs = (_lib.VpTriggerSpec * len(specs))(*[_lib.VpTriggerSpec(r, on, off) for r, _, on, off in specs])
        stacking = _lib.VP_STACK_AVG if args["stacking"] == "avg" else _lib.VP_STACK_MAX
        batch = max(1, min(int(args["batch_size"]), self._max_batch))
        _lib.check(lib.vp_classify_submit(h, 0, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE, n, args["overlap"],
                                          args["blinding"][0], args["blinding"][1], stacking, batch, c_specs,
                                          len(specs), None, _lib.VP_MEM_DEVICE, cap), "vp_classify_submit")
        return {"ctx": ctx, "x": x, "cap": cap, "data": data}  # x must outlive the submit

    def _collect_block(self, job, args, specs, columns=False):
        """Wait for a submitted block -> ([(spec_index, on, off, peak, value)], n_windows)
        (``columns=True``: the triggers as five arrays)."""
        lib = _lib.load()
        h = self._context(job["ctx"])
        cap = job["cap"]
        on, off, peak = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
        val, spec_of, found = (C.c_float * cap)(), (C.c_int32 * cap)(), C.c_int()
        fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.vp_classify_collect(h, 0, C.byref(fv), C.byref(lv), C.byref(nw), on, off, peak, val, spec_of,
                                           cap, C.byref(found)), "vp_classify_collect")
        if found.value > cap:  # rare: more triggers than the result block holds -> redo this block with room
            job2 = self._submit_block(job["ctx"], job["data"], args, specs, found.value)
            return self._collect_block(job2, args, specs, columns)
        m = found.value
        if columns:
            col = lambda a, dt: np.frombuffer(a, dtype=dt, count=m).copy() if m else np.empty(0, dt)
            return (col(spec_of, np.int32), col(on, np.int64), col(off, np.int64), col(peak, np.int64), col(val, np.float32)), nw.value
        return [(spec_of[i], on[i], off[i], peak[i], val[i]) for i in range(m)], nw.value

    def _classify_blocks(self, groups, args, specs, cap_per_row=256):
        """Blocks of several stations in ONE library call -> one trigger list per block."""
        torch = _torch()
        lib = _lib.load()
        h = self._context(0)
        dev = torch.device("cuda", self._device_index)
        K = len(groups)
        lens = np.array([g["data"].shape[1] for g in groups], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(3 * lens)[:-1]]).astype(np.int64)
        if all(torch.is_tensor(g["data"]) for g in groups):
            flat = torch.cat([g["data"].to(dev, torch.float32).reshape(-1) for g in groups])
        else:
            host = np.concatenate([(g["data"].cpu().numpy() if torch.is_tensor(g["data"]) else
                                    np.asarray(g["data"], dtype=np.float32)).reshape(-1) for g in groups])
            flat = torch.from_numpy(host).to(dev)
        torch.cuda.current_stream(dev).synchronize()
        c_specs = (_lib.VpTriggerSpec * len(specs))(*[_lib.VpTriggerSpec(r, on, off) for r, _, on, off in specs])
        stacking = _lib.VP_STACK_AVG if args["stacking"] == "avg" else _lib.VP_STACK_MAX
        batch = max(1, min(int(args["batch_size"]), self._max_batch))
        I64 = C.POINTER(C.c_int64)
        while True:
            cap = K * max(1, len(specs)) * cap_per_row
            on, off, peak = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
            val, spec_of, block_of = np.empty(cap, np.float32), np.empty(cap, np.int32), np.empty(cap, np.int32)
            found = C.c_int()
            _lib.check(lib.vp_classify_multi(
                h, C.c_void_p(flat.data_ptr()), _lib.VP_MEM_DEVICE, offsets.ctypes.data_as(I64), lens.ctypes.data_as(I64), K,
                args["overlap"], args["blinding"][0], args["blinding"][1], stacking, batch, c_specs, len(specs), None,
                _lib.VP_MEM_DEVICE, None, None, None, on.ctypes.data_as(I64), off.ctypes.data_as(I64),
                peak.ctypes.data_as(I64), val.ctypes.data_as(C.POINTER(C.c_float)),
                spec_of.ctypes.data_as(C.POINTER(C.c_int32)), block_of.ctypes.data_as(C.POINTER(C.c_int32)), cap_per_row,
                cap, C.byref(found)), "vp_classify_multi")
            if found.value <= cap:
                break
            cap_per_row *= 8  # rare: some row holds more triggers than its slot list
        out = [[] for _ in range(K)]
        for i in range(found.value):
            out[block_of[i]].append((int(spec_of[i]), int(on[i]), int(off[i]), int(peak[i]), float(val[i])))
        return out

    def _classify_block(self, data, args, specs, cap=8192):
        """(3,N) float32 ndarray -> ([(spec_index, on, off, peak, value)], n_windows); indices into the block."""
        return self._collect_block(self._submit_block(0, data, args, specs, cap), args, specs)

    def annotate(self, stream, copy=True, **kwargs):
        """Sliding-window probability traces, one per label, named ``<Model>_<label>``."""
        args = self._argdict(kwargs)
        out = Stream()
        for grp in _group_stream(stream, self.component_order, self.sampling_rate, copy, self.in_samples):
            long_block = self._is_long(grp["data"].shape[1], args)
            dev_out, fv, lv, nw = (self._annotate_segments if long_block else self._annotate_block)(grp["data"], args)
            if nw == 0 or fv < 0:
                continue
            host = dev_out[:, fv : lv + 1].cpu().numpy()
            for i, label in enumerate(self.labels):
                out.append(_make_trace(host[i], grp, fv, self.sampling_rate, f"{self.__class__.__name__}_{label}"))
        return _maybe_obspy(out, stream)

    def classify(self, stream, copy=True, **kwargs):
        """``annotate`` + trigger/peak extraction -> ``ClassifyOutput`` with ``.picks`` (and ``.detections``)."""
        args = self._argdict(kwargs)
        specs = self._trigger_specs(args)
        sr = self.sampling_rate
        # Triggers stay COLUMNS (numpy) from the library's result arrays to the sorted record lists: times in integer
        # microseconds, one stable lexsort by Pick / Detection order (start time, trace id, phase), and the records
        # themselves are built when the caller first touches the list (picks._PrintableList._deferred).
        cols = []  # (group index, spec indices, on, off, peak, value) per emitted trigger list
        tids, t0s = [], []

        def emit(grp, triggers):
            if not isinstance(triggers, tuple):  # [(spec, on, off, peak, value)] of the multi-block call
                if not triggers:
                    return
                z = list(zip(*triggers))
                triggers = (np.asarray(z[0], np.int32), np.asarray(z[1], np.int64), np.asarray(z[2], np.int64),
                            np.asarray(z[3], np.int64), np.asarray(z[4], np.float32))
            if len(triggers[0]):
                cols.append((len(tids),) + triggers)
                tids.append(grp["trace_id"])
                t0s.append(grp["starttime"]._us)

        # Blocks already on the device (read(..., device_resident=True)) are classified several at a time: their
        # windows share the forward batches (SeisBench's batch_size spans the whole stream) and stacking / trigger
        # scan are one launch per chunk of blocks.  Host blocks are pipelined one by one over the device contexts
        # instead -- block i+1 is assembled and enqueued while block i runs -- because for them the host-side
        # assembly and copy, not the GPU, set the pace (measured: 64 stations x 10 min, 8 ms of host assembly
        # against 2.5 ms of GPU work).
        torch = _torch()
        step = self.in_samples - int(args["overlap"])
        chunk, n_win, pending, n_host = [], 0, [], 0

        def flush_chunk():
            if len(chunk) == 1:
                emit(chunk[0], self._collect_block(self._submit_block(0, chunk[0]["data"], args, specs, 8192), args, specs, True)[0])
            elif chunk:
                for g0, triggers in zip(chunk, self._classify_blocks(chunk, args, specs)):
                    emit(g0, triggers)
            chunk.clear()

        tm = self._timing
        t_mark = time.perf_counter()
        long_pending, n_long = [], [0]

        def finish_long():
            while long_pending:
                g0, job = long_pending.pop(0)
                dev_out = job[0] if isinstance(job, tuple) else self._segments_collect(job)[0]
                t2 = time.perf_counter()
                found = self._pick_rows(dev_out, specs, columns=True)
                if tm is not None:
                    tm["pick_scan_d2h_ms"] = tm.get("pick_scan_d2h_ms", 0.0) + (time.perf_counter() - t2) * 1e3
                    t2 = time.perf_counter()
                emit(g0, found)
                if tm is not None:
                    tm["emit_records_ms"] = tm.get("emit_records_ms", 0.0) + (time.perf_counter() - t2) * 1e3

        for grp in _group_stream(stream, self.component_order, sr, copy, self.in_samples):
            if self._is_long(grp["data"].shape[1], args):  # a day-long block: its segments occupy all contexts
                for g0, job in pending:
                    emit(g0, self._collect_block(job, args, specs, True)[0])
                pending = []
                if tm is not None:  # profiled: the phases apart, one block at a time
                    finish_long()
                    tm["host_assembly_ms"] = tm.get("host_assembly_ms", 0.0) + (time.perf_counter() - t_mark) * 1e3
                # Many stations: block k + 1 is uploaded and enqueued (on the other pair of submit slots) BEFORE block k is
                # collected, scanned and emitted -- the host's upload, the longest part of a PhaseNet station-day, then runs
                # beside block k's last segments instead of behind them.
                two_sets = 2 * self._seg_per_context <= _lib.VP_MAX_INFLIGHT  # a second set of submit slots for the block behind
                if not two_sets:
                    finish_long()
                # (a block that is uploaded beside another one's compute goes as ONE segment per context: fewer, larger copies --
                # 2.82 -> 2.59 ms per station-day of eight; the first block of a call keeps the finer cut, whose first upload,
                # which nothing overlaps, is shorter: 3.09 vs 3.32 ms for a single station)
                job = self._segments_submit(grp["data"], args, self._seg_per_context * (n_long[0] & 1) if two_sets else 0,
                                            1 if (two_sets and long_pending) else None)
                n_long[0] += 1
                finish_long()
                long_pending.append((grp, job))
                if tm is not None:
                    finish_long()
                    t_mark = time.perf_counter()
                continue
            finish_long()  # (the short-block paths below use slot 0 of the same contexts)
            if self.batch_across_blocks and torch.is_tensor(grp["data"]):
                nw = (grp["data"].shape[1] - self.in_samples) // step + 2
                if chunk and n_win + nw > self._max_windows_per_call:
                    flush_chunk()
                    n_win = 0
                chunk.append(grp)
                n_win += nw
                continue
            if len(pending) == max(1, self.n_contexts):
                g0, job = pending.pop(0)
                emit(g0, self._collect_block(job, args, specs, True)[0])
            pending.append((grp, self._submit_block(n_host % max(1, self.n_contexts), grp["data"], args, specs, 8192)))
            n_host += 1  # host blocks only: device-resident blocks take the chunk path and must not advance the context
        finish_long()
        for g0, job in pending:
            emit(g0, self._collect_block(job, args, specs, True)[0])
        flush_chunk()
        t_mark = time.perf_counter()
        picks, detections = _records_from_columns(cols, tids, t0s, [sp[1] for sp in specs], sr)
        if tm is not None:
            tm["emit_records_ms"] = tm.get("emit_records_ms", 0.0) + (time.perf_counter() - t_mark) * 1e3
        return ClassifyOutput(self.name, picks=picks, detections=detections)


class PhaseNet(WaveformModel):
    name = "PhaseNet"
    _kind = _lib.VP_MODEL_PHASENET
    _weights_subdir = "phasenet"
    in_samples = 3001
    _annotate_args = dict(WaveformModel._annotate_args, overlap=1500, blinding=(0, 0))
    _annotate_args["*_threshold"] = 0.3

    def __init__(self, in_channels=3, classes=3, phases="NPS", sampling_rate=100, norm="std", **kwargs):
        if in_channels != 3 or classes != 3 or len(phases) != 3:
            raise ValueError("only the released 3-component, 3-class PhaseNet topology is implemented")
        if sampling_rate != 100:
            raise ValueError("only 100 Hz models are implemented")
        super().__init__(norm=norm, **kwargs)
        self.labels = phases
        self.in_channels, self.classes = in_channels, classes

    def get_model_args(self):
        return dict(super().get_model_args(), phases=self.labels)


class EQTransformer(WaveformModel):
    name = "EQTransformer"
    _kind = _lib.VP_MODEL_EQTRANSFORMER
    _weights_subdir = "eqtransformer"
    in_samples = 6000
    default_contexts = 4
    _annotate_args = dict(WaveformModel._annotate_args, overlap=1800, blinding=(500, 500))
    _annotate_args["*_threshold"] = 0.1
    _annotate_args["detection_threshold"] = 0.3

    def __init__(self, in_channels=3, in_samples=6000, classes=2, phases="PS", sampling_rate=100, norm="std",
                 norm_amp_per_comp=False, **kwargs):
        if in_channels != 3 or in_samples != 6000 or classes != 2 or len(phases) != 2:
            raise ValueError("only the released 3-component, 6000-sample, P/S EQTransformer topology is implemented")
        if kwargs.pop("original_compatible", False):
            raise ValueError("original_compatible variants are not part of the volpick path")
        kwargs.pop("lstm_blocks", None), kwargs.pop("drop_rate", None)
        super().__init__(norm=norm, **kwargs)
        self.phases = phases
        self.labels = ["Detection"] + list(phases)
        self.norm_amp_per_comp = bool(norm_amp_per_comp)

    def get_model_args(self):
        args = dict(super().get_model_args(), phases=self.phases)
        if self.norm_amp_per_comp:
            args["norm_amp_per_comp"] = True
        return args

    def _config(self):
        cfg = super()._config()
        cfg.norm_amp_per_comp = int(self.norm_amp_per_comp)
        return cfg

    def _format_output(self, y):
        return tuple(y[:, i] for i in range(3))  # (detection, P, S), each (B, T)


class _Rows:
    """The (C, N) block of one instrument as its C full-length host rows, NOT yet stacked: the upload copies each
    row straight into the device array (for a day-long stream the host-side ``np.stack`` alone costs ~12 ms)."""

    def __init__(self, parts):
        self.parts = parts
        self.shape = (len(parts), len(parts[0]))

    def __array__(self, dtype=None, copy=None):
        a = np.stack(self.parts).astype(np.float32, copy=False)
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return np.asarray(self)[idx]

    def upload(self, torch, dev, out=None, copy_stream=None):
        """-> (device array, temporaries to keep alive until the copies are done).  ``copy_stream``: the copies run under this
        stream while every allocation is made under the caller's current one (torch's caching allocator ties a block to the
        stream it was allocated under and touches that stream again when the block is reused or released)."""
        import contextlib

        x = out if out is not None else torch.empty(self.shape, dtype=torch.float32, device=dev)
        hosts = [torch.from_numpy(np.ascontiguousarray(p)) for p in self.parts]
        # int32 counts (what a miniSEED file decodes to) and float64 rows travel as they are and are cast on the device: the
        # host-side cast of a day-long row costs 20 ms, 30 x its upload.  Rows in page-locked memory
        # (volpick_amd.pinned_array) go by DMA without the runtime's staging copy.
        temps = [None if h.dtype == torch.float32 else torch.empty(h.shape, dtype=h.dtype, device=dev) for h in hosts]
        with (torch.cuda.stream(copy_stream) if copy_stream is not None else contextlib.nullcontext()):
            for c, (h, t) in enumerate(zip(hosts, temps)):
                pinned = h.is_pinned()
                if t is None:
                    x[c].copy_(h, non_blocking=pinned)
                else:
                    t.copy_(h, non_blocking=pinned)
                    x[c].copy_(t)
        return x, [t for t in temps if t is not None]


# --------------------------------------------------------------------------- stream handling
def _group_stream(stream, component_order, sampling_rate, copy, in_samples):
    """Yield one dict per contiguous block of one instrument: data (3,N) float32 in
    ``component_order`` (missing components / gaps inside a block zero-filled), start time and
    ids.  In-repo twin of the array assembly: volpick/data/convert.py:26-70."""
    from .resample import resample_trace

    # traces at the model's rate are never modified, so ``copy`` needs no deep copy for them; the others are
    # resampled as SeisBench's annotate() does (on copies, or in place with copy=False)
    traces = [resample_trace(tr, float(sampling_rate), copy) for tr in stream]
    if len(traces) == 0:
        return
    groups = {}
    for tr in traces:
        s = tr.stats
        groups.setdefault((s.network, s.station, s.location, s.channel[:-1]), []).append(tr)
    comp_alias = {"1": "N", "2": "E", "3": "Z"}  # flexible horizontal components
    for (net, sta, loc, cha), trs in sorted(groups.items()):
        t_start = min(UTCDateTime(t.stats.starttime) for t in trs)
        pieces = []  # (first sample, length, component index or -1, trace)
        for tr in trs:
            comp = tr.stats.channel[-1] if tr.stats.channel else ""
            comp = comp if comp in component_order else comp_alias.get(comp, comp)
            s0 = int(round((UTCDateTime(tr.stats.starttime) - t_start) * sampling_rate))
            npts = int(tr.stats.npts) if getattr(tr, "_dev", None) is not None else len(tr.data)
            pieces.append((s0, npts, component_order.index(comp) if comp in component_order else -1, tr))
        # contiguous covered runs (union over all components) become independent blocks; gaps are not bridged
        runs = []
        for s0, l, _, _ in sorted(pieces, key=lambda p: p[0]):
            if l <= 0:
                continue
            if runs and s0 <= runs[-1][1]:
                runs[-1][1] = max(runs[-1][1], s0 + l)
            else:
                runs.append([s0, s0 + l])
        for b0, b1 in runs:
            if b1 - b0 < in_samples:
                warnings.warn("Parts of the input stream consist of fragments shorter than the number of input "
                              "samples. Output might be empty.")
                continue
            used = [p for p in pieces if p[2] >= 0 and min(p[0] + p[1], b1) > max(p[0], b0)]
            on_device = bool(used) and all(getattr(p[3], "_dev", None) is not None for p in used)
            # the common case -- one full-length trace per component -- is a single stack, no zero fill
            full = sorted((p for p in used if p[0] == b0 and p[1] == b1 - b0), key=lambda p: p[2])
            if len(used) == len(component_order) and [p[2] for p in full] == list(range(len(component_order))):
                if on_device:
                    data = _torch().stack([p[3]._dev for p in full]).float()
                else:
                    data = _Rows([p[3].data.filled(0) if np.ma.isMaskedArray(p[3].data) else p[3].data for p in full])
                yield {
                    "data": data,
                    "starttime": t_start + b0 / sampling_rate,
                    "trace_id": f"{net}.{sta}.{loc}",
                    "network": net, "station": sta, "location": loc,
                }
                continue
            if on_device:  # traces decoded on the GPU (read(..., device_resident=True)): assemble there, no host copy
                torch = _torch()
                data = torch.zeros((len(component_order), b1 - b0), dtype=torch.float32, device=used[0][3]._dev.device)
            else:
                data = np.zeros((len(component_order), b1 - b0), dtype=np.float32)
            for s0, l, c, tr in sorted(pieces, key=lambda p: p[1]):  # shorter first: longer traces win overlaps
                lo, hi = max(s0, b0), min(s0 + l, b1)
                if c < 0 or hi <= lo:
                    continue
                if on_device:
                    data[c, lo - b0 : hi - b0] = tr._dev[lo - s0 : hi - s0]  # casts int counts to float32
                    continue
                d = tr.data[lo - s0 : hi - s0]
                if np.ma.isMaskedArray(d):
                    d = d.filled(0)
                data[c, lo - b0 : hi - b0] = d  # casts int / float64 counts to float32
            yield {
                "data": data,
                "starttime": t_start + b0 / sampling_rate,
                "trace_id": f"{net}.{sta}.{loc}",
                "network": net, "station": sta, "location": loc,
            }


def _make_trace(data, grp, first_valid, sampling_rate, channel):
    return Trace(data, {
        "network": grp["network"], "station": grp["station"], "location": grp["location"], "channel": channel,
        "starttime": grp["starttime"] + first_valid / sampling_rate, "sampling_rate": sampling_rate,
    })


def _maybe_obspy(out, like):
    """Return an obspy.Stream when the caller passed one (and obspy is importable)."""
    if type(like).__module__.startswith("obspy"):
        try:
            import obspy
        except ImportError:
            return out
        st = obspy.Stream()
        for tr in out:
            hdr = dict(tr.stats)
            hdr["starttime"] = obspy.UTCDateTime(tr.stats.starttime.timestamp)
            hdr.pop("npts", None)
            st.append(obspy.Trace(tr.data, hdr))
        return st
    return out
