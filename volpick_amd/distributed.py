"""Multi-GPU use of the path: one process per GPU, weights broadcast once, station streams
(or contiguous window ranges) partitioned with no data-path collective.

The reference is single-GPU; windows are independent given the weights,
so the only exchange is the start-up broadcast of the flat fp32 weight blob (1.08 MB PhaseNet /
1.52 MB EQTransformer) from rank 0 — ``torch.distributed`` backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU for the tests.  The payload is latency-bound (~10 us of one xGMI link), so a
single flat broadcast is the right collective; picks return through the host.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous [lo, hi) share of ``n_items`` for ``rank``; sizes differ by at most one."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(model, src: int = 0, group=None, create_handle: bool = True):
    """Broadcast ``model``'s flat weight blob from ``src`` and (on GPU) build the device plan
    straight from the broadcast buffer.  Every rank must hold a model of the same class; ranks
    other than ``src`` may hold arbitrary (e.g. zero) weights of the right size."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    backend = dist.get_backend(group)
    n = int(model._weights.size)
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.from_numpy(model._weights).to(dev) if dist.get_rank(group) == src else torch.empty(
            n, dtype=torch.float32, device=dev)
        dist.broadcast(buf, src=src, group=group)
        torch.cuda.current_stream(dev).synchronize()
        model._weights = buf.cpu().numpy()
        if create_handle:
            model._release()
            model._device_index = dev.index
            model._ensure_handle(weights_device_ptr=buf.data_ptr())
        return buf
    buf = torch.from_numpy(np.ascontiguousarray(model._weights).copy())
    dist.broadcast(buf, src=src, group=group)
    model._weights = buf.numpy().copy()
    model._release()
    return buf


def classify_sharded(model, streams, group=None, **kwargs):
    """Each rank classifies its contiguous share of ``streams`` (one Stream per station);
    rank 0 returns the concatenated, sorted pick list (other ranks return their own)."""
    import torch.distributed as dist

    from .picks import PickList

    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    lo, hi = shard_range(len(streams), rank, world)
    mine = PickList()
    for st in streams[lo:hi]:
        mine += model.classify(st, **kwargs).picks
    if world == 1:
        return PickList(sorted(mine))
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(mine), gathered, dst=0, group=group)
    if rank == 0:
        return PickList(sorted(p for part in gathered for p in part))
    return PickList(sorted(mine))


def annotate_stream_sharded(model, data, group=None, annotate_fn=None, **kwargs):
    """ONE long (3, N) block spread over the ranks (BASELINE config 4: a 24 h stream on 8 GPUs): rank r annotates
    segment r of ``segments.plan_segments`` and keeps the output samples it owns; one gather (RCCL over xGMI under
    the "nccl" backend -- 13 MB per rank for a station-day on 8 GPUs) brings the pieces to rank 0, which joins them
    into exactly the unsplit (n_out, N) result.  Every rank passes the same ``data``.

    Returns (array (n_out, N) with NaN outside the valid range, first_valid, last_valid) on rank 0 and
    (None, first_valid, last_valid) elsewhere.  ``annotate_fn(block) -> (n_out, len)`` replaces the model's GPU
    path (the CPU tests put the oracle there)."""
    import torch
    import torch.distributed as dist

    from .segments import plan_segments

    args = model._argdict(kwargs)
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    n = int(data.shape[1])
    T = model.in_samples
    segs = plan_segments(n, T, args["overlap"], args["blinding"], world)
    on_gpu = annotate_fn is None
    if on_gpu:
        dev = torch.device("cuda", model._device_index if model._device_index is not None else torch.cuda.current_device())

        def annotate_fn(block):
            fn = model._annotate_segments if model._is_long(block.shape[1], args) else model._annotate_block
            return fn(block, args)[0]
    else:
        dev = torch.device("cpu")
    width = max(sg["keep_hi"] - sg["keep_lo"] for sg in segs)
    mine = torch.full((3, width), float("nan"), dtype=torch.float32, device=dev)
    if rank < len(segs):  # with fewer segments than ranks (short stream) the surplus ranks send NaN
        sg = segs[rank]
        block = data[:, sg["lo"]:sg["hi"]]
        y = annotate_fn(block)
        y = y if torch.is_tensor(y) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))
        mine[:, : sg["keep_hi"] - sg["keep_lo"]] = y.to(dev)[:, sg["keep_lo"] - sg["lo"]:sg["keep_hi"] - sg["lo"]]
    # valid range of the whole stream
    step = T - args["overlap"]
    n_reg = (n - T) // step + 1 if n >= T else 0
    tail = 1 if n_reg > 0 and (n_reg - 1) * step + T < n else 0
    fv = args["blinding"][0] if n_reg > 0 else -1
    lv = ((n - T) if tail else (n_reg - 1) * step) + T - args["blinding"][1] - 1 if n_reg > 0 else -1
    if world == 1:
        pieces = [mine]
    else:
        pieces = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, pieces, dst=0, group=group)
    if rank != 0:
        return None, fv, lv
    out = torch.full((3, n), float("nan"), dtype=torch.float32, device=dev)
    for sg, piece in zip(segs, pieces):
        out[:, sg["keep_lo"]:sg["keep_hi"]] = piece[:, : sg["keep_hi"] - sg["keep_lo"]]
    return out, fv, lv


def classify_stream_sharded(model, data, starttime, trace_id, group=None, annotate_fn=None, pick_fn=None, **kwargs):
    """``annotate_stream_sharded`` + the trigger scan on rank 0 -> ``ClassifyOutput`` there, ``None`` elsewhere.
    ``pick_fn(rows (n_out, N), specs) -> [(spec_index, on, off, peak, value)]`` replaces the GPU scan in CPU tests."""
    from .picks import ClassifyOutput, Detection, DetectionList, Pick, PickList

    out, fv, lv = annotate_stream_sharded(model, data, group=group, annotate_fn=annotate_fn, **kwargs)
    if out is None:
        return None
    args = model._argdict(kwargs)
    specs = model._trigger_specs(args)
    triggers = (pick_fn or model._pick_rows)(out, specs)
    sr = model.sampling_rate
    picks, detections = PickList(), DetectionList()
    for si, on, off, pk, v in triggers:
        label = specs[si][1]
        if label == "Detection":
            detections.append(Detection(trace_id, starttime + on / sr, starttime + off / sr, v))
        else:
            picks.append(Pick(trace_id, starttime + on / sr, starttime + off / sr, starttime + pk / sr, v, label))
    return ClassifyOutput(model.name, picks=PickList(sorted(picks)), detections=DetectionList(sorted(detections)))
