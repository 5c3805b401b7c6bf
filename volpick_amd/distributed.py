"""Multi-GPU use of the path: one process per GPU, weights broadcast once, station streams
(or contiguous window ranges of one stream) partitioned with no data-path collective.

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


def _global_rank(group, group_rank: int) -> int:
    """torch.distributed's object / tensor collectives take the GLOBAL rank of the source; this module's ``src`` /
    ``root`` arguments are ranks WITHIN ``group`` (what RCCL's communicator numbers its members by)."""
    import torch.distributed as dist

    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _all_agree(ok: bool, group, device) -> bool:
    """True iff ``ok`` on every rank of the group (all-reduce MIN): the ranks decide TOGETHER which way to go, so a
    local failure never leaves one rank in a different collective than the others."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class RcclCommunicator:
    """An RCCL communicator owned by libvolpick_hip (``vp_rccl_comm_init``), spanning the ranks of a
    ``torch.distributed`` group.  torch.distributed only carries the 128-byte unique id to the other ranks (the
    out-of-band step any binder has to provide); the collective itself is the library's ``ncclBroadcast``.

    ``src`` / ``root`` are ranks within ``group``.  Construction is collective and every step that can fail on one rank
    alone is followed by an agreement over the torch group: either every rank ends up with a communicator, or every
    rank raises ``RcclUnavailable`` at the same point (and may fall back together)."""

    def __init__(self, device_index: int, src: int = 0, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _lib

        lib = _lib.load()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", int(device_index))
        self._comm = C.c_void_p()
        # 1. can every rank bind RCCL at all?  (a rank that cannot must not leave the others inside ncclCommInitRank)
        have = bool(lib.vp_rccl_available())
        why = "" if have else _lib.last_error()
        if not _all_agree(have, group, dev):
            raise RcclUnavailable("RCCL could not be bound on every rank" + (f" (this rank: {why})" if why else ""))
        # 2. the unique id: the source ALWAYS enters the object broadcast, with None if it could not make one
        ident = C.create_string_buffer(128)
        box = [None]
        if rank == src:
            box = [ident.raw if lib.vp_rccl_unique_id(ident) == 0 else None]
            why = "" if box[0] is not None else _lib.last_error()
        dist.broadcast_object_list(box, src=_global_rank(group, src), group=group)
        if box[0] is None:  # the same value on every rank
            raise RcclUnavailable("vp_rccl_unique_id failed on the source rank" + (f": {why}" if why else ""))
        # 3. the communicator itself (collective inside RCCL), then agreement on its outcome
        rc = lib.vp_rccl_comm_init(int(device_index), world, box[0], rank, C.byref(self._comm))
        why = "" if rc == 0 else _lib.last_error()
        if not _all_agree(rc == 0, group, dev):
            self.close()
            raise RcclUnavailable("vp_rccl_comm_init failed on some rank" + (f" (this rank: {why})" if why else ""))
        # 4. what RCCL itself reports (ncclCommCount / ncclCommUserRank) is part of the agreed outcome: a rank whose
        #    communicator came up with another shape must not leave alone while the others enter ncclBroadcast
        n, r = C.c_int(-1), C.c_int(-1)
        rc = lib.vp_rccl_comm_info(self._comm, C.byref(n), C.byref(r))
        self.n_ranks, self.rank = n.value, r.value
        mine = rc == 0 and (self.n_ranks, self.rank) == (world, rank)
        why = "" if mine else (_lib.last_error() if rc != 0 else
                               f"communicator spans {self.n_ranks} ranks (this one {self.rank}); expected {world} / {rank}")
        if not _all_agree(mine, group, dev):
            self.close()
            raise RcclUnavailable("vp_rccl_comm_info disagrees with the group on some rank" + (f" (this rank: {why})" if why else ""))
        path = C.create_string_buffer(4096)
        self.library_path = path.value.decode() if lib.vp_rccl_library_path(path, len(path)) == 0 else None

    def broadcast(self, tensor, root: int = 0):
        """In-place broadcast of a contiguous fp32 CUDA tensor (``vp_bcast_weights``); ``root`` = rank within the group."""
        import ctypes as C

        from . import _lib

        assert tensor.is_cuda and tensor.is_contiguous() and tensor.dtype.itemsize == 4
        _lib.check(_lib.load().vp_bcast_weights(self._comm, C.c_void_p(tensor.data_ptr()), tensor.numel(), int(root)),
                   "vp_bcast_weights")
        return tensor

    def close(self):
        """Never raises (every rank has to reach the agreement that follows a close)."""
        from . import _lib

        comm, self._comm = self._comm, None
        if comm:
            try:
                _lib.load().vp_rccl_comm_destroy(comm)
            except Exception:  # noqa: BLE001
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RcclUnavailable(RuntimeError):
    """Raised by ``RcclCommunicator`` on EVERY rank alike when the library's communicator cannot be set up."""


LAST_BROADCAST_PATH = None  # which collective the last "nccl" broadcast_weights used (bench.py reports it)
LAST_RCCL_RANKS = None      # ncclCommCount of the library's communicator in that broadcast (None: fallback path)
LAST_RCCL_LIBRARY = None    # the shared object vp_bcast_weights' RCCL entry points were bound from (vp_rccl_library_path)


def broadcast_weights(model, src: int = 0, group=None, create_handle: bool = True):
    """Broadcast ``model``'s flat weight blob from ``src`` (a rank within ``group``) and (on GPU) build the device plan
    straight from the broadcast buffer.  Every rank must hold a model of the same class; ranks
    other than ``src`` may hold arbitrary (e.g. zero) weights of the right size.

    Backend "nccl": the collective is the library's own ``vp_bcast_weights`` (one ``ncclBroadcast`` over RCCL,
    include/volpick_hip.h) and ``vp_create(VP_MEM_DEVICE)`` plans from the received device buffer.  If the library's
    communicator cannot be brought up, ALL ranks learn so together (``RcclCommunicator``) and ALL of them broadcast
    through the group's own communicator instead -- never one rank alone.  Backend "gloo" (CPU tests): a host broadcast
    of the same blob."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    backend = dist.get_backend(group)
    n = int(model._weights.size)
    src_global = _global_rank(group, src)
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.from_numpy(model._weights).to(dev) if dist.get_rank(group) == src else torch.zeros(
            n, dtype=torch.float32, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # the upload is done before RCCL touches the buffer
        global LAST_BROADCAST_PATH, LAST_RCCL_RANKS, LAST_RCCL_LIBRARY
        LAST_RCCL_RANKS = LAST_RCCL_LIBRARY = None
        comm, why = None, ""
        try:
            comm = RcclCommunicator(dev.index, src=src, group=group)
        except RcclUnavailable as e:  # raised on every rank at the same point
            why = str(e)
        done = False
        if comm is not None:
            try:
                comm.broadcast(buf, root=src)
                LAST_RCCL_RANKS, LAST_RCCL_LIBRARY = comm.n_ranks, comm.library_path
                done = True
            except Exception as e:  # noqa: BLE001 -- whatever went wrong here, the agreement below must be reached
                why = str(e)
            comm.close()  # never raises
            done = _all_agree(done, group, dev)  # a failed ncclBroadcast on one rank sends everybody to the fallback
        if done:
            LAST_BROADCAST_PATH = "vp_bcast_weights (ncclBroadcast through the C ABI)"
        else:
            import sys

            print(f"volpick_amd: vp_bcast_weights unavailable ({why}); all ranks broadcast through torch.distributed",
                  file=sys.stderr)
            dist.broadcast(buf, src=src_global, group=group)
            LAST_BROADCAST_PATH = "torch.distributed.broadcast (RCCL), after vp_bcast_weights failed: " + why[:200]
        model._weights = buf.cpu().numpy()
        if create_handle:
            model._release()
            model._device_index = dev.index
            model._ensure_handle(weights_device_ptr=buf.data_ptr())
        return buf
    buf = torch.from_numpy(np.ascontiguousarray(model._weights).copy())
    dist.broadcast(buf, src=src_global, group=group)
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


def stitch_triggers(parts, n_specs):
    """Join the per-rank trigger lists of ONE stream into the unsplit list.

    ``parts`` (in segment order): dicts with ``keep_lo``/``keep_hi`` (the owned output range), ``triggers``
    [(spec, on, off, peak, value)] in stream sample indices found by scanning the owned range only, and
    ``head_end`` [per spec: last sample of the run of samples > thr_off that starts at ``keep_lo``, or -1 if
    sample ``keep_lo`` is not above thr_off].  A trigger whose ``off`` is the last owned sample is open: it
    continues into the next part when that part's head run exists, taking the head run's end as its own, the
    earlier onset, and the larger peak (the earlier one on a tie -- first argmax, as the unsplit scan)."""
    out = []
    for si in range(n_specs):
        open_t = None  # [on, off, peak, value] ending at a cut
        for k, part in enumerate(parts):
            trig = sorted(list(t[1:]) for t in part["triggers"] if t[0] == si)
            he = part["head_end"][si] if k > 0 else -1
            if open_t is not None:
                if he >= 0:  # the run continues across the cut
                    m = next((t for t in trig if t[1] == he), None)  # the head run's own trigger, if it has one
                    if m is not None:
                        trig.remove(m)
                        if m[3] > open_t[3]:
                            open_t[2], open_t[3] = m[2], m[3]
                    open_t[1] = he
                    if he == part["keep_hi"] - 1 and k + 1 < len(parts):
                        continue  # the whole part is inside the run: still open
                out.append((si, *open_t))
                open_t = None
            if trig and trig[-1][1] == part["keep_hi"] - 1 and k + 1 < len(parts):
                open_t = trig.pop()
            out += [(si, *t) for t in trig]
        if open_t is not None:
            out.append((si, *open_t))
    return sorted(out, key=lambda t: (t[0], t[1]))


def _head_run_end(pick_fn, rows, spec, lo, hi):
    """Last sample (index into ``rows``) of the run of samples > thr_off that starts at ``lo``, or -1; scans a
    doubling prefix of [lo, hi) with the trigger scan at (thr_off, thr_off), where runs and triggers coincide."""
    row, label, _, thr_off = spec
    n = 4096
    while True:
        end = min(hi, lo + n)
        found = pick_fn(rows[:, lo:end], [(row, label, thr_off, thr_off)])
        first = min(found, key=lambda t: t[1]) if found else None
        if first is None or first[1] != 0:
            return -1
        if first[2] < end - lo - 1 or end == hi:
            return lo + first[2]
        n *= 4


_EXCHANGE_CAP = 16  # trigger rows per rank that travel WITH the header (classify_stream_sharded): enough for a quiet segment


def _trigger_columns(found):
    """[(spec, on, off, peak, value)] or the five arrays -> the five arrays (int32, int64 x 3, float32)."""
    if isinstance(found, tuple):
        return found
    if not found:
        return (np.empty(0, np.int32), np.empty(0, np.int64), np.empty(0, np.int64), np.empty(0, np.int64), np.empty(0, np.float32))
    z = list(zip(*found))
    return (np.asarray(z[0], np.int32), np.asarray(z[1], np.int64), np.asarray(z[2], np.int64), np.asarray(z[3], np.int64),
            np.asarray(z[4], np.float32))


def stitch_trigger_columns(parts, n_specs):
    """``stitch_triggers`` on columns.  ``parts`` (in segment order): (keep_lo, keep_hi, head_end[n_specs], columns) with the
    columns in stream sample indices.  Only the triggers that touch a cut -- the one that ends on a part's last owned sample
    and the one that ends where the part's head run ends: two per spec and cut at most -- go through the run-joining logic of
    ``stitch_triggers``; everything else stays arrays.  Returns the five columns sorted by (spec, onset)."""
    plain, edge_parts = [], []
    for k, (keep_lo, keep_hi, head_end, cols) in enumerate(parts):
        sp, on, off, pk, v = cols
        touch = np.zeros(len(sp), bool)
        if k + 1 < len(parts):
            touch |= off == keep_hi - 1
        if k > 0:
            he = np.asarray(head_end, np.int64)[sp] if len(sp) else np.empty(0, np.int64)
            touch |= (he >= 0) & (off == he)
        plain.append(tuple(c[~touch] for c in cols))
        idx = np.flatnonzero(touch)
        edge_parts.append(dict(keep_lo=keep_lo, keep_hi=keep_hi, head_end=list(head_end),
                               triggers=[(int(sp[i]), int(on[i]), int(off[i]), int(pk[i]), float(v[i])) for i in idx]))
    joined = _trigger_columns(stitch_triggers(edge_parts, n_specs))
    sp, on, off, pk, v = (np.concatenate([c[i] for c in plain] + [joined[i]]) for i in range(5))
    order = np.lexsort((on, sp))
    return sp[order], on[order], off[order], pk[order], v[order]


def classify_stream_sharded(model, data, starttime, trace_id, group=None, annotate_fn=None, pick_fn=None, timing=None, **kwargs):
    """ONE long (3, N) block spread over the ranks (BASELINE config 4: a 24 h stream on 8 GPUs) ->
    ``ClassifyOutput`` on rank 0, ``None`` elsewhere.

    Rank r takes segment r of ``segments.plan_segments`` -- only its samples [lo, hi) (owned range + halo) are read
    from ``data`` and uploaded -- annotates it, and scans the output range it owns: ONE trigger scan per rank, the trigger
    specs plus, per spec, a (thr_off, thr_off) row whose trigger at the first owned sample (if any) is the run that reaches
    in from the segment before (``head_end``).  What travels is fixed-width integer columns -- a header (count, owned range,
    head ends, the first few trigger rows) all-gathered, and, unless every list was that short, the trigger columns padded to
    the largest count and gathered on rank 0 -- a few KB, no probability rows, no data-path collective, no pickling (SURVEY.md
    section 8e).  Rank 0 joins the runs that cross a cut
    (``stitch_trigger_columns``) and returns record lists that are built on first access, as ``classify()`` does.
    ``data``: a (3, N) array / tensor, or ``(N, load)`` with ``load(lo, hi) -> (3, hi - lo)`` for callers that never hold the
    whole stream.  ``annotate_fn(block) -> (n_out, len)`` and ``pick_fn(rows, specs) -> [(spec, on, off, peak, value)]`` (or
    the five columns) replace the GPU path in the CPU tests (the oracle stands there).  ``timing``: a dict that receives
    this rank's ``total_ms``, ``gpu_ms`` (annotate, synchronised), ``scan_ms``, ``wait_ms`` (a barrier in front of the exchange:
    the wait for the slowest rank, measured only when ``timing`` is given), ``exchange_ms``, ``stitch_ms`` and ``fixed_ms`` =
    total - gpu - wait: the part of a call that does not shrink with the number of ranks."""
    import time

    import torch
    import torch.distributed as dist

    from .models import _records_from_columns
    from .picks import ClassifyOutput
    from .segments import plan_segments

    t_begin = time.perf_counter()
    args = model._argdict(kwargs)
    specs = model._trigger_specs(args)
    n_specs = len(specs)
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    if isinstance(data, tuple):
        n, load = int(data[0]), data[1]
    else:
        n, load = int(data.shape[1]), (lambda lo, hi: data[:, lo:hi])
    segs = plan_segments(n, model.in_samples, args["overlap"], args["blinding"], world)
    if annotate_fn is None:
        def annotate_fn(block):
            fn = model._annotate_segments if model._is_long(block.shape[1], args) else model._annotate_block
            return fn(block, args)[0]
    if pick_fn is None:
        def pick_fn(rows, sp):
            return model._pick_rows(rows, sp, columns=True)
    cols = _trigger_columns([])
    head, keep = [-1] * n_specs, (-1, -1)
    t_gpu = t_scan = 0.0
    if rank < len(segs):  # a short stream has fewer segments than ranks: the surplus ranks own nothing
        sg = segs[rank]
        t0 = time.perf_counter()
        rows = annotate_fn(load(sg["lo"], sg["hi"]))
        rows = rows if torch.is_tensor(rows) else torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32))
        t_gpu = time.perf_counter() - t0
        t0 = time.perf_counter()
        a, b = sg["keep_lo"] - sg["lo"], sg["keep_hi"] - sg["lo"]
        # the run of samples > thr_off that starts at the first owned sample = the first trigger of a (thr_off, thr_off) row
        head_specs = [(row, label, thr_off, thr_off) for row, label, _, thr_off in specs] if rank > 0 else []
        sp, on, off, pk, v = _trigger_columns(pick_fn(rows[:, a:b], list(specs) + head_specs))
        if head_specs:
            is_head = sp >= n_specs
            for i in np.flatnonzero(is_head & (on == 0)):
                head[int(sp[i]) - n_specs] = int(off[i]) + sg["keep_lo"]
            sp, on, off, pk, v = sp[~is_head], on[~is_head], off[~is_head], pk[~is_head], v[~is_head]
        cols = (sp, on + sg["keep_lo"], off + sg["keep_lo"], pk + sg["keep_lo"], v)
        keep = (sg["keep_lo"], sg["keep_hi"])
        t_scan = time.perf_counter() - t0
    t_wait = 0.0
    if timing is not None and world > 1:  # (diagnostic only) the wait for the slowest rank, kept apart from the exchange proper
        t0 = time.perf_counter()
        dist.barrier(group=group)
        t_wait = time.perf_counter() - t0
    t0 = time.perf_counter()
    if world == 1:
        parts = [(keep[0], keep[1], head, cols)]
    else:
        # Two small collectives of integer tensors: the headers (count, owned range, head ends) all-gathered, so that every rank
        # knows the largest count; the trigger columns, padded to it, gathered on rank 0.  (One all-gather of header + a fixed
        # 1024 rows per rank was measured too: 1.1 ms against 0.5 ms for these two over a four-rank gloo group -- 41 KB per rank
        # through the host's loopback; the inline rows are kept for short lists only.)
        CAP = int(_EXCHANGE_CAP)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        m = len(cols[0])

        def pack(width):
            q = np.zeros((width, 5), np.int64)
            k = min(m, width)
            q[:k, 0], q[:k, 1], q[:k, 2], q[:k, 3] = cols[0][:k], cols[1][:k], cols[2][:k], cols[3][:k]
            q[:k, 4] = cols[4][:k].astype(np.float32).view(np.int32)  # the value's bits: nothing is rounded on the way
            return q

        block = np.concatenate([np.asarray([m, keep[0], keep[1]] + head, np.int64), pack(CAP).ravel()])
        mine = torch.from_numpy(block).to(dev)
        blocks = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(blocks, mine, group=group)
        blocks = torch.stack(blocks).cpu().numpy()
        headers, rows_of = blocks[:, :3 + n_specs], [blocks[r, 3 + n_specs:].reshape(CAP, 5) for r in range(world)]
        width = int(headers[:, 0].max())
        if width > CAP:  # (decided by every rank alike, from the same headers)
            mine = torch.from_numpy(pack(width)).to(dev)
            pieces = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
            dist.gather(mine, pieces, dst=_global_rank(group, 0), group=group)
            if rank == 0:
                rows_of = [p_.cpu().numpy() for p_ in pieces]
        if rank == 0:
            parts = []
            for r in range(world):
                cnt = int(headers[r, 0])
                if headers[r, 1] < 0:  # a surplus rank
                    continue
                q = rows_of[r][:cnt]
                parts.append((int(headers[r, 1]), int(headers[r, 2]), headers[r, 3:].tolist(),
                              (q[:, 0].astype(np.int32), q[:, 1].copy(), q[:, 2].copy(), q[:, 3].copy(),
                               q[:, 4].astype(np.int32).view(np.float32))))
    t_exchange = time.perf_counter() - t0
    result, t_stitch = None, 0.0
    if rank == 0:
        t0 = time.perf_counter()
        sp, on, off, pk, v = parts[0][3] if len(parts) == 1 else stitch_trigger_columns(parts, n_specs)
        picks, detections = _records_from_columns([(0, sp, on, off, pk, v)] if len(sp) else [], [trace_id], [starttime._us],
                                                  [s_[1] for s_ in specs], model.sampling_rate)
        result = ClassifyOutput(model.name, picks=picks, detections=detections)
        t_stitch = time.perf_counter() - t0
    if timing is not None:
        total = time.perf_counter() - t_begin
        timing.update(total_ms=total * 1e3, gpu_ms=t_gpu * 1e3, scan_ms=t_scan * 1e3, wait_ms=t_wait * 1e3, exchange_ms=t_exchange * 1e3,
                      stitch_ms=t_stitch * 1e3, fixed_ms=(total - t_gpu - t_wait) * 1e3)
    return result
