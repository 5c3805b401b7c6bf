"""CPU, world_size 2, gloo: the multi-GPU path's host logic — weight broadcast, contiguous
sharding and pick gathering."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import pipeline as OP
from volpick_amd.distributed import _head_run_end, shard_range, stitch_triggers


def test_shard_range_partitions():
    for n in (0, 1, 7, 8, 17269, 100):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(5, 2, 2)


def test_stitch_triggers_equals_unsplit_scan():
    """Triggers found rank by rank on the owned ranges + the head-run ends, stitched, are exactly the triggers of
    the unsplit trace: runs crossing one or several cuts, thr_off < thr_on, NaN gaps, a run open at the end."""
    import torch

    rng = np.random.default_rng(5)

    def pick_fn(rows, specs):
        rows = rows.numpy() if torch.is_tensor(rows) else rows
        return [(si, *t) for si, (row, _, t_on, t_off) in enumerate(specs) for t in OP.picks_from_trace(rows[row], t_on, t_off)]

    for trial in range(120):
        n = int(rng.integers(200, 3000))
        x = np.clip(np.cumsum(rng.standard_normal(n)) * 0.12 + 0.35, 0, 1).astype(np.float32)
        if trial % 4 == 0:
            x[rng.integers(0, n, size=5)] = np.nan
        if trial % 5 == 0:
            x[int(n * 0.3):int(n * 0.9)] = 0.95  # one run across several cuts
        if trial % 7 == 0:
            x[-50:] = 0.9  # open at the end
        thr = float(rng.uniform(0.2, 0.7))
        specs = [(0, "P", thr, thr), (1, "Detection", thr, thr / 2)]
        rows = np.stack([x, x[::-1].copy()])
        parts_n = int(rng.integers(2, 7))
        cuts = [0] + sorted(rng.choice(np.arange(1, n), size=parts_n - 1, replace=False).tolist()) + [n]
        parts = []
        for k in range(parts_n):
            a, b = cuts[k], cuts[k + 1]
            found = pick_fn(rows[:, a:b], specs)
            head = [(-1 if k == 0 else _head_run_end(pick_fn, torch.from_numpy(rows), sp, a, b)) for sp in specs]
            parts.append(dict(keep_lo=a, keep_hi=b, head_end=head,
                              triggers=[(si, on + a, off + a, pk + a, v) for si, on, off, pk, v in found]))
        got = stitch_triggers(parts, len(specs))
        want = sorted(pick_fn(rows, specs), key=lambda t: (t[0], t[1]))
        assert got == want, (trial, cuts)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    import volpick_amd as va
    from volpick_amd.distributed import broadcast_weights, classify_sharded
    from volpick_amd.picks import ClassifyOutput, Pick, PickList

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = va.PhaseNet.from_pretrained("volpick")
        ref = model._weights.copy()
        if rank != 0:
            model._weights = np.zeros_like(model._weights)
        broadcast_weights(model, src=0, create_handle=False)
        same = bool(np.array_equal(model._weights, ref))

        # sharded classify with the device call stubbed out: station i yields i+1 picks
        def fake_classify(stream, **kw):
            i = stream
            return ClassifyOutput("PhaseNet", picks=PickList(
                [Pick(f"XX.S{i:02d}.", float(100 * i + k), None, float(100 * i + k), 0.9, "P") for k in range(i + 1)]))

        model.classify = fake_classify
        picks = classify_sharded(model, list(range(5)))
        q.put((rank, same, [(p.trace_id, p.start_time) for p in picks]))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharded_classify_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, same0, picks0), (r1, same1, picks1) = res
    assert same0 and same1, "weights differ after broadcast"
    want = [(f"XX.S{i:02d}.", float(100 * i + k)) for i in range(5) for k in range(i + 1)]
    assert picks0 == sorted(want, key=lambda t: (t[1], t[0]))  # rank 0 holds all 15 picks, sorted
    assert picks1 == [w for w in want if w[0] in ("XX.S03.", "XX.S04.")]  # rank 1 owns stations 3-4


def _stream_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import volpick_amd as va
    from oracle import pipeline as OP
    from oracle.models import load_pretrained
    from volpick_amd import UTCDateTime
    from volpick_amd.distributed import classify_stream_sharded
    from volpick_amd.synthetic import synthetic_stream_array

    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = va.PhaseNet.from_pretrained("volpick")
        net = load_pretrained("phasenet")
        T, overlap, blinding = 3001, 2000, (100, 150)
        n = 9 * T + 777
        data, _, _ = synthetic_stream_array(n, seed=31, n_events=4)

        def oracle_annotate(block):  # (3, len) -> stacked (3, len) with NaN where uncovered
            starts = OP.window_starts(block.shape[1], T, overlap)
            preds = OP.predict_windows(net, block, starts, blinding, 64)
            out = OP.reassemble(preds, starts, T, overlap, "avg")
            full = np.full((block.shape[1], 3), np.nan, dtype=np.float32)
            full[: out.shape[0]] = out
            return full.T

        def oracle_pick(rows, specs):
            rows = rows.numpy()
            res = []
            for si, (row, _, t_on, t_off) in enumerate(specs):
                x = np.nan_to_num(rows[row], nan=0.0)
                for a, b in OP.trigger_onset(x, t_on, t_off):
                    res.append((si, int(a), int(b), int(a + np.argmax(x[a:b + 1])), float(x[a:b + 1].max())))
            return res

        t0 = UTCDateTime("2020-02-02T00:00:00")
        kw = dict(overlap=overlap, blinding=blinding, P_threshold=0.3, S_threshold=0.3)
        touched = []

        def load(lo, hi):  # a rank is handed its segment + halo only
            touched.append((lo, hi))
            return data[:, lo:hi]

        got = classify_stream_sharded(model, (n, load), t0, "XX.ONE.", annotate_fn=oracle_annotate, pick_fn=oracle_pick,
                                      **kw)
        assert len(touched) == 1 and (touched[0][0] > 0 if rank else touched[0][1] < n)
        if rank == 0:
            want_rows = torch.from_numpy(np.ascontiguousarray(oracle_annotate(data)))
            want = oracle_pick(want_rows, model._trigger_specs(model._argdict(kw)))
            q.put((rank, [(p.phase, round((p.peak_time - t0) * 100), p.peak_value) for p in got.picks],
                   sorted((("P", "S")[si], pk, v) for si, on, off, pk, v in want)))
        else:
            q.put((rank, got, None))
    finally:
        dist.destroy_process_group()


def test_one_stream_sharded_over_two_ranks_gloo():
    """BASELINE config 4 in miniature: ONE stream split over the ranks by segments.plan_segments, every rank reads
    only its segment, scans the output range it owns, and rank 0 stitches the trigger lists -- equal to the unsplit
    result (oracle on CPU standing in for the GPU path)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, got, want), (_, other, _) = res
    assert other is None and len(got) == len(want) > 0
    for (ph, pk, v), (wph, wpk, wv) in zip(sorted(got), want):
        assert ph == wph and pk == wpk and abs(v - wv) < 1e-6
