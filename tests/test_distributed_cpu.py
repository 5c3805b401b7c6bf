"""CPU, world_size 2, gloo: the multi-GPU path's host logic — weight broadcast, contiguous
sharding and pick gathering."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from volpick_amd.distributed import shard_range


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
