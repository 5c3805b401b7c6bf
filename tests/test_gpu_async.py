"""GPU: asynchronous classify (submit / collect), device contexts and error behaviour of the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

import volpick_amd as va
from volpick_amd import _lib
from volpick_amd.synthetic import synthetic_stream_array

pytestmark = pytest.mark.gpu


def _specs(model):
    sp = model._trigger_specs({})
    return sp, (_lib.VpTriggerSpec * len(sp))(*[_lib.VpTriggerSpec(r, a, b) for r, _, a, b in sp])


def _collect(lib, h, slot, cap=512):
    on, off, pk = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
    val, which, n = (C.c_float * cap)(), (C.c_int32 * cap)(), C.c_int()
    fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.vp_classify_collect(h, slot, C.byref(fv), C.byref(lv), C.byref(nw), on, off, pk, val, which, cap,
                                       C.byref(n)))
    return [(which[i], on[i], off[i], pk[i], round(val[i], 6)) for i in range(n.value)], (fv.value, lv.value, nw.value)


def test_submit_collect_matches_sync_and_slots_are_independent():
    lib = _lib.load()
    model = va.PhaseNet.from_pretrained("volpick").cuda()
    h = model._handle
    sp, cs = _specs(model)
    streams = [synthetic_stream_array(20_000 + 3000 * k, seed=40 + k, n_events=3)[0] for k in range(4)]
    dev = [torch.from_numpy(s).cuda() for s in streams]
    want = [model._classify_block(s, model._argdict({}), sp)[0] for s in streams]
    for k in range(4):  # four submits in flight, collected in reverse order
        _lib.check(lib.vp_classify_submit(h, k, C.c_void_p(dev[k].data_ptr()), _lib.VP_MEM_DEVICE, streams[k].shape[1],
                                          1500, 0, 0, _lib.VP_STACK_AVG, 256, cs, len(sp), None, _lib.VP_MEM_DEVICE, 512))
    for k in (3, 1, 0, 2):
        got, (fv, lv, nw) = _collect(lib, h, k)
        assert got == [(a, b, c, d, round(e, 6)) for a, b, c, d, e in want[k]]
        assert (fv, lv) == (0, streams[k].shape[1] - 1) and nw > 0
    # error behaviour: busy slot, empty slot, bad arguments
    _lib.check(lib.vp_classify_submit(h, 0, C.c_void_p(dev[0].data_ptr()), _lib.VP_MEM_DEVICE, streams[0].shape[1], 1500,
                                      0, 0, _lib.VP_STACK_AVG, 256, cs, len(sp), None, _lib.VP_MEM_DEVICE, 512))
    rc = lib.vp_classify_submit(h, 0, C.c_void_p(dev[0].data_ptr()), _lib.VP_MEM_DEVICE, streams[0].shape[1], 1500, 0, 0,
                                _lib.VP_STACK_AVG, 256, cs, len(sp), None, _lib.VP_MEM_DEVICE, 512)
    assert rc < 0 and b"uncollected" in lib.vp_last_error()
    _collect(lib, h, 0)
    n = C.c_int()
    assert lib.vp_classify_collect(h, 0, None, None, None, None, None, None, None, None, 0, C.byref(n)) < 0
    assert lib.vp_classify_submit(h, 9, C.c_void_p(dev[0].data_ptr()), 1, 100, 0, 0, 0, 0, 256, cs, len(sp), None, 1, 8) < 0
    bad = (_lib.VpTriggerSpec * 1)(_lib.VpTriggerSpec(0, 0.1, 0.5))  # thr_off > thr_on
    assert lib.vp_classify_submit(h, 1, C.c_void_p(dev[0].data_ptr()), 1, streams[0].shape[1], 1500, 0, 0, 0, 256, bad, 1,
                                  None, 1, 8) < 0


def test_multi_station_classify_equals_per_station(monkeypatch):
    """classify() pipelines station blocks over three device contexts; the result must equal
    classifying every station on its own."""
    model = va.EQTransformer.from_pretrained("volpick").cuda()
    t0 = va.UTCDateTime("2021-06-01T00:00:00")
    stations, full = [], va.Stream()
    for k in range(5):
        data, _, _ = synthetic_stream_array(14_000 + 1000 * k, seed=90 + k, n_events=2)
        st = va.Stream([va.Trace(data[i], dict(network="XX", station=f"S{k}", channel=f"HH{c}", starttime=t0 + k,
                                               sampling_rate=100.0)) for i, c in enumerate("ZNE")])
        stations.append(st)
        full += st
    together = model.classify(full, overlap=3000)
    model.n_contexts = 1
    apart = [model.classify(st, overlap=3000) for st in stations]
    want = sorted(p for r in apart for p in r.picks)
    assert len(together.picks) == len(want) > 0
    for a, b in zip(together.picks, want):
        assert (a.trace_id, a.phase, a.peak_time, a.start_time, a.end_time) == (b.trace_id, b.phase, b.peak_time,
                                                                                b.start_time, b.end_time)
        assert a.peak_value == b.peak_value
    assert len(together.detections) == sum(len(r.detections) for r in apart)


def test_stage_timing_is_opt_in():
    lib = _lib.load()
    model = va.PhaseNet.from_pretrained("volpick").cuda()
    data, _, _ = synthetic_stream_array(30_000, seed=2)
    sp = model._trigger_specs({})
    total, stage = C.c_float(), (C.c_float * 4)()
    lib.vp_set_timing(model._handle, 1)
    model._classify_block(data, model._argdict({}), sp)
    lib.vp_last_timing(model._handle, C.byref(total), stage)
    assert stage[1] > 0 and stage[2] > 0 and stage[3] > 0 and abs(total.value - sum(stage)) < 1e-3
    lib.vp_set_timing(model._handle, 0)


def _many_stations(n_sta, seed0=300):
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.synthetic import synthetic_stream_array

    t0 = UTCDateTime("2021-03-04T05:06:07")
    trs, arrays = [], {}
    for s in range(n_sta):
        n = 30_000 + 4_321 * (s % 5)  # ragged lengths
        data, _, _ = synthetic_stream_array(n, seed=seed0 + s, n_events=3)
        arrays[f"S{s:02d}"] = data
        for i, c in enumerate("ZNE"):
            trs.append(Trace(data[i], dict(network="XX", station=f"S{s:02d}", location="", channel="HH" + c,
                                           starttime=t0 + 7.0 * s, sampling_rate=100.0)))
    return Stream(trs), arrays


@pytest.mark.parametrize("cls", [va.PhaseNet, va.EQTransformer])
def test_windows_of_many_stations_share_batches(cls):
    """classify() of a multi-station stream through vp_classify_multi (windows of all blocks fill the forward
    batches together) gives exactly the picks of the block-by-block path."""
    import torch

    from volpick_amd import Stream, Trace

    host_st, _ = _many_stations(9)
    # the same traces backed by device arrays (what read(..., device_resident=True) produces)
    st = Stream([Trace(header=dict(tr.stats), device_data=torch.from_numpy(tr.data).cuda()) for tr in host_st])
    m = cls.from_pretrained("volpick").cuda()
    kw = dict(overlap=m.in_samples // 2, blinding=(250, 250), P_threshold=0.2, S_threshold=0.2)
    a = m.classify(st, **kw)          # device blocks: vp_classify_multi
    b = m.classify(host_st, **kw)     # host blocks: one vp_classify_submit per block
    assert len(a.picks) == len(b.picks) > 0 and len(a.detections) == len(b.detections)
    for p, q in zip(a.picks, b.picks):
        assert (p.trace_id, p.phase, p.peak_time, p.start_time, p.end_time, p.peak_value) == (
            q.trace_id, q.phase, q.peak_time, q.start_time, q.end_time, q.peak_value)
    for p, q in zip(a.detections, b.detections):
        assert (p.trace_id, p.start_time, p.end_time, p.peak_value) == (q.trace_id, q.start_time, q.end_time, q.peak_value)
    # a tiny per-row capacity takes the retry path and ends in the same place
    groups = list(__import__("volpick_amd.models", fromlist=["_group_stream"])._group_stream(
        st, m.component_order, m.sampling_rate, True, m.in_samples))
    args = m._argdict(kw)
    specs = m._trigger_specs(args)
    small = m._classify_blocks(groups, args, specs, cap_per_row=1)
    big = m._classify_blocks(groups, args, specs, cap_per_row=512)
    assert small == big and sum(len(t) for t in big) == len(a.picks) + len(a.detections)
    # chunking by the window budget changes nothing either
    m._max_windows_per_call = 40
    c = m.classify(st, **kw)
    m.batch_across_blocks = False
    d = m.classify(st, **kw)
    assert len(d.picks) == len(a.picks)
    assert [(p.trace_id, p.phase, p.peak_time, p.peak_value) for p in c.picks] == [
        (p.trace_id, p.phase, p.peak_time, p.peak_value) for p in a.picks]


def test_classify_multi_rows_match_annotate_and_argument_errors():
    """C ABI: the stacked rows of every block equal vp_annotate of that block alone (NaN pattern included)."""
    import ctypes as C

    from volpick_amd import _lib

    lib = _lib.load()
    m = va.PhaseNet.from_pretrained("volpick").cuda()
    h = m._ensure_handle()
    _, arrays = _many_stations(4, seed0=700)
    blocks = [arrays[k] for k in sorted(arrays)]
    blocks.append(blocks[0][:, :2000])  # shorter than one window: no windows, all-NaN rows
    lens = np.array([b.shape[1] for b in blocks], np.int64)
    offs = np.concatenate([[0], np.cumsum(3 * lens)[:-1]]).astype(np.int64)
    flat = np.concatenate([b.reshape(-1) for b in blocks]).astype(np.float32)
    out = np.empty_like(flat)
    K = len(blocks)
    fv, lv, nw = np.zeros(K, np.int64), np.zeros(K, np.int64), np.zeros(K, np.int64)
    cap = 4096
    on, off, pk = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
    val, so, bo = np.empty(cap, np.float32), np.empty(cap, np.int32), np.empty(cap, np.int32)
    found = C.c_int()
    specs = (_lib.VpTriggerSpec * 2)(_lib.VpTriggerSpec(0, 0.3, 0.3), _lib.VpTriggerSpec(1, 0.3, 0.3))
    I64 = C.POINTER(C.c_int64)

    def call(k, overlap=1500, cap_row=256):
        return lib.vp_classify_multi(
            h, flat.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, offs.ctypes.data_as(I64), lens.ctypes.data_as(I64), k,
            overlap, 100, 200, _lib.VP_STACK_AVG, 64, specs, 2, out.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST,
            fv.ctypes.data_as(I64), lv.ctypes.data_as(I64), nw.ctypes.data_as(I64), on.ctypes.data_as(I64),
            off.ctypes.data_as(I64), pk.ctypes.data_as(I64), val.ctypes.data_as(C.POINTER(C.c_float)),
            so.ctypes.data_as(C.POINTER(C.c_int32)), bo.ctypes.data_as(C.POINTER(C.c_int32)), cap_row, cap, C.byref(found))

    assert call(K) == 0
    assert nw[-1] == 0 and fv[-1] == -1 and np.isnan(out[offs[-1]:]).all()
    n_trig = 0
    for k, b in enumerate(blocks[:-1]):
        n = b.shape[1]
        want = np.empty((3, n), np.float32)
        f1, l1, n1 = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.vp_annotate(h, np.ascontiguousarray(b).ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST, n, 1500, 100,
                                   200, _lib.VP_STACK_AVG, 64, want.ctypes.data_as(C.c_void_p), _lib.VP_MEM_HOST,
                                   C.byref(f1), C.byref(l1), C.byref(n1)))
        got = out[offs[k]:offs[k] + 3 * n].reshape(3, n)
        assert (fv[k], lv[k], nw[k]) == (f1.value, l1.value, n1.value)
        assert np.array_equal(got, want, equal_nan=True)
        sel = bo[:found.value] == k
        n_trig += int(sel.sum())
        for i in np.flatnonzero(sel):  # every trigger sits on a peak of its row, inside the valid range
            row = want[so[i]]
            assert fv[k] <= on[i] <= pk[i] <= off[i] <= lv[k] and row[pk[i]] == val[i] == np.nanmax(row[on[i]:off[i] + 1])
    assert n_trig == found.value > 0
    assert call(0) == -1 and call(K, overlap=3001) == -1 and call(K, cap_row=0) == -1


@pytest.mark.parametrize("cls,overlap,blinding,stacking", [(va.PhaseNet, 1500, (0, 0), "avg"),
                                                           (va.PhaseNet, 2700, (150, 250), "max"),
                                                           (va.EQTransformer, 5500, (500, 500), "avg")])
def test_long_block_over_the_contexts_is_bitwise_the_unsplit_result(cls, overlap, blinding, stacking):
    """A long block is cut into one segment per device context (volpick_amd/segments.py); outputs, valid range and
    picks equal the single-context result exactly."""
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.synthetic import synthetic_stream_array

    m = cls.from_pretrained("volpick")
    m._max_batch = 8
    m.cuda()
    T = m.in_samples
    n = T + (T - overlap) * 130 + 17  # > 2 * batch * contexts windows, tail window off the grid
    data, _, _ = synthetic_stream_array(n, seed=5, n_events=max(4, n // 60_000))
    args = m._argdict(dict(overlap=overlap, blinding=blinding, stacking=stacking))
    assert m._is_long(n, args)
    a, fa, la, na = m._annotate_block(data, args)
    b, fb, lb, nb = m._annotate_segments(data, args)
    assert (fa, la, na) == (fb, lb, nb)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)
    t0 = UTCDateTime("2022-02-02T02:02:02")
    st = Stream([Trace(data[i], dict(network="XX", station="LONG", location="", channel="HH" + c, starttime=t0,
                                     sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    kw = dict(overlap=overlap, blinding=blinding, stacking=stacking, P_threshold=0.2, S_threshold=0.2)
    long_out = m.classify(st, **kw)
    m.n_contexts = 1  # never "long": the plain single-block path
    ref_out = m.classify(st, **kw)
    m.n_contexts = 3
    assert len(long_out.picks) == len(ref_out.picks) > 0 and len(long_out.detections) == len(ref_out.detections)
    for p, q in zip(long_out.picks, ref_out.picks):
        assert (p.phase, p.start_time, p.end_time, p.peak_time, p.peak_value) == (
            q.phase, q.start_time, q.end_time, q.peak_time, q.peak_value)
    ann_a = m.annotate(st, **{k: kw[k] for k in ("overlap", "blinding", "stacking")})
    m.n_contexts = 1
    ann_b = m.annotate(st, **{k: kw[k] for k in ("overlap", "blinding", "stacking")})
    for x, y in zip(ann_a, ann_b):
        assert x.stats.starttime == y.stats.starttime and np.array_equal(x.data, y.data, equal_nan=True)


def test_stream_sharded_entry_point_on_one_rank_equals_classify():
    """distributed.classify_stream_sharded with the real GPU path (world size 1 here; the two-rank exchange is
    covered on CPU by tests/test_distributed_cpu.py)."""
    from volpick_amd import Stream, Trace, UTCDateTime
    from volpick_amd.distributed import classify_stream_sharded
    from volpick_amd.synthetic import synthetic_stream_array

    m = va.EQTransformer.from_pretrained("volpick").cuda()
    n = 6000 * 20 + 123
    data, _, _ = synthetic_stream_array(n, seed=9, n_events=5)
    t0 = UTCDateTime("2023-03-03T03:03:03")
    kw = dict(overlap=5500, blinding=(500, 500), P_threshold=0.2, S_threshold=0.2)
    got = classify_stream_sharded(m, data, t0, "XX.ONE.", **kw)
    st = Stream([Trace(data[i], dict(network="XX", station="ONE", location="", channel="HH" + c, starttime=t0,
                                     sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    want = m.classify(st, **kw)
    assert len(got.picks) == len(want.picks) > 0 and len(got.detections) == len(want.detections)
    for p, q in zip(got.picks, want.picks):
        assert (p.trace_id, p.phase, p.peak_time, p.peak_value) == (q.trace_id, q.phase, q.peak_time, q.peak_value)


@pytest.mark.parametrize("pinned", [False, True])
def test_many_long_stations_equal_per_station_calls(pinned):
    """classify() on a host Stream of several LONG station blocks (each spread over the device contexts segment by segment), from
    pageable rows or from rows in page-locked memory (volpick_amd.pinned_array: asynchronous DMA); short blocks and int32 counts
    mixed in.  Every station's pick list equals the one-station call's."""
    from volpick_amd import Stream, Trace, UTCDateTime, pinned_array
    from volpick_amd.synthetic import synthetic_stream_array

    m = va.PhaseNet.from_pretrained("volpick")
    m._max_batch = 8
    m.cuda()
    T, overlap = m.in_samples, 1500
    n_long = T + (T - overlap) * 90 + 5
    assert m._is_long(n_long, m._argdict(dict(overlap=overlap)))
    t0 = UTCDateTime("2023-03-03T03:03:03")
    stations, full = [], Stream()
    for k, n in enumerate([n_long, n_long + 777, 20_000, n_long, 3 * T]):
        data, _, _ = synthetic_stream_array(n, seed=700 + k, n_events=max(2, n // 50_000))
        rows = []
        for i in range(3):
            src = (data[i] * 1000).astype(np.int32) if k == 1 else data[i]  # one station as counts: cast on the device
            if pinned:
                row = pinned_array(n, src.dtype)
                row[:] = src
                src = row
            rows.append(src)
        st = Stream([Trace(rows[i], dict(network="XX", station=f"M{k}", location="", channel="HH" + c, starttime=t0 + 10 * k,
                                         sampling_rate=100.0)) for i, c in enumerate("ZNE")])
        stations.append(st)
        full += st
    kw = dict(overlap=overlap, P_threshold=0.25, S_threshold=0.25)
    together = m.classify(full, **kw)
    apart = sorted(p for st in stations for p in m.classify(st, **kw).picks)
    assert len(together.picks) == len(apart) > 10
    for a, b in zip(together.picks, apart):
        assert (a.trace_id, a.phase, a.start_time, a.end_time, a.peak_time, a.peak_value) == (
            b.trace_id, b.phase, b.start_time, b.end_time, b.peak_time, b.peak_value)
