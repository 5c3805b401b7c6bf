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
