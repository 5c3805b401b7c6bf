import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import volpick_amd as va
from volpick_amd import Stream, Trace, UTCDateTime
from volpick_amd.synthetic import synthetic_stream_array
t0 = UTCDateTime("2020-01-01T00:00:00")
trs = []
NS = 64
for s in range(NS):
    data, _, _ = synthetic_stream_array(60_000, seed=100 + s, n_events=4)
    for i, c in enumerate("ZNE"):
        trs.append(Trace(data[i], dict(network="XX", station=f"S{s:03d}", location="", channel="HH" + c, starttime=t0, sampling_rate=100.0)))
st = Stream(trs)
for cls in (va.PhaseNet, va.EQTransformer):
    m = cls.from_pretrained("volpick").cuda()
    m.classify(st)
    t = time.perf_counter()
    out = m.classify(st)
    dt = time.perf_counter() - t
    nwin = NS * (39 if cls is va.PhaseNet else len(range(0, 60000 - 6000 + 1, 6000 - 1800)) + 1)
    print(cls.__name__, f"{NS} stations x 10 min, host traces: classify {dt*1e3:.1f} ms, {len(out.picks)} picks, {dt/NS*1e6:.0f} us per station")
    dst = Stream([Trace(header=dict(tr.stats), device_data=torch.from_numpy(tr.data).cuda()) for tr in st])
    m.classify(dst)
    t = time.perf_counter()
    out2 = m.classify(dst)
    dt2 = time.perf_counter() - t
    m.batch_across_blocks = False
    m.classify(dst)
    t = time.perf_counter()
    out3 = m.classify(dst)
    dt3 = time.perf_counter() - t
    assert len(out2.picks) == len(out.picks) == len(out3.picks)
    print(cls.__name__, f"   device-resident traces: {dt2*1e3:.1f} ms batched across stations, {dt3*1e3:.1f} ms block by block")
