"""CPU: host-side logic of the product package and the C ABI surface (no GPU compute)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import volpick_amd as va
from oracle import pipeline as OP
from volpick_amd import _lib
from volpick_amd.models import _group_stream

ROOT = Path(__file__).resolve().parents[1]


def test_every_declared_symbol_is_exported_and_bound(lib):
    hdr = (ROOT / "include" / "volpick_hip.h").read_text()
    declared = set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in volpick_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) <= declared
    assert b"gfx950" in lib.vp_version()


def test_default_config_and_error_paths(lib):
    cfg = _lib.VpConfig()
    assert lib.vp_default_config(_lib.VP_MODEL_EQTRANSFORMER, C.byref(cfg)) == 0
    assert (cfg.taper_samples, cfg.max_batch, cfg.norm) == (6, 256, _lib.VP_NORM_PEAK)
    assert abs(cfg.bn_eps - 1e-3) < 1e-9 and abs(cfg.attention_eps - 1e-5) < 1e-12
    assert lib.vp_default_config(7, C.byref(cfg)) < 0 and b"unknown model" in lib.vp_last_error()
    h = C.c_void_p()
    w = np.zeros(10, np.float32)
    rc = lib.vp_create(0, _lib.VP_MODEL_PHASENET, w.ctypes.data_as(C.c_void_p), 10, 0, None, C.byref(h))
    assert rc < 0 and b"expected 269675" in lib.vp_last_error()
    assert lib.vp_weight_count(_lib.VP_MODEL_EQTRANSFORMER) == 378823


def test_constants_agree_with_product(lib):
    """oracle/constants.py (the named switches of SURVEY A.7) == what the product computes with."""
    from oracle import constants as K

    for kind, taper in ((_lib.VP_MODEL_PHASENET, 0), (_lib.VP_MODEL_EQTRANSFORMER, K.EQT_TAPER_SAMPLES)):
        cfg = _lib.VpConfig()
        assert lib.vp_default_config(kind, C.byref(cfg)) == 0
        assert cfg.bn_eps == np.float32(K.BN_EPS) and cfg.attention_eps == np.float32(K.EQT_ATTENTION_EPS)
        assert cfg.layernorm_eps == np.float32(K.EQT_LAYERNORM_EPS) and cfg.norm_eps == np.float32(K.NORM_EPS)
        assert cfg.taper_samples == taper and cfg.max_batch == K.DEFAULT_BATCH_SIZE
    pn, eq = va.PhaseNet._annotate_args, va.EQTransformer._annotate_args
    assert (pn["overlap"], tuple(pn["blinding"]), pn["*_threshold"]) == (
        K.PN_DEFAULTS["overlap"], K.PN_DEFAULTS["blinding"], K.PN_DEFAULTS["threshold"])
    assert (eq["overlap"], tuple(eq["blinding"]), eq["*_threshold"], eq["detection_threshold"]) == (
        K.EQT_DEFAULTS["overlap"], K.EQT_DEFAULTS["blinding"], K.EQT_DEFAULTS["threshold"],
        K.EQT_DEFAULTS["detection_threshold"])
    assert pn["batch_size"] == K.DEFAULT_BATCH_SIZE and pn["stacking"] == K.DEFAULT_STACKING
    assert (va.PhaseNet.in_samples, va.EQTransformer.in_samples) == (K.PN_IN_SAMPLES, K.EQT_IN_SAMPLES)
    assert va.PhaseNet.sampling_rate == K.SAMPLING_RATE


@pytest.mark.parametrize("N,T,ov", [(6890, 6000, 1000), (60_000, 3001, 1500), (3001, 3001, 0), (3000, 3001, 0),
                                    (8_640_000, 6000, 5500), (10_000, 3001, 3000)])
def test_window_starts_matches_oracle(lib, N, T, ov):
    want = OP.window_starts(N, T, ov)
    buf = (C.c_int64 * (len(want) + 4))()
    n = lib.vp_window_starts(N, T, ov, buf, len(buf))
    assert n == len(want) and list(buf[:n]) == want.tolist()
    assert lib.vp_window_starts(N, T, T, buf, len(buf)) < 0


def _pick_host(lib, x, on_thr, off_thr, cap=64):
    on, off, pk = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
    val = (C.c_float * cap)()
    n = C.c_int()
    x = np.ascontiguousarray(x, np.float32)
    assert lib.vp_pick_host(x.ctypes.data_as(C.c_void_p), len(x), on_thr, off_thr, on, off, pk, val, cap, C.byref(n)) == 0
    return [(on[i], off[i], pk[i], val[i]) for i in range(min(n.value, cap))], n.value


def test_pick_host_matches_trigger_onset(lib):
    z = np.load(ROOT / "tests" / "golden" / "trigger_cases.npz")
    got, n = _pick_host(lib, z["x"], .3, .3)
    assert [(a, b) for a, b, _, _ in got] == [(2, 3), (7, 8), (11, 14)] and [g[2] for g in got] == [3, 8, 11]
    rng = np.random.default_rng(0)
    for trial in range(200):
        n_s = int(rng.integers(1, 400))
        x = np.clip(np.cumsum(rng.standard_normal(n_s)) * 0.15 + 0.3, 0, 1).astype(np.float32)
        if trial % 3 == 0:
            x[rng.integers(0, n_s, size=max(1, n_s // 20))] = np.nan
        thr = float(rng.uniform(0.1, 0.8))
        for off_thr in (thr, thr / 2):
            want = OP.picks_from_trace(x, thr, off_thr)
            got, n = _pick_host(lib, x, thr, off_thr, cap=512)
            assert n == len(want)
            for g, w in zip(got, want):
                assert g[:3] == w[:3] and g[3] == pytest.approx(w[3])
    got, n = _pick_host(lib, np.ones(10), .5, .5, cap=0)  # count only
    assert n == 1 and got == []


def test_stream_types_and_grouping():
    t0 = va.UTCDateTime("2005-05-31T21:04:52.110000Z")
    assert str(t0) == "2005-05-31T21:04:52.110000Z" and str(t0 + 18.86) == "2005-05-31T21:05:10.970000Z"
    assert (t0 + 1.5) - t0 == 1.5 and va.UTCDateTime(t0.timestamp) == t0 and t0 < t0 + 0.01
    mk = lambda ch, start, n, sta="AAA": va.Trace(np.arange(n, dtype=np.float32) + 1,
                                                   dict(network="XX", station=sta, channel=ch, starttime=start,
                                                        sampling_rate=100.0))
    st = va.Stream([mk("HHZ", t0, 4000), mk("HHN", t0 + 1.0, 3500), mk("HHE", t0, 4000, sta="BBB")])
    assert st[0].id == "XX.AAA..HHZ" and st[0].stats.endtime == t0 + 39.99
    assert len(st.select(channel="HH?")) == 3 and len(st.select(station="BBB")) == 1 and len(st.copy()) == 3
    groups = list(_group_stream(st, "ZNE", 100.0, True, 3001))
    assert [g["trace_id"] for g in groups] == ["XX.AAA.", "XX.BBB."]
    a = groups[0]["data"]
    assert a.shape == (3, 4000) and a.dtype == np.float32
    assert a[0, 0] == 1 and a[1, 99] == 0 and a[1, 100] == 1 and (a[2] == 0).all()  # N starts 1 s late, E missing
    assert (groups[1]["data"][2] == np.arange(4000) + 1).all() and (groups[1]["data"][:2] == 0).all()
    # a gap splits the station into independent blocks; short fragments are dropped with a warning
    st2 = va.Stream([mk("HHZ", t0, 3200), mk("HHZ", t0 + 60, 1000)])
    with pytest.warns(UserWarning):
        blocks = list(_group_stream(st2, "ZNE", 100.0, True, 3001))
    assert len(blocks) == 1 and blocks[0]["data"].shape == (3, 3200)
    # a trace at another rate is resampled on a copy (SeisBench annotate() semantics), never in place with copy=True
    slow = va.Trace(np.zeros(2000), dict(channel="HHZ", sampling_rate=50.0))
    blocks = list(_group_stream(va.Stream([slow]), "ZNE", 100.0, True, 3001))
    assert len(blocks) == 1 and blocks[0]["data"].shape == (3, 4000) and slow.stats.sampling_rate == 50.0
    list(_group_stream(va.Stream([slow]), "ZNE", 100.0, False, 3001))
    assert slow.stats.sampling_rate == 100.0 and len(slow.data) == 4000  # copy=False: in place, as upstream
    merged = va.Stream([mk("HHZ", t0, 100), mk("HHZ", t0 + 1.0, 100)]).merge(-1)
    assert len(merged) == 1 and len(merged[0].data) == 200


def test_picker_api_surface_without_gpu():
    assert va.PhaseNet.list_pretrained() == ["volpick", "volpick_95train"] == va.EQTransformer.list_pretrained()
    pn = va.PhaseNet.from_pretrained("volpick")
    assert (pn.name, pn.labels, pn.norm, pn.component_order, pn.in_samples) == ("PhaseNet", "PSN", "peak", "ZNE", 3001)
    assert pn.default_args == {"P_threshold": 0.39, "S_threshold": 0.34} and "Zhong" in pn.weights_docstring
    assert str(pn.device) == "cpu" and pn.eval() is pn
    eq = va.EQTransformer.from_pretrained("volpick_95train")
    assert eq.labels == ["Detection", "P", "S"] and eq.default_args["detection_threshold"] == 0.3
    assert eq._threshold({}, "P") == 0.26 and eq._threshold({"P_threshold": 0.5}, "P") == 0.5
    assert eq._threshold({}, "detection") == 0.3
    args = eq._argdict({"overlap": 5500, "blinding": [500, 500], "parallelism": None, "P_threshold": 0.2})
    assert args["overlap"] == 5500 and args["blinding"] == (500, 500) and args["stacking"] == "avg"
    assert pn._argdict({})["overlap"] == 1500 and pn._argdict({})["blinding"] == (0, 0)
    with pytest.raises(ValueError):
        eq._argdict({"stacking": "median"})
    with pytest.raises(ValueError):
        eq._argdict({"overlap": 6000})
    with pytest.warns(UserWarning):
        eq._argdict({"not_an_arg": 1})
    with pytest.raises(ValueError):
        va.PhaseNet.from_pretrained("original")
    with pytest.raises(KeyError):
        va.PhaseNet().load_state_dict({"inc.weight": np.zeros((8, 3, 7))})
    p = va.Pick("NC.MMT.", va.UTCDateTime(0), va.UTCDateTime(1), va.UTCDateTime("2005-05-31T21:05:10.97"), 0.5, "P")
    assert str(va.PickList([p])) == "PickList with 1 entries:\n\nNC.MMT.\t2005-05-31T21:05:10.970000Z\tP"


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pn = va.PhaseNet.from_pretrained("volpick")
    with pytest.raises(va.VolpickHipError, match="no CPU fallback"):
        pn(np.zeros((1, 3, 3001), np.float32))


def test_product_never_imports_oracle():
    for f in (ROOT / "volpick_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("cls,name", [(va.PhaseNet, "volpick"), (va.EQTransformer, "volpick_95train")])
def test_save_round_trips_the_model_zoo_format(tmp_path, cls, name):
    """``save`` writes the ``<name>.json.v1`` / ``<name>.pt.v1`` pair of
    Final_models/ (model_training/tune.ipynb cell 7 ``export_model``); a strict torch load of the
    .pt into the oracle module and ``load`` of the pair reproduce the source weights exactly."""
    import json

    import torch

    from oracle import models as OM

    m = cls.from_pretrained(name)
    m.save(tmp_path / "zoo" / name, version_str="1")
    meta = json.loads((tmp_path / "zoo" / f"{name}.json.v1").read_text())
    assert set(meta) == {"docstring", "model_args", "seisbench_requirement", "version", "default_args"}
    assert meta["model_args"]["norm"] == "peak" and meta["model_args"]["component_order"] == "ZNE"
    assert meta["default_args"] == m.default_args and meta["version"] == "1"
    sd = torch.load(tmp_path / "zoo" / f"{name}.pt.v1", map_location="cpu", weights_only=True)
    net = {"phasenet": OM.PhaseNet, "eqtransformer": OM.EQTransformer}[m._weights_subdir](**meta["model_args"])
    net.load_state_dict(sd, strict=True)  # every key torch expects, including num_batches_tracked
    back = cls.load(tmp_path / "zoo" / name, version_str="1")
    assert np.array_equal(back._weights, m._weights)
    assert back.default_args == m.default_args and back.labels == m.labels
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), back.state_dict().items()):
        assert k1 == k2 and v1.shape == v2.shape and v1.dtype == v2.dtype and np.array_equal(v1, v2)


def test_training_host_pieces_without_gpu():
    """vector_cross_entropy (volpick/model/models.py:34-51), the label shape PhaseNetLit asks for and the
    500-step warm-up (models.py:168-175); the step itself needs the GPU and fails loudly without one."""
    import torch

    from volpick_amd.train import PhaseNetLit, PhaseNetTrainer, gaussian_labels, vector_cross_entropy

    rng = np.random.default_rng(0)
    logits = torch.from_numpy(rng.normal(size=(4, 3, 50)))
    p = torch.softmax(logits, 1)
    y = torch.from_numpy(rng.random((4, 3, 50)))
    want = -(y * torch.log(p + 1e-5)).mean(-1).sum(-1).mean()
    assert abs(vector_cross_entropy(p.numpy(), y.numpy()) - float(want)) < 1e-12
    lab = gaussian_labels([100.0, 2000.0], [400.0, np.nan], n_samples=3001, sigma=20)
    assert lab.shape == (2, 3, 3001) and lab.dtype == np.float32
    assert lab[0, 0, 100] == 1.0 and lab[0, 1, 400] == 1.0 and lab[1, 1].max() == 0.0
    assert abs(lab[0, 0, 120] - np.exp(-0.5)) < 1e-6 and np.allclose(lab.sum(1), 1.0, atol=1e-6)
    lit = PhaseNetLit(lr=5e-4, model=va.PhaseNet.from_pretrained("volpick"))
    lrs = [lit.learning_rate(k) for k in range(0, 600)]
    # step 0 at lr; the hook after 0-based step k (trainer.global_step == k there) sets lr * (k + 1) / 500 for step k + 1
    assert lrs[0] == 5e-4 and lrs[1] == pytest.approx(5e-4 * 1 / 500) and lrs[499] == pytest.approx(5e-4 * 499 / 500)
    assert lrs[500] == 5e-4 and lrs[599] == 5e-4
    assert all(b >= a for a, b in zip(lrs[1:500], lrs[2:501]))
    with pytest.raises(_lib.VolpickHipError):
        PhaseNetTrainer(va.PhaseNet.from_pretrained("volpick"), max_batch=4)


def test_resampling_rule_and_numerics():
    """volpick_amd/resample.py restates SeisBench's rule (integer ratio: zero-phase low-pass + decimation, otherwise
    ObsPy's Fourier resampling with a Hann window).  Unpinned like the rest of the stream handling; the checks here are
    its defining properties on band-limited signals."""
    from volpick_amd.resample import lowpass_zerophase, resample_array, resample_fourier

    rng = np.random.default_rng(5)
    t200 = np.arange(24_000) / 200.0
    sig = lambda t: np.sin(2 * np.pi * 3.0 * t + 0.3) + 0.5 * np.sin(2 * np.pi * 11.0 * t)  # well below 50 Hz
    # 200 -> 100 Hz: low-pass at 50 Hz (passes 3 / 11 Hz untouched, zero phase), every second sample
    y = resample_array(sig(t200), 200.0, 100.0)
    assert len(y) == 12_000
    assert np.abs(y[500:-500] - sig(np.arange(12_000) / 100.0)[500:-500]).max() < 2e-3
    # the filter is zero-phase: a symmetric pulse stays symmetric
    pulse = np.exp(-0.5 * ((np.arange(4001) - 2000) / 40.0) ** 2)
    lp = lowpass_zerophase(pulse, 25.0, 200.0)
    assert np.abs(lp - lp[::-1]).max() < 1e-9
    # 50 -> 100 Hz and 40 -> 100 Hz go through ObsPy's Fourier method: length int(npts / factor); its default Hann window
    # centred on DC scales a component at f by 0.5 (1 + cos(2 pi f / rate_in)) — the published behaviour, kept as is
    hann = lambda f, rate: 0.5 * (1 + np.cos(2 * np.pi * f / rate))
    t50, t100 = np.arange(3000) / 50.0, np.arange(6000) / 100.0
    up = resample_array(sig(t50), 50.0, 100.0)
    want = hann(3.0, 50.0) * np.sin(2 * np.pi * 3.0 * t100 + 0.3) + hann(11.0, 50.0) * 0.5 * np.sin(2 * np.pi * 11.0 * t100)
    assert len(up) == 6000 and np.abs(up[300:-300] - want[300:-300]).max() < 2e-3
    t40 = np.arange(4000) / 40.0
    up40 = resample_array(np.sin(2 * np.pi * 1.5 * t40), 40.0, 100.0)
    want40 = hann(1.5, 40.0) * np.sin(2 * np.pi * 1.5 * np.arange(10_000) / 100.0)
    assert len(up40) == 10_000 and np.abs(up40[500:-500] - want40[500:-500]).max() < 2e-3
    # identity rate: untouched; DC gain of the Fourier method is exactly the length ratio rule (mean preserved)
    x = rng.standard_normal(1000)
    assert resample_array(x, 100.0, 100.0) is not None and np.array_equal(resample_array(x, 100.0, 100.0), x)
    c = resample_fourier(np.full(1000, 2.5), 80.0, 100.0)
    assert len(c) == 1250 and np.abs(c - 2.5).max() < 1e-9
    with pytest.raises(NotImplementedError):
        resample_array(np.ma.masked_array(x, mask=x > 2), 50.0, 100.0)


def test_records_from_columns_sort_like_the_records_and_are_built_on_first_touch():
    """classify() keeps triggers as columns and defers the Pick / Detection objects: the lists must equal what sorting the
    records themselves gives (Pick order: start_time, trace_id, phase; stable), times rounded as UTCDateTime.__add__ rounds."""
    from volpick_amd.models import _records_from_columns
    from volpick_amd.picks import Detection, Pick

    rng = np.random.default_rng(5)
    labels = ["Detection", "P", "S"]
    tids = ["XX.B.", "XX.A.", "XX.B."]  # two blocks of one station + another station
    t0s = [va.UTCDateTime("2021-01-01T00:00:00")._us, va.UTCDateTime("2021-01-01T00:00:00.005")._us, va.UTCDateTime("2021-01-01T01:00:00")._us]
    cols, want_p, want_d = [], [], []
    for g in range(3):
        m = 40
        spec = rng.integers(0, 3, m).astype(np.int32)
        on = rng.integers(0, 50, m).astype(np.int64)  # many ties in the start time
        off, peak = on + rng.integers(1, 30, m), on + rng.integers(0, 5, m)
        val = rng.random(m).astype(np.float32)
        cols.append((g, spec, on, off, peak, val))
        t0 = va.UTCDateTime._from_us(t0s[g])
        for i in range(m):
            if labels[spec[i]] == "Detection":
                want_d.append(Detection(tids[g], t0 + int(on[i]) / 100.0, t0 + int(off[i]) / 100.0, float(val[i])))
            else:
                want_p.append(Pick(tids[g], t0 + int(on[i]) / 100.0, t0 + int(off[i]) / 100.0, t0 + int(peak[i]) / 100.0, float(val[i]),
                                   labels[spec[i]]))
    picks, dets = _records_from_columns(cols, tids, t0s, labels, 100.0)
    assert picks._lazy is not None and len(picks) == len(want_p) and picks._lazy is not None  # len() builds nothing
    assert list(picks) == sorted(want_p) and picks._lazy is None
    assert list(dets) == sorted(want_d)
    assert isinstance(picks.select(phase="P"), va.PickList) and {p.phase for p in picks.select(phase="P")} == {"P"}
    both = va.PickList()
    both += picks
    both.append(want_p[0])
    assert len(both) == len(want_p) + 1 and str(both).startswith(f"PickList with {len(want_p) + 1} entries")
    import copy
    import pickle

    again, _ = _records_from_columns(cols, tids, t0s, labels, 100.0)
    assert again._lazy is not None
    clone = pickle.loads(pickle.dumps(again))  # a deferred list travels as its records (multiprocessing, dist.gather_object)
    assert isinstance(clone, va.PickList) and list(clone) == sorted(want_p) and list(copy.deepcopy(again)) == sorted(want_p)
    assert list(again.copy()) == sorted(want_p) and list(again[:3]) == sorted(want_p)[:3]
    third, _ = _records_from_columns(cols, tids, t0s, labels, 100.0)
    shallow = copy.copy(third)  # UserList.__copy__ would look for an instance attribute `data` (KeyError in round 4)
    assert isinstance(shallow, va.PickList) and list(shallow) == sorted(want_p) and shallow.data is not third.data
    assert list(copy.copy(va.PickList(sorted(want_p)[:4]))) == sorted(want_p)[:4] and len(copy.copy(va.PickList())) == 0
    empty_p, empty_d = _records_from_columns([], [], [], labels, 100.0)
    assert len(empty_p) == 0 and list(empty_d) == []
