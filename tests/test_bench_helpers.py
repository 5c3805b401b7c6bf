"""CPU: the arithmetic behind bench.py's `roofline` object (no GPU, no library call)."""
import importlib.util
from pathlib import Path

spec = importlib.util.spec_from_file_location("bench", Path(__file__).resolve().parents[1] / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_pipe_time_adds_the_pipes_and_never_exceeds_a_real_duration():
    # eqt_tail3_kernel: 611.9424 MFLOP per window as bf16 MFMAs -> 256 windows at 2500 TFLOP/s
    t = bench.pipe_time_s({"mfma_f32": 0.0, "mfma_bf16": 611_942_400.0, "valu": 0.0}, 256)
    assert abs(t - 256 * 611_942_400.0 / 2.5e15) < 1e-12 and 62e-6 < t < 63e-6
    # pn_window_kernel: the three pipes share the SIMD's issue, their times add
    mix = {"mfma_f32": 20_762_624.0, "mfma_bf16": 70_385_664.0, "valu": 9_219_072.0}
    t = bench.pipe_time_s(mix, 256)
    want = 256 * ((20_762_624.0 + 9_219_072.0) / 157.3e12 + 70_385_664.0 / 2.5e15)
    assert abs(t - want) < 1e-12
    assert t / 97e-6 < 1.0  # frac of a measured 97 us launch: below one by construction of the peaks
    assert bench.PEAK_FP32_TFLOPS == 157.3 and bench.PEAK_BF16_TFLOPS == 2500.0 and bench.PEAK_HBM_GBS == 8000.0


# ---- `python bench.py --gpus N`: the launcher (no GPU needed: what is checked is process plumbing) ----------------
import os  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402

BENCH = str(Path(__file__).resolve().parents[1] / "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def test_flag_and_world_size_must_agree():
    """Under an external torch.distributed.run the flag is checked against WORLD_SIZE before torch is imported."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_clean_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and out.stdout == "" and "--gpus 8 but WORLD_SIZE=2" in out.stderr


def test_launcher_starts_the_ranks_itself_and_relays_their_exit_code(monkeypatch):
    """`--gpus 2` with no RANK in the environment: bench.py starts torch.distributed.run on itself.  Here there is no GPU,
    so both ranks fail; the launcher must hand back a non-zero code, print no JSON line, and leave no process behind."""
    import signal
    import time

    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--rehearse-gloo"], env=_clean_env(),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        raise
    import torch

    if torch.cuda.is_available():  # on a GPU box the same call is the rehearsal itself (tests/test_gpu_bench_rehearsal.py)
        assert p.returncode == 0
        return
    assert p.returncode not in (0, None) and out.strip() == ""
    assert "torch.distributed" in err or "elastic" in err or "ChildFailedError" in err  # the ranks were really started
    for _ in range(50):
        try:
            os.killpg(p.pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.killpg(p.pid, signal.SIGKILL)
        raise AssertionError("the launcher left processes of its rank group behind")


def test_launch_ranks_relays_the_one_json_line(tmp_path, monkeypatch):
    """The relay itself, with a stand-in for torch.distributed.run: everything that is not the JSON line goes to stderr."""
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import json, sys\nprint('noise from a rank')\nprint(json.dumps({'n_gpus': 2, 'argv': ' '.join(sys.argv[1:])}))\n")
    code = ("import sys, importlib.util; sys.argv=['bench.py'];"
            f"spec = importlib.util.spec_from_file_location('bench', {BENCH!r}); m = importlib.util.module_from_spec(spec);"
            "spec.loader.exec_module(m); sys.exit(m.launch_ranks(2, ['--gpus', '2', '--steps', '3']))")
    out = subprocess.run([sys.executable, "-c", code], env=_clean_env(PYTHONPATH=str(tmp_path)), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 1
    import json

    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "--nproc-per-node 2" in d["argv"] and "--master-addr 127.0.0.1" in d["argv"] and d["argv"].endswith("--gpus 2 --steps 3")
    assert "noise from a rank" in out.stderr


# ---- the compact line: what the driver parses (round 4's 20 KB line crossed its 16 KiB limit and was recorded as null) ----
def _required(d, path):
    cur = d
    for k in path.split("."):
        assert isinstance(cur, dict) and k in cur, f"compact line lacks {path}"
        cur = cur[k]
    return cur


def test_compact_line_from_round_4_detail_is_small_strict_and_complete():
    import json

    full = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "r04_b_bench.json").read_text())
    assert len(json.dumps(full)) > 16384  # the line that was cut
    line = bench.compact_line(full, "bench_detail.json")
    assert "\n" not in line and len(line) < 4096 < bench.COMPACT_LIMIT
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(AssertionError(f"non-strict JSON constant {c}")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config.workload", "config.batch", "timing.repeats", "timing.settle.steps", "sustained.value",
              "sustained.ms_per_step", "sustained.shader_clock_ghz", "roofline.bound", "roofline.kernel", "roofline.achieved",
              "roofline.peak", "roofline.unit", "roofline.frac", "roofline.kernel_ms", "roofline.step_bound", "roofline.traffic",
              "roofline.algorithmic.flop_over_fp32_peak", "cpu_baseline.value", "cpu_baseline.unit", "cpu_baseline.cores",
              "cpu_baseline.kind", "cpu_baseline.cpu_model", "cpu_baseline.sample", "pick_parity.picks_hip", "pick_parity.picks_oracle",
              "pick_parity.max_abs_dt_samples", "weight_broadcast_path", "eqtransformer.value", "eqtransformer.ms_per_step",
              "eqtransformer.config.workload", "eqtransformer.roofline.frac", "eqtransformer.roofline.kernel_ms",
              "eqtransformer.cpu_baseline.value", "eqtransformer.pick_parity.max_abs_dt_samples", "eqtransformer.sustained.value",
              "train.value", "train.ms_per_step", "train.batch", "train.dtype", "train.launches_per_step", "train.roofline.frac",
              "train.vs_torch_rocm", "mseed.value", "mseed.kernel_ms", "mseed.roofline.frac", "mseed.read_wall_ms", "detail"):
        _required(d, k)
    assert d["value"] == float(f"{full['value']:.6g}") and d["n_gpus"] == 1 and d["dtype"] == "f32"
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert abs(d["roofline"]["achieved"] / d["roofline"]["peak"] - d["roofline"]["frac"]) < 1e-4


def test_compact_line_never_carries_nan_and_shrinks_many_ranks():
    import json

    ranks = [{"rank": r, "device": r, "ms_per_step_own_median": 0.09, "weight_broadcast_path": "rccl", "rccl_comm_ranks": 8,
              "librccl": {"bound_by_vp": "/x" * 200, "mapped": ["/y" * 200], "one_copy": True}, "windows_per_step": 256} for r in range(8)]
    res = {"metric": "waveform-windows/sec", "value": float("nan"), "unit": "windows/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": float("inf"), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w", "batch": 256, "parallelism": "p"}, "ranks": ranks, "weight_broadcast_path": "rccl"}
    line = bench.compact_line(res)
    d = json.loads(line)
    assert d["value"] is None and d["ms_per_step"] is None and "NaN" not in line and "Infinity" not in line
    assert len(d["ranks"]) == 8 and d["rccl_comm_ranks"] == 8 and "librccl" not in d["ranks"][0] and len(line) < bench.COMPACT_LIMIT


def test_emit_prints_exactly_one_stdout_line_and_writes_the_detail(tmp_path, capsys):
    import json

    full = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "r04_b_bench.json").read_text())
    bench.emit(full, str(tmp_path / "d" / "bench_detail.json"))
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and json.loads(lines[0])["metric"] == "waveform-windows/sec"
    assert cap.err.startswith("bench.py detail: {")
    det = json.loads((tmp_path / "d" / "bench_detail.json").read_text())
    assert det["forward"]["kernels"] and det["eqtransformer"]["forward"]["kernels"]


def test_compact_line_sheds_optional_objects_rather_than_fail(monkeypatch):
    """If an object ever outgrows the limit again, the line drops optional parts (they stay in the detail file) and keeps the
    contract's keys, `roofline` and `cpu_baseline`."""
    import json

    full = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "r04_b_bench.json").read_text())
    monkeypatch.setattr(bench, "COMPACT_LIMIT", 2000)
    d = json.loads(bench.compact_line(full, "bench_detail.json"))
    assert len(json.dumps(d, separators=(",", ":"))) < 2000 and d["line_shrunk"] >= 1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert "api" not in d and "mseed" not in d


def test_compact_line_from_round_6_detail_stays_under_four_kilobytes():
    """The line with everything round 6 added (api.many_stations for both models, roofline.useful_frac, the long dtype label, the
    train / mseed objects from their child processes): still < 4 KB, strict JSON, the new keys present."""
    import json

    full = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "r06_k_bench_detail.json").read_text())
    line = bench.compact_line(full, "bench_detail.json")
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(AssertionError(f"non-strict JSON constant {c}")))
    for k in ("roofline.useful_frac", "roofline.frac", "roofline.kernel_ms", "api.many_stations.value", "api.many_stations.picks_equal",
              "eqtransformer.api.many_stations.value", "eqtransformer.roofline.useful_frac", "train.ms_per_step", "mseed.value",
              "cpu_baseline.cpu_model", "eqtransformer.cpu_baseline.value", "config.parallelism"):
        _required(d, k)
    assert d["dtype"].startswith("f32 (exact 3xbf16") and d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.0001
    assert "RCCL" not in d["config"]["parallelism"]  # one rank: nothing was broadcast
