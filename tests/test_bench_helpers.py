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
