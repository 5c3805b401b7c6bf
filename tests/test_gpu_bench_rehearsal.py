"""bench.py's N > 1 plumbing on a ONE-GPU box: `python bench.py --gpus 2 --rehearse-gloo` called DIRECTLY, the way the driver
calls it (no RANK in the environment): the script launches its two ranks itself (torch.distributed.run as a child process),
both on cuda:0, process group over gloo, weights through the host broadcast (no RCCL).  What is checked is the contract of
the line -- ONE JSON object on stdout, n_gpus = 2, whole-job value over the max-over-ranks span, `ranks` with what every
rank saw -- not the numbers; and that nothing of the rank group is left behind."""
import json
import os
import signal
import subprocess
import sys
import time
from pathlib import Path

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = Path(__file__).resolve().parents[1]


def _run(extra, timeout=420, n=2):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    detail = ROOT / "gpurun_out" / "rehearsal_detail.json"  # the full object; stdout carries its compact form
    detail.parent.mkdir(exist_ok=True)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), "--rehearse-gloo", "--detail-file", str(detail)] + extra
    # a session of its own: launcher, elastic agent and ranks share ONE process group that a timeout can kill whole
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        pytest.fail(f"bench.py --gpus {n} did not finish within {timeout} s; its process group was killed")
    finally:
        _reap_group(p.pid)
    assert p.returncode == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), "stdout carries ONE JSON line from rank 0 and nothing else"
    assert len(lines[0]) < 8000, "the driver keeps the last 8 KB of stdout: the line must fit"
    d = json.loads(lines[0])
    full = json.loads(detail.read_text())
    assert full["n_gpus"] == d["n_gpus"] and abs(full["value"] - d["value"]) <= 1e-5 * full["value"]
    d["_full"] = full
    return d


def _reap_group(pgid):
    """No process of the launcher's group may be left (a rank that survives holds the GPU for the steps after this one)."""
    for _ in range(50):
        try:
            os.killpg(pgid, 0)
        except ProcessLookupError:
            return
        time.sleep(0.1)
    os.killpg(pgid, signal.SIGKILL)
    pytest.fail("bench.py left processes of its rank group behind")


def test_weak_line_of_two_ranks():
    d = _run(["--steps", "5", "--warmup", "2", "--model", "phasenet", "--no-cpu-baseline", "--no-api", "--sustain-seconds", "0"])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] == pytest.approx(2 * 256 / (d["ms_per_step"] * 1e-3), rel=1e-4)  # whole job over the slowest rank's span (six digits in the line)
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["windows_per_step"] == 256 for r in d["ranks"])
    assert all(r["device"] == 0 and "weight_broadcast_path" in r for r in d["ranks"]) and all("librccl" in r for r in d["_full"]["ranks"])
    assert max(r["ms_per_step_own_median"] for r in d["ranks"]) <= d["ms_per_step"] * 1.25
    assert d["roofline"]["frac"] <= 1.0


def test_strong_line_of_two_ranks():
    d = _run(["--strong", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    segs = [r["segment"] for r in d["ranks"]]
    keeps = [r["keeps"] for r in d["ranks"]]
    assert keeps[0][0] == 0 and keeps[0][1] == keeps[1][0] and keeps[1][1] == 8_640_000  # the kept ranges tile the day
    assert segs[0][0] == 0 and segs[1][1] == 8_640_000 and segs[0][1] > keeps[0][1] and segs[1][0] < keeps[1][0]  # halos


@pytest.mark.slow
def test_four_ranks_weak_and_strong_on_one_gpu():
    """The widest rehearsal a one-GPU box allows (its process guard admits six processes with the GPU open: this test runner,
    the ranks, and one more of the launch -- six ranks were killed by it; the driver's real run is `--gpus 8` on eight GPUs):
    `bench.py --gpus 4 --rehearse-gloo`, weak and `--strong`.  Four ranks report, the line says n_gpus = 4, and the pick list
    stitched from four window ranges is the pick list of ONE rank over the whole day."""
    import subprocess

    d = _run(["--steps", "3", "--warmup", "1", "--model", "eqtransformer", "--no-cpu-baseline", "--no-api", "--sustain-seconds", "0",
              "--settle-seconds", "0", "--repeats", "3"], n=4, timeout=900)
    assert d["n_gpus"] == 4 and [r["rank"] for r in d["ranks"]] == list(range(4)) and all(r["windows_per_step"] == 256 for r in d["ranks"])
    assert d["value"] == pytest.approx(4 * 256 / (d["ms_per_step"] * 1e-3), rel=1e-4)
    s6 = _run(["--strong", "--steps", "1", "--warmup", "1", "--repeats", "1"], n=4, timeout=900)
    assert s6["n_gpus"] == 4 and s6["scaling"] == "strong" and len(s6["ranks"]) == 4
    keeps = [r["keeps"] for r in s6["ranks"]]
    assert keeps[0][0] == 0 and keeps[-1][1] == 8_640_000 and all(a[1] == b[0] for a, b in zip(keeps, keeps[1:]))
    one = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--strong", "--steps", "1", "--warmup", "1", "--repeats", "1",
                          "--detail-file", str(ROOT / "gpurun_out" / "rehearsal_detail_n1.json")], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    s1 = json.loads(one.stdout.splitlines()[-1])
    assert s1["n_gpus"] == 1 and s1["picks"] == s6["picks"] and s1["detections"] == s6["detections"] and s1["picks_digest"] == s6["picks_digest"]
    # what a call costs a rank beyond the annotate of its own segment and the wait for the slowest rank (trigger scan, two small
    # collectives of integer columns, stitching): it does not shrink with N, so it bounds the strong scaling (<= 1 ms)
    # Here four processes share ONE GPU and a gloo group on the host: a rank's trigger scan queues behind the other ranks' kernels,
    # (measured 0.7-1.8 ms, of which 0.3-1.2 ms is that queueing and the host's scheduling of four gloo ranks; a passing run
    # read 1.8 ms and one full-suite run failed in this test), so the rehearsal's bar is 4 ms -- it catches a return of the pickled exchange (8 ms); the
    # one-rank run, alone on the GPU, must meet the 1 ms the eight-GPU run is planned with.
    assert all(r["fixed_ms"] <= 4.0 for r in s6["ranks"]), [(r["rank"], r["gpu_ms"], r["fixed_ms"]) for r in s6["ranks"]]
    assert s1["call_split_ms"]["fixed_ms"] <= 1.0, s1["call_split_ms"]
