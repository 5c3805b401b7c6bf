"""bench.py's N > 1 plumbing on a ONE-GPU box: two ranks under torch.distributed.run over gloo, both on cuda:0
(`--rehearse-gloo`: no RCCL, weights through the host broadcast).  What is checked is the contract of the line -- one JSON
object from rank 0, whole-job value over the max-over-ranks span, `ranks` with what every rank saw -- not the numbers."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-gloo"] + extra
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, the other ranks nothing"
    return json.loads(lines[0])


def test_weak_line_of_two_ranks():
    d = _run(["--steps", "5", "--warmup", "2", "--model", "phasenet", "--no-cpu-baseline", "--no-api", "--sustain-seconds", "0"], 29541)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] == pytest.approx(2 * 256 / (d["ms_per_step"] * 1e-3), rel=1e-6)  # whole job over the slowest rank's span
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all(r["windows_per_step"] == 256 for r in d["ranks"])
    assert max(r["ms_per_step_own_median"] for r in d["ranks"]) <= d["ms_per_step"] * 1.25
    assert d["roofline"]["frac"] <= 1.0


def test_strong_line_of_two_ranks():
    d = _run(["--strong", "--steps", "2", "--warmup", "1"], 29542)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    segs = [r["segment"] for r in d["ranks"]]
    keeps = [r["keeps"] for r in d["ranks"]]
    assert keeps[0][0] == 0 and keeps[0][1] == keeps[1][0] and keeps[1][1] == 8_640_000  # the kept ranges tile the day
    assert segs[0][0] == 0 and segs[1][1] == 8_640_000 and segs[0][1] > keeps[0][1] and segs[1][0] < keeps[1][0]  # halos
