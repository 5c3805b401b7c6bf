#!/usr/bin/env python3
"""Headline benchmark: waveform-windows/s of the volpick picking path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for any N.  Run directly with N > 1 (no RANK in the environment), this script is the LAUNCHER: without
importing torch or touching HIP it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on itself as a
child process (127.0.0.1 rendezvous on a free port), relays rank 0's one JSON line and exits with the child's code.
Under an external torch.distributed.run, WORLD_SIZE must equal --gpus (else exit 2: a line that says n_gpus = 8 is
never produced by fewer ranks).

A "step" is one pass of the whole hot path over one batch of 256
windows cut from a device-resident synthetic 3-component stream: window gather +
annotate_batch_pre, model forward, blinding + overlap stacking, trigger/peak scan of the
phase traces.  The default run times BOTH single-GPU workloads of BASELINE.json:

  * configs[1]  PhaseNet volpick, batch 256, 3x3001, fp32                 -> the top-level line (`value`)
  * configs[2]  EQTransformer volpick, batch 256, 3x6000, overlap 5500,
                blinding (500, 500) -- the shape BASELINE's metric names   -> the "eqtransformer" object

(`--model phasenet|eqtransformer` times one of them alone.)  Inputs are resident in HBM before the
timed region.  The K steps are timed R times (`--repeats`, each bracketed by a barrier + device
synchronisation on both sides, max over ranks); `value` is the median repeat and min / max are
reported beside it, so that a 3 ms region does not decide the headline.
Multi-GPU: every rank owns whole station streams (weak scaling, no data-path collective); the
weights are broadcast once from rank 0 over RCCL (vp_bcast_weights) before the timed region.
`--strong` instead times BASELINE configs[3]: ONE 24 h stream (8,640,000 samples, 17,269
EQTransformer windows) split over the N ranks by window range, picks stitched on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time
from pathlib import Path

# hardware queues for the device contexts' streams (volpick_amd/__init__.py sets the same default; here it is set before
# anything can start the HIP runtime)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (same guide); an exact three-piece product costs six of them


def pipe_time_s(issued, batch):
    """Seconds the issued work of one launch over `batch` windows needs with every pipe at its dense peak.  fp32 MFMA,
    bf16 MFMA and packed fp32 FMA issue from the same SIMDs and do not overlap (tools/micro/micro_valu.hip: MFMA waves and
    packed-FMA waves on one SIMD each run at half rate), so the three times add."""
    return batch * (issued["mfma_f32"] / (PEAK_FP32_TFLOPS * 1e12) + issued["mfma_bf16"] / (PEAK_BF16_TFLOPS * 1e12) +
                    issued["valu"] / (PEAK_FP32_TFLOPS * 1e12))

# the arithmetic type of the path, said in full: fp32 results from exact bf16-piece products
DTYPE_LABEL = "f32 (exact 3xbf16 split products, fp32 accumulate)"
COMPACT_LIMIT = 8000  # bytes: the driver keeps the last 8 KB of stdout and parses the last line (round 4's 20 KB line was cut)


def _r(v, sig=6):
    """Round floats to `sig` significant digits (bytes matter in the compact line); NaN / inf -> None (strict JSON)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _compact_model(m, brief=False):
    """The short form of one model's result (the top-level PhaseNet line or, `brief`, the `eqtransformer` object: what the two
    share -- the host's CPU model, the quota -- is said once)."""
    out = _pick(m, "value", "unit", "ms_per_step")
    if isinstance(m.get("config"), dict):
        out["config"] = _pick(m["config"], "workload", "batch", "parallelism")
    t = m.get("timing") or {}
    if t:
        out["timing"] = _pick(t, "repeats", "windows_per_s_min", "windows_per_s_max")
        if isinstance(t.get("settle"), dict):
            out["timing"]["settle"] = _pick(t["settle"], "steps", "seconds")
    if m.get("sustained"):
        out["sustained"] = _pick(m["sustained"], "value", "ms_per_step", "seconds", "shader_clock_ghz")
    r = m.get("roofline")
    if r:
        rr = _pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "useful_frac", "kernel_ms", "pipe_time_ms", "traffic")
        if isinstance(rr.get("kernel"), str):
            rr["kernel"] = rr["kernel"][:72]
        if r.get("step_bound"):
            rr["step_bound"] = _pick(r["step_bound"], "ms", "frac")
        if r.get("algorithmic"):
            rr["algorithmic"] = _pick(r["algorithmic"], "tflops", "flop_over_fp32_peak")
        out["roofline"] = rr
    c = m.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = _pick(c, "value", "unit", "cores", "kind") if brief else \
            _pick(c, "value", "unit", "cores", "kind", "cpu_model", "physical_cores", "cgroup_cpu_quota")
        if isinstance(c.get("sample"), str):
            out["cpu_baseline"]["sample"] = c["sample"][:48 if brief else 110]
    if m.get("pick_parity"):
        out["pick_parity"] = _pick(m["pick_parity"], "picks_hip", "picks_oracle", "max_abs_dt_samples", "max_abs_dvalue")
    a = m.get("api")
    if a:
        out["api"] = _pick(a, "windows", "wall_ms", "wall_ms_records_built", "value", "picks")
        if isinstance(a.get("cpu_oracle_10min"), dict):
            out["api"]["picks_identical_10min"] = a["cpu_oracle_10min"].get("picks_identical")
        ms_ = a.get("many_stations")
        if isinstance(ms_, dict) and "pageable" in ms_:
            out["api"]["many_stations"] = {"stations": ms_["stations"], "value": ms_["pageable"]["value"], "per_station_ms": ms_["pageable"]["per_station_ms"],
                                           "value_pinned_rows": ms_["pinned"]["value"],
                                           "picks_equal": bool(ms_["pageable"]["picks_equal_one_station_call"] and ms_["pinned"]["picks_equal_one_station_call"])}
        elif isinstance(ms_, dict):
            out["api"]["many_stations"] = _pick(ms_, "error")
    if m.get("ranks"):
        out["ranks"] = [_pick(x, "rank", "device", "ms_per_step_own_median", "gpu_ms", "fixed_ms", "weight_broadcast_path", "rccl_comm_ranks", "windows_per_step", "segment", "keeps")
                        for x in m["ranks"]]
    return out


def compact_line(result, detail_path=None):
    """The ONE line the driver parses: the contract's keys plus the short form of every object of the full result
    (which goes to `detail_path` and to stderr).  Strict JSON (no NaN), < COMPACT_LIMIT bytes -- asserted."""
    out = _pick(result, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data")
    body = _compact_model(result)
    for k in ("value", "unit", "ms_per_step"):
        body.pop(k, None)
    out.update(body)
    for k in ("weight_broadcast_path", "weight_broadcast_s", "picks", "detections", "picks_digest", "call_split_ms"):
        if k in result:
            out[k] = result[k]
    if result.get("ranks"):
        out["rccl_comm_ranks"] = result["ranks"][0].get("rccl_comm_ranks")
    if isinstance(result.get("eqtransformer"), dict):
        out["eqtransformer"] = _compact_model(result["eqtransformer"], brief=True)
    tr = result.get("train")
    if isinstance(tr, dict):
        out["train"] = _pick(tr, "value", "unit", "ms_per_step", "batch", "dtype", "launches_per_step", "settle_steps", "loss_after",
                             "loss_torch_same_batch", "vs_torch_rocm", "error")
        if isinstance(tr.get("roofline"), dict):
            out["train"]["roofline"] = _pick(tr["roofline"], "bound", "achieved", "peak", "unit", "frac")
    ms = result.get("mseed")
    if isinstance(ms, dict):
        out["mseed"] = _pick(ms, "value", "unit", "kernel_ms", "bit_exact_vs_fixture_samples", "file_to_picks_ms", "error")
        if isinstance(ms.get("file_to_picks"), dict):
            out["mseed"]["file_to_picks_host_stream_ms"] = ms["file_to_picks"].get("host_stream_ms")
        if "read_wall_ms_host_file_to_host_stream" in ms:
            out["mseed"]["read_wall_ms"] = ms["read_wall_ms_host_file_to_host_stream"]
        if isinstance(ms.get("roofline"), dict):
            out["mseed"]["roofline"] = _pick(ms["roofline"], "bound", "achieved", "peak", "unit", "frac")
        if isinstance(ms.get("cpu_baseline"), dict):
            out["mseed"]["cpu_baseline"] = _pick(ms["cpu_baseline"], "value", "unit", "cores", "kind")
    if detail_path:
        out["detail"] = str(detail_path)
    out = _r(out)
    dumps = lambda o: json.dumps(o, allow_nan=False, separators=(",", ":"))  # noqa: E731
    line = dumps(out)
    # Never lose the headline to the size limit: what does not fit is dropped from the LINE in this order (it stays in the
    # detail file), the contract's keys, `roofline` and `cpu_baseline` last of all.
    shrink = [lambda o: o.__setitem__("ranks", [_pick(x, "rank", "ms_per_step_own_median") for x in o["ranks"]]) if "ranks" in o else None]
    for path in (("eqtransformer", "api"), ("api",), ("mseed",), ("train",), ("eqtransformer", "timing"), ("timing",),
                 ("eqtransformer", "pick_parity"), ("pick_parity",), ("ranks",), ("eqtransformer", "sustained"), ("sustained",),
                 ("eqtransformer", "cpu_baseline"), ("eqtransformer", "roofline"), ("eqtransformer",)):
        def drop(o, path=path):
            for k in path[:-1]:
                o = o.get(k) if isinstance(o, dict) else None
            if isinstance(o, dict):
                o.pop(path[-1], None)
        shrink.append(drop)
    dropped = 0
    while len(line) >= COMPACT_LIMIT and dropped < len(shrink):
        shrink[dropped](out)
        dropped += 1
        out["line_shrunk"] = dropped
        line = dumps(out)
    return line


def emit(result, detail_file):
    """Full result -> `detail_file` (+ stderr); compact line -> the LAST line of stdout."""
    full = json.dumps(_r(result, 9), allow_nan=False)
    path = None
    if detail_file:
        try:
            Path(detail_file).parent.mkdir(parents=True, exist_ok=True)
            Path(detail_file).write_text(full + "\n")
            path = detail_file
        except OSError as e:
            sys.stderr.write(f"bench.py: cannot write {detail_file}: {e}\n")
    sys.stderr.write("bench.py detail: " + full + "\n")
    sys.stderr.flush()
    rel = None
    if path:
        try:
            rel = str(Path(path).resolve().relative_to(ROOT))
        except ValueError:
            rel = str(path)
    print(compact_line(result, rel), flush=True)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks as ONE child process
    group (torch.distributed.run on this file), pass everything but the JSON line through to stderr, print the JSON line,
    return the child's exit code.  The parent never imports torch and never touches HIP (a process that has initialised
    the GPU must not replace itself, and does not need to: it only waits)."""
    import signal
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))  # torch.distributed.run would pin it to 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, str(Path(__file__).resolve())] + list(argv)

    def die_with_parent():  # the rank group must not outlive a launcher that is killed outright (SIGKILL cannot be relayed)
        try:
            C.CDLL(None, use_errno=True).prctl(1, signal.SIGTERM)  # PR_SET_PDEATHSIG
        except (OSError, AttributeError):
            pass

    # The child stays in the launcher's process group: whoever started the launcher in a session / group of its own
    # (the driver's timeout, tests/test_gpu_bench_rehearsal.py) takes every rank with one killpg.
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, preexec_fn=die_with_parent)

    def forward(signum, _frame):  # SIGTERM / Ctrl-C of the launcher alone: the elastic agent ends its workers on SIGTERM
        try:
            child.send_signal(signum)
        except ProcessLookupError:
            pass

    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT)}
    lines = []
    try:
        for ln in child.stdout:
            if ln.startswith("{") and ln.rstrip().endswith("}"):
                lines.append(ln.rstrip())
            else:
                sys.stderr.write(ln)
        rc = child.wait()
    finally:
        if child.poll() is None:
            child.terminate()
            try:
                child.wait(timeout=30)
            except subprocess.TimeoutExpired:
                child.kill()
                child.wait()
        for sg, h in old.items():
            signal.signal(sg, h)
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py launcher: expected ONE JSON line from rank 0, got {len(lines)}\n")
        rc = 3
    for ln in lines[-1:]:
        print(ln, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=21,
                    help="the K steps are timed this many times, each between two synchronisations; value = median "
                         "(the first ~6 regions after an idle GPU run at ramping clocks: timing.windows_per_s_first)")
    ap.add_argument("--settle-seconds", type=float, default=0.4,
                    help="untimed steps run for this long between the W warm-up steps and the timed regions, so that the shader "
                         "clock has climbed to what it holds under the load (0 = none)")
    ap.add_argument("--model", default="both", choices=["both", "phasenet", "eqtransformer"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU budget of the baseline legs, all models together")
    ap.add_argument("--contexts", type=int, default=0,
                    help="device contexts (stream + workspace) steps alternate over; 0 = the model's default (PhaseNet 3, EQTransformer 4)")
    ap.add_argument("--depth", type=int, default=0, help="submits in flight per device context (0 = the model's default)")
    ap.add_argument("--strong", action="store_true", help="time configs[3]: one 24 h stream sharded over the ranks")
    ap.add_argument("--sustain-seconds", type=float, default=5.2,
                    help="length of the one long timed region per model reported as `sustained` (0 = skip)")
    ap.add_argument("--no-api", action="store_true", help="skip the `api` object (classify() on a host 24 h stream)")
    ap.add_argument("--no-train", action="store_true", help="skip the `train` object (BASELINE configs[4]: bf16 training step, B = 512)")
    ap.add_argument("--no-train-torch", action="store_true",
                    help="`train` without the stock PyTorch-ROCm step beside it (MIOpen's first-use find costs ~35 s on a fresh box)")
    ap.add_argument("--no-mseed", action="store_true", help="skip the `mseed` object (SURVEY 8f-1: Steim-2 station-day decode)")
    ap.add_argument("--widened-only", choices=["train", "mseed"], default=None,
                    help="(internal) run ONE widened-row leg in this process and print its JSON object: the default run starts a child "
                         "for each, so that the leg sees a fresh process")
    ap.add_argument("--detail-file", default=str(ROOT / "bench_detail.json"),
                    help="the full result object goes here (and to stderr); stdout's last line is the compact (< 8 KB) form of it")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 plumbing on a ONE-GPU box: process group over gloo, every rank on cuda:0, weights "
                         "through the host broadcast (no RCCL); the line it prints is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "RANK" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')} in the environment: "
                         "the launcher and the flag must agree\n")
        sys.exit(2)

    if args.widened_only:  # a child of the default run: ONE leg in a process of its own, its object as the last stdout line
        import torch

        torch.cuda.set_device(0)
        fn = (lambda: bench_train(torch_baseline=not args.no_train_torch)) if args.widened_only == "train" else bench_mseed
        t0 = time.perf_counter()
        try:
            obj = fn()
        except Exception as e:  # noqa: BLE001 -- a widened row must not cost the headline line
            obj = {"error": repr(e)[:400]}
        obj["bench_seconds"] = time.perf_counter() - t0
        print(json.dumps(_r(obj, 9), allow_nan=False), flush=True)
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" in os.environ and not args.rehearse_gloo and local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} wants cuda:{local_rank}, this node shows {torch.cuda.device_count()} GPU(s)\n")
        sys.exit(2)
    # under torch.distributed.run (RANK set) the RCCL path is exercised even for one rank
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", str(world))
        if args.rehearse_gloo:
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    env = dict(args=args, world=world, rank=rank, use_dist=use_dist, dev=torch.device("cuda", torch.cuda.current_device()))

    if args.strong:
        result = bench_strong(env)
    else:
        names = ["phasenet", "eqtransformer"] if args.model == "both" else [args.model]
        cpu_budget = 0.0 if (args.no_cpu_baseline or rank != 0 or world != 1) else args.cpu_seconds / len(names)
        parts = {n: bench_model(n, env, cpu_budget) for n in names}
        result = parts[names[0]]
        if len(names) == 2:
            eq = parts["eqtransformer"]
            result["eqtransformer"] = {k: eq[k] for k in ("value", "unit", "ms_per_step", "timing", "sustained", "config",
                                                          "roofline", "forward", "api", "ranks", "cpu_baseline",
                                                          "pick_parity") if k in eq}
    # the two widened rows that have a throughput of their own (N = 1, the default run only): each a few seconds
    # Each in a CHILD process (this one stays alive, idle, and never execs): behind the forward legs -- pickers created, run and
    # released, station-days of buffers through torch's allocator -- a trainer created in the same process ran its step at
    # 2.4-6.6 instead of 1.38 ms, depending on what had been released before it (the hardware queues its two streams land on:
    # tools/train_after_api_probe.py, LOG.md round 6); a fresh process is the state a training job starts from.
    if world == 1 and not args.strong and args.model == "both" and not args.no_cpu_baseline:
        import subprocess

        for key, skip in (("train", args.no_train), ("mseed", args.no_mseed)):
            if not skip:
                t0 = time.perf_counter()
                try:
                    cmd = [sys.executable, str(Path(__file__).resolve()), "--widened-only", key] + (["--no-train-torch"] if args.no_train_torch else [])
                    cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                    result[key] = json.loads(lines[-1]) if cp.returncode == 0 and lines else {"error": f"rc {cp.returncode}: {cp.stderr[-300:]}"}
                    result[key]["process"] = "child of the bench process (fresh HIP state)"
                except Exception as e:  # noqa: BLE001 -- a widened row must not cost the headline line
                    result[key] = {"error": repr(e)[:400]}
                result[key]["bench_seconds_with_start"] = time.perf_counter() - t0
    if rank == 0:
        emit(result, args.detail_file)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _bcast_path():
    from volpick_amd import distributed

    return distributed.LAST_BROADCAST_PATH


def _rccl_ranks():
    from volpick_amd import distributed

    return distributed.LAST_RCCL_RANKS


def _rccl_libraries():
    """The librccl the C ABI bound by dlopen, and every librccl this process has mapped (torch's bundled copy carries
    the same SONAME, so the two must be ONE object: `one_copy`)."""
    from volpick_amd import distributed

    mapped = set()
    try:
        for ln in Path("/proc/self/maps").read_text().splitlines():
            f = ln.split()[-1]
            if "librccl" in f:
                mapped.add(os.path.realpath(f))
    except OSError:
        pass
    bound = distributed.LAST_RCCL_LIBRARY
    return {"bound_by_vp": bound, "mapped": sorted(mapped),
            "one_copy": (len(mapped) == 1 and os.path.realpath(bound) in mapped) if (mapped and bound) else None}


def clock_probe_create(cls, batch, dev):
    """A twin of the model on the debug plan whose dominant kernel stamps the shader clock and the 100 MHz constant
    clock at its two ends (plan_flags[1] bit 1); created BEFORE the region it is read behind."""
    import torch

    from volpick_amd.synthetic import synthetic_windows

    m = cls.from_pretrained("volpick")
    m._plan_flags = (0, 2)
    m._max_batch = batch
    m.cuda(dev)
    x = torch.from_numpy(synthetic_windows(batch, cls.in_samples, seed=1)).to(dev)
    m._forward_raw(x, preprocess=True)
    return m, x


def clock_probe_read(probe, model_name, batch):
    """Shader clock (GHz) under the dominant kernel: a few forward passes of the twin, then the stamps of the last launch."""
    from volpick_amd import _lib

    m, x = probe
    lib = _lib.load()
    for _ in range(6):
        m._forward_raw(x, preprocess=True)
    clk = np.zeros((batch, 32), np.uint64)
    try:
        if model_name == "phasenet":  # pn_window_kernel: [0] / [28] shader clock at start / end, [16] / [17] the 100 MHz clock
            _lib.check(lib.vp_debug_core_clock(m._handle, batch, clk.ctypes.data_as(C.c_void_p)))
            cyc, wall = clk[:, 28] - clk[:, 0], clk[:, 17] - clk[:, 16]
        else:  # eqt_tail3_kernel: [24] / [25] and [30] / [31]
            _lib.check(lib.vp_debug_tail_clock(m._handle, batch, clk.ctypes.data_as(C.c_void_p)))
            cyc, wall = clk[:, 25] - clk[:, 24], clk[:, 31] - clk[:, 30]
        ok = (wall > 0) & (cyc > 0)
        ghz = float(np.median(cyc[ok].astype(np.float64) / (wall[ok].astype(np.float64) / 100e6)) / 1e9) if ok.any() else None
    finally:
        m._release()
    return ghz


def bench_api(model_name, model, batch, oracle_threads=None):
    """The drop-in call itself, as /root/reference README.md:54-66 writes it: `picker.classify(stream, batch_size=256,
    overlap=..., blinding=..., stacking="avg", ...)` on a HOST Stream of one 24 h three-component station (BASELINE
    configs[3]'s workload on one GPU): wall time of the whole call (stream grouping, upload, forward passes, stacking,
    trigger scan, pick records), its phases from one extra profiled call (serialised by device synchronisations: their
    sum exceeds the pipelined wall time), and the CPU oracle on the first 10 minutes of the same stream beside it."""
    import torch

    import volpick_amd as va
    from oracle import pipeline as OP
    from oracle.models import load_pretrained
    from volpick_amd.synthetic import synthetic_stream_array

    n = 8_640_000
    if not hasattr(bench_api, "_day"):
        bench_api._day = synthetic_stream_array(n, seed=1004, n_events=600)[0]
    data = bench_api._day
    T = model.in_samples
    kw = (dict(overlap=1500, blinding=(0, 0)) if model_name == "phasenet" else dict(overlap=5500, blinding=(500, 500)))
    kw.update(batch_size=batch, stacking="avg")
    t0 = va.UTCDateTime("2021-01-01T00:00:00")
    st = va.Stream([va.Trace(data[i], dict(network="XX", station="DAY", location="", channel=f"HH{c}", starttime=t0,
                                           sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    n_windows = int(OP.window_starts(n, T, kw["overlap"]).shape[0])
    import gc

    res = model.classify(st, **kw)  # warm-up: contexts, buffers
    walls, pauses, t_gc = [], [], [0.0]

    def gc_watch(phase, info):  # full collections of CPython's cyclic collector inside the timed calls (round 3's 65 ms call)
        if phase == "start":
            t_gc[0] = time.perf_counter()
        elif info["generation"] == 2:
            pauses.append((len(walls), (time.perf_counter() - t_gc[0]) * 1e3))

    gc.callbacks.append(gc_watch)
    try:
        for _ in range(9):
            torch.cuda.synchronize()
            t = time.perf_counter()
            res = model.classify(st, **kw)
            walls.append(time.perf_counter() - t)
        walls_built = []  # the same call with every Pick / Detection record materialised inside the timed region (what the
        for _ in range(5):  # reference's classify() hands back: plain lists of records)
            torch.cuda.synchronize()
            t = time.perf_counter()
            r2 = model.classify(st, **kw)
            n_built = len(list(r2.picks)) + len(list(getattr(r2, "detections", []) or []))
            walls_built.append(time.perf_counter() - t)
    finally:
        gc.callbacks.remove(gc_watch)
    model._timing = {}
    model.classify(st, **kw)
    phases, model._timing = model._timing, None
    wall = statistics.median(walls)
    # the oracle on a 10-minute slice of the same stream (60,000 samples), and the HIP path on the same slice
    net = load_pretrained(model_name)
    ten = data[:, :60_000]
    threads_before = torch.get_num_threads()
    if oracle_threads:  # the thread count of cpu_baseline's fastest leg, not torch's default (all logical CPUs: 3x slower)
        torch.set_num_threads(int(oracle_threads))
    OP.classify_array(net, ten[:, :30_000], overlap=kw["overlap"], blinding=kw["blinding"], batch_size=batch)  # warm-up
    t = time.perf_counter()
    want = OP.classify_array(net, ten, overlap=kw["overlap"], blinding=kw["blinding"], batch_size=batch)
    t_cpu = time.perf_counter() - t
    oracle_threads_used = torch.get_num_threads()
    torch.set_num_threads(threads_before)
    st10 = va.Stream([va.Trace(ten[i], dict(network="XX", station="DAY", location="", channel=f"HH{c}", starttime=t0,
                                            sampling_rate=100.0)) for i, c in enumerate("ZNE")])
    got = model.classify(st10, **kw)
    many = None
    try:
        many = bench_many_stations(model, model_name, data, kw, n_windows, t0, res)
    except Exception as e:  # noqa: BLE001 -- this leg must not cost the line
        many = {"error": repr(e)[:300]}
    want_p = sorted((ph, pk) for ph, on, off, pk, v in want["picks"])
    got_p = sorted((p.phase, int(round((p.peak_time - t0) * 100))) for p in got.picks)
    w10 = int(OP.window_starts(60_000, T, kw["overlap"]).shape[0])
    return {
        "call": f"{model.name}.from_pretrained('volpick').classify(stream, batch_size={batch}, overlap={kw['overlap']}, "
                f"blinding={list(kw['blinding'])}, stacking='avg') on a HOST Stream: 3 traces x {n} samples (24 h at 100 Hz)",
        "windows": n_windows,
        "wall_ms": wall * 1e3,
        "wall_ms_all": [w * 1e3 for w in walls],
        "wall_statistic": f"median of {len(walls)} calls",
        "wall_ms_records_built": statistics.median(walls_built) * 1e3,
        "records_built": n_built,
        "value_records_built": n_windows / statistics.median(walls_built),
        "full_gc_collections_in_timed_calls": [{"call": i, "pause_ms": ms} for i, ms in pauses],
        "gc_note": "round 3's 65 ms call among 24-27 ms ones was a full (generation 2) collection of CPython's cyclic collector "
                   "(30-37 ms in a process that has torch imported), driven by the ~10 k record objects each call built; classify() "
                   "now keeps triggers as columns and builds Pick / Detection objects on first access (tools/api_outlier.py)",
        "value": n_windows / wall,
        "unit": "windows/s",
        "picks": len(res.picks),
        "phases_ms_serialised": {k: v for k, v in phases.items() if k.endswith("_ms")},
        "phases_note": "one extra call with the phases separated by device synchronisations: host_assembly (stream -> "
                       "rows), h2d (pageable host rows -> HBM), gpu (window cut .. stacking over all device contexts), "
                       "pick_scan_d2h (trigger scan + result copy), emit_records (Pick objects); the timed calls overlap "
                       "h2d with gpu segment by segment",
        "cpu_oracle_10min": {"windows": w10, "wall_ms": t_cpu * 1e3, "windows_per_s": w10 / t_cpu,
                             "picks_oracle": len(want_p), "picks_hip_same_slice": len(got_p), "picks_identical": want_p == got_p,
                             "threads": oracle_threads_used},
        "many_stations": many,
    }


def bench_many_stations(model, model_name, data, kw, n_windows, t0, one_station_result):
    """The call the reference's users make (README.md:54-66): ONE classify() on a host Stream of MANY stations -- 32 station-days
    for PhaseNet, 8 for EQTransformer -- host rows in, pick list out.  Every station-day is spread over the device contexts
    segment by segment (segment r + 1 travels while segment r computes).  Two host layouts: ordinary pageable numpy rows (what
    obspy.read gives), and rows in page-locked memory (volpick_amd.pinned_array: DMA from the caller's pages).  The stations
    share one day of samples (host memory: 104 MB, not 3.3 GB); every station must come back with the one-station call's pick
    list.  `host_gb_per_s` is what bounds the PhaseNet call: 104 MB per station-day over the host link."""
    import torch

    import volpick_amd as va

    n_st = 32 if model_name == "phasenet" else 8
    meta = lambda i, c: dict(network="XX", station=f"S{i:03d}", location="", channel=f"HH{c}", starttime=t0, sampling_rate=100.0)  # noqa: E731
    out = {"stations": n_st, "windows": n_st * n_windows,
           "call": f"one classify() on a host Stream of {n_st} station-days (3 x 8,640,000 samples each)"}
    want = sorted((p.phase, p.peak_time._us, float(np.float32(p.peak_value))) for p in one_station_result.picks)
    pinned = [va.pinned_array(data.shape[1], np.float32) for _ in range(3)]
    for row, src in zip(pinned, data):
        row[:] = src
    for label, rows in (("pageable", list(data)), ("pinned", pinned)):
        st = va.Stream([va.Trace(rows[i], meta(k, c)) for k in range(n_st) for i, c in enumerate("ZNE")])
        model.classify(st, **kw)  # warm-up
        walls = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            res = model.classify(st, **kw)
            walls.append(time.perf_counter() - t)
        wall = statistics.median(walls)
        per = {}
        for p in res.picks:
            per.setdefault(p.trace_id, []).append((p.phase, p.peak_time._us, float(np.float32(p.peak_value))))
        same = len(per) == n_st and all(sorted(v) == want for v in per.values())
        out[label] = {"wall_ms": wall * 1e3, "per_station_ms": wall * 1e3 / n_st, "value": n_st * n_windows / wall, "unit": "windows/s",
                      "host_gb_per_s": n_st * data.nbytes / wall / 1e9, "picks": len(res.picks), "picks_equal_one_station_call": same}
        del st, res
    out["value"] = out["pageable"]["value"]
    return out


def bench_train(batch=512, steps=30, warmup=5, torch_steps=6, torch_baseline=True):
    """BASELINE configs[4]: one PhaseNet training step (forward in training mode, vector cross-entropy, backward, Adam;
    /root/reference volpick/model/models.py:34-51,160-185) on VCSEIS-shaped synthetic batches resident in HBM, activation /
    gradient rows stored as bfloat16 with fp32 accumulation (vp_train_create_dtype(VP_TRAIN_BF16)).  Beside it, as the
    stated baseline, the same step through stock PyTorch-ROCm autograd (eager, bf16 autocast) on the same GPU."""
    import torch

    import volpick_amd as va
    from volpick_amd.synthetic import synthetic_windows
    from volpick_amd.train import PhaseNetTrainer, gaussian_labels

    rng = np.random.default_rng(1005)
    x = synthetic_windows(batch, 3001, seed=1005)
    x = x - x.mean(-1, keepdims=True)
    x = (x / (np.abs(x).max(-1, keepdims=True) + 1e-10)).astype(np.float32)
    p = rng.integers(300, 1500, batch).astype(float)
    y = gaussian_labels(p, p + rng.integers(200, 1200, batch))
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    # the timed loop alternates TWO resident batches, so that every step meets a tensor pair other than the previous step's and
    # pays what a loader's fresh batch pays (the producer event; no `inputs_unchanged` promise anywhere in the timed region)
    xd2, yd2 = torch.roll(xd, 1, 0).contiguous(), torch.roll(yd, 1, 0).contiguous()
    pairs = ((xd, yd), (xd2, yd2))
    torch.cuda.synchronize()
    tr = PhaseNetTrainer(va.PhaseNet.from_pretrained("volpick"), max_batch=batch, dtype="bf16")
    for _ in range(warmup):
        tr.step(xd, yd, 1e-4, want_loss=False)
    tr.synchronize()
    # untimed steps until the clock has settled, as for the forward regions (this leg starts behind ~25 s of CPU baseline legs
    # with the GPU idle: its first ~150 steps ran 15 % slower than tools/train_probe.py's steady 1.65-1.68 ms)
    settle_steps, t_settle = 0, time.perf_counter()
    while time.perf_counter() - t_settle < 0.4:
        for i in range(steps):
            tr.step(*pairs[i & 1], 1e-4, want_loss=False)
        tr.synchronize()
        settle_steps += steps
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(*pairs[i & 1], 1e-4, want_loss=False)
        tr.synchronize()
        times.append((time.perf_counter() - t0) / steps)
    dt = statistics.median(times)
    # beside it: one batch re-run under the caller's promise that nothing wrote to it (no producer event per step)
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(xd, yd, 1e-4, want_loss=False, inputs_unchanged=True)
    tr.synchronize()
    dt_same = (time.perf_counter() - t0) / steps
    # loss of the batch at the weights the timed steps arrived at (no update), and -- the comparator -- the same batch through
    # torch autograd's forward on the oracle module carrying THE SAME weights and the same bf16 storage points
    w_now = tr.weights()
    loss = tr.step(xd, yd, 0.0, update=False)
    launches_fwd_bwd = int(tr._lib.vp_train_launch_count(tr._h))
    tr.step(xd, yd, 1e-4, want_loss=False)
    tr.synchronize()
    launches = int(tr._lib.vp_train_launch_count(tr._h))
    loss_torch = None
    try:
        from oracle.bf16_emulation import bf16_storage
        from oracle.models import load_pretrained as _lp

        chk = _lp("phasenet")
        chk.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in w_now.items()}, strict=False)
        chk = chk.cuda().train()
        with torch.no_grad(), bf16_storage(chk) as cn:
            pr = cn(xd)
            loss_torch = float(-(yd * torch.log(pr + 1e-5)).mean(-1).sum(-1).mean())
        del chk, pr
    except Exception as e:  # noqa: BLE001 -- the comparator must not cost the line
        loss_torch = repr(e)[:200]
    flop = 3 * 38.93e6  # forward FLOP per 3x3001 window; backward = input gradient + weight gradient = 2x
    out = {
        "metric": "PhaseNet training windows/sec (fwd + loss + bwd + Adam)", "value": batch / dt, "unit": "windows/s",
        "ms_per_step": dt * 1e3, "ms_per_step_all": [t * 1e3 for t in times], "batch": batch, "steps": steps, "warmup": warmup,
        "inputs": "two resident batches alternating (a fresh tensor pair every step)", "ms_per_step_same_batch_promised": dt_same * 1e3,
        "settle_steps": settle_steps,
        "dtype": "bf16 storage / f32 accumulate", "data": "synthetic (VCSEIS-shaped: 3 x 3001, Gaussian P/S labels sigma 20)",
        "launches_per_step": launches, "launches_without_update": launches_fwd_bwd, "loss_after": loss,
        "loss_torch_same_batch": loss_torch,
        "loss_note": "loss_after: the batch at the weights the timed steps arrived at (vp_train_step, update = 0); "
                     "loss_torch_same_batch: the same batch and weights through the torch module (training-mode BatchNorm, "
                     "bf16 storage points of oracle/bf16_emulation.py) on this GPU",
        "roofline": {"bound": "mfma", "achieved": flop * batch / dt / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": flop * batch / dt / (PEAK_FP32_TFLOPS * 1e12),
                     "basis": "algorithmic 3 x forward FLOP of the whole step (all launches) / step time, against the dense fp32 "
                              "rate the reference arithmetic is priced in; since round 5 the convs run the bf16 MFMA with "
                              "three-piece weights on the bf16-stored rows (two layers the fp32 MFMA), the weight gradients "
                              "the bf16 MFMA (exact on bf16-stored rows)"},
    }
    tr.close()
    # the same step through stock PyTorch-ROCm (MIOpen / rocBLAS kernels, eager autograd) on this GPU.  On a fresh box MIOpen
    # has no kernel cache: its find + compile of the 19 conv layers' forward / backward kernels takes ~35 s before the first
    # step (MIOPEN_FIND_MODE=FAST takes 7 s and then runs the step at 573 ms instead of 12.5: not a baseline); --no-train-torch
    # skips the leg
    if not torch_baseline:
        return out
    try:
        from oracle.models import load_pretrained

        t_setup = time.perf_counter()
        net = load_pretrained("phasenet").cuda().train()
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)

        def torch_step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = net(xd)
            l = -(yd * torch.log(pred.float() + 1e-5)).mean(-1).sum(-1).mean()
            l.backward()
            opt.step()

        for _ in range(2):
            torch_step()
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t_setup
        t0 = time.perf_counter()
        for _ in range(torch_steps):
            torch_step()
        torch.cuda.synchronize()
        gt = (time.perf_counter() - t0) / torch_steps
        out["torch_rocm_same_gpu"] = {"value": batch / gt, "unit": "windows/s", "ms_per_step": gt * 1e3, "first_steps_s": t_setup,
                                      "note": f"torch {torch.__version__} eager autograd + torch.optim.Adam, bf16 autocast, batch {batch}: the "
                                              "stated baseline of this row (the reference trains through PyTorch Lightning on one GPU)"}
        out["vs_torch_rocm"] = gt / dt
        del net, opt
    except Exception as e:  # noqa: BLE001 -- the baseline is context, its failure must not cost the line
        out["torch_rocm_same_gpu"] = {"error": repr(e)[:300]}
    return out


def station_day_mseed(hours=24):
    """The committed six-minute Steim-2 fixture (tests/golden/bench_steim2_6min.mseed: 36 records, HHZ/HHN/HHE at 100 Hz)
    tiled to `hours` by patching the records' start times -> (file bytes, the fixture's bytes, its record table, its samples)."""
    import struct
    from datetime import datetime, timedelta, timezone

    import volpick_amd.io as vio

    blob0 = (ROOT / "tests" / "golden" / "bench_steim2_6min.mseed").read_bytes()
    want = np.load(ROOT / "tests" / "golden" / "bench_steim2_6min_samples.npz")
    recs0 = vio.scan_mseed(blob0)
    tiles = hours * 10
    epoch = datetime(1970, 1, 1, tzinfo=timezone.utc)
    parts = []
    for k in range(tiles):
        b = bytearray(blob0)
        for r in recs0:
            t = epoch + timedelta(microseconds=int(r["start_us"]) + k * 360_000_000)
            # the BTIME carries 100 us units; a blockette 1001 of the record keeps its microsecond offset
            struct.pack_into(">HHBBBBH", b, int(r["offset"]) + 20, t.year, t.timetuple().tm_yday, t.hour, t.minute, t.second, 0,
                             t.microsecond // 100)
        parts.append(bytes(b))
    return b"".join(parts), blob0, recs0, want


def bench_mseed(hours=24, iters=30):
    """SURVEY 8f-1 (what feeds A1; /root/reference volpick/data/convert.py:7 reads through obspy): Steim-2 decode of one
    three-component station-day (100 Hz, 4096-byte records) with the file resident in HBM -- vp_mseed_decode, HIP events
    around `iters` launches (vp_mseed_decode_bench).  Input: the committed six-minute fixture tests/golden/
    bench_steim2_6min.mseed tiled to 24 h by patching the records' start times (records then arrive out of time order,
    as archives deliver them).  Checked against the fixture's own samples; the CPU decoder of oracle/mseed.py beside it."""
    import struct
    from datetime import datetime, timedelta, timezone

    import torch

    import volpick_amd as va
    import volpick_amd.io as vio
    from volpick_amd import _lib

    lib = _lib.load()
    buf, blob0, recs0, want = station_day_mseed(hours)
    t0 = time.perf_counter()
    recs = vio.scan_mseed(buf)
    t_scan = time.perf_counter() - t0
    ns = recs["nsamples"].astype(np.int64)
    index = (np.cumsum(ns) - ns).astype(np.int64)
    total = int(ns.sum())
    dbuf = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
    dout = torch.empty(total, dtype=torch.int32, device="cuda")
    recs_c = (_lib.VpMseedRecord * len(recs)).from_buffer_copy(np.ascontiguousarray(recs).tobytes())
    ms = C.c_float(0)
    _lib.check(lib.vp_mseed_decode_bench(0, dbuf.data_ptr(), len(buf), recs_c, index.ctypes.data_as(C.POINTER(C.c_int64)),
                                         len(recs), _lib.VP_SAMPLES_INT32, dout.data_ptr(), total, iters, C.byref(ms)),
               "vp_mseed_decode_bench")
    got = dout.cpu().numpy()
    first_tile = np.concatenate([want[c] for c in ("HHZ", "HHN", "HHE")])
    exact = bool(np.array_equal(got[: first_tile.size], first_tile) and np.array_equal(got[-first_tile.size:], first_tile))
    payload = int((recs["reclen"] - recs["data_offset"]).sum())
    algo = payload + 4 * total  # every payload byte read once, every int32 sample written once
    va.read(buf[: len(blob0)])
    t0 = time.perf_counter()
    st = va.read(buf)
    t_read = time.perf_counter() - t0
    day_ok = len(st) == 3 and all(tr.stats.npts == total // 3 for tr in st)

    # ---- file -> picks: what a caller of obspy.read + classify waits for (/root/reference volpick/data/convert.py:7,150 ahead of
    # README.md:54-66): the station-day file in host memory -> read() -> PhaseNet.classify() -> pick list, median of 7 calls;
    # device-resident read (samples stay in HBM, trace.data copies on demand) and the plain host Stream, phases once
    def med(fn, n=7):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        return statistics.median(ts), r

    picker = va.PhaseNet.from_pretrained("volpick").cuda()
    ckw = dict(batch_size=256, overlap=1500, blinding=(0, 0), stacking="avg")
    picker.classify(va.read(buf, device_resident=True), **ckw)  # warm-up: contexts, scratch
    f2p_dev, n_picks_dev = med(lambda: len(picker.classify(va.read(buf, device_resident=True), **ckw).picks))
    f2p_host, n_picks_host = med(lambda: len(picker.classify(va.read(buf), **ckw).picks))
    t_scan2, recs2 = med(lambda: vio.scan_mseed(buf))
    t_seg, _ = med(lambda: vio._segments(recs2))
    t_read_dev, st_dev = med(lambda: va.read(buf, device_resident=True))
    t_cls_dev, _ = med(lambda: len(picker.classify(st_dev, **ckw).picks))
    t_read_host, st_host = med(lambda: va.read(buf))
    t_cls_host, _ = med(lambda: len(picker.classify(st_host, **ckw).picks))
    picker._release()
    out = {
        "metric": "miniSEED samples decoded per second (Steim-2, one 3-component station-day, file resident in HBM)",
        "value": total / (ms.value * 1e-3), "unit": "samples/s", "kernel_ms": ms.value, "records": int(len(recs)), "samples": total,
        "file_bytes": len(buf), "bytes_per_sample": len(buf) / total, "bit_exact_vs_fixture_samples": exact,
        "read_gives_three_day_long_traces": day_ok,
        "roofline": {"bound": "hbm", "achieved": algo / (ms.value * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": algo / (ms.value * 1e-3) / (PEAK_HBM_GBS * 1e9), "algorithmic_bytes": algo,
                     "basis": "payload bytes read + 4 B per decoded sample written, per launch / its duration (HIP events, "
                              f"{iters} launches back to back); 36 MB file + 104 MB of samples: latency of the serial Steim "
                              "difference chain inside a record, not bandwidth, bounds it.  The input is a 36-record fixture tiled 240x and "
                              "read by back-to-back launches: its 35 MB sit in the 256 MB Infinity Cache, so the READ side of `achieved` is "
                              "not an HBM measurement (the 104 MB written per launch are)"},
        "read_wall_ms_host_file_to_host_stream": t_read_host, "read_wall_ms_first_call": t_read * 1e3, "scan_ms": t_scan * 1e3,
        "file_to_picks_ms": f2p_dev,
        "file_to_picks": {
            "call": "PhaseNet.classify(va.read(file_bytes, device_resident=True), batch_size=256, overlap=1500) on the station-day, host file "
                    "bytes in, pick list out; median of 7",
            "device_resident_ms": f2p_dev, "host_stream_ms": f2p_host, "picks": n_picks_dev, "picks_host_stream": n_picks_host,
            "phases_ms": {"scan": t_scan2, "order_and_chain_records": t_seg, "read_device_resident (scan + upload 35 MB + decode)": t_read_dev,
                          "classify_device_resident": t_cls_dev, "read_host_stream (+ 104 MB to fresh host memory)": t_read_host,
                          "classify_host_stream (int32 counts uploaded, cast on the device)": t_cls_host},
            "note": "the host Stream pays for 104 MB of decoded samples landing in fresh host memory (first-touch page faults: 9 of its "
                    "13 ms; huge pages / MADV_POPULATE_WRITE measured the same) and for their way back up; the device-resident Stream "
                    "keeps them in HBM and copies to the host only if trace.data is read",
        },
    }
    from oracle import mseed as OM

    orecs = OM.scan_records(blob0)
    n_cpu, reps, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        n_cpu += sum(len(OM.decode_record(blob0, r)) for r in orecs)
        reps += 1
    t_cpu = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": n_cpu / t_cpu, "unit": "samples/s", "cores": 1, "kind": "port",
                           "sample": f"{reps} x the fixture's {len(orecs)} records ({n_cpu} samples) through oracle/mseed.py (numpy / Python)"}
    return out


def timed_repeats(run_once, sync_all, repeats, use_dist, dev):
    """R x (barrier + synchronize, K steps, synchronize + barrier) -> list of seconds, each the max over ranks.
    The clock stops when this rank's device has drained, in front of the closing barrier: the ranks started together, so
    the maximum over ranks of those spans is the job's span, without the latency of the barrier collective itself
    (0.1-0.2 ms, 5-8 % of a 20-step PhaseNet region)."""
    import torch
    import torch.distributed as dist

    times = []
    for _ in range(max(1, repeats)):
        sync_all()
        t0 = time.perf_counter()
        run_once()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        sync_all()
    OWN_TIMES[:] = times
    if use_dist:
        tt = torch.tensor(times, dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times = [float(v) for v in tt.tolist()]
    return times


OWN_TIMES = []  # this rank's own spans of the last timed_repeats call (its return value is the max over ranks)


def bench_model(model_name, env, cpu_budget_s):
    import torch
    import torch.distributed as dist

    import volpick_amd as va
    from volpick_amd import _lib
    from volpick_amd.distributed import broadcast_weights
    from volpick_amd.synthetic import synthetic_stream_array

    args, world, rank, use_dist, dev = env["args"], env["world"], env["rank"], env["use_dist"], env["dev"]
    lib = _lib.load()
    cls = va.PhaseNet if model_name == "phasenet" else va.EQTransformer
    model = cls.from_pretrained("volpick")
    model._max_batch = args.batch
    t_bcast = 0.0
    if use_dist:
        if rank != 0:
            model._weights = np.zeros_like(model._weights)  # only rank 0's copy is real
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        broadcast_weights(model, src=0)
        if model._handle is None:  # --rehearse-gloo: the host broadcast leaves the plan to be built
            model.cuda(dev)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    else:
        model.cuda(dev)
    h = model._handle

    # ---- workload: one stream per rank that cuts into exactly `batch` windows -------------
    T = model.in_samples
    if model_name == "phasenet":
        overlap, blinding = 1500, (0, 0)  # class defaults of the reference API
    else:
        overlap, blinding = 5500, (500, 500)  # README.md:57-58
    n_samples = T + (T - overlap) * (args.batch - 1)
    data, _, _ = synthetic_stream_array(n_samples, seed=1002 + rank)
    x = torch.from_numpy(data).to(dev)
    out = torch.empty((3, n_samples), dtype=torch.float32, device=dev)
    specs = model._trigger_specs({})
    c_specs = (_lib.VpTriggerSpec * len(specs))(*[_lib.VpTriggerSpec(r, t_on, t_off) for r, _, t_on, t_off in specs])
    cap = 8192
    on, off, peak = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int64 * cap)()
    val, spec_of = (C.c_float * cap)(), (C.c_int32 * cap)()
    found = C.c_int()
    fv, lv, nw = C.c_int64(), C.c_int64(), C.c_int64()

    # Three (PhaseNet) or four (EQTransformer) device contexts (HIP stream + workspace each; tools/ctx_sweep.sh: more
    # only add queueing, and four need more than HIP's default four hardware queues -- GPU_MAX_HW_QUEUES above), two
    # submits in flight per context: the host enqueues ahead of the GPU, and one context's latency-bound stages
    # (LSTM/attention, small tail kernels) overlap the other's MFMA-bound ones.  Every step is still one full pass
    # over one batch, and every step is collected inside the timed region.
    NCTX, DEPTH = (args.contexts if args.contexts > 0 else model.n_contexts), (args.depth if args.depth > 0 else 2)
    ctxs = [model._context(k) for k in range(NCTX)]
    outs = [out] + [torch.empty_like(out) for _ in range(NCTX - 1)]

    def submit(i):
        k, slot = i % NCTX, (i // NCTX) % DEPTH
        _lib.check(lib.vp_classify_submit(ctxs[k], slot, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE, n_samples,
                                          overlap, blinding[0], blinding[1], _lib.VP_STACK_AVG, args.batch, c_specs,
                                          len(specs), C.c_void_p(outs[k].data_ptr()), _lib.VP_MEM_DEVICE, cap),
                   "vp_classify_submit")

    def collect(i):
        k, slot = i % NCTX, (i // NCTX) % DEPTH
        _lib.check(lib.vp_classify_collect(ctxs[k], slot, C.byref(fv), C.byref(lv), C.byref(nw), on, off, peak, val,
                                           spec_of, cap, C.byref(found)), "vp_classify_collect")
        return found.value

    def run_steps(k):
        """k steps, each one full pass of the path over one 256-window batch; every step is collected."""
        inflight, n_picks = [], 0
        for i in range(k):
            if len(inflight) == NCTX * DEPTH:
                n_picks = collect(inflight.pop(0))
            submit(i)
            inflight.append(i)
        while inflight:
            n_picks = collect(inflight.pop(0))
        return n_picks

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    n_picks = run_steps(args.warmup)
    assert nw.value == args.batch, (nw.value, args.batch)
    # Clock settling, untimed: after an idle GPU the shader clock climbs for tens of milliseconds (round 4: the 21 regions of
    # 20 PhaseNet steps ran 100.3, 103.8, 99.9 ... 87.3 us per step in order, still falling at the end, against 84.9 us in the
    # 5 s region behind them), so W warm-up steps of 0.09 ms leave the timed regions on the ramp.  More warm-up steps of the same
    # kind run until --settle-seconds have passed; the timed regions are K steps each, exactly as before.
    settle_steps, t_settle = 0, time.perf_counter()
    while args.settle_seconds > 0 and time.perf_counter() - t_settle < args.settle_seconds:
        run_steps(max(args.steps, 20))
        settle_steps += max(args.steps, 20)
    torch.cuda.synchronize()
    t_settle = time.perf_counter() - t_settle
    times = timed_repeats(lambda: run_steps(args.steps), sync_all, args.repeats, use_dist, dev)
    env["own_times"] = list(OWN_TIMES)
    n_picks = found.value
    dt = statistics.median(times)
    windows = args.batch * args.steps * world
    value = windows / dt

    # ---- ONE long region (>= --sustain-seconds): long enough for DVFS to settle and for an outside sampler (the driver's
    # gpu_busy) to see the load; the shader clock right behind it from the in-kernel stamps of a debug-plan twin
    sustained = None
    if args.sustain_seconds > 0:
        k_long = max(args.steps, int(args.sustain_seconds / (dt / args.steps)) + 1)
        probe = clock_probe_create(cls, args.batch, dev)
        t_long = timed_repeats(lambda: run_steps(k_long), sync_all, 1, use_dist, dev)[0]
        ghz = clock_probe_read(probe, model_name, args.batch)
        sustained = {
            "seconds": t_long,
            "steps": k_long,
            "value": args.batch * k_long * world / t_long,
            "unit": "windows/s",
            "ms_per_step": t_long / k_long * 1e3,
            "vs_median_of_short_regions": (args.batch * k_long * world / t_long) / value,
            "shader_clock_ghz": ghz,
            "shader_clock_basis": "shader-cycle stamps at the two ends of the dominant kernel / the 100 MHz constant clock at "
                                  "the same two points (median over the workgroups of one launch of a debug-plan twin of the "
                                  "model, run right behind the region while the chip is hot)",
        }

    # ---- per-kernel HIP-event timing on the handle's stream -> roofline of the dominant kernel
    n_steps = lib.vp_step_count(h)
    ms = (C.c_float * n_steps)()
    _lib.check(lib.vp_profile_steps(h, args.batch, 20, ms, n_steps), "vp_profile_steps")
    kernels = []
    for i in range(n_steps):
        name, fl, iss = C.c_char_p(), C.c_double(), _lib.VpIssuedWork()
        lib.vp_step_info(h, i, C.byref(name), C.byref(fl))
        _lib.check(lib.vp_step_issued_work(h, i, C.byref(iss)), "vp_step_issued_work")
        k = {"name": name.value.decode(), "ms": float(ms[i]), "flop_per_window": fl.value,
             "issued_flop_per_window": {"mfma_f32": iss.mfma_f32_flop, "mfma_bf16": iss.mfma_bf16_flop,
                                        "valu": iss.valu_flop}}
        k["pipe_time_ms"] = pipe_time_s(k["issued_flop_per_window"], args.batch) * 1e3
        k["frac_of_pipe_peaks"] = k["pipe_time_ms"] / k["ms"] if k["ms"] > 0 else None
        if k["name"].startswith("fused.mid"):
            k["note"] = ("four windows per 1024-thread workgroup (teams of four waves): a 256-window launch holds 64 of the 256 CUs "
                         "(the other contexts' kernels run on the rest); two windows per workgroup took 91 us on 128 CUs, one 66 us "
                         "on all 256; frac_of_pipe_peaks is against the whole chip")
        kernels.append(k)
    fwd_ms = sum(k["ms"] for k in kernels)
    # Roofline candidates are the launches that hold >= 5 % of the forward FLOPs (the MFMA / packed-FMA bound ones).
    # EQTransformer's fused.mid (BiLSTM recurrences + attention, 2 % of the FLOPs) is a serial dependency chain bound
    # by instruction latency, not by a throughput roof; it stays in forward.kernels and is named in
    # roofline.latency_bound with its duration, so a reader sees when it is as long as the dominant compute launch.
    fl_total = sum(k["flop_per_window"] for k in kernels) or 1.0
    cand = [k for k in kernels if k["flop_per_window"] >= 0.05 * fl_total] or kernels
    dom = max(cand, key=lambda k: k["ms"])
    lat = [{"name": k["name"], "ms_back_to_back": k["ms"], "flop_share": k["flop_per_window"] / fl_total}
           for k in kernels if k not in cand and k["ms"] >= 0.5 * dom["ms"]]
    # The dominant launch is re-timed IN the pipeline (whole step list in order, events around it only): its inputs
    # then come from the preceding kernel instead of a warm re-run of itself -- the duration rocprofv3 reports
    # for it under this same command (profiles/).  200 passes enqueued back to back, one synchronisation: the launch
    # runs at the clock it has in the timed loop (a synchronisation per pass, as in round 1, lets the GPU idle in
    # between and times it 5-8 % slower).  The back-to-back figure stays in forward.kernels.
    dom_ms = C.c_float()
    _lib.check(lib.vp_profile_step_in_pipeline(h, args.batch, 200, kernels.index(dom), C.byref(dom_ms)),
               "vp_profile_step_in_pipeline")
    # The duration the fractions are taken on is one the timed loop pays: a one-launch plan's kernel cannot take longer than a
    # whole step of that loop, so where the HIP-event figure (one queue, a gap between two launches of it) exceeds the step,
    # the step is the duration (never a kernel_ms above ms_per_step).
    step_ms = dt / args.steps * 1e3
    dom_dur_ms = min(dom_ms.value, step_ms) if (len(kernels) == 1 and world == 1) else dom_ms.value
    per_s = args.batch / (dom_dur_ms * 1e-3) / 1e12 if dom_dur_ms > 0 else 0.0
    dom_algorithmic = dom["flop_per_window"] * per_s
    dom_issued_flop = sum(dom["issued_flop_per_window"].values())
    dom_pipe_s = pipe_time_s(dom["issued_flop_per_window"], args.batch)
    dom_achieved = dom_issued_flop * per_s                                              # issued TFLOP/s, all pipes
    dom_peak = dom_issued_flop * args.batch / dom_pipe_s / 1e12 if dom_pipe_s > 0 else 0.0  # the same mix at its pipes' peaks
    flop_w = lib.vp_flops_per_window(h)
    stage = (C.c_float * 4)()
    total_ms = C.c_float()
    lib.vp_set_timing(h, 1)  # stage split from one extra, un-timed, synchronous step
    _lib.check(lib.vp_classify(h, C.c_void_p(x.data_ptr()), _lib.VP_MEM_DEVICE, n_samples, overlap, blinding[0],
                               blinding[1], _lib.VP_STACK_AVG, args.batch, c_specs, len(specs),
                               C.c_void_p(out.data_ptr()), _lib.VP_MEM_DEVICE, C.byref(fv), C.byref(lv), C.byref(nw),
                               on, off, peak, val, spec_of, cap, C.byref(found)), "vp_classify")
    lib.vp_last_timing(h, C.byref(total_ms), stage)
    lib.vp_set_timing(h, 0)

    kname = dom["name"]
    kernel_label = ("pn_window_kernel: " if kname.startswith("fused.window") else
                    "eqt fused kernel: " if kname.startswith("fused.") else "conv_mfma_kernel: ") + kname
    result = {
        "metric": "waveform-windows/sec",
        "value": value,
        "unit": "windows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_LABEL,
        "data": "synthetic",
        "sustained": sustained,
        "timing": {
            "repeats": len(times),
            "statistic": "median over the repeats of the max-over-ranks time of K steps",
            "windows_per_s_min": windows / max(times),
            "windows_per_s_max": windows / min(times),
            "windows_per_s_first": windows / times[0],
            "settle": {"seconds": t_settle, "steps": settle_steps,
                       "note": "untimed steps between the W warm-up steps and the timed regions (clock settling, --settle-seconds)"},
            "ms_per_step_all": [t / args.steps * 1e3 for t in times],
        },
        "config": {
            "workload": f"{model.name} volpick, batch={args.batch}, 3x{T} windows, fp32, overlap={overlap}, "
                        f"blinding={list(blinding)}, stacking=avg, full path A2-A8 per step",
            "batch": args.batch,
            "in_samples": T,
            "parallelism": f"stream-sharded x{world}, weights broadcast once (RCCL)" if world > 1 else "one rank (nothing broadcast)",
            "device_contexts": NCTX,
            "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "inflight_steps_per_context": DEPTH,
            "input_residency": "every step re-reads the same device-resident stream (18 / 1.6 MB): Infinity-Cache "
                               "resident; the kernels are compute-bound, so this does not flatter the number",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": kernel_label,
            "achieved": dom_achieved,
            "peak": dom_peak,
            "unit": "TFLOP/s",
            "frac": (dom_achieved / dom_peak) if dom_peak > 0 else None,
            "basis": "ISSUED work of this launch (vp_step_issued_work: whole MFMA tiles, padded channels, recomputed halos, "
                     "folded taps; every one of the six bf16 MFMAs of an exact three-piece product counted) / its duration "
                     "(HIP events, in the pipeline) = achieved; peak = the same instruction mix with every pipe at its dense "
                     "peak, sum_i issued_i / (sum_i issued_i / peak_i) with fp32 MFMA 157.3, bf16 MFMA 2500 and packed fp32 "
                     "FMA 157.3 TFLOP/s (MI355X_MICROARCH.md) -- the pipes share the SIMD's issue, so their times add; "
                     "frac = pipe time / kernel duration <= 1 by construction",
            "frac_note": ("the fraction moves with the instruction mix as well as with the duration: moving a layer from the fp32 "
                          "to the bf16 matrix cores cuts its pipe time to 6/16, so a kernel can get faster while frac falls "
                          "(pn_window_kernel, round 2 -> 3: pipe time 56 -> 46 us per 256 windows with up1.same / up2.same on the "
                          "bf16 pipes, duration 102 -> 94-98 us at the same clock); read it beside kernel_ms"),
            "issued_flop_per_window": dom["issued_flop_per_window"],
            "useful_frac": dom["flop_per_window"] * args.batch * 6 / (PEAK_BF16_TFLOPS * 1e12) / (dom_dur_ms * 1e-3) if dom_dur_ms > 0 else None,
            "useful_frac_basis": "ALGORITHMIC FLOP (SURVEY 8d) x 6 piece products at the dense bf16 MFMA peak / the same duration: what "
                                 "frac would be without padded tiles, recomputed halos and the fp32-MFMA / VALU layers",
            "pipe_time_ms": dom_pipe_s * 1e3,
            "kernel_ms": dom_dur_ms,
            "kernel_ms_hip_events": dom_ms.value,
            "kernel_ms_back_to_back": dom["ms"],
            "kernel_ms_note": "HIP events around the launch, 200 passes on ONE stream: every pass pays the event records and the "
                              "drain / refill of the chip between two launches of one queue (what rocprofv3's AverageNs of the same "
                              "command shows, profiles/); in the timed loop the device contexts' launches follow each other without "
                              "that gap, so a whole STEP there can be shorter than this figure (step_bound)",
            "step_bound": ({"ms": sustained["ms_per_step"] * (len(kernels) == 1), "frac": dom_pipe_s * 1e3 / sustained["ms_per_step"],
                            "note": "one-launch plan: the launch cannot take longer than a whole step of the sustained region"}
                           if sustained and len(kernels) == 1 and world == 1 else None),
            "algorithmic": {
                "tflops": dom_algorithmic,
                "flop_over_fp32_peak": dom_algorithmic / PEAK_FP32_TFLOPS,
                "note": "2 x MAC of the REFERENCE layers of this launch / its duration, against the dense fp32 rate: the "
                        "price of the reference arithmetic, not a ceiling of this kernel (bf16-piece layers cost 6/16 of "
                        "the fp32 MFMA time, the decoder fold removes 29-45 % of the taps): may exceed 1",
            },
            "traffic": traffic_bytes(model_name, kname),
            "latency_bound": lat,
        },
        "forward": {
            "flop_per_window": flop_w,
            "launches": len(kernels),
            "sum_kernel_ms": fwd_ms,
            "tflops_kernels": flop_w * args.batch / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0,
            "fp32_frac_end_to_end": value / world * flop_w / (PEAK_FP32_TFLOPS * 1e12),
            "hbm_frac_compulsory": value / world * (2 * 3 * T * 4) / (PEAK_HBM_GBS * 1e9),
            "traffic_bytes_per_step": traffic_bytes(model_name, "_step_total"),
            "stage_ms_one_sync_step": {"forward": stage[1], "stack": stage[2], "trigger_scan": stage[3]},
            "picks_per_step": n_picks,
            "kernels": kernels,
        },
        "weight_broadcast_s": t_bcast,
        "weight_broadcast_path": _bcast_path(),
    }
    if use_dist:  # what every rank saw, so that a multi-GPU line checks itself
        info = {"rank": rank, "device": torch.cuda.current_device(), "ms_per_step_own_median": None,
                "weight_broadcast_s": t_bcast, "weight_broadcast_path": _bcast_path(), "rccl_comm_ranks": _rccl_ranks(),
                "librccl": _rccl_libraries(), "windows_per_step": args.batch}
        info["ms_per_step_own_median"] = statistics.median(env.get("own_times", [dt])) / args.steps * 1e3
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
        result["ranks"] = gathered
    if cpu_budget_s > 0:
        result["cpu_baseline"] = cpu_baseline(model_name, data, overlap, blinding, args.batch, cpu_budget_s)
        result["pick_parity"] = pick_parity(model, model_name, data, overlap, blinding, args.batch)
    if cpu_budget_s > 0 and not args.no_api:
        result["api"] = bench_api(model_name, model, args.batch, oracle_threads=result["cpu_baseline"]["cores"])
    model._release()
    return result


def bench_strong(env):
    """BASELINE configs[3]: ONE 24 h three-component stream (8,640,000 samples at 100 Hz, overlap 5500, blinding
    (500, 500): 17,269 EQTransformer windows) classified by N ranks, each on its own window range
    (volpick_amd.distributed.classify_stream_sharded: segment + halo per rank, trigger lists stitched on rank 0).
    A step = the whole day once; the rank's segment is resident in HBM before the timed region."""
    import torch
    import torch.distributed as dist

    import volpick_amd as va
    from volpick_amd import UTCDateTime
    from volpick_amd.distributed import broadcast_weights, classify_stream_sharded
    from volpick_amd.segments import plan_segments
    from volpick_amd.synthetic import synthetic_stream_array

    args, world, rank, use_dist, dev = env["args"], env["world"], env["rank"], env["use_dist"], env["dev"]
    model = va.EQTransformer.from_pretrained("volpick")
    model._max_batch = args.batch
    t_bcast = 0.0
    if use_dist:
        if rank != 0:
            model._weights = np.zeros_like(model._weights)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        broadcast_weights(model, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    else:
        model.cuda(dev)
    T, overlap, blinding, n = 6000, 5500, (500, 500), 8_640_000
    kw = dict(overlap=overlap, blinding=blinding, batch_size=args.batch, stacking="avg")
    segs = plan_segments(n, T, overlap, blinding, world)
    sg = segs[rank] if rank < len(segs) else None
    data, _, _ = synthetic_stream_array(n, seed=1004, n_events=600)  # every rank generates the day, uploads its segment only
    mine = torch.from_numpy(np.ascontiguousarray(data[:, sg["lo"]:sg["hi"]])).to(dev) if sg else None
    del data
    t_start = UTCDateTime("2021-01-01T00:00:00")
    res, tms = [None], []

    def one_day():
        tm = {}
        res[0] = classify_stream_sharded(model, (n, lambda lo, hi: mine), t_start, "XX.DAY.", timing=tm, **kw)
        tms.append(tm)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    steps, warm = max(1, min(args.steps, 20)), max(1, min(args.warmup, 3))
    for _ in range(warm):
        one_day()
    tms.clear()
    times = timed_repeats(lambda: [one_day() for _ in range(steps)], sync_all, min(args.repeats, 5), use_dist, dev)
    dt = statistics.median(times)
    n_windows = 17_269
    # this rank's split of a call (medians over the timed calls): gpu_ms = annotate of its segment, synchronised; fixed_ms = the
    # rest (segment plan, trigger scan + result copy, header / column exchange, stitching and record columns on rank 0)
    split = {k: statistics.median(t[k] for t in tms) for k in ("total_ms", "gpu_ms", "scan_ms", "wait_ms", "exchange_ms", "stitch_ms", "fixed_ms")}
    out = {
        "metric": "waveform-windows/sec",
        "value": n_windows * steps / dt,
        "unit": "windows/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warm,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "scaling_note": "LATENCY of one station-day: per rank and call, gpu_ms (annotate of its segment) shrinks with N, "
                        "fixed_ms (trigger scan, two small collectives of integer columns, stitching on rank 0) does not; "
                        "the >= 7x target of BASELINE.json is the weak-scaling (many streams) default mode of this script",
        "call_split_ms": {k: round(v, 3) for k, v in split.items()},
        "vs_baseline": None,
        "dtype": DTYPE_LABEL,
        "data": "synthetic",
        "timing": {"repeats": len(times), "windows_per_s_min": n_windows * steps / max(times),
                   "windows_per_s_max": n_windows * steps / min(times)},
        "config": {
            "workload": "EQTransformer volpick, ONE 24 h 3-component stream (8,640,000 samples), overlap=5500, "
                        "blinding=[500, 500], 17,269 windows, split by window range over the ranks; a step = the whole "
                        "day through annotate + trigger scan + stitching of the pick lists on rank 0",
            "batch": args.batch,
            "parallelism": (f"window-range sharding x{world} (segment + halo per rank), weights broadcast once, trigger columns "
                            "gathered as fixed-width integer tensors (no pickling)") if world > 1 else
                           "one rank: the whole day on one GPU (no broadcast, no exchange)",
            "segment_samples_this_rank": (sg["hi"] - sg["lo"]) if sg else 0,
        },
        "weight_broadcast_s": t_bcast,
        "weight_broadcast_path": _bcast_path(),
    }
    if use_dist:  # what every rank saw (its own segment, its own median), so that a multi-GPU line checks itself
        info = {"rank": rank, "device": torch.cuda.current_device(), "segment": [sg["lo"], sg["hi"]] if sg else None,
                "keeps": [sg["keep_lo"], sg["keep_hi"]] if sg else None,
                "ms_per_step_own_median": statistics.median(OWN_TIMES) / steps * 1e3 if OWN_TIMES else None,
                "gpu_ms": round(split["gpu_ms"], 3), "fixed_ms": round(split["fixed_ms"], 3), "call_split_ms": {k: round(v, 3) for k, v in split.items()},
                "weight_broadcast_s": t_bcast, "weight_broadcast_path": _bcast_path(), "rccl_comm_ranks": _rccl_ranks(),
                "librccl": _rccl_libraries()}
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
        out["ranks"] = gathered
    if rank == 0:
        import hashlib

        out["picks"] = len(res[0].picks)
        out["detections"] = len(res[0].detections)
        # what the N ranks stitched together, as one number a second run at another N can be compared with
        rows = sorted((p.phase, p.trace_id, p.start_time._us, p.end_time._us, p.peak_time._us, float(np.float32(p.peak_value))) for p in res[0].picks)
        out["picks_digest"] = hashlib.sha1(repr(rows).encode()).hexdigest()[:16]
    return out


def traffic_bytes(model_name, step_name):
    """HBM bytes per launch of a kernel (or "_step_total": all launches of one step) from the newest committed rocprofv3
    PMC summary (profiles/r*_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md;
    separate --pmc passes of tools/run_forward.py on that round's build), else None."""
    files = sorted((ROOT / "profiles").glob("r*_traffic.json"))
    for f in reversed(files):
        try:
            v = json.loads(f.read_text()).get(model_name, {}).get(step_name)
        except (ValueError, OSError):
            continue
        if v is not None:
            return v
    return None


def pick_parity(model, model_name, data, overlap, blinding, batch, n_windows=64):
    """The second half of BASELINE.json's metric ("P/S pick delta-t vs ref"): the picks of the HIP path on a
    64-window prefix of the bench stream against the CPU oracle's on the same samples (the oracle stands in for the
    un-installable SeisBench reference: parity unpinned)."""
    from oracle import pipeline as OP
    from oracle.models import load_pretrained

    net = load_pretrained(model_name)
    T = net.in_samples
    seg = data[:, : T + (T - overlap) * (n_windows - 1)]
    want = OP.classify_array(net, seg, overlap=overlap, blinding=blinding, batch_size=batch)["picks"]
    args = model._argdict(dict(overlap=overlap, blinding=blinding, stacking="avg", batch_size=batch))
    specs = [s for s in model._trigger_specs(args) if s[1] != "Detection"]
    got, _ = model._classify_block(seg, args, specs)
    got = sorted((specs[si][1], on, off, pk, v) for si, on, off, pk, v in got)
    want = sorted(want)
    out = {"vs": "CPU oracle (port; parity unpinned)", "windows": n_windows, "picks_hip": len(got), "picks_oracle": len(want)}
    if len(got) == len(want) and all(g[0] == w[0] for g, w in zip(got, want)):
        out["max_abs_dt_samples"] = max([abs(g[3] - w[3]) for g, w in zip(got, want)], default=0)
        out["max_abs_dt_s"] = out["max_abs_dt_samples"] / 100.0
        out["max_abs_dvalue"] = float(max([abs(g[4] - w[4]) for g, w in zip(got, want)], default=0.0))
    return out


def host_cpu():
    """(model name, physical cores, logical CPUs) from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            key, _, v = line.partition(":")
            key, v = key.strip(), v.strip()
            if key == "processor":
                logical += 1
            elif key == "model name":
                model = v
            elif key == "physical id":
                phys = v
            elif key == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    n_phys = len(cores) or logical or (os.cpu_count() or 1)
    try:  # a container may be pinned to fewer CPUs than the host has
        n_phys = max(1, min(n_phys, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        pass
    return model, n_phys, logical


def cpu_quota():
    """CPUs' worth of time the cgroup grants this job (cgroup v2 cpu.max), or None: a one-GPU box hands out 16 of the
    host's 256 logical CPUs that way -- affinity and /proc/cpuinfo still show them all, and threads beyond the quota are
    throttled, not run."""
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return None if quota == "max" else max(1, int(quota) // int(period))
    except (OSError, ValueError):
        return None


def cpu_baseline(model_name, data, overlap, blinding, batch, budget_s):
    """The CPU oracle (torch-CPU restatement of the reference path, kind "port": the stand-in for SeisBench on
    the CPU, which cannot be installed here) timed on this host over whole `batch`-window chunks of the bench
    stream (batch_size filled, as the reference's classify would), with 1 thread, with one thread per physical
    core, and with two counts in between -- torch's intra-op threading of these small 1-D convolutions does not
    scale to a two-socket host, so the fastest leg is the one reported as `value` (its thread count in `cores`).
    Bounded: each leg runs chunks until its share of `budget_s` is used (at least one chunk)."""
    import torch

    from oracle import pipeline as OP
    from oracle.models import load_pretrained

    net = load_pretrained(model_name)
    cpu_model, n_phys, n_logical = host_cpu()
    T = net.in_samples
    step = T - overlap
    seg = data[:, : T + step * (batch - 1)]
    threads_before = torch.get_num_threads()
    quota = cpu_quota()
    counts = sorted({1, min(8, n_phys), min(quota or 16, n_phys), min(32, n_phys), n_phys})
    share = budget_s / len(counts)
    legs = []
    for threads in counts:
        torch.set_num_threads(threads)
        OP.classify_array(net, data[:, : T + step * 7], overlap=overlap, blinding=blinding, batch_size=batch)  # warm-up
        t0 = time.perf_counter()
        OP.classify_array(net, data[:, : T + step * 15], overlap=overlap, blinding=blinding, batch_size=batch)
        per_win = (time.perf_counter() - t0) / 16
        # a full chunk may need far longer than this leg's share: then a bounded prefix of it is timed instead
        n_win = batch if per_win * batch <= share else max(16, int(share / per_win) // 16 * 16)
        part = seg[:, : T + step * (n_win - 1)]
        done, t_used = 0, 0.0
        while True:
            t0 = time.perf_counter()
            OP.classify_array(net, part, overlap=overlap, blinding=blinding, batch_size=batch)
            t_used += time.perf_counter() - t0
            done += n_win
            if t_used >= share:
                break
        legs.append({"threads": threads, "value": done / t_used, "windows": done, "seconds": t_used, "chunk_windows": n_win})
    torch.set_num_threads(threads_before)
    best = max(legs, key=lambda l: l["value"])
    one = legs[0]
    return {
        "value": best["value"],
        "unit": "windows/s",
        "cores": best["threads"],
        "kind": "port",
        "cpu_model": cpu_model,
        "physical_cores": n_phys,
        "logical_cpus": n_logical,
        "cgroup_cpu_quota": quota,
        "one_thread": {"value": one["value"], "cores": 1},
        "all_physical_cores": {"value": legs[-1]["value"], "cores": legs[-1]["threads"]},
        "legs": legs,
        "sample": f"{best['windows']} windows ({best['windows'] // best['chunk_windows']} x {best['chunk_windows']}-window "
                  f"chunk of the bench stream, batch_size={batch}) through oracle.pipeline.classify_array, torch "
                  f"{torch.__version__} CPU; fastest of the legs with {counts} threads on {n_phys} physical cores",
    }


if __name__ == "__main__":
    main()
