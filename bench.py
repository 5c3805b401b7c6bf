#!/usr/bin/env python3
"""Headline benchmark: waveform-windows/s of the volpick picking path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1: run directly)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path over one batch of 256
windows cut from a device-resident synthetic 3-component stream: window gather +
annotate_batch_pre, model forward, blinding + overlap stacking, trigger/peak scan of the
phase traces.  Workload = BASELINE.json configs[1]: PhaseNet volpick, batch 256, 3x3001, fp32
(--model eqtransformer runs configs[2]).  Inputs are resident in HBM before the timed region.
Multi-GPU: every rank owns whole station streams (weak scaling, no data-path collective); the
weights are broadcast once from rank 0 over RCCL before the timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_FP32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="phasenet", choices=["phasenet", "eqtransformer"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--contexts", type=int, default=3, help="device contexts (stream + workspace) steps alternate over")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK set) the RCCL path is exercised even for one rank
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import volpick_amd as va
    from volpick_amd import _lib
    from volpick_amd.distributed import broadcast_weights
    from volpick_amd.synthetic import synthetic_stream_array

    lib = _lib.load()
    cls = va.PhaseNet if args.model == "phasenet" else va.EQTransformer
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
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    else:
        model.cuda(dev)
    h = model._handle

    # ---- workload: one stream per rank that cuts into exactly `batch` windows -------------
    T = model.in_samples
    if args.model == "phasenet":
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

    # Three device contexts (HIP stream + workspace each; 3 measured best, 4+ share hardware queues), two
    # submits in flight per context: the host
    # enqueues ahead of the GPU, and one context's latency-bound stages (LSTM/attention, small tail
    # kernels) overlap the other's MFMA-bound ones.  Every step is still one full pass over one batch.
    NCTX, DEPTH = max(1, args.contexts), 2
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
    sync_all()
    t0 = time.perf_counter()
    n_picks = run_steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    windows = args.batch * args.steps * world
    value = windows / dt

    # ---- per-kernel HIP-event timing on the handle's stream -> roofline of the dominant kernel
    n_steps = lib.vp_step_count(h)
    ms = (C.c_float * n_steps)()
    _lib.check(lib.vp_profile_steps(h, args.batch, 20, ms, n_steps), "vp_profile_steps")
    kernels = []
    for i in range(n_steps):
        name, fl = C.c_char_p(), C.c_double()
        lib.vp_step_info(h, i, C.byref(name), C.byref(fl))
        kernels.append({"name": name.value.decode(), "ms": float(ms[i]), "flop_per_window": fl.value})
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
    # for it under this same command (profiles/).  The back-to-back figure stays in forward.kernels.
    dom_ms = C.c_float()
    _lib.check(lib.vp_profile_step_in_pipeline(h, args.batch, 50, kernels.index(dom), C.byref(dom_ms)),
               "vp_profile_step_in_pipeline")
    dom_tflops = dom["flop_per_window"] * args.batch / (dom_ms.value * 1e-3) / 1e12 if dom_ms.value > 0 else 0.0
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
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{model.name} volpick, batch={args.batch}, 3x{T} windows, fp32, overlap={overlap}, "
                        f"blinding={list(blinding)}, stacking=avg, full path A2-A8 per step",
            "batch": args.batch,
            "in_samples": T,
            "parallelism": f"stream-sharded x{world}, weights broadcast once (RCCL)",
            "device_contexts": NCTX,
            "inflight_steps_per_context": DEPTH,
        },
        "roofline": {
            "bound": "mfma",
            "kernel": ("pn_window_kernel: " if dom["name"].startswith("fused.window") else "conv_mfma_kernel: ") + dom["name"],
            "achieved": dom_tflops,
            "peak": PEAK_FP32_TFLOPS,
            "unit": "TFLOP/s",
            "frac": dom_tflops / PEAK_FP32_TFLOPS,
            "traffic": traffic_bytes(args.model, dom["name"]),
            "kernel_ms": dom_ms.value,
            "kernel_ms_back_to_back": dom["ms"],
            "latency_bound": lat,
        },
        "forward": {
            "flop_per_window": flop_w,
            "sum_kernel_ms": fwd_ms,
            "tflops_kernels": flop_w * args.batch / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0,
            "fp32_frac_end_to_end": value / world * flop_w / (PEAK_FP32_TFLOPS * 1e12),
            "hbm_frac_compulsory": value / world * (2 * 3 * T * 4) / (PEAK_HBM_GBS * 1e9),
            "stage_ms_one_sync_step": {"forward": stage[1], "stack": stage[2], "trigger_scan": stage[3]},
            "picks_per_step": n_picks,
            "kernels": kernels,
        },
        "weight_broadcast_s": t_bcast,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.model, data, overlap, blinding, args.batch, args.cpu_seconds)
        result["pick_parity"] = pick_parity(model, args.model, data, overlap, blinding, args.batch)
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def traffic_bytes(model_name, step_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md; separate --pmc runs of tools/run_forward.py on this build), else None."""
    f = ROOT / "profiles" / "r01_traffic.json"
    if not f.exists():
        return None
    try:
        return json.loads(f.read_text()).get(model_name, {}).get(step_name)
    except (ValueError, OSError):
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


def cpu_baseline(model_name, data, overlap, blinding, batch, budget_s):
    """The CPU oracle (torch-CPU restatement of the reference path, kind "port") timed on this
    host's cores over a bounded prefix of the same stream."""
    import torch

    from oracle import pipeline as OP
    from oracle.models import load_pretrained

    net = load_pretrained(model_name)
    cores = torch.get_num_threads()
    T = net.in_samples
    step = T - overlap
    done, t_used = 0, 0.0
    chunk = 64
    # warm-up (thread pools, oneDNN primitives)
    OP.classify_array(net, data[:, : T + step * 7], overlap=overlap, blinding=blinding, batch_size=batch)
    while t_used < budget_s and done < batch * 8:
        n = T + step * (chunk - 1)
        seg = data[:, :n]
        t0 = time.perf_counter()
        OP.classify_array(net, seg, overlap=overlap, blinding=blinding, batch_size=batch)
        t_used += time.perf_counter() - t0
        done += chunk
    return {
        "value": done / t_used,
        "unit": "windows/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} windows ({done // chunk} x {chunk}-window prefix of the bench stream) through "
                  f"oracle.pipeline.classify_array, torch {torch.__version__} CPU, {cores} threads",
    }


if __name__ == "__main__":
    main()
