#!/usr/bin/env python3
"""bench.py -- variables sampled per second of the Gibbs sweep hot path on MI355X.

Workload (BASELINE.json config 3, the graph north_star quotes its target on): 10M
boolean variables x 10 unary ISTRUE factors each, 1M learnable weights, 50 % evidence;
one STEP = one learning sweep (sample_sgd: two chains + weight SGD) followed by one
inference sweep (sample) over all variables of the rank's block, i.e. 2 V variables
sampled per step, counted as the reference's own `vars/sec` print counts them
(src/dimmwitted.cc:152,232).  With --gpus N > 1 the graph is sharded by variable block
(config 5: 12.5M variables per GPU, weights global); each learning sweep all-reduces
the int64 gradient vector over RCCL.  Inputs are resident in HBM before timing starts.

Prints ONE JSON line (rank 0).  See DESIGN.md §6 for every field.
"""
import argparse
import json
import os
import re
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
B_INFER, B_LEARN = 340.0, 404.0  # algorithmic bytes per variable, SURVEY.md §8(d) cfg 3


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(n_vars, stepsize, decay, reg):
    """Time the CPU side on a bounded sample of the same workload: the same generator at
    n_vars variables, 3 learning + 10 inference epochs on all host cores.  Uses the real
    reference binary (oracle/_ref/dw) when it travelled with the repo, else the oracle's
    threaded restatement."""
    from oracle import binding as orc
    from sampler_amd import binary_format, synthetic
    raw = synthetic.cfg3(n_vars, n_weights=max(1, n_vars // 10), seed=1234)
    cores = os.cpu_count() or 1
    sample = ("config-3 generator at V=%d (10 unary ISTRUE factors/var, W=V/10, 50%% evidence); "
              "3 learning + 10 inference epochs, all host cores; value = 2V / (median learn "
              "epoch + median inference epoch)" % n_vars)
    if orc.have_reference():
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
            binary_format.write_graph(raw, d)
            out = orc.run_reference_dw(d, ["-l", "3", "-i", "10", "--alpha", str(stepsize),
                                           "--diminish", str(decay), "--reg_param", str(reg)],
                                       d, quiet=False)
        tl = [float(x) for x in re.findall(r"LEARNING EPOCH[^\n]*?\.\.\.\.([0-9.eE+-]+) sec\.", out)]
        ti = [float(x) for x in re.findall(r"INFERENCE EPOCH[^\n]*?\.\.\.\.([0-9.eE+-]+) sec\.", out)]
        if tl and ti:
            t = statistics.median(tl) + statistics.median(ti)
            return {"value": 2.0 * n_vars / t, "unit": "variables/s", "cores": cores,
                    "kind": "reference", "sample": sample,
                    "learn_vars_per_sec": n_vars / statistics.median(tl),
                    "infer_vars_per_sec": n_vars / statistics.median(ti)}
    o = orc.Oracle(raw, reg_param=reg)
    o.set_workers(cores)
    tl, ti = [], []
    cur = stepsize
    for _ in range(3):
        t0 = time.perf_counter(); o.sample_sgd(cur, threaded=True); tl.append(time.perf_counter() - t0)
        cur *= decay
    for _ in range(10):
        t0 = time.perf_counter(); o.sample(threaded=True); ti.append(time.perf_counter() - t0)
    t = statistics.median(tl) + statistics.median(ti)
    return {"value": 2.0 * n_vars / t, "unit": "variables/s", "cores": cores, "kind": "port",
            "sample": sample, "learn_vars_per_sec": n_vars / statistics.median(tl),
            "infer_vars_per_sec": n_vars / statistics.median(ti)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vars-per-gpu", type=int, default=0,
                    help="override the per-GPU variable count (default 10M at 1 GPU, 12.5M else)")
    ap.add_argument("--cpu-sample-vars", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repeat-infer", action="store_true",
                    help="skip the untimed repeated-inference leg (profiling runs: keeps the kernel stats clean)")
    ap.add_argument("--weights", type=int, default=0, help="override the weight count (experiments)")
    ap.add_argument("--tile-vars", type=int, default=0, help="graph-compile knob (experiments)")
    ap.add_argument("--tile-edges", type=int, default=0, help="graph-compile knob (experiments)")
    ap.add_argument("--wide-records", action="store_true",
                    help="graph-compile knob (experiments): 16-byte records even though the graph is all-unary")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line.  Libraries that print to the process's stdout
    # (RCCL's version banner: NCCL_DEBUG=VERSION is exported on the GPU boxes) are sent to
    # stderr: descriptor 1 becomes a copy of descriptor 2, the result goes to the saved one.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from sampler_amd import dwx, synthetic
    from sampler_amd.dist import HipEngine, ShardedDimmWitted

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench.py: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    n_gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the dwx sampler has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # DWX_BENCH_FORCE_DIST=1 runs the multi-GPU code path (process group, RCCL
    # all-reduces, barriers) even at world size 1 -- a self-check for single-GPU boxes
    use_dist = world > 1 or os.environ.get("DWX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    V = args.vars_per_gpu or (10_000_000 if n_gpus == 1 else 12_500_000)
    W = args.weights or (1_000_000 if V >= 1_000_000 else max(1, V // 10))
    stepsize, decay, reg = 0.001, 0.95, 0.01       # SURVEY.md §8(d) config-3 run flags
    t0 = time.time()
    raw = synthetic.cfg3(V, n_weights=W, seed=1234, shard=rank)
    graph = dwx.Graph(raw, tile_vars=args.tile_vars, tile_edges=args.tile_edges,
                      no_compact_records=1 if args.wide_records else 0)
    sampler = dwx.GibbsSampler(graph, device=local_rank, reg_param=reg, seed=20260103,
                               var_id_offset=rank * V)
    if rank == 0:
        log("setup: V/GPU=%d W=%d tiles=%d colours=%d device_bytes=%.2f GB (%.1f s)"
            % (V, W, graph.info.num_tiles, graph.info.num_colors,
               graph.info.device_bytes / 1e9, time.time() - t0))
    del raw
    engine = HipEngine(sampler)
    drv = ShardedDimmWitted(engine, 0, 0, stepsize, decay)
    if use_dist and not drv.distributed:      # forced self-check at world size 1
        drv.distributed = True
        engine.allreduce_static_counts()
    if os.environ.get("DWX_BENCH_PLAN_WORLD"):
        # experiment: one rank of an N-GPU run (its mini-batch plan and collectives, minus the
        # xGMI time): plan as if N equal shards contributed to every weight
        drv.plan_world = int(os.environ["DWX_BENCH_PLAN_WORLD"])

    drv.prepare(stepsize)         # one-off planning work, whatever --warmup says

    def step(cur):
        drv.learn_epoch(cur)      # sample_sgd (+ RCCL all-reduce of the gradient vector)
        engine.sample()           # sample

    def fence():
        engine.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    cur = stepsize
    for _ in range(args.warmup):
        step(cur); cur *= decay
    fence()
    sampler.kernel_time_reset(True)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(cur); cur *= decay
    fence()
    elapsed = time.perf_counter() - t_start
    ms_i, nl_i, ns_i = sampler.kernel_time("infer")
    ms_l, nl_l, ns_l = sampler.kernel_time("learn")
    ms_p, nl_p, ns_p = sampler.kernel_time("pull")
    # outside the timed region: inference sweeps that FOLLOW EACH OTHER on unchanged weights
    # (what `dw gibbs -i N` runs) stream tabulated potential terms instead of gathering
    # weights; the step above alternates learning and inference and never gets there
    ms_r = nl_r = ns_r = 0
    if not args.no_repeat_infer:
        sampler.kernel_time_reset(True)
        for _ in range(6):
            engine.sample()
        engine.wait()
        sampler.kernel_time_reset(True)
        for _ in range(10):
            engine.sample()
        engine.wait()
        ms_r, nl_r, ns_r = sampler.kernel_time("infer")
    sampler.kernel_time_reset(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    w = sampler.weights
    assert np.isfinite(w).all()
    # mini-batch plan at the LARGEST step of the run: 1 = un-split sweeps
    plan_batches, _, plan_eta = sampler.sgd_plan(stepsize * n_gpus)

    if rank == 0:
        total_vars = 2.0 * V * n_gpus * args.steps
        # units one launch processes: a learning launch visits every variable of the
        # block (both chains); an inference launch visits the query variables only
        # (evidence is skipped, as in the reference, but still counted in vars/sec)
        Vq = int(graph.info.num_query_variables)
        # config 3 is all-unary: its sweeps run sweep8_kernel (8-byte records) unless --wide-records
        kern = "sweep_kernel" if args.wide_records else "sweep8_kernel"
        if ms_l >= ms_i:
            kname, per_launch_ms, bpv, units = kern + "<LEARN=true>", ms_l / max(nl_l, 1), B_LEARN, V
        else:
            kname, per_launch_ms, bpv, units = kern + "<LEARN=false>", ms_i / max(nl_i, 1), B_INFER, Vq
        achieved = bpv * units / (per_launch_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(kname)
            except Exception:
                traffic = None
        out = {
            "metric": "variables sampled/sec (whole node)",
            "value": total_vars / elapsed,
            "unit": "variables/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "cfg3: %d boolean vars/GPU x 10 unary ISTRUE factors, W=%d learnable "
                            "weights, 50%% evidence; step = 1 learning sweep (sample_sgd) + 1 "
                            "inference sweep (sample)" % (V, W),
                "vars_per_gpu": V, "factors_per_var": 10, "weights": W,
                "parallelism": "variable-block shards x%d, int64 gradient all-reduce per "
                               "learning sweep" % n_gpus if n_gpus > 1 else "single GPU",
                "stepsize": stepsize, "diminish": decay, "reg_param": reg,
                "sgd_batches_per_sweep": plan_batches, "effective_stepsize": plan_eta / n_gpus,
            },
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_var": bpv,
                         "vars_per_launch": units, "avg_launch_ms": per_launch_ms},
            "infer_roofline_frac": (B_INFER * Vq / (ms_i / max(nl_i, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS) if nl_i else None,
            "learn_roofline_frac": (B_LEARN * V / (ms_l / max(nl_l, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS) if nl_l else None,
            "sampled_vars_per_sec": (V + Vq) * n_gpus * args.steps / elapsed,
            "infer_vars_per_sec": V * n_gpus / (ms_i / max(ns_i, 1) * 1e-3) if ns_i else None,
            "learn_vars_per_sec": V * n_gpus / (ms_l / max(ns_l, 1) * 1e-3) if ns_l else None,
            "infer_kernel_ms": ms_i / max(nl_i, 1), "learn_kernel_ms": ms_l / max(nl_l, 1),
            "pull_grad_kernel_ms": (ms_p / nl_p) if nl_p else None,
            "infer_repeat_kernel_ms": (ms_r / nl_r) if nl_r else None,
            "infer_repeat_vars_per_sec": V * n_gpus / (ms_r / max(ns_r, 1) * 1e-3) if ns_r else None,
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_vars, stepsize, decay, reg)
            except Exception as e:  # the baseline must never lose the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "variables/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: %r" % (e,)}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    sampler.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
