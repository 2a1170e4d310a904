#!/usr/bin/env python3
"""Throughput of every BASELINE.json single-GPU config (2, 3, 3b, 4) -- per-kernel device
time from HIP events, variables/s as the reference counts them.  Not the driver's
bench (that is bench.py); used to steer optimisation and for DESIGN.md tables."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sampler_amd import dwx, synthetic  # noqa: E402


COMPILE = {}


def run(name, raw, n_learn, n_infer, stepsize=0.001, **kw):
    t0 = time.time()
    g = dwx.Graph(raw, **COMPILE)
    s = dwx.GibbsSampler(g, seed=7, **kw)
    t_setup = time.time() - t0
    for _ in range(2):
        if n_learn:
            s.sample_sgd(stepsize)
        s.sample()
    s.wait()
    s.kernel_time_reset(True)
    for _ in range(n_learn):
        s.sample_sgd(stepsize)
    for _ in range(n_infer):
        s.sample()
    s.wait()
    ms_i, nl_i, ns_i = s.kernel_time("infer")
    ms_l, nl_l, ns_l = s.kernel_time("learn")
    ms_p, nl_p, ns_p = s.kernel_time("pull")
    # (per SWEEP: all its colour launches and mini-batches; the figure includes the pull-gradient
    # kernels, not apply_kernel)
    ns_l = n_learn
    ms_l = ms_l + ms_p
    # wall time per sweep with the event timing off: what `dw gibbs` sees
    s.kernel_time_reset(False)
    wall = {}
    for kind, n in (("learn", n_learn), ("infer", n_infer)):
        if not n:
            continue
        for _ in range(2):
            s.sample_sgd(stepsize) if kind == "learn" else s.sample()
        s.wait()
        t1 = time.perf_counter()
        for _ in range(n):
            s.sample_sgd(stepsize) if kind == "learn" else s.sample()
        s.wait()
        wall[kind] = (time.perf_counter() - t1) / n * 1e3
    # n inference sweeps in ONE call (dwx_sample_n_async: one launch on an all-unary graph,
    # n queued sweeps on any other), device time per sweep and wall time per sweep
    multi = {}
    for n in (10, 100, 1000):
        s.sample_n(n); s.wait()
        s.kernel_time_reset(True)
        s.sample_n(n); s.wait()
        ms_m, nl_m, ns_m = s.kernel_time("infer")
        s.kernel_time_reset(False)
        t1 = time.perf_counter()
        s.sample_n(n); s.wait()
        multi[str(n)] = {"ms_per_sweep": ms_m / n, "wall_ms_per_sweep": (time.perf_counter() - t1) / n * 1e3,
                         "launches": int(nl_m)}
        if ms_m > 2000.0:
            break
    batches, n_chunks, eta = s.sgd_plan(stepsize) if n_learn else (None, None, None)
    V = raw.num_variables
    out = {"config": name, "V": V, "colors": int(g.info.num_colors), "tiles": int(g.info.num_tiles),
           "giant_tiles": int(g.info.num_giant_tiles), "records": int(g.info.num_index_entries),
           "setup_s": round(t_setup, 2),
           "infer_ms_per_sweep": ms_i / max(ns_i, 1), "infer_vars_per_s": V / (ms_i / max(ns_i, 1) * 1e-3) if ns_i else None,
           "learn_ms_per_sweep": ms_l / max(ns_l, 1) if ns_l else None,
           "learn_vars_per_s": V / (ms_l / ns_l * 1e-3) if ns_l else None,
           "learn_wall_ms_per_sweep": wall.get("learn"), "infer_wall_ms_per_sweep": wall.get("infer"),
           "infer_n_in_one_call": multi,
           "sgd_batches": batches, "sgd_chunks": n_chunks, "min_weight_stepsize": eta}
    print(json.dumps(out), flush=True)
    s.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--only", default="")
    ap.add_argument("--tile-vars", type=int, default=0)
    ap.add_argument("--tile-edges", type=int, default=0)
    ap.add_argument("--tile-rows", type=int, default=0)
    a = ap.parse_args()
    sc = a.scale
    COMPILE.update(tile_vars=a.tile_vars, tile_edges=a.tile_edges, tile_rows=a.tile_rows)
    todo = a.only.split(",") if a.only else ["cfg2", "cfg3", "cfg3b", "cfg4", "cfg4learn"]
    if "cfg2" in todo:
        run("cfg2", synthetic.cfg2(int(1_000_000 * sc)), 0, 20)
    if "cfg3" in todo:
        run("cfg3", synthetic.cfg3(int(10_000_000 * sc), n_weights=int(1_000_000 * sc)), 10, 20, reg_param=0.01)
    if "cfg3b" in todo:
        run("cfg3b", synthetic.cfg3b(int(10_000_000 * sc), n_weights=int(1_000_000 * sc)), 10, 20, reg_param=0.01)
    if "cfg3c" in todo:   # (not a BASELINE config: ternary factors, the generic path)
        run("cfg3c", synthetic.cfg3c(int(10_000_000 * sc), n_weights=int(1_000_000 * sc)), 5, 10, reg_param=0.01)
    if "cfg4" in todo:
        run("cfg4", synthetic.cfg4(int(5_000_000 * sc), card=8, learn=False), 0, 20)
    if "cfg4learn" in todo:
        run("cfg4learn", synthetic.cfg4(int(5_000_000 * sc), card=8, learn=True), 10, 10, stepsize=0.001)
    if "cfg4b" in todo:   # (not a BASELINE config: categorical chain, pairwise agreement factors)
        run("cfg4b", synthetic.cfg4b(int(2_000_000 * sc), card=8, n_weights=int(200_000 * sc)), 5, 10, stepsize=0.001)
