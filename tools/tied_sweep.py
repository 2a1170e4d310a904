#!/usr/bin/env python3
"""Learning / inference sweep time against the number of weights (how heavily they are tied):
config 3 (all unary), 3b (pairwise), 3c (ternary) shapes with W from 10^2 to V/10."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sampler_amd import dwx, synthetic


def run(name, raw, W, stepsize, n=6):
    g = dwx.Graph(raw)
    s = dwx.GibbsSampler(g, seed=7, reg_param=0.01)
    for _ in range(2):
        s.sample_sgd(stepsize); s.sample()
    s.wait()
    out = {"shape": name, "V": raw.num_variables, "W": W}
    for kind in ("learn", "infer"):
        t0 = time.perf_counter()
        for _ in range(n):
            s.sample_sgd(stepsize) if kind == "learn" else s.sample()
        s.wait()
        out[kind + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 3)
    # device time of the learning sweep kernels and of the pull-gradient kernels (events)
    s.kernel_time_reset(True)
    for _ in range(n):
        s.sample_sgd(stepsize)
    s.wait()
    out["learn_kernel_ms"] = round(s.kernel_time("learn")[0] / n, 3)
    out["pull_kernel_ms"] = round(s.kernel_time("pull")[0] / n, 3)
    s.kernel_time_reset(False)
    b, c, eta = s.sgd_plan(stepsize)
    out.update(batches=b, chunks=c)
    print(json.dumps(out), flush=True)
    s.close(); g.close() if hasattr(g, "close") else None


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=4_000_000)
    ap.add_argument("--stepsize", type=float, default=1e-5)
    ap.add_argument("--shapes", default="cfg3,cfg3b,cfg3c")
    ap.add_argument("--weights", default="100,1000,2000,10000,100000,400000")
    a = ap.parse_args()
    for shape in a.shapes.split(","):
        for W in [int(x) for x in a.weights.split(",")]:
            if shape == "cfg4b":   # W = 0: the 2 * card tied weights
                run(shape, synthetic.cfg4b(a.V, n_weights=W or None), W or 16, a.stepsize)
            else:
                run(shape, getattr(synthetic, shape)(a.V, n_weights=W), W, a.stepsize)
