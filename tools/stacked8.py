"""tools/stacked8.py -- BASELINE config 5's 8-way variable-block split at size, on the ONE GPU of
the test box: `dw gibbs --gpus 8 --devices 0,0,0,0,0,0,0,0 --comm host` (eight rank threads,
eight samplers on one device, host-staged collectives in place of RCCL -- everything of the
8-GPU path except xGMI) against the single-rank run on the same files.

  5a (10 unary factors per variable, empty halo): with one mini-batch per sweep (`--step_cap 0`)
     the eight shards must reproduce the single-rank result files BYTE FOR BYTE (Philox counters
     use global ids, gradient sums are integers); with the default (global) mini-batch plan the
     chunk boundaries differ, so the weights are compared statistically.
  5b (6 unary + 4 pairwise EQUAL, dense halo): scan order and one-sweep-stale ghosts differ, so
     the comparison is statistical, with the criteria of
     tests/test_dw_multi.py::test_product_dw_cross_shard_graph_at_size_matches_single_gpu_statistically
     (the reference's own run-to-run spread on that mix).

Prints one JSON line; the runs' stdout/stderr (DWX_TIMING phase lines, per-epoch vars/sec of the
non-quiet 8-rank run) go to --log-dir.  Semantics held: src/dimmwitted.cc:199-216,264-265.
Run on the GPU box: python3 tools/stacked8.py --workload cfg5a --vars 100000000
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from sampler_amd import binary_format, synthetic  # noqa: E402

DW = os.path.join(ROOT, "sampler_amd", "csrc", "dw")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def column(path, k):
    return np.loadtxt(path, usecols=(k,), dtype=np.float64) if os.path.getsize(path) else np.zeros(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["cfg5a", "cfg5b"], default="cfg5a")
    ap.add_argument("--vars", type=int, default=100_000_000)
    ap.add_argument("--weights", type=int, default=1_000_000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--learn", type=int, default=10)
    ap.add_argument("--infer", type=int, default=100)
    ap.add_argument("--log-dir", default=os.path.join(ROOT, "gpurun_out", "stacked8"))
    ap.add_argument("--skip-default-plan", action="store_true")
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--dw", default=DW, help="the dw binary (tests: tests/hipemu/build/dw_emu)")
    ap.add_argument("--cfg3b-generator", action="store_true",
                    help="cfg5b: the same mix from synthetic.cfg3b (sequential PRNG, 2.5x faster than the per-shard hash generator)")
    a = ap.parse_args()
    os.makedirs(a.log_dir, exist_ok=True)
    tag = "%s_%d" % (a.workload, a.vars)
    res = {"workload": a.workload, "vars": a.vars, "weights": a.weights, "ranks": a.ranks,
           "learn_epochs": a.learn, "inference_epochs": a.infer}
    devs = ",".join(["0"] * a.ranks)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        t0 = time.time()
        if a.workload == "cfg5a":
            raw = synthetic.cfg3(a.vars, n_weights=a.weights, seed=1234)
        else:
            raw = (synthetic.cfg3b(a.vars, n_weights=a.weights, seed=1234) if a.cfg3b_generator else
                   synthetic.cfg5b(a.vars, a.weights, seed=1234))
        binary_format.write_graph(raw, d)
        res["factors"] = int(raw.num_factors)
        del raw
        res["write_s"] = round(time.time() - t0, 1)
        res["bytes_on_disk"] = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
        print("graph written: %.1f GB in %.0f s" % (res["bytes_on_disk"] / 1e9, res["write_s"]), file=sys.stderr, flush=True)
        files = ["-m", d + "/graph.meta", "-v", d + "/graph.variables", "-w", d + "/graph.weights",
                 "-f", d + "/graph.factors"]
        flags = ["-l", str(a.learn), "-i", str(a.infer), "--alpha", "0.001", "--diminish", "0.95",
                 "--reg_param", "0.01", "--seed", "9"]
        multi = ["--gpus", str(a.ranks), "--devices", devs, "--comm", "host"]
        runs = [("single_cap0", ["-q", "--step_cap", "0"]),
                ("stacked_cap0", ["-q", "--step_cap", "0"] + multi)]
        if not a.skip_default_plan:
            runs += [("single_plan", ["-q"]), ("stacked_plan", multi)]      # (the last one prints per-epoch lines)
        outs = {}
        env = dict(os.environ, DWX_TIMING="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for name, extra in runs:
            out = os.path.join(d, "out_" + name)
            os.makedirs(out)
            t0 = time.time()
            r = subprocess.run([a.dw, "gibbs"] + files + ["-o", out] + flags + extra, capture_output=True, text=True,
                               env=env, timeout=a.timeout)
            wall = time.time() - t0
            open(os.path.join(a.log_dir, "%s_%s.stdout" % (tag, name)), "w").write(r.stdout)
            open(os.path.join(a.log_dir, "%s_%s.stderr" % (tag, name)), "w").write(r.stderr)
            res[name] = {"rc": r.returncode, "wall_s": round(wall, 2)}
            for line in r.stdout.splitlines():
                if line.startswith("TOTAL"):
                    res[name][line.split(":")[0].replace(" ", "_").lower()] = line.split(":")[1].strip()
                if line.startswith("Gradient all-reduce"):
                    res[name]["gradient_allreduce"] = line.split(":", 1)[1].strip()
            print("%s: rc %d, %.1f s" % (name, r.returncode, wall), file=sys.stderr, flush=True)
            if r.returncode != 0:
                res[name]["stderr_tail"] = r.stderr[-2000:]
                print(json.dumps(res))
                sys.exit(1)
            outs[name] = (os.path.join(out, "inference_result.out.weights.text"),
                          os.path.join(out, "inference_result.out.text"))
            res[name]["sha256_weights"], res[name]["sha256_marginals"] = sha(outs[name][0]), sha(outs[name][1])
            res[name]["marginal_lines"] = sum(1 for _ in open(outs[name][1]))

        def compare(x, y):
            w1, w2 = column(outs[x][0], 1), column(outs[y][0], 1)
            p1, p2 = column(outs[x][1], 2), column(outs[y][1], 2)
            from scipy.stats import ks_2samp
            c = {"weights_mean_diff": float(abs(w1.mean() - w2.mean())),
                 "weights_max_abs_diff": float(np.abs(w1 - w2).max()),
                 "weights_corr": float(np.corrcoef(w1, w2)[0, 1]) if w1.std() > 0 and w2.std() > 0 else None,
                 "marginals_mean_diff": float(abs(p1.mean() - p2.mean())),
                 "marginals_ks_D": float(ks_2samp(p1[:5_000_000], p2[:5_000_000]).statistic),
                 "n_marginals": int(len(p1))}
            return c

        ok = True
        same = (res["single_cap0"]["sha256_weights"] == res["stacked_cap0"]["sha256_weights"] and
                res["single_cap0"]["sha256_marginals"] == res["stacked_cap0"]["sha256_marginals"])
        res["cap0_byte_identical"] = same
        if a.workload == "cfg5a":
            ok = ok and same
        else:
            res["cap0_statistical"] = c = compare("single_cap0", "stacked_cap0")
            # the criteria of tests/test_dw_multi.py (the reference's own run-to-run spread)
            ok = ok and c["weights_mean_diff"] < 0.01 and c["marginals_mean_diff"] < 0.012 and c["marginals_ks_D"] < 0.065
        if not a.skip_default_plan:
            res["plan_statistical"] = c = compare("single_plan", "stacked_plan")
            ok = ok and c["weights_mean_diff"] < 0.01 and c["marginals_mean_diff"] < 0.012 and c["marginals_ks_D"] < 0.065
        res["pass"] = bool(ok)
    print(json.dumps(res))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
