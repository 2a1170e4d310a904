#!/usr/bin/env python3
"""End-to-end wall time of `dw gibbs` (load + index + learn + infer + dump) on a config-3
graph written to disk: this build's drop-in binary vs the real reference binary
(oracle/_ref/dw) on the same files and flags.  Run on the GPU box."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sampler_amd import binary_format, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=2_000_000)
ap.add_argument("--learn", type=int, default=10)
ap.add_argument("--infer", type=int, default=100)
ap.add_argument("--skip-ref", action="store_true")
a = ap.parse_args()

with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
    t0 = time.time()
    raw = synthetic.cfg3(a.vars, n_weights=max(1, a.vars // 10), seed=1234)
    binary_format.write_graph(raw, d)
    del raw
    t_write = time.time() - t0
    size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    flags = ["-l", str(a.learn), "-i", str(a.infer), "--alpha", "0.001", "--diminish", "0.95",
             "--reg_param", "0.01", "--quiet"]
    files = ["-m", d + "/graph.meta", "-v", d + "/graph.variables", "-w", d + "/graph.weights",
             "-f", d + "/graph.factors"]
    res = {"vars": a.vars, "bytes_on_disk": size, "write_s": round(t_write, 1)}
    for name, exe in (("dwx", os.path.join(ROOT, "sampler_amd", "csrc", "dw")),
                      ("reference", os.path.join(ROOT, "oracle", "_ref", "dw"))):
        if name == "reference" and (a.skip_ref or not os.path.exists(exe)):
            continue
        out = os.path.join(d, "out_" + name)
        os.makedirs(out)
        t0 = time.time()
        r = subprocess.run([exe, "gibbs"] + files + ["-o", out] + flags, capture_output=True, text=True)
        t1 = time.time()
        res[name + "_wall_s"] = round(t1 - t0, 2)
        res[name + "_rc"] = r.returncode
        for l in r.stderr.splitlines():   # DWX_TIMING=1: what lies before the first and after the last phase
            if l.startswith("[dw timing] epoch at start:"):
                res[name + "_startup_s"] = round(float(l.split(":")[1]) - t0, 2)
            if l.startswith("[dw timing] epoch at exit:"):
                res[name + "_process_exit_s"] = round(t1 - float(l.split(":")[1]), 2)
        phases = [l for l in r.stderr.splitlines() if l.startswith(("[dw", "[devb"))]   # DWX_TIMING=1
        if phases:
            res[name + "_phases"] = phases
        for line in r.stdout.splitlines():
            if line.startswith("TOTAL"):
                res[name + "_" + line.split(":")[0].replace(" ", "_").lower()] = line.split(":")[1].strip()
        res[name + "_marginal_lines"] = sum(1 for _ in open(os.path.join(out, "inference_result.out.text")))
    print(json.dumps(res))
