#!/usr/bin/env python3
"""Every config on PHYSICAL bytes: from tools/profile_gpu.sh / profile_cfg.sh
output directories, per kernel: HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, the MI355X
guide's gfx950 correction; separate --pmc passes), average launch duration from the kernel trace
of the same command, TB/s and the fraction of the 8 TB/s peak.

    python tools/physical_table.py NAME=DIR [NAME=DIR ...] > physical.json
    python tools/physical_table.py --check-bench profiles/r04/bench_final.json [--traffic profiles/traffic.json]
        recomputes the bench line's roofline.frac from the committed PMC bytes per launch and the
        line's own avg_launch_ms (exit 1 unless they agree to 3 digits)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
PEAK = 8.0e12


def short(name):
    for k in ("sorted_sweep_kernel", "sweep8_kernel", "sweep_kernel"):
        if k + "<" in name or k + "IL" in name:
            learn = (k + "<true" in name) or (k + "ILb1" in name)
            tab = k == "sweep8_kernel" and (", true," in name.split(k)[1][:24] or "Lb1ELi" in name.split(k)[1][:30])
            return "%s<%s%s>" % (k, "LEARN" if learn else "INFER", ",TAB" if tab else "")
    for k in ("pull_ell_kernel", "pull_grad_kernel", "fold_partials_kernel", "apply_kernel", "giant_pot_kernel",
              "giant_decide_kernel", "giant_grad_kernel", "giant_kernel", "wide_kernel", "build_terms8_kernel",
              "build_terms_kernel", "refresh_w32_kernel"):
        if k in name:
            return k
    return None


def load(d):
    dur, fetch, write = defaultdict(list), defaultdict(list), defaultdict(list)
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            n = short(row["Kernel_Name"])
            if n:
                dur[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for sub, store in (("pmc_FETCH_SIZE", fetch), ("pmc_WRITE_SIZE", write)):
        for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                n = short(row["Kernel_Name"])
                if n:
                    store[n].append(float(row["Counter_Value"]))
    out = {}
    for n, v in dur.items():
        if n not in fetch:
            continue
        ms = sum(v) / len(v) / 1e6
        b = (2.0 * sum(fetch[n]) / len(fetch[n]) + (sum(write[n]) / len(write[n]) if write.get(n) else 0.0)) * 1024.0
        out[n] = {"launches": len(v), "ms": round(ms, 4), "bytes": round(b), "TB_per_s": round(b / (ms * 1e-3) / 1e12, 3),
                  "frac_of_8TBs": round(b / (ms * 1e-3) / PEAK, 3)}
    return out


def check_bench(bench_path, traffic_path):
    d = json.loads([l for l in open(bench_path).read().splitlines() if l.startswith("{")][-1])
    rf = d["roofline"]
    tj = json.load(open(traffic_path))
    wl = d["config"]["workload"].split(":")[0]
    t = tj.get(wl, {}).get(rf["kernel"])
    if t is None or rf.get("traffic") is None:
        print("no PMC figure in the line or in %s for %s / %s" % (traffic_path, wl, rf["kernel"]))
        return 1
    frac = t / (rf["avg_launch_ms"] * 1e-3) / PEAK
    ok = abs(frac - rf["frac"]) < 5e-4
    print(json.dumps({"kernel": rf["kernel"], "bytes_per_launch": t, "avg_launch_ms": rf["avg_launch_ms"],
                      "frac_recomputed": round(frac, 4), "frac_in_line": round(rf["frac"], 4), "agree": ok,
                      "stamp": tj.get("_stamp")}))
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--check-bench":
        tp = sys.argv[sys.argv.index("--traffic") + 1] if "--traffic" in sys.argv else os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
        sys.exit(check_bench(sys.argv[2], tp))
    res = {}
    for a in sys.argv[1:]:
        name, d = a.split("=", 1)
        res[name] = load(d)
    json.dump(res, sys.stdout, indent=1, sort_keys=True)
    print()
