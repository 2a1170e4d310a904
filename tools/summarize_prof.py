#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel time from the kernel
trace, per-kernel counter averages from each PMC pass."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    if "sorted_sweep_kernel<true" in name or "sorted_sweep_kernelILb1" in name:
        return "sorted_sweep_kernel<LEARN=true>"
    if "sorted_sweep_kernel<false" in name or "sorted_sweep_kernelILb0" in name:
        return "sorted_sweep_kernel<LEARN=false>"
    import re
    if re.search(r"sweep8_kernel<false, \d+, false, \d+, true>", name) or re.search(r"sweep8_kernelILb0ELi\d+ELb0ELi\d+ELb1E", name):
        return "sweep8_kernel<INFER,MULTI>"
    if "sweep8_kernel<true" in name or "sweep8_kernelILb1" in name:
        return "sweep8_kernel<LEARN=true>"
    if "sweep8_kernel<false" in name or "sweep8_kernelILb0" in name:
        return "sweep8_kernel<LEARN=false>"
    if "pull_grad_kernel" in name:
        return "pull_grad_kernel"
    if "pull_ell_kernel" in name:
        return "pull_ell_kernel"
    if "fold_partials_kernel" in name:
        return "fold_partials_kernel"
    if "sweep_kernel<true" in name or "sweep_kernelILb1" in name:
        return "sweep_kernel<LEARN=true>"
    if "sweep_kernel<false" in name or "sweep_kernelILb0" in name:
        return "sweep_kernel<LEARN=false>"
    for k in ("apply_kernel", "giant_pot_kernel", "giant_decide_kernel", "giant_grad_kernel", "giant_kernel",
              "wide_kernel", "build_terms8_kernel", "build_terms_kernel", "refresh_w32_kernel"):
        if k in name:
            return k
    return name[:60]


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-28s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (
            short(row.get("Name", "")), row.get("Calls"), row.get("TotalDurationNs"),
            row.get("AverageNs"), row.get("Percentage")))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = defaultdict(list)
    regs = {}
    for row in csv.DictReader(open(f)):
        n = short(row["Kernel_Name"])
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        regs[n] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                   row.get("Grid_Size"), row.get("Workgroup_Size"))
    print("== kernel trace durations (ns): name, n, min, median, mean | vgpr sgpr lds grid wg")
    for n, v in d.items():
        v.sort()
        print("  %-28s %5d %10d %10d %10.0f | %s" % (n, len(v), v[0], v[len(v) // 2], sum(v) / len(v), regs[n]))
for pdir in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(pdir):
        continue
    for f in glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== %s" % os.path.basename(pdir))
        for k, cs in acc.items():
            print("  %-28s " % k + "  ".join("%s=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in cs.items()))

# HBM traffic per launch for bench.py's roofline.traffic: the guide's gfx950 correction
# (FETCH_SIZE tallies 128-B requests at 64 B -> double it; WRITE_SIZE is exact), KB -> B
import json
fetch, write = {}, {}
for pdir, store in (("pmc_FETCH_SIZE", fetch), ("pmc_WRITE_SIZE", write)):
    for f in glob.glob(os.path.join(out, pdir, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            store[k] = sum(v) / len(v)
traffic = {k: (2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0 for k in fetch if "sweep" in k}
if traffic:
    # stamped with the sha of the kernel sources it was measured on: bench.py reports `traffic:
    # null` when the tree has moved on (tools/update_traffic.py merges it into profiles/traffic.json)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = ("sweep_kernels.h", "tile_walk.h", "aux_kernels.h", "persist_kernels.h", "device_types.h")
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(root, "sampler_amd", "csrc", f), "rb").read())
    traffic["_stamp"] = {"kernel_sources_sha16": h.hexdigest()[:16], "files": list(files)}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1, sort_keys=True)
    del traffic["_stamp"]
    print("== traffic.json (bytes per launch, 2*FETCH_SIZE + WRITE_SIZE)")
    for k, v in traffic.items():
        print("  %-28s %.4g" % (k, v))
