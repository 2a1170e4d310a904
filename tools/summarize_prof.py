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
    for k in ("sweep_kernelILb0", "sweep_kernelILb1", "apply_kernel"):
        if k in name:
            return {"sweep_kernelILb0": "sweep_kernel<LEARN=false>",
                    "sweep_kernelILb1": "sweep_kernel<LEARN=true>", "apply_kernel": "apply_kernel"}[k]
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
