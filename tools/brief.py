#!/usr/bin/env python3
"""Print the interesting numbers of bench.py JSON lines (stdin or files)."""
import json, sys
for f in sys.argv[1:] or ["/dev/stdin"]:
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        print("%-28s value %.3e  step %.3f ms  infer %.3f ms (%.2f)  learn %.3f ms (%.2f)  pull %.3f ms" % (
            f.split("/")[-1], d["value"], d["ms_per_step"], d["infer_kernel_ms"],
            d.get("infer_roofline_frac") or 0, d["learn_kernel_ms"], d.get("learn_roofline_frac") or 0,
            d.get("pull_grad_kernel_ms") or 0))
