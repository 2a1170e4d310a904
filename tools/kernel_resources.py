#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in libdwx (hipcc -Rpass-analysis).

    python tools/kernel_resources.py [extra hipcc flags...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sampler_amd", "csrc")


def main():
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
           "-ffp-contract=off", "-x", "hip", "-c", "-o", "/dev/null", "dwx_api.cc",
           "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
    out = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)],
                                  capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("dwx::", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("%-52s %5s %5s %6s %6s %8s %4s" % ("kernel", "VGPR", "AGPR", "spillV", "spillS", "LDS", "occ"))
    for r in rows:
        print("%-52s %5d %5d %6d %6d %8d %4d" % (
            r["name"][:52], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
            r.get("SGPRs Spill", -1), r.get("LDS Size [bytes/block]", -1),
            r.get("Occupancy [waves/SIMD]", -1)))


if __name__ == "__main__":
    main()
