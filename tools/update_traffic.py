#!/usr/bin/env python3
"""tools/update_traffic.py WORKLOAD=PROFILE_DIR [...] -- merge the traffic.json files that
tools/summarize_prof.py wrote (PMC bytes per launch of the sweep kernels, stamped with the sha of
the kernel sources they were measured on) into profiles/traffic.json, which bench.py reads for
`roofline.traffic` / `roofline.frac`.  All inputs must carry the SAME stamp, and it becomes the
file's stamp: bench.py reports `traffic: null` whenever that stamp differs from the tree's.

    python tools/update_traffic.py cfg3=gpurun_out/prof_r04 cfg5b=gpurun_out/prof_r04_cfg5b
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, stamp = {}, None
for a in sys.argv[1:]:
    wl, d = a.split("=", 1)
    t = json.load(open(os.path.join(d, "traffic.json")))
    st = t.pop("_stamp", None)
    if st is None:
        sys.exit("%s/traffic.json carries no stamp (written by an old summarize_prof.py)" % d)
    if stamp is not None and st != stamp:
        sys.exit("the profile directories were measured on different kernel sources: %r vs %r" % (stamp, st))
    stamp = st
    out[wl] = t
out["_stamp"] = stamp
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
