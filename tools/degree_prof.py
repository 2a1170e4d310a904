#!/usr/bin/env python3
"""Sweeps over the power-law test graph (tests/randgraph.degree_graph_fast), for
`rocprofv3 --kernel-trace --stats -- python3 tools/degree_prof.py [wide_min]`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from randgraph import degree_graph_fast
from sampler_amd import dwx
wide = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
max_degree = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
n_high = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
raw = degree_graph_fast(7, W=W, max_degree=max_degree, n_high=n_high)
g = dwx.Graph(raw, wide_min_records=wide)
s = dwx.GibbsSampler(g, device=0, seed=77)
print("tiles", g.info.num_tiles, "wide", g.info.num_wide_tiles, "giant", g.info.num_giant_tiles, "colours", g.info.num_colors)
for learn in (False, True):
    for _ in range(2):
        s.sample_sgd(0.0005) if learn else s.sample()
    s.wait()
    t0 = time.perf_counter()
    for _ in range(5):
        s.sample_sgd(0.0005) if learn else s.sample()
    s.wait()
    print("learn" if learn else "infer", "%.3f ms per sweep" % ((time.perf_counter() - t0) / 5 * 1e3))
