"""n inference sweeps in one launch (dwx_sample_n_async) on configs 2 / 3 / 4:
device time per sweep against the single-sweep path, draws per second.  The profiled command of
tools/profile_cfg.sh multiN (PROFILE_CMD)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sampler_amd import dwx, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg2")
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
sc = a.scale
raw = {"cfg2": lambda: synthetic.cfg2(int(1_000_000 * sc)),
       "cfg3": lambda: synthetic.cfg3(int(10_000_000 * sc), n_weights=int(1_000_000 * sc)),
       "cfg4": lambda: synthetic.cfg4(int(5_000_000 * sc), card=8, learn=False)}[a.config]()
g = dwx.Graph(raw)
s = dwx.GibbsSampler(g, seed=7)
V = raw.num_variables
sampled = int((raw.var_role < 1).sum())
for _ in range(4):
    s.sample()
s.wait()
s.kernel_time_reset(True)
for _ in range(20):
    s.sample()
s.wait()
ms1, _, ns1 = s.kernel_time("infer")
s.sample_n(a.n); s.wait()
s.kernel_time_reset(True)
t0 = time.perf_counter()
for _ in range(a.reps):
    s.sample_n(a.n)
s.wait()
wall = time.perf_counter() - t0
msn, nl, nsn = s.kernel_time("infer")
per = msn / nsn
print(json.dumps({"config": a.config, "V": V, "sampled_vars": sampled, "n_sweeps_per_launch": a.n, "launches": int(nl),
                  "single_sweep_ms": ms1 / ns1, "ms_per_sweep": per, "speedup": (ms1 / ns1) / per,
                  "wall_ms_per_sweep": wall / (a.reps * a.n) * 1e3,
                  "vars_per_s": V / (per * 1e-3), "draws_per_s": sampled / (per * 1e-3)}), flush=True)
s.close()
