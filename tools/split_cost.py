import sys, time, numpy as np
sys.path.insert(0, '.')
from sampler_amd import dwx, synthetic
V = 10_000_000
raw = synthetic.cfg3(V, n_weights=1_000_000, seed=1234)
g = dwx.Graph(raw); s = dwx.GibbsSampler(g, seed=1, reg_param=0.01)
del raw
for step in (0.001, 0.004, 0.01, 0.03):
    b, n, eta = s.sgd_plan(step)
    for _ in range(2): s.sample_sgd(step)
    s.wait(); t0 = time.perf_counter()
    for _ in range(10): s.sample_sgd(step)
    s.wait(); dt = (time.perf_counter() - t0) / 10
    print("stepsize %.3f: batches %d chunks %d eta %.4g  %.3f ms/sweep" % (step, b, n, eta, dt * 1e3), flush=True)
