"""Shared parity driver: run the dwx sampler (HIP library on a GPU, or the test-only
host emulation of the same kernel source) and the CPU oracle in schedule mode on the
same graph, seed and sweeps, and compare state exactly."""
import os
import subprocess

import numpy as np

from oracle import binding as orc
from sampler_amd import dwx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "hipemu")


def emu_library(asan=False):
    subprocess.run(["make", "-s", "-j4", "-C", EMU_DIR], check=True)
    name = "libdwx_emu_asan.so" if asan else "libdwx_emu.so"
    return dwx.Library(os.path.join(EMU_DIR, "build", name))


def gpu_library():
    return dwx.default_library()


def check_index_parity(graph: dwx.Graph, oracle: orc.Oracle):
    """construct_index parity: same value rows, same factor lists per row."""
    base, sparse = graph.values()
    assert np.array_equal(base, oracle.var_val_base)
    assert np.array_equal(sparse, oracle.value_sparse)
    ib, il, fi = graph.index()
    ob, ol, of = oracle.value_index_base, oracle.value_index_len, oracle.factor_index
    assert np.array_equal(il, ol)
    assert len(fi) == len(of)
    for r in range(len(il)):
        n = int(il[r])
        if n:
            assert np.array_equal(fi[int(ib[r]):int(ib[r]) + n], of[int(ob[r]):int(ob[r]) + n])


def learn_sweep_both(s, o, order, seed, sweep, stepsize):
    """One learning sweep on the device and, mirrored, on the oracle: the device cuts a
    sweep into mini-batches (dwx_sgd_plan); the oracle follows the same chunk boundaries --
    accumulate over each chunk's variables, apply where the device applies."""
    batches, n_chunks, _ = s.sgd_plan(stepsize)
    chunk_off = s.sgd_chunks(n_chunks)
    # a split plan without per-chunk tables (more chunks than the table limit) applies every
    # chunk with the curvature bounds of the whole sweep
    hess = None
    if batches > 1 and s.device_buffer(dwx.BUF_TSTATIC_PLAN)[1] == 0:
        hess = o.sched_curvature(order)
    s.sample_sgd(stepsize); s.wait()
    for c in range(n_chunks):
        sl = order[int(chunk_off[c, 0]):int(chunk_off[c, 1])]
        o.sched_accumulate(sl, np.array([0, len(sl)], np.uint64), seed, sweep)
        if batches > 1 or c + 1 == n_chunks:
            o.sched_apply(stepsize, hess)
    return batches


def run_parity(lib, raw, n_learn=3, n_infer=5, stepsize=0.05, decay=0.9, seed=77,
               sample_evidence=False, learn_non_evidence=False, noise_aware=False,
               regularization="l2", reg_param=0.01, step_cap=1.0, compile_opts=None, plan_layouts=0,
               check_index=True, wtol=1e-12):
    """Returns (sampler, oracle) after asserting exact parity of assignments and tallies
    after every sweep and weights within wtol."""
    kw = dict(sample_evidence=sample_evidence, learn_non_evidence=learn_non_evidence,
              noise_aware=noise_aware, regularization=regularization, reg_param=reg_param)
    g = dwx.Graph(raw, lib=lib, **(compile_opts or {}))
    o = orc.Oracle(raw, **kw)
    o.set_fixed_point_mask(g.fixed_point_mask())   # (boolean variables of compact-record graphs)
    if check_index:
        check_index_parity(g, o)
    order, off = g.schedule()
    assert sorted(order.tolist()) == list(range(raw.num_variables - raw.num_ghost_variables))
    assert o.sched_check_independent(order, off), "a launch is not an independent set"
    s = dwx.GibbsSampler(g, seed=seed, step_cap=step_cap, plan_layouts=plan_layouts, **kw)
    assert np.array_equal(s.assignments("free"), o.assignments("free"))
    assert np.array_equal(s.assignments("evid"), o.assignments("evid"))
    sweep = 0
    cur = stepsize
    for _ in range(n_learn):
        learn_sweep_both(s, o, order, seed, sweep, cur)
        sweep += 1
        cur *= decay
        assert np.array_equal(s.assignments("free"), o.assignments("free")), "free chain differs"
        assert np.array_equal(s.assignments("evid"), o.assignments("evid")), "evid chain differs"
        np.testing.assert_allclose(s.weights, o.weights, rtol=wtol, atol=wtol)
    s.clear_tallies(); o.clear_tallies()
    for _ in range(n_infer):
        s.sample(); s.wait()
        o.sched_sample(order, off, seed, sweep)
        sweep += 1
        assert np.array_equal(s.assignments("evid"), o.assignments("evid")), "inference chain differs"
    t, n = s.tallies()
    assert np.array_equal(t, o.tallies[:len(t)]), "tallies differ"
    assert np.array_equal(n, o.nsamples), "nsamples differ"
    assert s.sweep == sweep
    return s, o
