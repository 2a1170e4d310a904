"""dwx_sample_n_async (include/dwx.h): n inference sweeps in one call.  On a graph whose factors
are all unary the sweeps share ONE launch (sweep8_kernel<..., MULTI> / infer_variable_multi:
potentials summed once, n draws per variable with the uniforms of sweeps s, s + 1, ...); the
contract is that state, tallies and the sweep counter afterwards are bit for bit those of n
calls of dwx_sample_async -- checked here against the CPU oracle stepped sweep by sweep and
against the single-sweep path of the same library.  Emulated kernels on the CPU, the HIP
library under -m gpu.  The loop it replaces: /root/reference/src/dimmwitted.cc:131-156."""
import os

import numpy as np
import pytest

from oracle import binding as orc
from parity import emu_library, gpu_library
from sampler_amd import dwx, synthetic


def _both(lib, raw, n, learn=0, stepsize=0.05, seed=77, compile_opts=None, **kw):
    """n sweeps in one call (+ a second call, + a single sweep) against the oracle and against
    the same library sweeping one by one; returns the sampler that ran them in one call."""
    g = dwx.Graph(raw, lib=lib, **(compile_opts or {}))
    o = orc.Oracle(raw, **kw)
    o.set_fixed_point_mask(g.fixed_point_mask())
    order, off = g.schedule()
    many = dwx.GibbsSampler(g, seed=seed, **kw)
    one = dwx.GibbsSampler(g, seed=seed, **kw)
    sweep = 0
    for _ in range(learn):      # (new weights first: the sweeps gather them)
        from parity import learn_sweep_both
        learn_sweep_both(many, o, order, seed, sweep, stepsize)
        one.sample_sgd(stepsize); one.wait()
        sweep += 1
    many.clear_tallies(); one.clear_tallies(); o.clear_tallies()
    many.kernel_time_reset(True)
    for k in (n, 1, 3):
        many.sample_n(k); many.wait()
        for _ in range(k):
            one.sample(); one.wait()
            o.sched_sample(order, off, seed, sweep)
            sweep += 1
        assert many.sweep == sweep == one.sweep
        assert np.array_equal(many.assignments("evid"), o.assignments("evid")), "assignments differ from the oracle's"
        assert np.array_equal(many.assignments("evid"), one.assignments("evid"))
        t, ns = many.tallies()
        assert np.array_equal(t, o.tallies[:len(t)]), "tallies differ from the oracle's"
        assert np.array_equal(ns, o.nsamples)
        t1, ns1 = one.tallies()
        assert np.array_equal(t, t1) and np.array_equal(ns, ns1)
    return many


def _cases(scale):
    from randgraph import random_graph
    yield "cfg2", synthetic.cfg2(int(1500 * scale), n_weights=100, seed=3), dict(), 0
    yield "cfg3 after learning", synthetic.cfg3(int(1500 * scale), n_weights=100, seed=4), dict(), 2
    yield "cfg3 sample_evidence", synthetic.cfg3(int(900 * scale), n_weights=50, seed=5), dict(sample_evidence=True), 1
    yield "cfg4 card 8", synthetic.cfg4(int(700 * scale), card=8, seed=6, learn=False), dict(), 0
    yield "cfg4 card 5 learned", synthetic.cfg4(int(700 * scale), card=5, seed=7, learn=True), dict(), 2
    yield "cfg4 card 12 (LDS scratch draws)", synthetic.cfg4(int(300 * scale), card=12, seed=8, learn=False), dict(), 0
    # every sign class, zero and negative feature values, fixed weights, boolean + categorical owners
    raw = random_graph(21, V=int(90 * scale), F=int(400 * scale), W=9, max_arity=1, exact_fvals=True, with_domains=False)
    raw.fac_feature_value[::17] = 0.0
    yield "random all-unary", raw, dict(), 1


@pytest.fixture(scope="module")
def emu():
    return emu_library(asan=bool(os.environ.get("DWX_EMU_ASAN")))


def test_n_sweeps_in_one_launch_equal_n_sweeps_emulated(emu):
    for name, raw, kw, learn in _cases(1):
        s = _both(emu, raw, 7, learn=learn, **kw)
        _, launches, sweeps = s.kernel_time(0)
        assert sweeps == 11 and launches <= 5, (name, launches, sweeps)     # (7 + 1 + 3 sweeps in three calls)


def test_small_tiles_and_16_byte_records_emulated(emu):
    """Several tiles per workgroup (the persistent loop re-stages) -- and a graph compiled with
    16-byte records is not eligible: the call falls back to n single sweeps, same contract."""
    raw = synthetic.cfg3(700, n_weights=40, seed=9)
    _both(emu, raw, 5, learn=1, compile_opts=dict(tile_vars=9, tile_edges=48))
    _both(emu, raw, 5, learn=1, compile_opts=dict(no_compact_records=1))


def test_long_runs_are_sliced_over_idle_lanes_emulated(emu):
    """256 sweeps and more over a tile with fewer variables than lanes: four (variable, slice)
    items per variable, dealt out over all lanes (sweep8_kernel<MULTI>) -- same contract."""
    _both(emu, synthetic.cfg4(400, card=8, seed=6, learn=False), 300)                 # 192 variables per tile
    _both(emu, synthetic.cfg4(150, card=12, seed=8, learn=False), 257)                # LDS-scratch draws
    _both(emu, synthetic.cfg3(300, n_weights=40, seed=9), 263, learn=1, compile_opts=dict(tile_vars=9, tile_edges=48))
    _both(emu, synthetic.cfg3(700, n_weights=40, seed=9), 256, sample_evidence=True)  # (a last tile that is not full)


def test_graphs_with_pairwise_factors_run_their_sweeps_one_by_one_emulated(emu):
    raw = synthetic.cfg3b(600, n_weights=32, seed=5)
    s = _both(emu, raw, 4, learn=1)
    assert s.graph.info.num_colors >= 2
    _, launches, sweeps = s.kernel_time(0)
    assert sweeps == 8 and launches >= 16


@pytest.mark.gpu
def test_n_sweeps_in_one_launch_equal_n_sweeps_gpu():
    lib = gpu_library()
    for name, raw, kw, learn in _cases(40):
        _both(lib, raw, 25, learn=learn, **kw)
    # long runs: sliced over idle lanes
    _both(lib, synthetic.cfg4(20_000, card=8, seed=6, learn=False), 300)
    _both(lib, synthetic.cfg4(5_000, card=12, seed=8, learn=False), 257)
    _both(lib, synthetic.cfg3(30_000, n_weights=400, seed=9), 263, learn=1, compile_opts=dict(tile_vars=100, tile_edges=1100))


@pytest.mark.gpu
def test_one_launch_is_faster_than_n_gpu():
    """Config 2 (1 M x 10 unary), 64 sweeps: one launch against 64 (terms table and all)."""
    import time
    lib = gpu_library()
    raw = synthetic.cfg2(1_000_000, seed=1)
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, seed=5)
    for _ in range(4):
        s.sample()
    s.wait()
    t0 = time.perf_counter()
    for _ in range(64):
        s.sample()
    s.wait()
    t_single = time.perf_counter() - t0
    s.sample_n(64); s.wait()
    t0 = time.perf_counter()
    s.sample_n(64); s.wait()
    t_multi = time.perf_counter() - t0
    assert t_multi < 0.6 * t_single, (t_multi, t_single)
