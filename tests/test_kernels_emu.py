"""CPU-side (no GPU) parity of the PRODUCT'S KERNEL SOURCE against the oracle: the
file sampler_amd/csrc/sweep_kernels.h is compiled for the host against the test-only
HIP emulation (tests/hipemu) and must reproduce the oracle's schedule-mode state
exactly.  The same comparisons run on the real GPU in tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, parse_dw_args
from parity import emu_library, run_parity
from sampler_amd import binary_format, dwx, synthetic


@pytest.fixture(scope="module")
def lib():
    # DWX_EMU_ASAN=1 (with libasan preloaded) selects the ASan+UBSan build; see
    # test_sanitizers.py, which re-runs this file that way in a subprocess
    return emu_library(asan=bool(os.environ.get("DWX_EMU_ASAN")))


@pytest.mark.parametrize("fx", FIXTURES)
def test_fixture_parity(lib, fx):
    d = os.path.join(GOLDEN, fx)
    o = parse_dw_args(open(os.path.join(d, "dw-args")).read())
    raw = binary_format.read_graph_dir(d)
    run_parity(lib, raw, n_learn=0 if o["l"] == 0 else 25, n_infer=25, stepsize=o["alpha"],
               decay=o["diminish"], sample_evidence=o["sample_evidence"],
               noise_aware=o["noise_aware"], reg_param=o["reg_param"])


@pytest.mark.parametrize("fx", ["biased_coin", "sparse_domains", "partial_observation"])
def test_fixture_parity_flags(lib, fx):
    raw = binary_format.read_graph_dir(os.path.join(GOLDEN, fx))
    run_parity(lib, raw, n_learn=10, n_infer=10, learn_non_evidence=True, sample_evidence=True,
               regularization="l1", reg_param=0.001)
    run_parity(lib, raw, n_learn=10, n_infer=10, noise_aware=True, step_cap=0.0)


def test_synth_cfg2(lib):
    run_parity(lib, synthetic.cfg2(1500, n_weights=100, seed=3), n_learn=0, n_infer=6)


def test_synth_cfg3(lib):
    run_parity(lib, synthetic.cfg3(1500, n_weights=100, seed=4), n_learn=4, n_infer=4)


def test_synth_cfg3b_colouring(lib):
    raw = synthetic.cfg3b(1200, n_weights=64, seed=5)
    s, _ = run_parity(lib, raw, n_learn=4, n_infer=4)
    assert s.graph.info.num_colors >= 2


@pytest.mark.parametrize("name", ["synth_cfg2", "synth_cfg3", "synth_cfg3b", "synth_cfg4"])
def test_ks_against_reference_golden_marginals_emulated(lib, name):
    """The KS comparison with the real reference's committed marginals (tests/ks_golden.py),
    on the emulated kernels: the CPU suite covers it too, not only the -m gpu leg."""
    import ks_golden
    ks_golden.check(lib, name, {})


def test_synth_cfg4(lib):
    run_parity(lib, synthetic.cfg4(700, card=8, seed=6, learn=False), n_learn=0, n_infer=5)
    run_parity(lib, synthetic.cfg4(700, card=5, seed=7, learn=True), n_learn=5, n_infer=3,
               stepsize=0.01)


def test_categorical_draws_through_the_second_tier():
    """cat_draw's second tier (the reference's logadd sequence restated in linear space: eight
    independent f64 exps instead of sixteen dependent transcendentals) must return the exact
    sequence's verdict.  libdwx_emu_tier2.so is built with DWX_DRAW_GUARD = 2: the f32 tier never
    decides, EVERY small-domain categorical draw goes through the second tier -- half a million of
    them here, against the oracle's reference sequence, including potentials spread over the logadd
    cut-off (terms dropped below exp(-18.42) of the running sum, src/common.h:118-132)."""
    import subprocess
    from parity import EMU_DIR
    subprocess.run(["make", "-s", "-C", EMU_DIR, "build/libdwx_emu_tier2.so"], check=True)
    lib2 = dwx.Library(os.path.join(EMU_DIR, "build", "libdwx_emu_tier2.so"))
    run_parity(lib2, synthetic.cfg4(20_000, card=8, seed=6, learn=False), n_learn=0, n_infer=20, check_index=False)
    raw = synthetic.cfg4(6000, card=8, seed=8, learn=False)
    raw.w_initial_value[:] = [0.0, -18.3, -18.5, -30.0, 5.0, -13.4199, 2.0, -18.42]
    run_parity(lib2, raw, n_learn=0, n_infer=10, check_index=False)
    raw.w_initial_value[:] = [-40.0, -58.41, -58.43, -40.0 - 18.42, -77.0, -40.0, -41.0, -900.0]
    run_parity(lib2, raw, n_learn=0, n_infer=10, check_index=False)
    run_parity(lib2, synthetic.cfg4(4000, card=5, seed=7, learn=True), n_learn=6, n_infer=6, stepsize=0.01)
    run_parity(lib2, synthetic.cfg4(3000, card=3, seed=9, learn=True), n_learn=4, n_infer=6, stepsize=0.02)
    for fx in ("biased_coin_with_multinomial", "sparse_multinomial2", "sparse_domains"):
        run_parity(lib2, binary_format.read_graph_dir(os.path.join(GOLDEN, fx)), n_learn=20, n_infer=40, stepsize=0.01)


def test_split_sweep_with_one_launch_per_mini_batch(lib, monkeypatch):
    """A split learning sweep of an all-unary graph with few weights (LDS gradient accumulators) runs ONE
    launch per mini-batch (sweep8_merged_kernel): the update of mini-batch c - 1 is the prologue of
    mini-batch c's sweep kernel -- every workgroup for itself, into its LDS copy of the f32 weights --
    with three gradient buffers and two weight buffers in turn.  Exact against the oracle stepped chunk by
    chunk (categorical, boolean, L2 and L1, 8 to 900 weights) and bit for bit the two-launch path
    (DWX_NO_MERGED_APPLY)."""
    cases = [(synthetic.cfg4(700, card=5, seed=7, learn=True), dict(stepsize=0.01, decay=1.0, compile_opts=dict(tile_vars=16))),
             (synthetic.cfg3(3000, n_weights=20, seed=4), dict(stepsize=0.01, decay=0.9, compile_opts=dict(tile_vars=32),
                                                              regularization="l1", reg_param=0.002)),
             (synthetic.cfg3(3000, n_weights=900, seed=4), dict(stepsize=0.3, decay=0.9, compile_opts=dict(tile_vars=32),
                                                               step_cap=0.05))]
    for raw, kw in cases:
        s, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        assert s.sgd_plan(kw["stepsize"])[0] >= 2
        assert s.kernel_time("merged")[1] == 4, s.kernel_time("merged")
        monkeypatch.setenv("DWX_NO_MERGED_APPLY", "1")
        s2, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        monkeypatch.delenv("DWX_NO_MERGED_APPLY")
        assert s2.kernel_time("merged")[1] == 0
        assert np.array_equal(s.weights, s2.weights) and np.array_equal(s.assignments("evid"), s2.assignments("evid"))


def test_split_sweep_as_one_persistent_launch(lib, monkeypatch):
    """DWX_PERSIST=1 (opt-in: measured slower than the plain launches, persist_kernels.h): a split
    learning sweep of an all-unary graph with few weights (>= 8 mini-batches) runs as ONE
    launch of persist_learn8_kernel: chunk loop, a gradient row per workgroup, grid barrier, the update
    by every workgroup on its LDS copy of the weights (here with one workgroup: the harness runs a
    grid's blocks one after another; tests/test_gpu_parity.py has the full grid) -- exact against the
    oracle stepped chunk by chunk, categorical (config 4's shape) and boolean, L2 and L1."""
    monkeypatch.setenv("DWX_PERSIST", "1")
    raw = synthetic.cfg4(700, card=5, seed=7, learn=True)
    s, _ = run_parity(lib, raw, n_learn=5, n_infer=1, stepsize=0.01, decay=1.0, compile_opts=dict(tile_vars=16))
    batches, n_chunks, _ = s.sgd_plan(0.01)
    assert batches >= 8 and n_chunks >= 8, (batches, n_chunks)
    assert s.kernel_time("persist")[1] == 5, s.kernel_time("persist")
    raw = synthetic.cfg3(3000, n_weights=20, seed=4)
    s, _ = run_parity(lib, raw, n_learn=4, n_infer=2, stepsize=0.01, decay=0.9, compile_opts=dict(tile_vars=32),
                      regularization="l1", reg_param=0.002)
    assert s.kernel_time("persist")[1] >= 3, s.kernel_time("persist")
    # timing on: the sweep is still one launch, counted as such
    s.kernel_time_reset(True)
    s.sample_sgd(0.005); s.wait()
    assert s.kernel_time("learn")[1:] == (1, 1), s.kernel_time("learn")
    s.kernel_time_reset(False)


def test_split_sweep_through_the_graph_replay_path(lib, monkeypatch):
    """DWX_GRAPH=n: dwx_sample_sgd_async captures a split sweep and replays it as one graph launch
    (the harness's "capture" runs the launches eagerly and its graph launch is a no-op: the host
    logic -- which sweeps are captured, the per-level graph, the counter -- is what runs here;
    tests/test_gpu_parity.py has the real thing)."""
    raw = synthetic.cfg4(700, card=5, seed=7, learn=True)
    monkeypatch.setenv("DWX_GRAPH", "2")
    s, _ = run_parity(lib, raw, n_learn=5, n_infer=1, stepsize=0.01, decay=1.0,
                      compile_opts=dict(tile_vars=16))
    batches, n_chunks, _ = s.sgd_plan(0.01)
    assert batches > 1 and n_chunks >= 2, (batches, n_chunks)
    assert s.kernel_time("graph")[1] == 4, s.kernel_time("graph")       # every sweep but the level's first
    s.kernel_time_reset(True)                                           # timed sweeps are never captured
    s.sample_sgd(0.01); s.wait()
    assert s.kernel_time("graph")[1] == 4
    s.kernel_time_reset(False)
    monkeypatch.delenv("DWX_GRAPH")
    s2, _ = run_parity(lib, raw, n_learn=5, n_infer=1, stepsize=0.01, decay=1.0,
                       compile_opts=dict(tile_vars=16))
    assert s2.kernel_time("graph")[1] == 0                              # off by default
    s2.sample_sgd(0.01); s2.wait()
    assert np.array_equal(s.weights, s2.weights)


@pytest.mark.parametrize("compact", [True, False])
def test_all_unary_compact_records(lib, compact):
    # all-unary graphs stream 8-byte records (EdgeRec8) by default: every sign class
    # (+f/-f, f/0, f/f), negative and zero feature values, fixed weights, boolean and
    # categorical owners -- and the same graph with 16-byte records
    from randgraph import random_graph
    for seed in (21, 22):
        raw = random_graph(seed, V=90, F=400, W=9, max_arity=1, exact_fvals=True, with_domains=False)
        raw.fac_feature_value[::17] = 0.0
        run_parity(lib, raw, n_learn=5, n_infer=5, stepsize=0.1, learn_non_evidence=seed == 22,
                   compile_opts=dict(no_compact_records=0 if compact else 1))
        run_parity(lib, raw, n_learn=4, n_infer=4, stepsize=0.1, sample_evidence=True,
                   compile_opts=dict(no_compact_records=0 if compact else 1, tile_vars=9, tile_edges=24,
                                     tile_rows=12))
    run_parity(lib, synthetic.cfg3(1500, n_weights=2000, seed=4), n_learn=4, n_infer=4,
               compile_opts=dict(no_compact_records=0 if compact else 1))


@pytest.mark.parametrize("min_w", [0, 10 ** 9])
def test_weight_sorted_super_tiles(lib, monkeypatch, min_w):
    """sorted_sweep_kernel (boolean all-unary tiles of compact-record graphs; here forced onto
    small graphs, and -- min_w = 10^9 -- switched off: the tile sweep must give the same state,
    the potential sums are fixed point either way): every sign class and mixed feature values
    (several distinct record deltas, records that add nothing), fixed weights, ragged tiles and
    super-tiles cut at the query/evidence boundary, boolean next to categorical tiles, few
    weights (learning falls back to the LDS accumulators), split sweeps (the sorted kernel serves
    whole super-tiles inside a chunk, the tile sweep the rest), sample_evidence /
    learn_non_evidence."""
    from randgraph import random_graph
    monkeypatch.setenv("DWX_SORTED_MIN_W", str(min_w))
    want = (lambda s: s.graph.info.num_super_tiles > 0) if min_w == 0 else (lambda s: s.graph.info.num_super_tiles == 0)
    for seed in (31, 32):
        raw = random_graph(seed, V=700, F=5000, W=1500, max_arity=1, exact_fvals=True, with_domains=False)
        raw.fac_feature_value[::13] = 0.0
        s, _ = run_parity(lib, raw, n_learn=4, n_infer=4, stepsize=0.05, learn_non_evidence=seed == 32,
                          compile_opts=dict(tile_vars=32, super_tiles=4))
        assert want(s)
        s, _ = run_parity(lib, raw, n_learn=3, n_infer=3, stepsize=0.05, sample_evidence=True,
                          compile_opts=dict(tile_vars=9, tile_edges=48, tile_rows=12, super_tiles=5))
        assert want(s)
    s, _ = run_parity(lib, synthetic.cfg3(3000, n_weights=3000, seed=4), n_learn=2, n_infer=2)
    assert want(s)
    # few weights: inference sorted, learning through the tile sweep's LDS accumulators
    s, _ = run_parity(lib, synthetic.cfg3(3000, n_weights=40, seed=5), n_learn=3, n_infer=3, compile_opts=dict(tile_vars=64))
    assert want(s)
    # a split sweep: chunks cut inside super-tiles
    s, _ = run_parity(lib, synthetic.cfg3(3200, n_weights=1200, seed=6), n_learn=3, n_infer=2, stepsize=0.5,
                      compile_opts=dict(tile_vars=32, super_tiles=6))
    assert want(s) and s.sgd_plan(0.5)[0] > 1
    if min_w == 0:
        # ... and with chunks big enough for super-tiles of their own (8 tiles x sorted_slots and
        # more): the plan level builds a layout cut ALONG its chunks, every chunk is one launch of
        # the sorted kernel and nothing is left to the tile sweep
        raw = synthetic.cfg3(3200, n_weights=1200, seed=6)
        s, _ = run_parity(lib, raw, n_learn=3, n_infer=2, stepsize=0.5, decay=1.0, step_cap=48.0, plan_layouts=1,
                          compile_opts=dict(tile_vars=32, super_tiles=6, sorted_slots=2))
        batches, n_chunks, _ = s.sgd_plan(0.5)
        assert batches > 1 and s.graph.info.num_tiles // n_chunks >= 16
        s.kernel_time_reset(True)
        s.sample_sgd(0.5); s.wait()
        _, launches, sweeps = s.kernel_time("learn")
        assert sweeps == 1 and launches == n_chunks, (launches, n_chunks)
        s.kernel_time_reset(False)
        # ... and so does an un-split sweep whose launch holds query and evidence tiles (one run
        # per launch instead of the default layout's two): one launch per sweep
        s, _ = run_parity(lib, raw, n_learn=3, n_infer=2, stepsize=0.01, step_cap=0.0, plan_layouts=1,
                          compile_opts=dict(tile_vars=32, super_tiles=6, sorted_slots=2))
        s.kernel_time_reset(True)
        s.sample_sgd(0.01); s.wait()
        _, launches, sweeps = s.kernel_time("learn")
        assert sweeps == 1 and launches == 1, launches
        s.kernel_time_reset(False)
        # the default policy builds the level's layout once the level has run 2048 sweeps (here: 3,
        # test hook) -- in the middle of a run, with the same results before and after the switch
        monkeypatch.setenv("DWX_LAYOUT_AFTER_SWEEPS", "3")
        s, _ = run_parity(lib, raw, n_learn=5, n_infer=1, stepsize=0.5, decay=1.0, step_cap=48.0,
                          compile_opts=dict(tile_vars=32, super_tiles=6, sorted_slots=2))
        s.kernel_time_reset(True)
        s.sample_sgd(0.5); s.wait()
        assert s.kernel_time("learn")[1] == s.sgd_plan(0.5)[1]          # (one launch per chunk: the layout is in)
        s.kernel_time_reset(False)
        monkeypatch.delenv("DWX_LAYOUT_AFTER_SWEEPS")
        # (plan_layouts = 2: never -- the default layout serves, with the tile sweep at the seams)
        s, _ = run_parity(lib, raw, n_learn=2, n_infer=1, stepsize=0.5, decay=1.0, step_cap=48.0, plan_layouts=2,
                          compile_opts=dict(tile_vars=32, super_tiles=6, sorted_slots=2))
    # categorical rows stay with the tile sweep (config 4's shape beside boolean variables)
    raw = random_graph(33, V=600, F=4000, W=1300, max_arity=1, exact_fvals=True, p_cat=0.4)
    s, _ = run_parity(lib, raw, n_learn=3, n_infer=3, compile_opts=dict(tile_vars=32, super_tiles=3))
    assert want(s)
    if min_w == 0:
        # a split plan WITHOUT per-chunk tables scatters its gradient and counts updates with
        # atomics: such sweeps stay with the tile sweep (the sorted kernel only publishes ballots),
        # inference sweeps of the same sampler take the sorted one
        monkeypatch.setenv("DWX_PLAN_TABLE_CHUNKS", "1")
        s, _ = run_parity(lib, synthetic.cfg3(3200, n_weights=1200, seed=7), n_learn=3, n_infer=2, stepsize=0.5,
                          compile_opts=dict(tile_vars=32, super_tiles=6))
        assert want(s) and s.sgd_plan(0.5)[0] > 1
        monkeypatch.delenv("DWX_PLAN_TABLE_CHUNKS")
        # more distinct record deltas than the LDS table holds (1024): no sorted copy, same results
        raw = synthetic.cfg3(3000, n_weights=1500, seed=8)
        raw.fac_feature_value[:] = 1.0 + (np.arange(raw.num_factors) % 1500) / 1024.0      # f32-exact
        s, _ = run_parity(lib, raw, n_learn=2, n_infer=2)
        assert s.graph.info.num_super_tiles == 0
        raw.fac_feature_value[:] = 1.0 + (np.arange(raw.num_factors) % 400) / 1024.0       # 400 of them: sorted
        s, _ = run_parity(lib, raw, n_learn=2, n_infer=2)
        assert s.graph.info.num_super_tiles > 0


def test_gradient_unit_reported_for_multi_gpu_drivers(lib):
    """dwx_graph_info.grad_shift / grad_unit_max / max_records_per_weight (what lets
    sampler_amd.dist send the gradient all-reduce as 32-bit counts): ISTRUE with f = 1 moves a
    potential by d = 2, i.e. 2^31 in the 2^-30 fixed point of the gradient sums; mixed feature
    values lower the common power of two; categorical variables or non-unary factors: unknown (0).
    And the sums a learning sweep leaves in DWX_BUF_GRAD are indeed multiples of it."""
    from randgraph import random_graph
    raw = synthetic.cfg3(2000, n_weights=300, seed=3)
    g = dwx.Graph(raw, lib=lib)
    assert (g.info.grad_shift, g.info.grad_unit_max) == (31, 1)
    per_w = np.bincount(raw.fac_weight_id.astype(np.int64), minlength=300)
    assert g.info.max_records_per_weight == per_w.max()
    raw.fac_feature_value[::3] = 0.75       # d = 1.5 = 3 * 2^-1  ->  q = 3 * 2^29
    g2 = dwx.Graph(raw, lib=lib)
    assert (g2.info.grad_shift, g2.info.grad_unit_max) == (29, 4)
    assert dwx.Graph(synthetic.cfg4(500, card=4, learn=True), lib=lib).info.grad_shift == 0
    assert dwx.Graph(synthetic.cfg3b(600, n_weights=50), lib=lib).info.grad_shift == 0
    s = dwx.GibbsSampler(g2, seed=5, step_cap=0.0)
    s.sgd_plan(0.01)
    s.sgd_accumulate(0); s.wait()
    ptr, nbytes = s.device_buffer(dwx.BUF_GRAD)
    grad = np.frombuffer((__import__("ctypes").c_char * nbytes).from_address(ptr), np.int64)[:300].copy()
    s.sgd_apply(); s.sgd_finish(); s.wait()
    assert np.any(grad != 0) and np.all(grad % (1 << 29) == 0)


@pytest.mark.parametrize("block_tiles, depth_hint", [(8, 1), (32, 2), (1024, 2)])
def test_block_pull(lib, monkeypatch, block_tiles, depth_hint):
    # pull_ell_kernel (un-split sweeps of graphs with many weights; here forced onto a small
    # graph): several variable blocks, one- and two-row tables, entries that overflow onto the
    # list of pull_grad_kernel, mixed deltas (2f, f, -f ...), fixed weights
    from randgraph import random_graph
    monkeypatch.setenv("DWX_BLOCK_PULL_MIN_W", "0")
    monkeypatch.setenv("DWX_BLOCK_PULL_TILES", str(block_tiles))
    run_parity(lib, synthetic.cfg3(6000, n_weights=1100, seed=14), n_learn=4, n_infer=2, step_cap=0.0,
               compile_opts=dict(tile_vars=32))
    raw = random_graph(31, V=700, F=9000, W=1500, p_cat=0.0, max_arity=1, exact_fvals=True, with_domains=False)
    run_parity(lib, raw, n_learn=4, n_infer=2, stepsize=0.02, step_cap=0.0, compile_opts=dict(tile_vars=16))
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.02, step_cap=0.0, learn_non_evidence=True,
               compile_opts=dict(tile_vars=16, no_compact_records=1))
    # mixed graph: pull-gradient tiles next to tiles with binary factors and categorical
    # variables (those scatter their gradient), several colours, an oversized variable
    raw = random_graph(8, V=1500, F=9000, W=1300, p_cat=0.3)
    run_parity(lib, raw, n_learn=4, n_infer=2, stepsize=0.05, step_cap=0.0, learn_non_evidence=True,
               compile_opts=dict(tile_vars=32))
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.05, step_cap=0.0,
               compile_opts=dict(tile_vars=8, tile_edges=40, tile_rows=12))


def test_small_tiles_and_giant_variable(lib):
    # tiny LDS budgets force ragged tiles and the direct-from-HBM path for
    # variables that do not fit one tile
    raw = synthetic.cfg3b(300, n_weights=16, seed=8)
    s, _ = run_parity(lib, raw, n_learn=3, n_infer=3,
                      compile_opts=dict(tile_vars=7, tile_edges=16, tile_rows=7))
    raw = synthetic.cfg4(60, card=9, seed=9, learn=True)
    s, _ = run_parity(lib, raw, n_learn=3, n_infer=3,
                      compile_opts=dict(tile_vars=5, tile_edges=8, tile_rows=8))
    assert s.graph.info.num_giant_tiles > 0


@pytest.mark.parametrize("seed", range(6))
def test_random_mixed_graphs(lib, seed):
    from randgraph import random_graph
    raw = random_graph(seed, truthy=bool(seed % 2))
    s, o = run_parity(lib, raw, n_learn=6, n_infer=6, stepsize=0.1,
                      noise_aware=bool(seed % 2), learn_non_evidence=seed in (2, 3),
                      sample_evidence=seed in (1, 2))
    assert not np.array_equal(s.weights, raw.w_initial_value)      # learning moved weights
    assert np.array_equal(s.weights[raw.w_is_fixed == 1], raw.w_initial_value[raw.w_is_fixed == 1])
    assert s.tallies()[0].sum() > 0


def test_learning_is_doing_something(lib):
    raw = synthetic.cfg3(1500, n_weights=100, seed=4)
    s, o = run_parity(lib, raw, n_learn=4, n_infer=4)
    assert np.abs(s.weights).max() > 1e-3
    t, n = s.tallies()
    assert t.sum() > 0 and n.max() == 4


def test_ghost_variables_parity(lib):
    """A shard's local graph: ghost variables hold assignments (set from outside, as the
    halo exchange does) but are never sampled; owned variables read them."""
    from sampler_amd.shard import make_shard
    raw = synthetic.cfg3b(600, n_weights=24, seed=21)
    local, ghosts = make_shard(raw, 150, 420)
    assert local.num_ghost_variables > 0
    s, o = run_parity(lib, local, n_learn=0, n_infer=0)
    rng = np.random.default_rng(0)
    n_owned = local.num_variables - local.num_ghost_variables
    order, off = s.graph.schedule()
    sweep = 0
    for it in range(4):
        for chain in ("free", "evid"):     # "halo exchange": new ghost values on both sides
            a = s.assignments(chain)
            a[n_owned:] = rng.integers(0, 2, local.num_ghost_variables)
            s.set_assignments(chain, a)
            o.assignments(chain)[:] = a
        if it % 2 == 0:
            from parity import learn_sweep_both
            learn_sweep_both(s, o, order, 77, sweep, 0.05)
        else:
            s.sample(); s.wait()
            o.sched_sample(order, off, 77, sweep)
        sweep += 1
        assert np.array_equal(s.assignments("free"), o.assignments("free"))
        assert np.array_equal(s.assignments("evid"), o.assignments("evid"))
        np.testing.assert_allclose(s.weights, o.weights, rtol=1e-12, atol=1e-12)
    t, n = s.tallies()
    assert np.array_equal(t, o.tallies[:len(t)]) and (n[n_owned:] == 0).all()


def test_many_weights_global_atomics_path(lib):
    # W > LDS_AGG_MAX_W (1024): gradients go straight to memory-side atomics; the small-W
    # tests above all take the per-workgroup LDS accumulators
    s, _ = run_parity(lib, synthetic.cfg3(3000, n_weights=1500, seed=12), n_learn=4, n_infer=2)
    assert s.graph.info.num_weights == 1500
    run_parity(lib, synthetic.cfg4(400, card=6, seed=13, learn=True), n_learn=4, n_infer=2, stepsize=0.01)
    # pairwise factors + many weights: mixed pull tiles / direct global atomics
    s, _ = run_parity(lib, synthetic.cfg3b(2500, n_weights=1400, seed=14), n_learn=4, n_infer=2)
    from randgraph import random_graph
    run_parity(lib, random_graph(5, V=900, F=5000, W=1300), n_learn=4, n_infer=2, stepsize=0.05,
               learn_non_evidence=True)


@pytest.mark.parametrize("seed", range(4))
def test_binary_factor_tiles_all_functions(lib, seed):
    """Boolean graphs whose factors all have arity <= 2 and f32-exact feature values: the
    inference sweep evaluates them edge-parallel in the staging pass (TILE_TERMS2); every
    factor function, duplicated variables inside a factor, duplicate factors."""
    from randgraph import random_graph
    raw = random_graph(100 + seed, V=300, F=1500, W=20, p_cat=0.0, max_arity=2, exact_fvals=True)
    run_parity(lib, raw, n_learn=3, n_infer=8, stepsize=0.05, sample_evidence=bool(seed % 2),
               learn_non_evidence=seed >= 2)


def test_high_degree_hub_variables(lib):
    """Variables whose records exceed any tile: one workgroup per variable, lane-strided
    partial sums + LDS reduction (giant_kernel)."""
    from randgraph import hub_graph
    raw = hub_graph(3)
    s, _ = run_parity(lib, raw, n_learn=4, n_infer=6, stepsize=0.001, learn_non_evidence=True)
    assert s.graph.info.num_giant_tiles == 2
    run_parity(lib, hub_graph(4, W=2000), n_learn=3, n_infer=3, stepsize=0.001, sample_evidence=True)


def test_arity3_tiles_evaluated_edge_parallel(lib):
    """TILE_TERMS3: tiles whose factors have arity <= 3 (the IMPLY-with-two-body-atoms shape of
    DeepDive rules) evaluate every record in the staging pass through the general sign functions
    on batched loads -- inference (two scenarios) and learning (four: both chains), every boolean
    function, owners appearing twice in a factor, mixed with pre-signed and arity-2 records."""
    from randgraph import random_graph
    s, _ = run_parity(lib, synthetic.cfg3c(3000, n_weights=60, seed=3), n_learn=3, n_infer=3, stepsize=0.01)
    assert s.graph.info.num_staged_tiles == s.graph.info.num_tiles
    run_parity(lib, synthetic.cfg3c(1500, n_weights=40, seed=4), n_learn=3, n_infer=2, stepsize=0.01,
               learn_non_evidence=True, sample_evidence=True)
    # the 768-record and 3072-record kernel builds (K = 3: staged like K = 6; K = 12: generic walk)
    for te in (700, 3000):
        run_parity(lib, synthetic.cfg3c(900, n_weights=40, seed=8), n_learn=2, n_infer=2, stepsize=0.01,
                   compile_opts=dict(tile_edges=te))
        run_parity(lib, synthetic.cfg4b(300, card=4, seed=8), n_learn=2, n_infer=2, stepsize=0.01,
                   compile_opts=dict(tile_edges=te))
    # (factor->variable entries once per factor instead of per record: the layout huge graphs fall back to)
    run_parity(lib, synthetic.cfg3c(1500, n_weights=40, seed=5), n_learn=2, n_infer=2, stepsize=0.01,
               compile_opts=dict(no_record_vifs=1))
    for seed in (31, 32, 33):
        raw = random_graph(seed, V=800, F=4000, W=30, p_cat=0.0, max_arity=3, exact_fvals=True)
        run_parity(lib, raw, n_learn=3, n_infer=4, stepsize=0.05, learn_non_evidence=seed == 32,
                   sample_evidence=seed == 33, compile_opts=dict(tile_vars=64))


def test_unary_records_of_mixed_tiles_take_the_pull_gradient(lib):
    """TILE_PULL_UNARY: in a graph with more than 1024 weights the pre-signed records of boolean
    TERMS tiles are pulled through the ballots like all-unary tiles (two lanes per variable: half
    a ballot word per wave), the others scatter; split sweeps, the block pull, every option, and
    the compile option that turns it off."""
    from randgraph import random_graph
    for kw in (dict(), dict(learn_non_evidence=True, sample_evidence=True), dict(noise_aware=True)):
        run_parity(lib, synthetic.cfg3b(2500, n_weights=2000, seed=3), n_learn=3, n_infer=2, stepsize=0.01, **kw)
    run_parity(lib, synthetic.cfg3c(2000, n_weights=1500, seed=4), n_learn=3, n_infer=2, stepsize=0.01)
    run_parity(lib, synthetic.cfg3b(2000, n_weights=1200, seed=5), n_learn=2, n_infer=1, stepsize=0.01,
               compile_opts=dict(no_pull_unary=1))
    for seed in (51, 52, 53):   # (about ten records per variable: tiles of more than 128 variables, one lane each)
        raw = random_graph(seed, V=900, F=4000, W=1300, p_cat=0.3 if seed == 53 else 0.0, max_arity=3, exact_fvals=True)
        run_parity(lib, raw, n_learn=3, n_infer=2, stepsize=0.05, learn_non_evidence=seed == 52)
        run_parity(lib, raw, n_learn=2, n_infer=1, stepsize=0.05, compile_opts=dict(tile_vars=64))
    # heavily tied weights cut the sweep into mini-batches: per-chunk lists; and the block pull
    run_parity(lib, synthetic.cfg3b(3000, n_weights=1100, seed=6), n_learn=3, n_infer=1, stepsize=0.5, step_cap=0.05)
    os.environ["DWX_BLOCK_PULL_MIN_W"] = "1000"
    try:
        run_parity(lib, synthetic.cfg3b(3000, n_weights=1100, seed=7), n_learn=3, n_infer=1, stepsize=0.01)
    finally:
        del os.environ["DWX_BLOCK_PULL_MIN_W"]


def test_categorical_tiles_evaluated_edge_parallel(lib):
    """Categorical tiles whose factors have arity <= 3 take TILE_TERMS3 too: a record's proposal
    is its row's value ("the owner's own predicate holds"), learning stages LearnRecs (hit / miss
    x both chains) and runs the tiles' process_variable on them (W_LREC) -- every option, mixed
    with boolean tiles, owners that sit twice in a factor kept on the generic path."""
    from randgraph import random_graph
    s, _ = run_parity(lib, synthetic.cfg4b(600, card=5), n_learn=3, n_infer=3, stepsize=0.01)
    assert s.graph.info.num_staged_tiles == s.graph.info.num_tiles
    run_parity(lib, synthetic.cfg4b(500, card=4, n_weights=2000, seed=9), n_learn=3, n_infer=2, stepsize=0.01,
               learn_non_evidence=True, sample_evidence=True)
    run_parity(lib, synthetic.cfg4b(400, card=3, learn=False), n_learn=0, n_infer=4)
    for seed in (41, 42, 43, 44):
        raw = random_graph(seed, V=600, F=3000, W=30, p_cat=0.6, max_arity=3, exact_fvals=True)
        run_parity(lib, raw, n_learn=3, n_infer=3, stepsize=0.05, learn_non_evidence=seed == 42,
                   sample_evidence=seed == 43, noise_aware=seed == 44, compile_opts=dict(tile_vars=64))


def test_halo_lists_travel_as_bits_bytes_or_words(lib):
    """dwx_halo_*: the listed variables' values are packed as 1 bit (all boolean), 8 bits
    (cardinalities <= 256) or 32 bits each, chain blocks padded to 8 bytes; a list's own
    unpack restores exactly what its pack wrote (what the peer's list of the same variables
    receives)."""
    import ctypes as C
    from randgraph import random_graph
    rng = np.random.default_rng(5)

    def roundtrip(raw, ids, bits):
        g = dwx.Graph(raw, lib=lib)
        s = dwx.GibbsSampler(g, seed=3)
        V = raw.num_variables
        card = np.asarray(raw.var_cardinality, np.int64)
        vals = {c: (rng.integers(0, 1 << 30, V) % card).astype(np.uint64) for c in ("free", "evid")}
        for c in vals:
            s.set_assignments(c, vals[c])
        h = dwx.HaloList(s, ids)
        n = len(ids)
        block = (n * bits + 63) // 64 * 8
        assert h.message_bytes(1) == block and h.message_bytes(2) == block and h.message_bytes(3) == 2 * block
        ptr, nbytes = h.buffer()
        assert nbytes == 2 * block
        for mask in (3, 1, 2):
            h.pack(mask)
            host = np.zeros(nbytes, np.uint8)
            lib.check(lib.L.dwx_buffer_copy(s.h, host.ctypes.data, ptr, nbytes, 0))
            for k, c in enumerate([c for c, b in (("free", 1), ("evid", 2)) if mask & b]):
                blk = host[k * block:(k + 1) * block]
                if bits == 1:
                    got = np.unpackbits(blk, bitorder="little")[:n]
                elif bits == 8:
                    got = blk[:n]
                else:
                    got = blk.view(np.uint32)[:n]
                assert np.array_equal(got.astype(np.uint64), vals[c][ids.astype(np.int64)]), (bits, mask, c)
            # scramble the variables, then unpack: the listed ones come back, nobody else moves
            for c in vals:
                s.set_assignments(c, (vals[c] + 1) % card.astype(np.uint64))
            h.unpack(mask)
            for c, b in (("free", 1), ("evid", 2)):
                want = (vals[c] + 1) % card.astype(np.uint64)
                if mask & b:
                    want[ids.astype(np.int64)] = vals[c][ids.astype(np.int64)]
                assert np.array_equal(s.assignments(c), want), (bits, mask, c)
                s.set_assignments(c, vals[c])
        s.close()

    b = synthetic.cfg3b(700, n_weights=20, seed=2)
    roundtrip(b, np.sort(rng.choice(700, 333, replace=False)).astype(np.uint64), 1)
    roundtrip(b, np.array([699], np.uint64), 1)
    c = synthetic.cfg4b(300, card=7)
    roundtrip(c, np.sort(rng.choice(300, 77, replace=False)).astype(np.uint64), 8)
    m = random_graph(12, V=300, F=900, W=10, p_cat=0.5, max_arity=2)
    roundtrip(m, np.arange(0, 300, 3, dtype=np.uint64), 8)
    big = synthetic.cfg4(40, card=300)
    roundtrip(big, np.arange(5, 40, dtype=np.uint64), 32)


def test_degree_bins_lane_wave_workgroup(lib):
    """Degrees from 1 to thousands in one graph: low-degree variables in lane-per-variable
    tiles, mid-degree ones walked by a wave each (TILE_WIDE, wide_kernel), the largest by a
    workgroup each (giant_kernel) -- boolean and categorical, unary and pairwise factors, both
    chains, every learning flag; and the same graph with the wave bin switched off."""
    from sampler_amd import dwx
    from randgraph import degree_graph
    raw = degree_graph(5, n_low=1500, n_high=60, max_degree=6000, W=120)
    g = dwx.Graph(raw, lib=lib)
    assert g.info.num_wide_tiles >= 10 and g.info.num_giant_tiles >= 2
    run_parity(lib, raw, n_learn=3, n_infer=3, stepsize=0.002)
    run_parity(lib, raw, n_learn=2, n_infer=2, stepsize=0.002, learn_non_evidence=True, sample_evidence=True)
    run_parity(lib, raw, n_learn=2, n_infer=1, stepsize=0.002, noise_aware=True)
    g_off = dwx.Graph(raw, lib=lib, wide_min_records=0xFFFFFFFF)
    assert g_off.info.num_wide_tiles == 0 and g_off.info.num_tiles < g.info.num_tiles
    run_parity(lib, raw, n_learn=2, n_infer=2, stepsize=0.002, compile_opts=dict(wide_min_records=0xFFFFFFFF))
    # a lower threshold moves more variables into the bin; an all-unary graph keeps its compact
    # record stream next to wide tiles
    run_parity(lib, raw, n_learn=2, n_infer=2, stepsize=0.002, compile_opts=dict(wide_min_records=40))
    un = degree_graph(6, n_low=800, n_high=40, max_degree=3000, W=2000, p_cat=0.0)
    keep = np.diff(un.fac_edge_offset.astype(np.int64)) == 1
    from sampler_amd.rawgraph import RawGraph
    off = np.zeros(int(keep.sum()) + 1, np.uint64); off[1:] = np.arange(1, int(keep.sum()) + 1)
    first = un.fac_edge_offset[:-1].astype(np.int64)[keep]
    un = RawGraph(un.var_role, un.var_init_value, un.var_dtype, un.var_cardinality, un.fac_func[keep], off,
                  un.fac_weight_id[keep], un.fac_feature_value[keep], un.edge_vid[first], un.edge_equal_to[first],
                  un.w_initial_value, un.w_is_fixed)
    s, _ = run_parity(lib, un, n_learn=3, n_infer=3, stepsize=0.002)
    assert s.graph.info.num_wide_tiles > 0


def test_split_sweep_parity_and_heavy_tying_learns_like_the_reference(lib):
    """Heavily tied weights (hundreds of SGD updates per weight and sweep, every variable
    couples 10 weights) with a large step: one batched update per sweep would be outside
    its stability region (it learns mean weight 0.3 where the reference learns 0.04).  The
    plan must split the sweep into mini-batches; the split sweep must (a) match the oracle
    exactly when the oracle follows the same chunks and (b) learn what the REFERENCE
    semantics (sequential in-place updates, oracle reference mode) learn."""
    from oracle import binding as orc
    from sampler_amd import dwx
    raw = synthetic.cfg3(20_000, n_weights=200, seed=2024)
    s, o = run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.01, decay=0.95)
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, seed=5, reg_param=0.01)
    batches, n_chunks, eta = s.sgd_plan(0.01)
    # (cuts balance the SGD work: the query tiles, which learn nothing, ride along with the
    # first evidence chunk, so there can be fewer chunks than requested batches)
    # (eta: the smallest step any weight takes -- this much tying saturates it below 0.01)
    assert batches >= 8 and 8 <= n_chunks <= batches and 0.002 < eta < 0.01
    assert s.sgd_plan(1e-5)[0] == 1                      # tiny step: no split
    dwx.DimmWitted(s, 30, 0, 0.01, 0.95).learn()
    ref = orc.Oracle(raw, reg_param=0.01)
    ref.set_workers(1)
    ref.learn(30, 0.01, 0.95)
    w, wr = s.weights, ref.weights
    assert abs(w.mean() - wr.mean()) < 0.01 and abs(w.std() - wr.std()) < 0.015, (w.mean(), wr.mean())
    assert np.corrcoef(w, wr)[0, 1] > 0.6
    # with splitting disabled (step_cap <= 0) every sweep is one batch; the saturating step
    # keeps even that stable (a plain batched step of this size learns mean weight 0.3)
    s1 = dwx.GibbsSampler(g, seed=5, reg_param=0.01, step_cap=0.0)
    assert s1.sgd_plan(0.01)[0] == 1
    dwx.DimmWitted(s1, 30, 0, 0.01, 0.95).learn()
    w1 = s1.weights
    assert np.isfinite(w1).all() and abs(w1.mean() - wr.mean()) < 0.03, (w1.mean(), wr.mean())


def test_replica_weight_averaging(lib):
    """dwx_average_weights_async: after an in-place sum over n replicas, non-fixed weights
    are divided by n, fixed weights restored verbatim, and the f32 sampling copy follows
    (the next sweep draws exactly what a sampler with those weights set draws)."""
    from sampler_amd import dwx
    from randgraph import random_graph
    raw = random_graph(41, V=500, F=2500, W=25, p_cat=0.2, max_arity=2, exact_fvals=True)
    raw.w_is_fixed[::4] = 1
    raw.w_initial_value[::4] = 0.1 * np.arange(len(raw.w_initial_value[::4])) + 0.3   # 3*x/3 != x for some
    g = dwx.Graph(raw, lib=lib)
    a, b = dwx.GibbsSampler(g, seed=9), dwx.GibbsSampler(g, seed=9)
    a.sample_sgd(0.05); b.sample_sgd(0.05); a.wait(); b.wait()
    w = a.weights
    assert np.array_equal(w, b.weights) and np.abs(w[raw.w_is_fixed == 0]).max() > 0
    fixed = raw.w_is_fixed.astype(bool)
    other = w + np.where(fixed, 0.0, 0.25)             # what a second and third replica hold
    third = w - np.where(fixed, 0.0, 0.125)
    a.weights = w + other + third                      # stands for the in-place all-reduce(SUM)
    a.average_weights(3)
    expect = np.where(fixed, raw.w_initial_value, (w + other + third) / 3)
    assert np.array_equal(a.weights, expect)
    assert np.array_equal(a.weights[fixed], raw.w_initial_value[fixed])
    b.weights = expect
    for _ in range(3):
        a.sample(); b.sample()
    a.wait(); b.wait()
    assert np.array_equal(a.assignments("evid"), b.assignments("evid"))
    assert np.array_equal(a.tallies()[0], b.tallies()[0])


def test_split_sweep_with_pull_tiles_and_static_counts(lib, monkeypatch):
    """Split learning sweeps on graphs whose tiles use the pull-based gradient (W > 1024):
    every chunk pulls its own part of the (chunk, weight)-sorted incidence list and applies
    with its own row of static update counts -- same result as the oracle following the
    same chunks.  Also a two-colour graph (pairwise factors), categorical variables (their
    counts stay dynamic), learn_non_evidence, and a plan with more chunks than the table
    limit (atomics fallback)."""
    from sampler_amd import dwx
    from randgraph import random_graph
    raw = synthetic.cfg3(8000, n_weights=1300, seed=7)
    g = dwx.Graph(raw, lib=lib, tile_vars=64)
    s = dwx.GibbsSampler(g, seed=3)
    assert 2 <= s.sgd_plan(0.05)[0] <= 64                   # split, with per-chunk tables
    run_parity(lib, raw, n_learn=4, n_infer=2, stepsize=0.05, compile_opts=dict(tile_vars=64))
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.08, learn_non_evidence=True,
               compile_opts=dict(tile_vars=64))
    run_parity(lib, synthetic.cfg3b(4000, n_weights=1400, seed=15), n_learn=4, n_infer=2, stepsize=0.05,
               compile_opts=dict(tile_vars=64))
    run_parity(lib, random_graph(8, V=1500, F=9000, W=1300, p_cat=0.3), n_learn=4, n_infer=2, stepsize=0.2,
               learn_non_evidence=True, compile_opts=dict(tile_vars=32))
    # never more than 64 batches per launch, and no finer than still lowers the batches'
    # curvature (a batch cannot be smaller than a variable); an absurd step just saturates
    s2 = dwx.GibbsSampler(dwx.Graph(raw, lib=lib, tile_vars=16), seed=3)
    b2, n2, eta2 = s2.sgd_plan(5.0)
    assert 16 <= b2 <= 64 and n2 <= 64 and eta2 < 5.0
    assert b2 == 64 or s2.sgd_curvature(2 * b2) > 0.8 * s2.sgd_curvature(b2)
    run_parity(lib, raw, n_learn=2, n_infer=1, stepsize=5.0, compile_opts=dict(tile_vars=16))
    # plans too large for the per-chunk tables: per-record atomics and dynamic counts
    monkeypatch.setenv("DWX_PLAN_TABLE_CHUNKS", "4")
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.05, compile_opts=dict(tile_vars=64))


@pytest.mark.parametrize("block_tiles", [4, 2048])
def test_block_pull_in_split_sweeps(lib, monkeypatch, block_tiles):
    """Split plans with the block pull: every chunk has its own entry rows (and the rest of
    its list); chunks whose rows would be nearly empty keep the plain list.  The decaying step
    walks through several plan levels; also a two-colour graph and learn_non_evidence."""
    from sampler_amd import dwx
    monkeypatch.setenv("DWX_BLOCK_PULL_MIN_W", "0")
    monkeypatch.setenv("DWX_BLOCK_PULL_TILES", str(block_tiles))
    raw = synthetic.cfg3(8000, n_weights=1300, seed=7)
    s = dwx.GibbsSampler(dwx.Graph(raw, lib=lib, tile_vars=64), seed=3)
    assert 2 <= s.sgd_plan(0.05)[0] <= 64
    run_parity(lib, raw, n_learn=5, n_infer=1, stepsize=0.05, decay=0.6, compile_opts=dict(tile_vars=64))
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.08, learn_non_evidence=True,
               compile_opts=dict(tile_vars=64))
    run_parity(lib, synthetic.cfg3b(4000, n_weights=1400, seed=15), n_learn=4, n_infer=2, stepsize=0.05,
               compile_opts=dict(tile_vars=64))


def test_tabulated_inference_terms_follow_every_weight_change(lib):
    """From the second consecutive inference sweep on unchanged weights on, all-unary tiles
    stream a table of their potential terms instead of gathering weights.  The table must be
    rebuilt after anything that changes the weights -- a learning sweep, dwx_set_weights,
    replica averaging -- and must not be used by learning sweeps.  Exact parity with the
    oracle through an interleaved sequence; mixed graph (tabulated and generic tiles)."""
    from oracle import binding as orc
    from sampler_amd import dwx
    from parity import learn_sweep_both
    from randgraph import random_graph
    for raw in (synthetic.cfg3(3000, n_weights=1500, seed=5),
                random_graph(21, V=1200, F=6000, W=60, p_cat=0.3, max_arity=2, exact_fvals=True)):
        g = dwx.Graph(raw, lib=lib, tile_vars=64)
        s, o = dwx.GibbsSampler(g, seed=13), orc.Oracle(raw)
        o.set_fixed_point_mask(g.fixed_point_mask())
        order, off = g.schedule()
        sweep = 0

        def infer(n):
            nonlocal sweep
            for _ in range(n):
                s.sample(); s.wait()
                o.sched_sample(order, off, 13, sweep); sweep += 1
                assert np.array_equal(s.assignments("evid"), o.assignments("evid"))

        def learn(n, step):
            nonlocal sweep
            for _ in range(n):
                learn_sweep_both(s, o, order, 13, sweep, step); sweep += 1
                assert np.array_equal(s.assignments("free"), o.assignments("free"))
                assert np.array_equal(s.assignments("evid"), o.assignments("evid"))
                np.testing.assert_allclose(s.weights, o.weights, rtol=1e-12, atol=1e-12)

        infer(4)                      # gather, build table, table, table
        learn(2, 0.05)                # weights move: table stale
        infer(3)
        learn(1, 0.02); infer(1); learn(1, 0.02); infer(3)      # alternating, then repeated
        w = o.weights.copy(); w[raw.w_is_fixed == 0] *= -0.5
        s.weights = w; o.weights[:] = w
        infer(3)
        s.weights = 3 * w; s.average_weights(3)      # stands for an all-reduce + averaging
        o.weights[:] = s.weights
        infer(3)
        assert np.array_equal(s.tallies()[0], o.tallies[:len(s.tallies()[0])])
