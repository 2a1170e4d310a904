"""GPU parity tests proper (-m gpu): every case calls the HIP library through the C ABI
and compares with the CPU oracle on the same seeded inputs -- exactly (assignments,
tallies: integer work) and to 1e-12 (weights: f64 sums of identical fixed-point
gradients) -- then checks BASELINE-size runs through size-independent properties and
against the real reference's marginals (KS, alpha = 0.01)."""
import os
import tempfile

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, parse_dw_args
from parity import gpu_library, run_parity
from sampler_amd import binary_format, dwx, synthetic
import stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return gpu_library()


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


@pytest.mark.parametrize("seed", range(6))
def test_random_mixed_graphs(lib, seed):
    from randgraph import random_graph
    raw = random_graph(seed, truthy=bool(seed % 2))
    run_parity(lib, raw, n_learn=6, n_infer=6, stepsize=0.1, noise_aware=bool(seed % 2),
               learn_non_evidence=seed in (2, 3), sample_evidence=seed in (1, 2))


def test_random_bigger_graph(lib):
    from randgraph import random_graph
    raw = random_graph(42, V=3000, F=12000, W=200)
    run_parity(lib, raw, n_learn=4, n_infer=4, stepsize=0.05, learn_non_evidence=True)


def test_synth_exact(lib):
    run_parity(lib, synthetic.cfg2(50_000, seed=3), n_learn=0, n_infer=4)
    run_parity(lib, synthetic.cfg3(50_000, seed=4), n_learn=3, n_infer=3, stepsize=0.01)
    s, _ = run_parity(lib, synthetic.cfg3b(30_000, seed=5), n_learn=3, n_infer=3, stepsize=0.01)
    assert s.graph.info.num_colors >= 2
    run_parity(lib, synthetic.cfg4(20_000, card=8, seed=6), n_learn=0, n_infer=3)
    run_parity(lib, synthetic.cfg4(20_000, card=8, seed=7, learn=True), n_learn=3, n_infer=2,
               stepsize=0.001)


def test_device_build_scratch_is_given_back(lib):
    """The device builds take their large temporaries from a block cache instead of hipMalloc / hipFree
    (device_build.hip: a hipFree of gigabytes is paid by a later hipMalloc).  The cache must not grow
    with the number of samplers built, and destroying the samplers must return the device to where it was:
    five create / sweep / destroy rounds on a 1.2 M-variable graph (120 MB of sort buffers each)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")      # (the runtime the library itself is linked against)

    def free_bytes():
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    raw = synthetic.cfg3(1_200_000, n_weights=150_000, seed=22)
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, seed=3)          # (context, kernels, the runtime's own pools: before the baseline)
    s.close()
    free0 = free_bytes()
    held = []
    for _ in range(5):
        s = dwx.GibbsSampler(g, seed=3)
        s.sample_sgd(0.01); s.sample(); s.wait()
        held.append(free0 - free_bytes())
        s.close()
    free1 = free_bytes()
    assert max(held) - min(held) < (64 << 20), held          # every round holds the same: blocks are reused
    assert abs(free0 - free1) < (64 << 20), (free0, free1)   # ... and all of it comes back


def test_device_built_sorted_records_equal_the_host_builder(lib, monkeypatch):
    """The weight-sorted copy of the records is built on the device (device_build.hip: ordered emit +
    radix sort by (super-tile, weight id)) from the uploaded columns; the host builder
    (graph_compile.cc: build_sorted_layout, DWX_HOST_BUILD=1) is its checker: the same bytes, for the
    graph's default layout and for a split plan level's own layout (several feature values and sign
    classes, fixed weights, query and evidence parts; 12 M records -- a level builds a layout of its
    own only while a chunk still holds 2048 tiles)."""
    raw = synthetic.cfg3(1_200_000, n_weights=150_000, seed=21)
    rng = np.random.default_rng(5)
    raw.fac_feature_value[:] = rng.choice([1.0, 0.5, 2.0, -1.0, 0.25], size=raw.num_factors)
    raw.edge_equal_to[:] = rng.integers(0, 2, size=raw.num_edges)          # ISTRUE on "== 0" too: other sign classes
    raw.w_is_fixed[:] = rng.random(raw.num_weights) < 0.1
    got = {}
    for mode in ("device", "host"):
        if mode == "host":
            monkeypatch.setenv("DWX_HOST_BUILD", "1")
        g = dwx.Graph(raw, lib=lib)
        assert g.info.num_super_tiles > 0 and g.info.num_sorted_records > 0
        s = dwx.GibbsSampler(g, seed=3, plan_layouts=1)
        base = s.read_buffer(dwx.BUF_SORTED_RECORDS, np.uint64)
        assert len(base) == g.info.num_sorted_records
        s.sgd_plan(0.01, 1)                    # the un-split level: one run per launch (query + evidence tiles)
        lvl1 = s.read_buffer(dwx.BUF_SORTED_RECORDS_PLAN, np.uint64)
        s.sgd_plan(0.01, 2)                    # a split plan: super-tiles cut along its chunks
        lvl2 = s.read_buffer(dwx.BUF_SORTED_RECORDS_PLAN, np.uint64)
        s.sample_sgd(0.01); s.sample(); s.wait()
        got[mode] = (base, lvl1, lvl2, s.weights.copy(), s.assignments("evid").copy())
        s.close()
    monkeypatch.delenv("DWX_HOST_BUILD")
    assert len(got["device"][1]) > 0 and len(got["device"][2]) > 0, "a plan level built no layout of its own"
    for k, what in enumerate(("default layout", "un-split level layout", "split level layout", "weights", "chain")):
        assert np.array_equal(got["device"][k], got["host"][k]), what


def test_device_built_static_tables_equal_the_host_builder(lib, monkeypatch):
    """A plan level's static update counts T and curvature bounds h per chunk, built on the device
    (device_build.hip: a lane per variable, integer atomics) against the host builder
    (DWX_HOST_BUILD=1; dwx_api.cc build_level): the same tables bit for bit -- unary and pairwise
    and ternary factors, boolean and categorical variables, fixed weights, several feature values,
    un-split and split levels."""
    rng = np.random.default_rng(9)
    graphs = []
    raw = synthetic.cfg3(300_000, n_weights=9000, seed=21)
    raw.fac_feature_value[:] = rng.choice([1.0, 0.5, 2.0, -1.0, 0.3], size=raw.num_factors)   # (0.3: not f32-exact -> f64 side array)
    raw.w_is_fixed[:] = rng.random(raw.num_weights) < 0.1
    graphs.append(raw)
    graphs.append(synthetic.cfg3b(60_000, n_weights=5000, seed=5))
    graphs.append(synthetic.cfg3c(30_000, n_weights=3000, seed=6))
    graphs.append(synthetic.cfg4(150_000, card=8, seed=7, learn=True))
    graphs.append(synthetic.cfg4b(20_000, card=8, n_weights=2000, seed=8))
    for gi, raw in enumerate(graphs):
        got, lam = {}, {}
        for mode in ("device", "host"):
            if mode == "host":
                monkeypatch.setenv("DWX_HOST_BUILD", "1")
            g = dwx.Graph(raw, lib=lib)
            s = dwx.GibbsSampler(g, seed=3)
            t1 = s.read_buffer(dwx.BUF_TSTATIC, np.int64)
            _, n_chunks, min_step1 = s.sgd_plan(0.01, 1)
            b, n_chunks, min_step = s.sgd_plan(0.01, 4)
            t4 = s.read_buffer(dwx.BUF_TSTATIC_PLAN, np.int64)
            got[mode] = (t1, t4, np.array([min_step1, min_step, n_chunks], np.float64))
            # the curvature estimates (three power steps per mini-batch, 64-bit fixed-point sums on both sides:
            # the same integers; the closing dot products are added up in another order)
            lam[mode] = [s.sgd_curvature(bb) for bb in (1, 2, 4)]
            s.close()
            if mode == "host":
                monkeypatch.delenv("DWX_HOST_BUILD")
        assert len(got["device"][0]) == 2 * raw.num_weights and np.abs(got["device"][0]).max() > 0
        for k, what in enumerate(("un-split tables", "split tables", "smallest steps")):
            assert np.array_equal(got["device"][k], got["host"][k]), (gi, what)
        np.testing.assert_allclose(lam["device"], lam["host"], rtol=1e-9, err_msg="curvature estimates, graph %d" % gi)
        assert min(lam["device"]) > 0


@pytest.mark.parametrize("block_pull", [False, True])
def test_device_built_incidence_lists_and_block_tables_equal_the_host_builder(lib, monkeypatch, block_pull):
    """The pull gradient's structures of a plan level -- the incidence list sorted by (chunk, weight), the
    block-pull tables with their overflow lists -- built on the device (device_build.hip: ordered emit,
    one radix sort, a lane per weight filling the rows) against the host builder (DWX_HOST_BUILD=1):
    integer gradient sums, so the learned weights and both chains must be bit for bit the same after
    un-split and split learning sweeps, and both must equal the oracle.  Several feature values and sign
    classes, fixed weights, unary records of mixed tiles (config 3b), block tables forced at this size."""
    if block_pull:
        monkeypatch.setenv("DWX_BLOCK_PULL_MIN_W", "0")
        monkeypatch.setenv("DWX_BLOCK_PULL_TILES", "64")      # (several variable blocks at this size)
    rng = np.random.default_rng(11)
    raw = synthetic.cfg3(200_000, n_weights=6000, seed=21)
    raw.fac_feature_value[:] = rng.choice([1.0, 0.5, 2.0, -1.0, 0.25], size=raw.num_factors)
    raw.edge_equal_to[:] = rng.integers(0, 2, size=raw.num_edges)
    raw.w_is_fixed[:] = rng.random(raw.num_weights) < 0.1
    for gi, (g_raw, forced) in enumerate(((raw, 1), (raw, 4), (synthetic.cfg3b(80_000, n_weights=5000, seed=5), 2))):
        got = {}
        for mode in ("device", "host"):
            if mode == "host":
                monkeypatch.setenv("DWX_HOST_BUILD", "1")
            g = dwx.Graph(g_raw, lib=lib)
            s = dwx.GibbsSampler(g, seed=3)
            for k in range(3):
                batches, n_chunks, _ = s.sgd_plan(0.01, forced)
                for c in range(n_chunks):
                    s.sgd_accumulate(c)
                    if batches > 1 or c + 1 == n_chunks:
                        s.sgd_apply()
                s.sgd_finish()
            s.sample(); s.wait()
            got[mode] = (s.weights.copy(), s.assignments("free").copy(), s.assignments("evid").copy())
            s.close()
            if mode == "host":
                monkeypatch.delenv("DWX_HOST_BUILD")
        assert np.abs(got["device"][0]).max() > 0
        for k, what in enumerate(("weights", "free chain", "evidence chain")):
            assert np.array_equal(got["device"][k], got["host"][k]), (gi, forced, what)
    # ... and against the oracle (the device build is the default: every other parity test runs on it too)
    run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.01, check_index=False)


def test_split_learning_sweep_with_one_launch_per_mini_batch(lib, monkeypatch):
    """Config 4 with learning (8 weights tied to 10^5 evidence factors each: 64 mini-batches per sweep) and a
    boolean graph with 20 weights: every mini-batch is ONE launch (sweep8_merged_kernel: the previous
    mini-batch's update as the kernel's prologue, every workgroup for itself) -- the default.  Exact
    against the oracle stepped chunk by chunk, bit for bit the two-launch path (DWX_NO_MERGED_APPLY)."""
    for raw, kw in ((synthetic.cfg4(200_000, card=8, seed=7, learn=True), dict(stepsize=0.001, decay=0.9)),
                    (synthetic.cfg3(300_000, n_weights=20, seed=4), dict(stepsize=0.002, decay=0.8, regularization="l1",
                                                                        reg_param=0.002))):
        s, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        batches, n_chunks, _ = s.sgd_plan(kw["stepsize"])
        assert batches >= 8 and n_chunks >= 8, (batches, n_chunks)
        assert s.kernel_time("merged")[1] == 4, s.kernel_time("merged")
        monkeypatch.setenv("DWX_NO_MERGED_APPLY", "1")
        s2, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        monkeypatch.delenv("DWX_NO_MERGED_APPLY")
        assert s2.kernel_time("merged")[1] == 0
        assert np.array_equal(s.weights, s2.weights)
        assert np.array_equal(s.assignments("free"), s2.assignments("free"))
        assert np.array_equal(s.assignments("evid"), s2.assignments("evid"))


def test_split_learning_sweep_as_one_persistent_launch(lib, monkeypatch):
    """DWX_PERSIST=1 (opt-in: built, exact, measured slower than the plain launches -- persist_kernels.h).
    Config 4 with learning (8 weights tied to 10^5 evidence factors each: 64 mini-batches per sweep)
    and a boolean graph with 20 weights: every split sweep is ONE launch of persist_learn8_kernel --
    up to one workgroup per CU, a gradient row per workgroup and chunk, a grid barrier between the
    chunks, the update applied by every workgroup to its LDS copy of the weights.  Exact against the
    oracle stepped chunk by chunk; bit for bit the chunk-by-chunk launches (DWX_NO_PERSIST)."""
    for raw, kw in ((synthetic.cfg4(200_000, card=8, seed=7, learn=True), dict(stepsize=0.001, decay=0.9)),
                    (synthetic.cfg3(300_000, n_weights=20, seed=4), dict(stepsize=0.002, decay=0.8, regularization="l1",
                                                                        reg_param=0.002))):
        monkeypatch.setenv("DWX_PERSIST", "1")
        s, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        batches, n_chunks, _ = s.sgd_plan(kw["stepsize"])
        assert batches >= 8 and n_chunks >= 8, (batches, n_chunks)
        assert s.kernel_time("persist")[1] >= 3, s.kernel_time("persist")
        monkeypatch.delenv("DWX_PERSIST")
        s2, _ = run_parity(lib, raw, n_learn=4, n_infer=2, **kw)
        assert s2.kernel_time("persist")[1] == 0
        assert np.array_equal(s.weights, s2.weights)
        assert np.array_equal(s.assignments("free"), s2.assignments("free"))
        assert np.array_equal(s.assignments("evid"), s2.assignments("evid"))


def test_split_learning_sweep_replayed_as_one_graph(lib, monkeypatch):
    """DWX_GRAPH=4: dwx_sample_sgd_async hands a split sweep (>= 4 mini-batches: config 4's tied
    weights) over as ONE hipGraph launch from the plan level's second sweep on -- captured every
    sweep, patched into the level's instantiated graph (only the Philox sweep counter and the step
    change).  Exact against the oracle sweep by sweep, identical to the plain launches (the default),
    and a graph with side streams inside (wave / workgroup bins fork beside the tile sweep) replays
    as well."""
    raw = synthetic.cfg4(20_000, card=8, seed=7, learn=True)
    monkeypatch.setenv("DWX_GRAPH", "4")
    s, _ = run_parity(lib, raw, n_learn=6, n_infer=2, stepsize=0.001, decay=1.0)
    batches, n_chunks, _ = s.sgd_plan(0.001)
    assert batches > 1 and n_chunks >= 4, (batches, n_chunks)
    assert s.kernel_time("graph")[1] == 5, s.kernel_time("graph")
    # (timing on: plain launches, the events must not land inside a graph)
    s.kernel_time_reset(True)
    s.sample_sgd(0.001); s.wait()
    assert s.kernel_time("learn")[1] >= n_chunks and s.kernel_time("graph")[1] == 5
    s.kernel_time_reset(False)
    monkeypatch.delenv("DWX_GRAPH")
    s2, _ = run_parity(lib, raw, n_learn=6, n_infer=2, stepsize=0.001, decay=1.0)
    assert s2.kernel_time("graph")[1] == 0
    s2.sample_sgd(0.001); s2.wait()
    assert np.array_equal(s.weights, s2.weights)
    assert np.array_equal(s.assignments("free"), s2.assignments("free"))
    assert np.array_equal(s.assignments("evid"), s2.assignments("evid"))
    monkeypatch.setenv("DWX_GRAPH", "4")
    # a decaying step walks down the plan levels: every level gets a graph of its own
    s3, _ = run_parity(lib, raw, n_learn=12, n_infer=1, stepsize=0.002, decay=0.7)
    assert s3.kernel_time("graph")[1] >= 3, s3.kernel_time("graph")
    # forks: lanes, waves and workgroups per variable in one split sweep
    from randgraph import degree_graph_fast
    s4, _ = run_parity(lib, degree_graph_fast(8, n_low=20_000, n_high=300, max_degree=20_000, W=500),
                       n_learn=4, n_infer=1, stepsize=0.0005, decay=1.0, check_index=False)
    assert s4.graph.info.num_wide_tiles > 0 and s4.graph.info.num_giant_tiles > 0
    if s4.sgd_plan(0.0005)[1] >= 4:
        assert s4.kernel_time("graph")[1] == 3, s4.kernel_time("graph")


def test_small_tiles_and_giant_variable(lib):
    raw = synthetic.cfg3b(300, n_weights=16, seed=8)
    run_parity(lib, raw, n_learn=3, n_infer=3, compile_opts=dict(tile_vars=7, tile_edges=16, tile_rows=7))
    raw = synthetic.cfg4(60, card=9, seed=9, learn=True)
    s, _ = run_parity(lib, raw, n_learn=3, n_infer=3, compile_opts=dict(tile_vars=5, tile_edges=8, tile_rows=8))
    assert s.graph.info.num_giant_tiles > 0


# ---------------- BASELINE-size runs: size-independent properties ----------------

def _boolean_marginals(s):
    t, n = s.tallies()
    return t.astype(np.float64) / np.maximum(n, 1)


def test_cfg2_full_size_closed_form(lib):
    """Config 2 (1M boolean x 10 ISTRUE, inference only): unary graph => draws are
    i.i.d. per variable, so z-scores against the closed form sigmoid(2 sum w) are
    exactly N(0,1); KS at alpha = 0.01, and the mean tally must match the mean
    probability to Monte-Carlo accuracy."""
    raw = synthetic.cfg2(1_000_000, seed=1234)
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, seed=99)
    N = 100
    drv = dwx.DimmWitted(s, 0, N)
    drv.inference()
    p_hat = _boolean_marginals(s)
    p = synthetic.cfg2_closed_form(raw)
    z = stats.z_scores_vs_exact(p_hat, p, N)
    ok = (p > 0.05) & (p < 0.95)           # normal approximation of the binomial
    assert stats.ks_normal(z[ok][::7]) > 0.01
    assert abs(p_hat.mean() - p.mean()) < 5 * np.sqrt(0.25 / (N * len(p)))
    # idempotence of the state read-out and determinism of the chain
    s2 = dwx.GibbsSampler(g, seed=99)
    dwx.DimmWitted(s2, 0, N).inference()
    assert np.array_equal(s2.tallies()[0], s.tallies()[0])


def test_cfg3_full_size_exact_against_oracle(lib):
    """The bench's own workload at full size (config 3: 10 M variables, 100 M factors, 1 M
    weights): two learning sweeps (sweep8_kernel, block pull, apply) and two inference sweeps
    (weight gathers, then the 8-byte terms table) -- assignments of both chains and tallies
    bit for bit, weights to 1e-12, against the oracle."""
    raw = synthetic.cfg3(10_000_000, n_weights=1_000_000, seed=1234)
    s, o = run_parity(lib, raw, n_learn=2, n_infer=2, stepsize=0.001, decay=0.95, check_index=False)
    assert s.sgd_plan(0.001)[0] == 1                      # un-split sweeps, as in bench.py
    assert np.abs(s.weights).max() > 0


def test_cfg3b_full_size_exact_against_oracle(lib):
    """Config 3b at full size (10 M variables, 6 unary + 4 pairwise EQUAL factors each, two
    colours: the shape of config 5b's shards): one learning and two inference sweeps, exact."""
    raw = synthetic.cfg3b(10_000_000, n_weights=1_000_000, seed=1234)
    s, _ = run_parity(lib, raw, n_learn=1, n_infer=2, stepsize=0.001, check_index=False)
    assert s.graph.info.num_colors >= 2


def _host_memory_available_gb():
    """What this process may still allocate: the cgroup's limit minus its use, or MemAvailable."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            room = int(lim) - int(open("/sys/fs/cgroup/memory.current").read())
            avail = room if avail is None else min(avail, room)
    except (OSError, ValueError):
        pass
    return (avail or 0) / 1e9


def test_config5_whole_graph_on_one_gpu_exact_against_oracle(lib):
    """BASELINE config 5's WHOLE graph -- 100 M boolean variables x 10 unary factors = 10^9 records,
    1 M weights tied to 1 000 factors each -- on ONE MI355X (35 GB of its 288): one learning sweep
    (a split plan: mini-batches, block pull and apply per chunk) and one inference sweep, both
    chains' assignments and the tallies bit for bit, weights to 1e-12, against the oracle that
    follows the same chunk boundaries.  (run_parity's body with numpy in place of Python lists.)

    The full size takes 185 GB of host memory and three minutes (graph + compile + the oracle's
    copy): it runs wherever the box has 220 GB to spare (or DWX_BIG_TESTS=1); else HALF of it --
    50 M variables, 5 x 10^8 records: the first size whose record streams pass 4 GB, a split plan --
    in 65 s and 94 GB, skipped on a box without the room."""
    from oracle import binding as orc
    from parity import learn_sweep_both
    # the full 10^9-record graph where the box has the room for it (220 GB: graph + compile + the
    # oracle's copy take 185), else the half that still passes 4 GB of records
    have = _host_memory_available_gb()
    V = int(os.environ.get("DWX_BIG_VARS", "100000000" if (os.environ.get("DWX_BIG_TESTS") == "1" or have >= 220) else "50000000"))
    need = 2.0e-6 * V + 20
    if have < need:
        pytest.skip("needs %.0f GB of host memory" % need)
    raw = synthetic.cfg3(V, n_weights=1_000_000, seed=1234)
    g = dwx.Graph(raw, lib=lib)
    o = orc.Oracle(raw)
    o.set_fixed_point_mask(g.fixed_point_mask())
    order, off = g.schedule()
    assert np.array_equal(np.sort(order), np.arange(V, dtype=order.dtype))
    assert o.sched_check_independent(order, off)
    s = dwx.GibbsSampler(g, seed=77)
    assert g.info.num_index_entries == 10 * V and g.info.num_super_tiles > 0
    batches = learn_sweep_both(s, o, order, 77, 0, 0.001)
    print("config 5 on one GPU: V = %d, %d mini-batches per learning sweep, device bytes %.1f GB"
          % (V, batches, g.info.device_bytes / 1e9))
    assert np.array_equal(s.assignments("free"), o.assignments("free")), "free chain differs"
    assert np.array_equal(s.assignments("evid"), o.assignments("evid")), "evid chain differs"
    np.testing.assert_allclose(s.weights, o.weights, rtol=1e-12, atol=1e-12)
    assert np.abs(s.weights).max() > 0
    s.clear_tallies(); o.clear_tallies()
    s.sample(); s.wait()
    o.sched_sample(order, off, 77, 1)
    assert np.array_equal(s.assignments("evid"), o.assignments("evid")), "inference chain differs"
    t, n = s.tallies()
    assert np.array_equal(t, o.tallies[:len(t)]) and np.array_equal(n, o.nsamples)


def test_cfg2_cfg4_full_size_exact_against_oracle(lib):
    """BASELINE configs 2 (1 M boolean x 10 ISTRUE) and 4 (5 M categorical, domain 8) at full
    size, inference only, three sweeps each (gathers, table build, table): bit for bit."""
    run_parity(lib, synthetic.cfg2(1_000_000, seed=1234), n_learn=0, n_infer=3, check_index=False)
    run_parity(lib, synthetic.cfg4(5_000_000, card=8, seed=1234, learn=False), n_learn=0, n_infer=3,
               check_index=False)


def test_cfg4_closed_form(lib):
    raw = synthetic.cfg4(200_000, card=8, seed=1234, learn=False)
    s = dwx.GibbsSampler(dwx.Graph(raw, lib=lib), seed=5)
    N = 50
    dwx.DimmWitted(s, 0, N).inference()
    t, n = s.tallies()
    assert (n == N).all() and (t.reshape(-1, 8).sum(1) == N).all()   # one draw per sweep
    p = synthetic.cfg4_closed_form(raw, 8)
    freq = t.reshape(-1, 8).sum(0) / (N * 200_000.0)
    assert np.abs(freq - p).max() < 5 * np.sqrt(0.25 / (N * 200_000))


def test_cfg3_learning_recovers_evidence_rate(lib):
    """Config 3 shape at 1M variables: every weight is shared by ~100 unary ISTRUE
    factors whose evidence is Bernoulli(0.7): after learning, query-variable
    marginals must sit near 0.7 on average (the MLE), weights stay finite, evidence
    variables are never resampled during inference, fixed state is untouched."""
    raw = synthetic.cfg3(1_000_000, seed=1234)
    s = dwx.GibbsSampler(dwx.Graph(raw, lib=lib), seed=11, reg_param=0.01)
    drv = dwx.DimmWitted(s, 30, 50, stepsize=0.01, decay=0.95)
    drv.learn()
    w = s.weights
    assert np.isfinite(w).all() and np.abs(w).max() < 5
    drv.inference()
    t, n = s.tallies()
    q = raw.var_role == 0
    assert (n[q] == 50).all() and (n[~q] == 0).all() and (t[~q] == 0).all()
    ev = s.assignments("evid")
    assert np.array_equal(ev[~q], raw.var_init_value[~q])
    m = (t[q] / 50.0).mean()
    assert 0.6 < m < 0.8, m


# ---------------- against the real reference (KS on per-variable marginals) --------

def _parse_marginals(txt, V):
    import ks_golden
    return ks_golden.parse_marginals(txt, V)


@pytest.mark.parametrize("name", ["synth_cfg2", "synth_cfg3", "synth_cfg3b", "synth_cfg4"])
def test_ks_against_reference_golden_marginals(lib, name):
    """Committed marginals of the REAL reference (multi-threaded run in the build
    container, tests/golden/make_golden.py) on 1/500-scale configs 2, 3, 3b, 4: with this
    build's OWN learned weights the distribution of the marginals must match (two-sample KS);
    with the reference's weights substituted the per-variable z-scores must be N(0,1) (KS,
    alpha 0.01).  tests/ks_golden.py."""
    import ks_golden
    ks_golden.check(lib, name, dict(device=0))


def test_ks_against_reference_live(lib):
    """When oracle/_ref/dw travelled to this box: run the real reference here on a
    100k-variable config-2 graph and KS-compare per-variable marginals."""
    from oracle import binding as orc
    if not orc.have_reference():
        pytest.skip("oracle/_ref/dw not present")
    raw = synthetic.cfg2(100_000, seed=77)
    N = 200
    with tempfile.TemporaryDirectory() as d:
        binary_format.write_graph(raw, d)
        orc.run_reference_dw(d, ["-l", "0", "-i", str(N)], d)
        p_ref = _parse_marginals(open(os.path.join(d, "inference_result.out.text")).read(), 100_000)
    s = dwx.GibbsSampler(dwx.Graph(raw, lib=lib), seed=8)
    dwx.DimmWitted(s, 0, N).inference()
    p_gpu = _boolean_marginals(s)
    z = stats.z_scores_two_sample(p_gpu, N, p_ref, N)
    pbar = 0.5 * (p_gpu + p_ref)
    z = z[(pbar > 0.1) & (pbar < 0.9)]
    n = len(z)
    # z lives on a lattice with an atom at 0 (difference of two binomials over N), so
    # at n ~ 5e4 a KS test against the continuous normal rejects on discreteness
    # alone; test the first two moments on the full set (5 sigma) and KS on a
    # subsample small enough for the lattice step to sit below the critical distance
    assert abs(z.mean()) < 5.0 / np.sqrt(n)
    assert abs(z.var() - 1.0) < 5.0 * np.sqrt(2.0 / n) + 0.02
    assert stats.ks_normal(z[:: max(1, n // 1500)]) > 0.01
    assert stats.ks_two_sample(p_gpu, p_ref) > 0.01


def test_ghost_variables_parity(lib):
    """Shard-local graph with ghost variables (config 5b): see tests/test_kernels_emu.py."""
    import test_kernels_emu as E
    E.test_ghost_variables_parity.__wrapped__(lib) if hasattr(E.test_ghost_variables_parity, "__wrapped__") \
        else E.test_ghost_variables_parity(lib)


def test_replica_weight_averaging(lib):
    import test_kernels_emu as E
    E.test_replica_weight_averaging(lib)


def test_tabulated_inference_terms_follow_every_weight_change(lib):
    import test_kernels_emu as E
    E.test_tabulated_inference_terms_follow_every_weight_change(lib)


def test_split_sweep_with_pull_tiles_and_static_counts(lib, monkeypatch):
    import test_kernels_emu as E
    E.test_split_sweep_with_pull_tiles_and_static_counts(lib, monkeypatch)


@pytest.mark.parametrize("compact", [True, False])
def test_all_unary_compact_records(lib, compact):
    import test_kernels_emu as E
    E.test_all_unary_compact_records(lib, compact)


@pytest.mark.parametrize("min_w", [0, 10 ** 9])
def test_weight_sorted_super_tiles(lib, monkeypatch, min_w):
    import test_kernels_emu as E
    E.test_weight_sorted_super_tiles(lib, monkeypatch, min_w)


@pytest.mark.parametrize("block_tiles, depth_hint", [(8, 1), (32, 2), (2048, 2)])
def test_block_pull(lib, monkeypatch, block_tiles, depth_hint):
    import test_kernels_emu as E
    E.test_block_pull(lib, monkeypatch, block_tiles, depth_hint)


@pytest.mark.parametrize("block_tiles", [4, 2048])
def test_block_pull_in_split_sweeps(lib, monkeypatch, block_tiles):
    import test_kernels_emu as E
    E.test_block_pull_in_split_sweeps(lib, monkeypatch, block_tiles)


def test_block_pull_at_size_equals_list_pull_and_oracle(lib, monkeypatch):
    # 2 M variables, 300 k weights: the block pull engages on its own (>= 131 072 weights);
    # several variable blocks, two-row tables, entries left on the list.  Exact against the
    # oracle, and bit-identical to the same run with the block pull switched off.
    raw = synthetic.cfg3(2_000_000, n_weights=300_000, seed=21)
    s, _ = run_parity(lib, raw, n_learn=3, n_infer=1, stepsize=0.01, step_cap=0.0, check_index=False)
    w_block = s.weights.copy()
    monkeypatch.setenv("DWX_BLOCK_PULL_MIN_W", str(10**9))
    g = dwx.Graph(raw, lib=lib)
    s2 = dwx.GibbsSampler(g, device=0, seed=77, reg_param=0.01, step_cap=0.0)
    cur = 0.01
    for _ in range(3):
        s2.sample_sgd(cur); cur *= 0.9
    s2.wait()
    assert np.array_equal(s2.weights, w_block)
    assert np.abs(w_block).max() > 0


@pytest.mark.parametrize("seed", range(3))
def test_binary_factor_tiles_all_functions(lib, seed):
    from randgraph import random_graph
    raw = random_graph(100 + seed, V=3000, F=15000, W=40, p_cat=0.0, max_arity=2, exact_fvals=True)
    run_parity(lib, raw, n_learn=3, n_infer=6, stepsize=0.05, sample_evidence=bool(seed % 2),
               learn_non_evidence=seed >= 1)


def test_high_degree_hub_variables(lib):
    from randgraph import hub_graph
    s, _ = run_parity(lib, hub_graph(3, V=4000, hub_degree=200_000), n_learn=3, n_infer=4,
                      stepsize=0.0001, learn_non_evidence=True)
    assert s.graph.info.num_giant_tiles == 2
    run_parity(lib, hub_graph(4, W=2000), n_learn=3, n_infer=3, stepsize=0.001, sample_evidence=True)


def test_degree_bins_lane_wave_workgroup(lib):
    import test_kernels_emu as E
    E.test_degree_bins_lane_wave_workgroup(lib)


def test_arity3_tiles_evaluated_edge_parallel(lib):
    import test_kernels_emu as E
    E.test_arity3_tiles_evaluated_edge_parallel(lib)
    # and at size: 2 M variables, 9 colours, two learning and two inference sweeps, exact
    run_parity(lib, synthetic.cfg3c(2_000_000, seed=5), n_learn=2, n_infer=2, stepsize=0.001, check_index=False)


def test_halo_lists_travel_as_bits_bytes_or_words(lib):
    import test_kernels_emu as E
    E.test_halo_lists_travel_as_bits_bytes_or_words(lib)


def test_unary_records_of_mixed_tiles_take_the_pull_gradient(lib):
    import test_kernels_emu as E
    E.test_unary_records_of_mixed_tiles_take_the_pull_gradient(lib)
    # at size, block pull engaged (300 k weights) and list pull (20 k)
    run_parity(lib, synthetic.cfg3b(2_000_000, n_weights=300_000, seed=8), n_learn=2, n_infer=1, stepsize=0.001, check_index=False)
    run_parity(lib, synthetic.cfg3c(1_000_000, n_weights=20_000, seed=9), n_learn=2, n_infer=1, stepsize=0.001, check_index=False)


def test_categorical_tiles_evaluated_edge_parallel(lib):
    import test_kernels_emu as E
    E.test_categorical_tiles_evaluated_edge_parallel(lib)
    # and at size: 500 k categorical variables in a chain (12 M records), tied and untied weights
    run_parity(lib, synthetic.cfg4b(500_000, seed=6), n_learn=2, n_infer=2, stepsize=0.001, check_index=False)
    run_parity(lib, synthetic.cfg4b(300_000, n_weights=30_000, seed=7), n_learn=2, n_infer=2, stepsize=0.001,
               check_index=False, learn_non_evidence=True)


def test_degree_histogram_1_to_1e5_exact_and_3x_faster_than_round_1(lib):
    """A power-law-shaped graph: 200k variables with 1-8 factors, 3000 with 16 ... 100 000
    (log-uniform), 35 M factors, 20 colours.  Exact against the oracle through learning and
    inference sweeps -- lanes, waves and (several) workgroups per variable side by side.

    Speed (">= 3x faster than today"): the round-1 build
    (commit 293a388: lane-per-variable tiles + one 256-lane workgroup per oversized variable)
    took 15.5 ms per inference sweep and 236 ms per learning sweep on this very graph on the same
    box (gpurun_out of this round, profiles/r02/degree_bins.md); this build must stay under a
    third of that.  Where the time went: oversized variables now spread over one workgroup per
    8192 records (a single CU turns around one scattered request per ~2.3 cycles, so a 10^5-record
    hub kept ITS workgroup busy for 0.3 ms per colour), their walks batch four records' loads per
    lane, mid-degree variables left the tiles for a wave each (the wave bin alone: learning
    sweep 1.3x on this graph, 2.6x on one without hubs), and the mini-batch plan cuts colours in
    proportion to their work (1.3 ms / 16 ms per sweep at the end of round 2)."""
    import time
    from randgraph import degree_graph_fast
    raw = degree_graph_fast(7)
    s, _ = run_parity(lib, raw, n_learn=2, n_infer=2, stepsize=0.0005, check_index=False)
    info = s.graph.info
    assert info.num_wide_tiles > 500 and info.num_giant_tiles > 100
    deg = np.bincount(raw.edge_vid.astype(np.int64), minlength=raw.num_variables)
    assert deg.min() >= 1 and deg.max() > 50_000

    def sweep_ms(sampler, learn):
        for _ in range(2):
            sampler.sample_sgd(0.0005) if learn else sampler.sample()
        sampler.wait()
        t0 = time.perf_counter()
        for _ in range(5):
            sampler.sample_sgd(0.0005) if learn else sampler.sample()
        sampler.wait()
        return (time.perf_counter() - t0) / 5 * 1e3

    off = dwx.GibbsSampler(dwx.Graph(raw, lib=lib, wide_min_records=0xFFFFFFFF), device=0, seed=77)
    assert off.graph.info.num_wide_tiles == 0
    t_on = (sweep_ms(s, False), sweep_ms(s, True))
    t_off = (sweep_ms(off, False), sweep_ms(off, True))
    print("degree bins: inference %.3f ms (wave bin off: %.3f; round 1: 15.5); learning %.3f ms (off: %.3f; round 1: 236)"
          % (t_on[0], t_off[0], t_on[1], t_off[1]))
    assert t_on[0] <= 15.5 / 3 and t_on[1] <= 236.0 / 3, t_on
    # (the wave bin must not cost anything here -- on this graph the hubs dominate, the bin's own gain
    # shows on the hub-less histogram: DESIGN.md 3.3)
    assert t_off[1] >= 0.9 * t_on[1], (t_on, t_off)


def test_full_pipeline_learn_infer_vs_reference_live(lib):
    """Config-3 shape at 1M variables, the WHOLE pipeline on both sides (the real
    reference binary vs this build's `dw` drop-in, same files, same flags).  Learned
    weights are noisy SGD estimates of the same optimum, so the yardstick is the
    reference's own run-to-run spread (parity item 4): the reference is run
    twice (16 and 5 Hogwild threads: different interleavings and seeds), and this build
    must sit as close to a reference run as the reference runs sit to each other."""
    import subprocess
    from oracle import binding as orc
    if not orc.have_reference():
        pytest.skip("oracle/_ref/dw not present")
    V, N = 1_000_000, 200
    raw = synthetic.cfg3(V, n_weights=10_000, seed=2024)
    args = ["-l", "40", "-i", str(N), "--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"]
    dw = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sampler_amd", "csrc", "dw")
    q = raw.var_role == 0

    def read(o):
        w = np.array([float(l.split()[1]) for l in open(os.path.join(o, "inference_result.out.weights.text"))])
        return w, _parse_marginals(open(os.path.join(o, "inference_result.out.text")).read(), V)

    with tempfile.TemporaryDirectory() as d:
        binary_format.write_graph(raw, d)
        outs = [os.path.join(d, n) for n in ("ref_a", "ref_b", "mine")]
        for o in outs:
            os.makedirs(o)
        orc.run_reference_dw(d, args + ["-t", "16", "-c", "1"], outs[0])
        orc.run_reference_dw(d, args + ["-t", "5", "-c", "1"], outs[1])
        r = subprocess.run([dw, "gibbs", "-m", d + "/graph.meta", "-v", d + "/graph.variables",
                            "-w", d + "/graph.weights", "-f", d + "/graph.factors", "-o", outs[2], "-q",
                            "--seed", "5"] + args, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        (wa, pa), (wb, pb), (wm, pm) = [read(o) for o in outs]

    def spread(w0, p0, w1, p1):
        z = stats.z_scores_two_sample(p1[q], N, p0[q], N)
        pbar = 0.5 * (p1[q] + p0[q])
        z = z[(pbar > 0.1) & (pbar < 0.9)]
        return dict(dw_mean=abs(w0.mean() - w1.mean()), dw_std=(w0 - w1).std(),
                    corr=np.corrcoef(w0, w1)[0, 1], dp_mean=abs(p0[q].mean() - p1[q].mean()),
                    z_mean=abs(z.mean()), z_var=z.var())

    ref, mine = spread(wa, pa, wb, pb), spread(wa, pa, wm, pm)
    print("reference vs reference:", ref)
    print("this build vs reference:", mine)
    # ~1000 factors (~500 evidence) per weight; a variable's 10 weights share the MLE
    # logit(0.7)/2 = 0.42, i.e. 0.042 per weight on average
    assert 0.02 < wm.mean() < 0.07 and 0.02 < wa.mean() < 0.07
    assert np.array_equal(np.isnan(pa), np.isnan(pm)) and np.array_equal(~np.isnan(pm), q)
    assert mine["dw_mean"] < 1.5 * ref["dw_mean"] + 0.005
    assert mine["dw_std"] < 1.5 * ref["dw_std"] + 0.01
    assert mine["corr"] > ref["corr"] - 0.15
    # (two reference runs differ by 0.002-0.005 in the mean marginal and 0.05-0.11 in the mean
    # z-score from one invocation to the next -- their thread interleaving is not reproducible;
    # this build's distance to a reference run is 0.004-0.006 / 0.10-0.13: the additive terms
    # keep the comparison from failing on a lucky pair of reference runs)
    assert mine["dp_mean"] < 1.5 * ref["dp_mean"] + 0.006
    assert mine["z_mean"] < 1.5 * ref["z_mean"] + 0.15
    assert 0.9 < mine["z_var"] < 1.5 * ref["z_var"] + 0.3
