"""`dw gibbs` over several ranks (sampler_amd/csrc/dw_multi.cc): variable-block shards
(--gpus N: int64 gradient all-reduce per mini-batch, halo exchange of boundary assignments)
and the reference's replica mode (-c N: weight averaging per round, ceil(n / N) epochs,
summed tallies; /root/reference/src/dimmwitted.cc:97-119,199-216,264-265,280-282).

CPU legs run the same host sources over the emulated library with the host-staged test
communicator (--comm host); the -m gpu legs run the product binary: two ranks stacked on the
one GPU of the test box through the host-staged communicator, and the RCCL communicator itself
with a single rank (RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import check_result
from conftest import FIXTURES, GOLDEN
from sampler_amd import binary_format, synthetic
from test_dw_cli import DW, DW_EMU, outputs, run_dw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dw_emu():
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tests", "hipemu")], check=True)
    return DW_EMU


MODES = [["--gpus", "2"], ["--gpus", "3"], ["-c", "2"]]
# CPU leg (emulated kernels, 2000-epoch fixtures): every fixture over two shards, a few more
# combinations on the side; the -m gpu leg runs every fixture in both modes
CPU_CASES = [(fx, ["--gpus", "2"]) for fx in FIXTURES] + [
    ("sparse_domains", ["--gpus", "3"]), ("partial_observation", ["--gpus", "3"]),
    ("biased_coin", ["-c", "2"]), ("sparse_domains", ["-c", "2"]), ("biased_coin_truthiness", ["-c", "3"]),
    ("biased_coin_continuous", ["-c", "2"])]


def _check(fx, mode, weights_text, marginals_text):
    """The fixture's own acceptance check -- except where its tolerance was written for -c 1:
    biased_coin_continuous under -c N runs ceil(2000 / N) learning rounds, the last step is
    0.1 * 0.995^1000 = 7e-4 instead of 4e-6 and the learned weight scatters by +-0.011 instead of
    +-0.003 (the real reference does the same: a property of its replica arithmetic; its own check
    runs -c 1).  There the yardstick is the reference's OWN spread under -c N, from eight runs of its
    replica arithmetic on the byte-pinned oracle (tests/replica_spread.py): the weight within 4 sigma
    (+0.005) of their mean, every marginal within 4 sigma (+0.02) of its mean over those runs."""
    if not (fx == "biased_coin_continuous" and mode[0] == "-c"):
        check_result.check(fx, weights_text, marginals_text)
        return
    import replica_spread
    from conftest import parse_dw_args
    d = os.path.join(GOLDEN, fx)
    o = parse_dw_args(open(os.path.join(d, "dw-args")).read())
    raw = binary_format.read_graph_dir(d)
    W, P = replica_spread.reference_replica_runs(raw, int(mode[1]), o["l"], o["i"], o["alpha"], o["diminish"], n_runs=8,
                                                 sample_evidence=o["sample_evidence"], reg_param=o["reg_param"])
    w = np.array([float(l.split()[1]) for l in weights_text.splitlines()])
    p = np.array([float(l.split()[2]) for l in marginals_text.splitlines()])
    assert np.all(np.abs(w - W.mean(axis=0)) <= 4 * W.std(axis=0) + 0.005), (w, W.mean(axis=0), W.std(axis=0))
    assert len(p) == P.shape[1]
    assert np.all(np.abs(p - P.mean(axis=0)) <= 4 * P.std(axis=0) + 0.02), (p, P.mean(axis=0), P.std(axis=0))


@pytest.mark.parametrize("fx,mode", CPU_CASES, ids=lambda x: x if isinstance(x, str) else "".join(x))
def test_fixtures_pass_the_reference_checks_over_several_ranks(dw_emu, fx, mode):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, fx, out, ["--quiet", "--seed", "3", "--comm", "host"] + mode)
        assert r.returncode == 0, r.stderr
        _check(fx, mode, *outputs(out))


@pytest.mark.parametrize("fx,n", [("biased_coin", 2), ("biased_coin_with_multinomial", 3), ("biased_coin_truthiness", 5)])
def test_shards_of_a_unary_graph_reproduce_the_single_rank_run_byte_for_byte(dw_emu, fx, n):
    """No factor crosses a block: the ranks draw with Philox counters of GLOBAL variable ids
    and sum integer gradients, so N shards = one rank, whatever N (mini-batch plan off: the
    multi-rank plan works with the global curvature and may cut differently)."""
    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        common = ["--quiet", "--seed", "11", "--step_cap", "0"]
        r1 = run_dw(dw_emu, fx, a, common)
        r2 = run_dw(dw_emu, fx, b, common + ["--gpus", str(n), "--comm", "host"])
        assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
        assert outputs(a) == outputs(b)


@pytest.mark.parametrize("mode", ["throw", "block"])
def test_a_rank_that_fails_mid_epoch_takes_the_run_down_within_seconds(dw_emu, mode):
    """VERDICT r02 #7b / ADVICE: a rank that throws in the middle of learning must not leave
    `dw --gpus N` hanging.  DWX_DW_TEST_FAULT makes rank 1 throw at the start of its second
    learning epoch: the other ranks are released from their host barriers (HostAgree::abort) and
    collectives (Comm::abort; ncclCommAbort under RCCL) and the run ends with exit code 1 and the
    first error on stderr; with `block` the other ranks sit in a call that never returns -- what a
    collective without its dead peer does -- and only the watchdog can end the run, 5 s after the
    failure."""
    import time
    env = dict(os.environ, DWX_DW_TEST_FAULT="1:1" + (":block" if mode == "block" else ""))
    d = os.path.join(GOLDEN, "biased_coin")
    with tempfile.TemporaryDirectory() as out:
        cmd = [dw_emu, "gibbs", "-m", d + "/graph.meta", "-w", d + "/graph.weights", "-v", d + "/graph.variables",
               "-f", d + "/graph.factors", "-o", out, "-l", "50", "-i", "10", "--alpha", "0.1", "-q",
               "--gpus", "3", "--comm", "host"]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=60)
        dt = time.time() - t0
    assert r.returncode == 1, (r.returncode, r.stderr)
    assert "rank 1: injected fault" in r.stderr
    assert dt < (12.0 if mode == "block" else 5.0), dt
    if mode == "block":
        assert "did not stop within 5 s" in r.stderr and dt > 4.0


def test_a_shard_with_a_single_tile_runs_the_split_plan_of_its_peers(dw_emu):
    """three shards of 230 variables, the first with 40 factors per variable
    (several tiles), the others with one (ONE tile per colour launch each); three weights tied to
    thousands of factors and a large step force a split plan.  The one-tile shards used to fall
    back to plan_batches = 1 -- other collectives than their peers, whole-sweep counts applied at
    chunk 0, a host-staged sum reading past a short buffer.  Now a forced batch count is honoured
    whatever the shard's size: the run ends -- under ASan / UBSan too -- with finite, regularised
    weights.  (They are not the single-rank run's: tied this heavily every batch's update is the
    equilibrium of THAT batch, the weights after a sweep are those of its last batch, and the two
    runs cut their sweeps differently.)"""
    from sampler_amd.rawgraph import RawGraph, FUNC_ISTRUE, DTYPE_BOOLEAN
    rng = np.random.default_rng(3)
    V, per = 690, 230
    deg = np.where(np.arange(V) < per, 40, 1)
    F = int(deg.sum())
    owner = np.repeat(np.arange(V, dtype=np.uint64), deg)
    wid = (owner % 3).astype(np.uint64)
    role = np.ones(V, np.uint8)
    val = (rng.random(V) < np.array([0.8, 0.3, 0.6])[np.arange(V) % 3]).astype(np.uint64)
    raw = RawGraph(var_role=role, var_init_value=val, var_dtype=np.full(V, DTYPE_BOOLEAN, np.uint16),
                   var_cardinality=np.full(V, 2, np.uint64), fac_func=np.full(F, FUNC_ISTRUE, np.uint16),
                   fac_edge_offset=np.arange(F + 1, dtype=np.uint64), fac_weight_id=wid,
                   fac_feature_value=np.ones(F), edge_vid=owner, edge_equal_to=np.ones(F, np.uint64),
                   w_initial_value=np.zeros(3), w_is_fixed=np.zeros(3, np.uint8))
    args = ["-l", "30", "-i", "0", "--alpha", "0.05", "--diminish", "0.95", "--reg_param", "0.01", "--seed", "5"]
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    with tempfile.TemporaryDirectory() as d:
        files = _write(raw, d)
        for exe, env in ((dw_emu, None),
                         (dw_emu + "_asan", dict(os.environ, LD_PRELOAD=libasan,
                                                 ASAN_OPTIONS="detect_leaks=0:abort_on_error=1"))):
            with tempfile.TemporaryDirectory() as out:
                r = subprocess.run([exe, "gibbs"] + files + ["-o", out] + args + ["--gpus", "3", "--comm", "host"],
                                   capture_output=True, text=True, timeout=600, env=env)
                assert r.returncode == 0, r.stderr[-3000:]
                assert "batches=" in r.stdout, r.stdout[-2000:]      # the plan was split
                w = np.array([float(l.split()[1]) for l in outputs(out)[0].splitlines()])
                assert np.all(np.isfinite(w)) and np.abs(w).max() < 1.0, w


def test_gradient_travels_as_narrow_counts_where_the_graph_allows_it(dw_emu):
    """All-boolean all-unary blocks (biased_coin: ISTRUE, f = 1: every contribution is +-2^31):
    the ranks agree on the shift and all-reduce counts -- 16-bit ones packed two per word while the
    sum over all ranks of a weight's records stays below 2^15, else 32-bit (dwx_grad_pack_async); a graph with
    categorical variables keeps the int64 vector.  Either way the result files are the single
    rank's (test_shards_of_a_unary_graph_reproduce_the_single_rank_run_byte_for_byte)."""
    for fx, want, env in (("biased_coin", "16-bit counts, shift 31", None),
                          ("biased_coin", "32-bit counts, shift 31", dict(os.environ, DWX_NO_16BIT_ALLREDUCE="1")),
                          ("biased_coin_with_multinomial", "int64 sums", None)):
        with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
            common = ["--seed", "11", "--step_cap", "0", "-l", "150", "-i", "50"]    # (last value wins)
            r1 = run_dw(dw_emu, fx, a, common + ["--quiet"])
            r2 = run_dw(dw_emu, fx, b, common + ["--gpus", "2", "--comm", "host"], env=env)
            assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
            assert "Gradient all-reduce: " + want in r2.stdout, r2.stdout[-1500:]
            assert outputs(a) == outputs(b)


def _write(raw, d):
    binary_format.write_graph(raw, d)
    return ["-m", d + "/graph.meta", "-w", d + "/graph.weights", "-v", d + "/graph.variables", "-f", d + "/graph.factors"]


def test_cross_shard_factors_equal_the_lockstep_emulation_of_the_python_driver(dw_emu):
    """Pairwise factors across the block boundary (config 5b shape): the C++ shard builder, halo
    lists, pack / exchange / unpack and collective order against an independent implementation
    -- sampler_amd.shard.make_shard + CPU oracles stepped in lockstep (tests/test_halo_gloo.py).
    Same seed, same flags: the two result files must agree to the last printed digit."""
    import halo_worker as hw
    from test_halo_gloo import _lockstep
    from sampler_amd.dwx import fmt_g
    total, world = 640, 2
    engines = _lockstep(total, world)
    with tempfile.TemporaryDirectory() as d, tempfile.TemporaryDirectory() as out:
        files = _write(hw.build(total), d)
        r = subprocess.run([dw_emu, "gibbs"] + files + ["-o", out, "-l", str(hw.N_LEARN), "-i", str(hw.N_INFER),
                                                        "--alpha", str(hw.STEP), "--diminish", str(hw.DECAY),
                                                        "--reg_param", str(hw.REG), "--seed", str(hw.SEED),
                                                        "--step_cap", "0", "-q", "--gpus", str(world), "--comm", "host"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
    assert w == "".join("%d %s\n" % (j, fmt_g(x)) for j, x in enumerate(engines[0].o.weights))
    want = []
    raw = hw.build(total)
    for e in engines:
        t = e.o.tallies[:e.n_owned]
        for i in range(e.n_owned):
            v = e.begin + i
            if raw.var_role[v] == 0:
                want.append("%d 1 %s\n" % (v, fmt_g(float(t[i]) / hw.N_INFER)))
    assert m == "".join(want) and len(want) > 100


def test_categorical_chain_across_shards_equals_the_lockstep_emulation(dw_emu):
    """The same comparison for categorical variables whose pairwise factors cross the block
    boundaries (synthetic.cfg4b: a chain, every block edge is crossed): ghost values travel as
    bytes (dwx_halo_message_bytes), update counts are dynamic ([G|T] all-reduced), the tiles are
    staged edge-parallel -- three ranks."""
    import halo_worker as hw
    from test_halo_gloo import _lockstep
    from sampler_amd import synthetic
    from sampler_amd.dwx import fmt_g
    total, world, card = 330, 3, 5

    def build(n):
        return synthetic.cfg4b(n, card=card, seed=21)

    engines = _lockstep(total, world, build=build)
    raw = build(total)
    with tempfile.TemporaryDirectory() as d, tempfile.TemporaryDirectory() as out:
        files = _write(raw, d)
        r = subprocess.run([dw_emu, "gibbs"] + files + ["-o", out, "-l", str(hw.N_LEARN), "-i", str(hw.N_INFER),
                                                        "--alpha", str(hw.STEP), "--diminish", str(hw.DECAY),
                                                        "--reg_param", str(hw.REG), "--seed", str(hw.SEED),
                                                        "--step_cap", "0", "-q", "--gpus", str(world), "--comm", "host"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
    assert w == "".join("%d %s\n" % (j, fmt_g(x)) for j, x in enumerate(engines[0].o.weights))
    want = []
    for e in engines:
        t = e.o.tallies
        for i in range(e.n_owned):
            v = e.begin + i
            if raw.var_role[v] == 0:
                for k in range(card):
                    want.append("%d %d %s\n" % (v, k, fmt_g(float(t[i * card + k]) / hw.N_INFER)))
    assert m == "".join(want) and len(want) > 100


def test_replicas_follow_the_reference_epoch_arithmetic(dw_emu):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "7", "-i", "5", "-a", "0.1", "-c", "2", "--comm", "host"])
        assert r.returncode == 0, r.stderr
        # 7 epochs over 2 copies = 4 rounds of 2 (src/dimmwitted.cc:280-282), printed as ranges
        assert r.stdout.count("LEARNING EPOCH") == 4 and "LEARNING EPOCH 6~7" in r.stdout
        assert r.stdout.count("INFERENCE EPOCH") == 3 and "INFERENCE EPOCH 4~5" in r.stdout
        assert "2 x replica" in r.stdout
        _, m = outputs(out)
        # 3 rounds x 2 copies = 6 samples per query variable: marginals are multiples of 1/6
        vals = [float(l.split()[2]) for l in m.strip().splitlines()]
        assert all(abs(v * 6 - round(v * 6)) < 1e-4 for v in vals), vals


def test_multi_rank_errors_are_loud(dw_emu):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "1", "-q", "--gpus", "2"])
        assert r.returncode == 1 and "--gpus 2 but 1 HIP device(s) visible" in r.stderr
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "1", "-q", "--gpus", "2", "--devices", "0,0"])
        assert r.returncode == 1 and "no RCCL" in r.stderr           # the emulated build has none
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "1", "-q", "--gpus", "2", "--comm", "mpi"])
        assert r.returncode != 0 and "--comm" in r.stderr
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "1", "-q", "--gpus", "2", "--comm", "host",
                                                     "--devices", "0"])
        assert r.returncode == 1 and "one device per rank" in r.stderr
    import torch
    if not torch.cuda.is_available():
        with tempfile.TemporaryDirectory() as out:
            r = run_dw(DW, "biased_coin", out, args=["-l", "1", "-i", "1", "-q", "--gpus", "2"])
            assert r.returncode == 1 and "--gpus 2 but 0 HIP device(s) visible" in r.stderr


def test_make_shard_progress_output_names_the_ranks(dw_emu):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "sparse_domains", out, args=["-l", "2", "-i", "2", "--gpus", "2", "--comm", "host"])
        assert r.returncode == 0, r.stderr
        assert "2 x shard" in r.stdout and "host-staged" in r.stdout
        assert r.stdout.count("LEARNING EPOCH") == 2 and "TOTAL INFERENCE TIME" in r.stdout


# ------------------------------------------------------------------------ sharded loader
def _write_graph(tmp, recs, n_vars, n_weights, split=None, n_factors=None, n_edges=None, raw=None):
    """A boolean graph (every third variable evidence) whose factor file(s) hold `recs`
    ([(func, [(vid, equal_to), ...], wid, fval)]); returns the `dw gibbs` file arguments."""
    import struct
    from test_text2bin import _factor_bytes
    n_factors = len(recs) if n_factors is None else n_factors
    n_edges = sum(len(r[1]) for r in recs) if n_edges is None else n_edges
    open(os.path.join(tmp, "graph.meta"), "w").write("%d,%d,%d,%d" % (n_weights, n_vars, n_factors, n_edges))
    open(os.path.join(tmp, "graph.variables"), "wb").write(
        b"".join(struct.pack(">QBQHQ", v, v % 3 == 0, v % 2, 0, 2) for v in range(n_vars)))
    open(os.path.join(tmp, "graph.weights"), "wb").write(
        b"".join(struct.pack(">QBd", w, 0, 0.0) for w in range(n_weights)))
    cmd = ["-m", os.path.join(tmp, "graph.meta"), "-v", os.path.join(tmp, "graph.variables"),
           "-w", os.path.join(tmp, "graph.weights")]
    cuts = [0] + (split or []) + [len(recs)]
    for i in range(len(cuts) - 1):
        fn = os.path.join(tmp, "graph.factors.%d" % i)
        open(fn, "wb").write(raw if raw is not None else _factor_bytes(recs[cuts[i]:cuts[i + 1]]))
        cmd += ["-f", fn]
    return cmd


def _dw_files(binary, files, out, extra, env=None):
    cmd = [binary, "gibbs"] + files + ["-o", out] + extra
    return subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, **(env or {})))


def test_sharded_loader_equals_make_shard(dw_emu):
    """`dw gibbs --gpus N` decodes the factor files once, straight into the ranks' shards
    (load_factors_sharded) -- no whole-graph factor columns.  DWX_DW_VERIFY_SHARDS compares every
    column of every shard with load_factors + make_shard: fixtures, and a graph of mixed arities
    (fixed-stride file, hop-cut file, look-alike file, > 64 k records = several pieces) whose
    pairwise and ternary factors cross the block boundaries."""
    import random
    for fx, n in [("biased_coin", 3), ("partial_observation", 2), ("sparse_domains", 3),
                  ("biased_coin_with_multinomial", 4), ("categorical_noise_aware", 2)]:
        if fx not in FIXTURES:
            continue
        with tempfile.TemporaryDirectory() as out:
            r = subprocess.run(
                [dw_emu, "gibbs"] + _fixture_files(fx) + ["-o", out, "-l", "2", "-i", "2", "--gpus", str(n), "--comm", "host"],
                capture_output=True, text=True, env=dict(os.environ, DWX_DW_VERIFY_SHARDS="1"))
            assert r.returncode == 0, r.stderr
            assert "%d shards equal make_shard" % n in r.stdout, fx
    rnd = random.Random(11)
    V, W = 3000, 50

    def rec(arity):
        v0 = rnd.randrange(V)
        vs = [v0] + [(v0 + rnd.choice([1, 7, 101, V // 3 + 3])) % V for _ in range(arity - 1)]
        return (rnd.choice([0, 1, 2, 3]) if arity > 1 else 4, [(v, 1) for v in vs], rnd.randrange(W), 1.0)

    recs = ([rec(1) for _ in range(70_000)] + [rec(rnd.randint(1, 3)) for _ in range(3000)] +
            [rec(1)] + [rec(3), rec(3), rec(3)] * 7 + [rec(1)] * 20 + [rec(2) for _ in range(2000)])
    split = [70_000, 73_000, 73_000 + 42]
    for n, exe, env in ((2, dw_emu, {}), (3, dw_emu, {}), (7, dw_emu, {}), (3, dw_emu + "_asan", _asan_env())):
        with tempfile.TemporaryDirectory() as t:
            files = _write_graph(t, recs, V, W, split=split)
            r = _dw_files(exe, files, t, ["-l", "1", "-i", "1", "--gpus", str(n), "--comm", "host"],
                          env=dict(env, DWX_DW_VERIFY_SHARDS="1", DWX_HOST_THREADS="8"))
            assert r.returncode == 0, r.stderr[-3000:]
            assert "%d shards equal make_shard" % n in r.stdout


def _asan_env():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    return {"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1"}


@pytest.mark.parametrize("asan", [False, True])
def test_sharded_loader_rejects_malformed_files(dw_emu, asan):
    """The errors of load_factors (tests/test_text2bin.py), through `dw gibbs --gpus 2`; plain and
    under ASan/UBSan (a rejected file must not have written past a column)."""
    import random
    from test_text2bin import _factor_bytes
    if asan:
        dw_emu = dw_emu + "_asan"
    asan_env = dict(_asan_env(), DWX_HOST_THREADS="8") if asan else None
    rnd = random.Random(6)
    recs = [(4, [(rnd.randrange(8), 1)], 0, 1.0) for _ in range(100)]
    blob = _factor_bytes(recs)
    extra = ["-l", "1", "-i", "1", "-q", "--gpus", "2", "--comm", "host"]
    with tempfile.TemporaryDirectory() as t:      # cut in the middle of a record
        r = _dw_files(dw_emu, _write_graph(t, recs, 8, 4, raw=blob[:-5]), t, extra, asan_env)
        assert r.returncode == 1 and "truncated" in r.stderr
    with tempfile.TemporaryDirectory() as t:      # fewer records than graph.meta announces
        r = _dw_files(dw_emu, _write_graph(t, recs, 8, 4, n_factors=101, n_edges=101), t, extra, asan_env)
        assert r.returncode == 1 and "factor count" in r.stderr
    with tempfile.TemporaryDirectory() as t:      # more records than graph.meta announces
        r = _dw_files(dw_emu, _write_graph(t, recs, 8, 4, n_factors=99, n_edges=99), t, extra, asan_env)
        assert r.returncode == 1 and "count" in r.stderr
    with tempfile.TemporaryDirectory() as t:      # an arity field pointing far past the file
        bad = bytearray(blob)
        bad[42 * 50 + 2:42 * 50 + 10] = (1 << 40).to_bytes(8, "big")
        r = _dw_files(dw_emu, _write_graph(t, recs, 8, 4, raw=bytes(bad)), t, extra, asan_env)
        assert r.returncode == 1 and "truncated" in r.stderr
    with tempfile.TemporaryDirectory() as t:      # a variable id the meta file does not know
        bad = recs[:50] + [(4, [(8, 1)], 0, 1.0)] + recs[51:]
        r = _dw_files(dw_emu, _write_graph(t, bad, 8, 4), t, extra, asan_env)
        assert r.returncode == 1 and "unknown variable" in r.stderr


def _fixture_files(fx):
    d = os.path.join(GOLDEN, fx)
    cmd = ["-m", os.path.join(d, "graph.meta"), "-w", os.path.join(d, "graph.weights"),
           "-v", os.path.join(d, "graph.variables"), "-f", os.path.join(d, "graph.factors")]
    if os.path.exists(os.path.join(d, "graph.domains")):
        cmd += ["--domains", os.path.join(d, "graph.domains")]
    return cmd


# ------------------------------------------------------------------------ GPU box
@pytest.mark.gpu
@pytest.mark.parametrize("mode", [["--gpus", "2"], ["-c", "2"]], ids=lambda m: "".join(m))
@pytest.mark.parametrize("fx", FIXTURES)
def test_product_dw_two_ranks_on_one_gpu(fx, mode):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(DW, fx, out, ["--quiet", "--seed", "3", "--comm", "host", "--devices", "0,0"] + mode)
        assert r.returncode == 0, r.stderr
        _check(fx, mode, *outputs(out))


@pytest.mark.gpu
def test_product_dw_shards_equal_single_rank_and_rccl_path_with_one_rank():
    env = dict(os.environ, DWX_DW_FORCE_MULTI="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for fx in ("biased_coin", "biased_coin_with_multinomial"):
        with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b, tempfile.TemporaryDirectory() as c:
            common = ["--quiet", "--seed", "11", "--step_cap", "0"]
            r1 = run_dw(DW, fx, a, common)
            r2 = run_dw(DW, fx, b, common + ["--gpus", "2", "--comm", "host", "--devices", "0,0"])
            assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
            assert outputs(a) == outputs(b)
            # one rank through the RCCL communicator (ncclCommInitAll, ncclAllReduce on the
            # sampler's stream): the same files again
            d = os.path.join(GOLDEN, fx)
            cmd = [DW, "gibbs", "-m", d + "/graph.meta", "-w", d + "/graph.weights", "-v", d + "/graph.variables",
                   "-f", d + "/graph.factors", "-o", c] + open(d + "/dw-args").read().split() + common + ["--gpus", "1"]
            if os.path.exists(d + "/graph.domains"):
                cmd += ["--domains", d + "/graph.domains"]
            r3 = subprocess.run(cmd, capture_output=True, text=True, env=env)
            assert r3.returncode == 0, r3.stderr
            assert outputs(a) == outputs(c)


@pytest.mark.gpu
def test_product_dw_cross_shard_graph_at_size_matches_single_gpu_statistically():
    """200k variables of the config-5b mix over two ranks (dense halo between them) vs one rank:
    different scan orders and one-sweep-stale ghosts, so the comparison is statistical."""
    import stats
    raw = synthetic.cfg5b(200_000, 2_000, seed=5)
    N = 100
    args = ["-l", "10", "-i", str(N), "--alpha", "0.01", "--diminish", "0.95", "-q", "--seed", "9"]
    res = []
    with tempfile.TemporaryDirectory() as d:
        files = _write(raw, d)
        for extra in ([], ["--gpus", "2", "--comm", "host", "--devices", "0,0"]):
            with tempfile.TemporaryDirectory() as out:
                r = subprocess.run([DW, "gibbs"] + files + ["-o", out] + args + extra, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
                w, m = outputs(out)
                res.append((np.array([float(l.split()[1]) for l in w.splitlines()]),
                            np.array([float(l.split()[2]) for l in m.splitlines()])))
    (w1, p1), (w2, p2) = res
    assert len(p1) == len(p2) > 90_000
    # (2 000 weights x 1 000 factors after ten epochs at 0.01 are mostly SGD noise: two runs of the
    # REFERENCE on this graph -- other thread counts -- correlate at 0.30 .. 0.35, sigma 0.10 per
    # weight; this build's batched updates average more, so its two runs agree better than that)
    assert abs(w1.mean() - w2.mean()) < 0.01 and np.corrcoef(w1, w2)[0, 1] > 0.5
    # marginals computed with two noisy sets of learned weights: four runs of the REFERENCE on this
    # graph (-t 2 / 3 / 5 / 8) differ pairwise by 0.001 .. 0.011 in the mean marginal and by
    # D = 0.017 .. 0.063 in the two-sample KS distance of the marginals; one rank against two ranks
    # of this build must not be further apart than the reference is from itself
    from scipy.stats import ks_2samp
    assert abs(p1.mean() - p2.mean()) < 0.012
    assert ks_2samp(p1, p2).statistic < 0.065


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg5a", "cfg5b"])
def test_config5_eight_way_split_stacked_on_one_gpu(workload):
    """BASELINE config 5's decomposition at eight ranks x 1 M variables each, every rank on the ONE GPU
    of the test box (`dw gibbs --gpus 8 --devices 0,0,0,0,0,0,0,0 --comm host`: the sharded loader, the
    global mini-batch plan, the narrow gradient counts, eight-peer halo lists, the block-ordered dump --
    everything of the 8-GPU path except xGMI), against the single-rank run on the same files
    (tools/stacked8.py; the full-size runs -- 100 M variables -- are logged under profiles/r04/):
    5a (10 unary factors per variable, no halo): byte-identical result files with one mini-batch per
    sweep, statistically equal under the ranks' own global plan; 5b (6 unary + 4 pairwise, dense halo):
    within the reference's own run-to-run spread (the criteria of
    test_product_dw_cross_shard_graph_at_size_matches_single_gpu_statistically).
    Semantics held: src/dimmwitted.cc:199-216,264-265."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as logs:
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "stacked8.py"), "--workload", workload,
                            "--vars", "8000000", "--weights", "800000", "--learn", "5", "--infer", "20",
                            "--cfg3b-generator", "--log-dir", logs], capture_output=True, text=True, timeout=800)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["pass"] and d["ranks"] == 8 and d["stacked_cap0"]["marginal_lines"] == d["single_cap0"]["marginal_lines"] > 3_900_000
    if workload == "cfg5a":
        assert d["cap0_byte_identical"]
        assert "16-bit counts" in d["stacked_plan"]["gradient_allreduce"]
    else:
        assert d["stacked_plan"]["gradient_allreduce"] == "int64 sums"
