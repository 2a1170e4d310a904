"""Multi-process path on CPU (gloo, world_size 2): the sharded epoch driver with an
oracle-backed engine.  Two variable-block shards + one int64 gradient all-reduce per
learning sweep must reproduce, bit for bit, a single process sweeping the union graph:
weights identical on both ranks and equal to the single-process weights; per-variable
state equal to the corresponding block of the single-process state."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import binding as orc
from sampler_amd.rawgraph import RawGraph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _concat(shards):
    """Union of variable-block shards that share one weight table."""
    cols = {}
    voff = foff = eoff = 0
    parts = {k: [] for k in ("var_role", "var_init_value", "var_dtype", "var_cardinality", "fac_func",
                             "fac_weight_id", "fac_feature_value", "edge_vid", "edge_equal_to")}
    offs = [np.zeros(1, np.uint64)]
    for g in shards:
        for k in parts:
            a = getattr(g, k)
            parts[k].append(a + np.uint64(voff) if k == "edge_vid" else a)
        offs.append(g.fac_edge_offset[1:] + np.uint64(eoff))
        voff += g.num_variables; foff += g.num_factors; eoff += g.num_edges
    cols = {k: np.concatenate(v) for k, v in parts.items()}
    return RawGraph(fac_edge_offset=np.concatenate(offs), w_initial_value=shards[0].w_initial_value,
                    w_is_fixed=shards[0].w_is_fixed, **cols)


@pytest.mark.parametrize("mixed", [False, True])
def test_two_rank_sharded_learning_equals_single_process(mixed):
    """mixed: rank 1's block is all categorical, rank 0's all boolean (tests/dist_worker.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import shard_graph as _shard_graph

    def shard_graph(*a):
        return _shard_graph(*a, mixed=mixed)
    total, W, world = 1200, 40, 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_RANK=str(r), DWX_TEST_MIXED="1" if mixed else "0")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"),
                                           out, str(total), str(W)], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
    # weights bit-identical on every rank (no broadcast needed)
    assert np.array_equal(res[0]["weights"], res[1]["weights"])
    # single process over the union graph
    shards = [shard_graph(total, W, r, world, 1234)[0] for r in range(world)]
    union = _concat(shards)
    o = orc.Oracle(union, reg_param=0.01)
    order = np.arange(union.num_variables, dtype=np.uint64)
    off = np.array([0, union.num_variables], np.uint64)
    sweep, cur = 0, 0.05
    for _ in range(6):
        o.sched_sample_sgd(order, off, 4242, sweep, cur); sweep += 1; cur *= 0.9
    o.clear_tallies()
    for _ in range(4):
        o.sched_sample(order, off, 4242, sweep); sweep += 1
    assert np.array_equal(res[0]["weights"], o.weights)
    assert np.abs(o.weights).max() > 0
    row0 = 0
    for r in range(world):
        b = int(res[r]["begin"]); n = len(res[r]["free"]); rows = len(res[r]["tallies"])
        assert np.array_equal(res[r]["free"], o.assignments("free")[b:b + n])
        assert np.array_equal(res[r]["evid"], o.assignments("evid")[b:b + n])
        assert np.array_equal(res[r]["tallies"], o.tallies[row0:row0 + rows])
        row0 += rows


def test_shard_range_covers_everything():
    from sampler_amd.dist import shard_range
    for total in (1, 7, 100, 12_500_001):
        for world in (1, 2, 3, 8):
            per = (total + world - 1) // world
            if per * (world - 1) >= total:
                # some rank would own nothing (9 variables over 4 ranks: blocks of 3, the last
                # empty): refused loudly instead of running a rank that skips collectives
                with pytest.raises(ValueError):
                    [shard_range(total, k, world) for k in range(world)]
                continue
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert all(e > b for b, e in r)


def test_two_rank_split_sweeps_agree_on_the_plan_and_match_single_process():
    """Mini-batched learning across ranks: steps above a threshold are cut into 4 batches,
    rank 1 can only cut into 3 (fewer tiles) and idles through the 4th collective; once the
    step has decayed below the threshold the sweep is un-split and the driver stops
    negotiating.  Result = one process running the same chunk sequence on the union graph."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import shard_graph
    total, W, world, thr = 1200, 40, 2, 0.07
    port = _free_port()
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_RANK=str(r), DWX_TEST_SPLIT_ABOVE=str(thr))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"),
                                           out, str(total), str(W)], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(res[0]["weights"], res[1]["weights"])
    shards = [shard_graph(total, W, r, world, 1234)[0] for r in range(world)]
    union = _concat(shards)
    o = orc.Oracle(union, reg_param=0.01)
    nv = [g.num_variables for g in shards]
    base = [0, nv[0]]
    max_chunks = (4, 3)
    order = np.arange(union.num_variables, dtype=np.uint64)
    sweep, cur, n_split = 0, 0.05, 0
    for _ in range(6):
        split = cur * world > thr          # the driver plans for the whole graph's step
        n_split += split
        n_chunks = 4 if split else 1
        for c in range(n_chunks):
            parts = []
            for r in range(world):
                n_r = min(4 if split else 1, max_chunks[r])
                if c < n_r:
                    parts.append(order[base[r] + nv[r] * c // n_r:base[r] + nv[r] * (c + 1) // n_r])
            sl = np.concatenate(parts)
            o.sched_accumulate(sl, np.array([0, len(sl)], np.uint64), 4242, sweep)
            if split or c + 1 == n_chunks:
                o.sched_apply(cur)
        sweep += 1; cur *= 0.9
    assert 0 < n_split < 6                 # both regimes and the transition were exercised
    o.clear_tallies()
    all_order, off = order, np.array([0, union.num_variables], np.uint64)
    for _ in range(4):
        o.sched_sample(all_order, off, 4242, sweep); sweep += 1
    assert np.array_equal(res[0]["weights"], o.weights)
    for r in range(world):
        b = int(res[r]["begin"]); n = len(res[r]["free"])
        assert np.array_equal(res[r]["free"], o.assignments("free")[b:b + n])
        assert np.array_equal(res[r]["evid"], o.assignments("evid")[b:b + n])
    # the ranks negotiate per batch COUNT (curvature of 1, 2 and 4 batches: three rounds for
    # six sweeps), never per sweep, and every sweep runs with the agreed count
    plans = res[0]["plans"]
    assert int(res[0]["curv_calls"]) == 3 and len(plans) == 6
    assert [int(p[1]) for p in plans] == [4] * n_split + [1] * (6 - n_split)


def test_two_rank_replicas_average_weights_like_n_datacopy():
    """Replica mode (the reference's -c n_datacopy): both ranks hold the whole graph and
    sample with their own seeds; after every learning round the weights are summed and
    halved (fixed weights untouched); 7 requested epochs = 4 rounds; tallies add up.  Must
    equal two oracles run in lockstep in one process."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from replica_worker import make_graph
    world = 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "replica_worker.py"), out],
                                          env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(res[0]["weights"], res[1]["weights"])
    assert np.array_equal(res[0]["tallies"], res[1]["tallies"])
    raw = make_graph()
    os_ = [orc.Oracle(raw, reg_param=0.01) for _ in range(world)]
    order = np.arange(raw.num_variables, dtype=np.uint64)
    off = np.array([0, raw.num_variables], np.uint64)
    fixed = raw.w_is_fixed.astype(bool)
    sweep, cur = 0, 0.05
    for _ in range(4):
        for r, o in enumerate(os_):
            o.sched_sample_sgd(order, off, 900 + r, sweep, cur)
        avg = np.where(fixed, os_[0].weights, (os_[0].weights + os_[1].weights) / 2)
        for o in os_:
            o.weights[:] = avg
        sweep += 1; cur *= 0.9
    for o in os_:
        o.clear_tallies()
    for _ in range(3):
        for r, o in enumerate(os_):
            o.sched_sample(order, off, 900 + r, sweep)
        sweep += 1
    assert np.array_equal(res[0]["weights"], os_[0].weights) and np.abs(os_[0].weights).max() > 0
    assert np.array_equal(res[0]["tallies"], os_[0].tallies + os_[1].tallies)
    assert np.all(res[0]["nsamples"] == 6)
