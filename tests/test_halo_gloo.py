"""Variable-block shards WITH cross-shard factors: ghost variables + halo exchange
(SURVEY.md §8e, config 5b), world size 2 over gloo on CPU.  The 2-process run must equal,
bit for bit, an in-process lockstep emulation (two shard oracles stepped together with
ghost values copied between sweeps), and both shards must agree on the weights."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

from oracle import binding as orc
from sampler_amd import dwx
from sampler_amd.dist import shard_range
from sampler_amd.shard import make_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_make_shard_structure():
    import halo_worker as hw
    raw = hw.build(400)
    b, e = 100, 250
    local, ghosts = make_shard(raw, b, e)
    assert local.num_ghost_variables == len(ghosts) > 0
    assert ((ghosts < b) | (ghosts >= e)).all()
    n_owned = e - b
    # every kept factor touches an owned variable; every owned variable keeps all its factors
    ev = local.edge_vid.astype(np.int64)
    off = local.fac_edge_offset.astype(np.int64)
    for f in range(local.num_factors):
        assert (ev[off[f]:off[f + 1]] < n_owned).any()
    deg_local = np.bincount(ev[ev < n_owned], minlength=n_owned)
    gv = raw.edge_vid.astype(np.int64)
    deg_global = np.bincount(gv, minlength=raw.num_variables)[b:e]
    assert np.array_equal(deg_local, deg_global)
    # the compiled graph samples exactly the owned variables
    g = dwx.Graph(local)
    order, launch_off = g.schedule()
    assert sorted(order.tolist()) == list(range(n_owned))
    assert g.info.num_owned_variables == n_owned and g.info.num_values == n_owned


def _lockstep(total, world, build=None):
    """Two shard oracles in ONE process, stepped together; ghosts copied between sweeps."""
    import halo_worker as hw
    raw = (build or hw.build)(total)
    bounds = [shard_range(total, k, world) for k in range(world)]
    engines, ghosts = [], []
    for k in range(world):
        local, gh = make_shard(raw, *bounds[k])
        engines.append(hw.OracleShardEngine.__new__(hw.OracleShardEngine))
        eng = engines[-1]
        eng.o = orc.Oracle(local, reg_param=hw.REG)
        g = dwx.Graph(local)
        eng.order_local, eng.off = g.schedule()
        eng.begin = bounds[k][0]
        eng.n_owned = local.num_variables - local.num_ghost_variables
        eng.sweep = 0
        ghosts.append(gh.astype(np.int64))

    def exchange(chains):
        for chain in chains:
            glob = np.concatenate([e.o.assignments(chain)[:e.n_owned] for e in engines])
            for e, gh in zip(engines, ghosts):
                e.o.assignments(chain)[e.n_owned:] = glob[gh]

    exchange(("free", "evid"))
    cur = hw.STEP
    for _ in range(hw.N_LEARN):
        for e in engines:
            e.sgd_plan(cur)
            e.sgd_accumulate(0)
        total_grad = sum(e.o.grad.copy() for e in engines)
        for e in engines:
            e.o.grad[:] = total_grad
            e.sgd_apply()
            e.sgd_finish()
        exchange(("free", "evid"))
        cur *= hw.DECAY
    for e in engines:
        e.o.clear_tallies()
    for _ in range(hw.N_INFER):
        for e in engines:
            e.sample()
        exchange(("evid",))
    return engines


def test_two_rank_halo_exchange_equals_lockstep_emulation():
    total, world = 640, 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "halo_worker.py"),
                                           out, str(total)], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]
    assert np.array_equal(res[0]["weights"], res[1]["weights"])
    assert np.abs(res[0]["weights"]).max() > 0
    assert int(res[0]["n_boundary"]) > 0 and int(res[1]["n_boundary"]) > 0
    engines = _lockstep(total, world)
    for r, e in enumerate(engines):
        n = e.n_owned
        assert np.array_equal(res[r]["weights"], e.o.weights)
        assert np.array_equal(res[r]["free"], e.o.assignments("free")[:n])
        assert np.array_equal(res[r]["evid"], e.o.assignments("evid")[:n])
        assert np.array_equal(res[r]["tallies"], e.o.tallies[:n])
    assert sum(int(r_["tallies"].sum()) for r_ in res) > 0
