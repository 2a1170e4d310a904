"""Worker of tests/test_halo_gloo.py: one rank of a 2-shard lattice graph (config 3b
shape: pairwise EQUAL factors cross the shard boundary) with halo exchange, gloo, CPU,
oracle-backed engine."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as orc  # noqa: E402
from sampler_amd import dwx, synthetic  # noqa: E402
from sampler_amd.dist import HaloExchange, ShardedDimmWitted, shard_range  # noqa: E402
from sampler_amd.shard import make_shard  # noqa: E402

SEED, RNG_SEED, REG = 99, 1234, 0.01
N_LEARN, N_INFER, STEP, DECAY = 5, 4, 0.05, 0.9


class OracleShardEngine:
    """HipEngine's interface over the CPU oracle; the schedule (independent sets) comes
    from the product's host-side graph compiler, which needs no GPU."""

    def __init__(self, local_raw, begin):
        self.o = orc.Oracle(local_raw, reg_param=REG)
        self.o.set_var_id_offset(0)
        g = dwx.Graph(local_raw)
        self.order_local, self.off = g.schedule()
        assert self.o.sched_check_independent(self.order_local, self.off)
        self.begin = begin
        self.n_owned = local_raw.num_variables - local_raw.num_ghost_variables
        self.sweep = 0
        self.grad = torch.from_numpy(self.o.grad)
        self._views = {c: torch.from_numpy(self.o.assignments(c).view(np.int64)) for c in ("free", "evid")}

    # Philox counters must use GLOBAL ids: owned local id + begin.  The oracle adds one
    # offset to every scheduled (= owned) variable.
    def _prep(self):
        self.o.set_var_id_offset(self.begin)

    step_cap = 0.0          # no mini-batch plan: one batch per sweep

    def allreduce_static_counts(self, group=None):
        pass

    def sgd_plan(self, stepsize, force_batches=0):
        self.eta = stepsize
        return 1, 1, stepsize

    def sgd_accumulate(self, chunk):
        self._prep()
        self.o.sched_accumulate(self.order_local, self.off, SEED, self.sweep)

    def sgd_finish(self):
        self.sweep += 1

    def allreduce_grad(self, group=None):
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def sgd_apply(self):
        self.o.sched_apply(self.eta)

    def sample(self):
        self._prep()
        self.o.sched_sample(self.order_local, self.off, SEED, self.sweep)
        self.sweep += 1

    def wait(self):
        pass

    def assign_tensor(self, chain):
        return self._views["free" if chain in (0, "free") else "evid"]

    def positions(self, local_vids):
        return torch.as_tensor(np.asarray(local_vids, np.int64))

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def halo_list(self, local_vids):
        return _HostHaloList(self, np.asarray(local_vids, np.int64))


class _HostHaloList:
    """HipEngine.halo_list's interface on the oracle's host arrays."""

    def __init__(self, engine, ids):
        self.e, self.pos, self.n = engine, torch.as_tensor(ids), len(ids)
        self.tensor = torch.zeros(2 * len(ids), dtype=torch.int64)

    def message(self, mask):
        return self.tensor[:len(self._views(mask)) * self.n]

    def _views(self, mask):
        return [self.e._views[c] for c, bit in (("free", 1), ("evid", 2)) if mask & bit]

    def pack(self, mask):
        for c, a in enumerate(self._views(mask)):
            self.tensor[c * self.n:(c + 1) * self.n] = a[self.pos]

    def unpack(self, mask):
        for c, a in enumerate(self._views(mask)):
            a[self.pos] = self.tensor[c * self.n:(c + 1) * self.n]


def build(total):
    return synthetic.cfg3b(total, n_weights=30, seed=RNG_SEED, offsets=[1, 7, 101, total // 8 + 3])


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out, total = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw = build(total)
    bounds = [shard_range(total, k, world) for k in range(world)]
    b, e = bounds[rank]
    local, ghosts = make_shard(raw, b, e)
    eng = OracleShardEngine(local, b)
    halo = HaloExchange(eng, b, e, ghosts, bounds)
    assert halo.n_boundary > 0
    drv = ShardedDimmWitted(eng, N_LEARN, N_INFER, STEP, DECAY, halo=halo)
    drv.learn()
    eng.o.clear_tallies()
    drv.inference()
    n = eng.n_owned
    np.savez(os.path.join(out, "rank%d.npz" % rank), weights=eng.o.weights,
             free=eng.o.assignments("free")[:n], evid=eng.o.assignments("evid")[:n],
             tallies=eng.o.tallies[:n], ghosts=ghosts, n_boundary=halo.n_boundary)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
