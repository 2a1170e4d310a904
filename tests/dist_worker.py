"""Worker of tests/test_dist_gloo.py: one rank of the sharded epoch driver
(sampler_amd.dist.ShardedDimmWitted) on CPU with the gloo backend and an ORACLE-backed
engine (test infrastructure; the product engine is sampler_amd.dist.HipEngine)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from sampler_amd import synthetic  # noqa: E402
from sampler_amd.dist import ShardedDimmWitted, shard_range  # noqa: E402


class OracleEngine:
    """Same interface as HipEngine, state in the CPU oracle (schedule mode)."""

    def __init__(self, raw, seed, var_id_offset, reg_param, split_above=None, max_chunks=1):
        # split_above: steps larger than this are cut into 4 mini-batches (a stand-in for
        # the curvature plan of dwx_sgd_plan); max_chunks: this rank's tile count, so that
        # ranks can end up with different chunk counts
        self.split_above, self.max_chunks = split_above, max_chunks
        self.n_chunks = 1
        self.plans = []
        self.curv_calls = 0
        # what the library would report (a stand-in): one batch is fine up to the threshold
        # step, two batches are no better than one (query half / evidence half), four cut the
        # curvature by four
        self.step_cap = 1.5 if split_above is not None else 0.0
        self.max_batches = 4
        self.o = orc.Oracle(raw, reg_param=reg_param)
        self.o.set_var_id_offset(var_id_offset)
        self.order = np.arange(raw.num_variables, dtype=np.uint64)
        self.off = np.array([0, raw.num_variables], np.uint64)
        self.seed = seed
        self.sweep = 0
        self.grad = torch.from_numpy(self.o.grad)      # int64 view of [G | T]
        self.static_reduced = False

    def curvature(self, batches):
        self.curv_calls += 1
        lam1 = self.step_cap / self.split_above
        return {1: lam1, 2: lam1}.get(batches, lam1 / batches)

    def allreduce_static_counts(self, group=None):
        self.static_reduced = True                     # the oracle counts T dynamically

    def sgd_plan(self, stepsize, force_batches=0):
        self.eta = stepsize          # the driver's last call carries this rank's true step
        batches = force_batches or 1     # the driver plans (ShardedDimmWitted._plan) and forces
        self.n_chunks = min(batches, self.max_chunks)
        self.cur_batches = batches
        return batches, self.n_chunks, stepsize

    def chunk_vars(self, chunk, n_chunks=None):
        n, V = n_chunks or self.n_chunks, len(self.order)
        return self.order[V * chunk // n:V * (chunk + 1) // n]

    def sgd_accumulate(self, chunk):
        if chunk >= self.n_chunks:
            return                   # fewer chunks than the slowest rank: idle
        sl = self.chunk_vars(chunk)
        self.o.sched_accumulate(sl, np.array([0, len(sl)], np.uint64), self.seed, self.sweep)

    def sgd_finish(self):
        self.plans.append((self.eta, self.cur_batches))    # the plan this sweep ran with
        self.sweep += 1

    def allreduce_grad(self, group=None):
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)

    def sgd_apply(self):
        self.o.sched_apply(self.eta)

    def sample(self):
        self.o.sched_sample(self.order, self.off, self.seed, self.sweep)
        self.sweep += 1

    def wait(self):
        pass


def shard_graph(total_vars, n_weights, rank, world, seed, mixed=False):
    """mixed: the odd ranks' blocks hold CATEGORICAL variables only (cardinality 8, one unary
    factor per value on weights 0..7 of the shared table), the even ranks' boolean ones --
    DeepDive numbers variables per relation, so a variable-block shard can easily have no
    categorical variable while its neighbour has nothing else."""
    b, e = shard_range(total_vars, rank, world)
    if mixed and rank % 2 == 1:
        g = synthetic.cfg4(e - b, card=8, seed=seed, learn=True, shard=rank)
        g.w_initial_value = np.zeros(n_weights)
        g.w_is_fixed = np.zeros(n_weights, np.uint8)
        return g, b
    return synthetic.cfg3(e - b, n_weights=n_weights, seed=seed, shard=rank), b


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    total, W, seed = int(sys.argv[2]), int(sys.argv[3]), 4242
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw, begin = shard_graph(total, W, rank, world, 1234, mixed=os.environ.get("DWX_TEST_MIXED") == "1")
    split = os.environ.get("DWX_TEST_SPLIT_ABOVE")
    eng = OracleEngine(raw, seed, begin, 0.01, split_above=float(split) if split else None,
                       max_chunks=(4, 3)[rank % 2] if split else 1)
    drv = ShardedDimmWitted(eng, n_learning_epoch=6, n_inference_epoch=4, stepsize=0.05, decay=0.9)
    assert drv.distributed and eng.static_reduced
    drv.learn()
    eng.o.clear_tallies()
    drv.inference()
    np.savez(os.path.join(out, "rank%d.npz" % rank), weights=eng.o.weights, tallies=eng.o.tallies,
             free=eng.o.assignments("free"), evid=eng.o.assignments("evid"), begin=begin,
             plans=np.array(eng.plans, np.float64), curv_calls=eng.curv_calls)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
