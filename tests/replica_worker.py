"""Worker of tests/test_dist_gloo.py::test_two_rank_replicas...: one rank of the replica
driver (sampler_amd.dist.ReplicatedDimmWitted) on CPU with gloo and an oracle-backed engine
(test infrastructure; the product engine is sampler_amd.dist.HipEngine)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dist_worker import OracleEngine  # noqa: E402
from randgraph import random_graph  # noqa: E402
from sampler_amd.dist import ReplicatedDimmWitted  # noqa: E402


class ReplicaOracleEngine(OracleEngine):
    def sample_sgd(self, stepsize):
        self.o.sched_sample_sgd(self.order, self.off, self.seed, self.sweep, stepsize)
        self.sweep += 1

    def clear_tallies(self):
        self.o.clear_tallies()

    def average_weights(self, group=None):
        w = torch.from_numpy(self.o.weights)          # in-place view of the oracle's weights
        fixed = self.o.weights.copy()
        dist.all_reduce(w, op=dist.ReduceOp.SUM, group=group)
        n = dist.get_world_size(group)
        keep = self.fixed_mask
        self.o.weights[:] = np.where(keep, fixed, self.o.weights / n)

    def allreduce_tallies(self, group=None):
        dist.all_reduce(torch.from_numpy(self.o.tallies.view(np.int64)), op=dist.ReduceOp.SUM, group=group)

    def tallies(self):
        return self.o.tallies.copy(), np.full(len(self.order), self.n_infer, np.uint64)


def make_graph():
    return random_graph(77, V=600, F=2500, W=30, p_cat=0.2, max_arity=2, exact_fvals=True)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw = make_graph()
    eng = ReplicaOracleEngine(raw, 900 + rank, 0, 0.01)     # whole graph, own seed
    eng.fixed_mask = raw.w_is_fixed.astype(bool)
    drv = ReplicatedDimmWitted(eng, n_learning_epoch=7, n_inference_epoch=5, stepsize=0.05, decay=0.9)
    assert drv.n_learning_rounds == 4 and drv.n_inference_rounds == 3    # ceil(n / 2)
    eng.n_infer = drv.n_inference_rounds
    drv.learn()
    drv.inference()
    t, n = drv.marginals()
    t2, _ = drv.marginals()                                   # idempotent: summed once
    assert np.array_equal(t, t2)
    np.savez(os.path.join(out, "rank%d.npz" % rank), weights=eng.o.weights, tallies=t, nsamples=n)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
