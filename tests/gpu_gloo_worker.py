"""Worker of tests/test_gpu_dist.py::test_two_ranks_share_one_gpu_over_gloo: one rank of the
sharded epoch driver with the PRODUCT engine (HipEngine: real kernels, raw device buffers
wrapped as torch tensors, collectives on the sampler's stream).  The single-GPU test box
cannot host two RCCL ranks, so the two ranks share cuda:0 and talk over gloo (which moves CUDA
tensors through the host); everything above the transport is what an N-GPU run executes."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dist_worker import shard_graph  # noqa: E402
from sampler_amd import dwx  # noqa: E402
from sampler_amd.dist import HipEngine, ShardedDimmWitted  # noqa: E402


class RecordingEngine(HipEngine):
    """Remembers, per learning sweep, the plan it ran with and this rank's chunk boundaries."""

    def __init__(self, sampler):
        super().__init__(sampler)
        self.record = []

    def sgd_plan(self, stepsize, force_batches=0):
        r = super().sgd_plan(stepsize, force_batches)
        self._last = (r[0], r[1], stepsize)     # (the third value returned is a diagnostic: the
        return r                                 #  smallest saturated step, not the step applied)

    def sgd_finish(self):
        batches, n_mine, eta = self._last
        self.record.append((batches, eta, self.s.sgd_chunks(n_mine).copy()))
        super().sgd_finish()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out, total, W, stepsize = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mixed = os.environ.get("DWX_TEST_MIXED") == "1"
    raw, begin = shard_graph(total, W, rank, world, 1234, mixed=mixed)
    g = dwx.Graph(raw, tile_vars=(64, 32)[rank % 2])          # different tilings per rank
    if mixed:      # the situation under test: this shard's own view differs from the graph's
        assert bool(g.info.has_categorical) == (rank % 2 == 1)
    s = dwx.GibbsSampler(g, device=0, seed=4242, reg_param=0.01, var_id_offset=begin)
    eng = RecordingEngine(s)
    drv = ShardedDimmWitted(eng, n_learning_epoch=6, n_inference_epoch=3, stepsize=stepsize, decay=0.7)
    assert drv.distributed
    if mixed:      # ... and the engines agreed on the whole graph's
        assert eng.has_categorical and eng.grad_reduced.numel() == 2 * W
        assert eng._narrow_shift is None           # (counts and truthiness-weighted sums travel as int64)
    else:
        # all-boolean all-unary shards: the gradient sums travel as counts (every contribution is
        # +-2^31: dist.HipEngine.agree) -- 16-bit ones, two per word, on a graph this small (a weight
        # has a few dozen records), 32-bit ones with DWX_NO_16BIT_ALLREDUCE; packed, checked (dwx_wait
        # fails on a sum that does not fit) and unpacked by the library (dwx_grad_pack_async)
        assert eng._narrow_shift == 31
        assert eng._narrow_bits == (32 if os.environ.get("DWX_NO_16BIT_ALLREDUCE") else 16)
    drv.learn()
    s.clear_tallies()
    drv.inference()
    order, _ = g.schedule()
    t, n = s.tallies()
    np.savez(os.path.join(out, "rank%d.npz" % rank), weights=s.weights, tallies=t,
             free=s.assignments("free"), evid=s.assignments("evid"), begin=begin, order=order,
             fixed_mask=g.fixed_point_mask(),
             batches=np.array([r[0] for r in eng.record]), eta=np.array([r[1] for r in eng.record]),
             **{"chunks%d" % i: r[2] for i, r in enumerate(eng.record)})
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
