"""GPU leg of the multi-GPU plumbing (run by tests/test_gpu_dist.py with world size 1 on
the single-GPU test box): HipEngine wraps the sampler's raw device buffers as torch
tensors, runs RCCL all-reduces on the sampler's own HIP stream, and must leave exactly
the state a plain sampler reaches."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sampler_amd import dwx, synthetic  # noqa: E402
from sampler_amd.dist import HipEngine, ShardedDimmWitted  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    raw = synthetic.cfg3(200_000, n_weights=5000, seed=5)
    g = dwx.Graph(raw)
    a = dwx.GibbsSampler(g, seed=31)
    b = dwx.GibbsSampler(g, seed=31)
    eng = HipEngine(a)
    # the wrapped tensors alias the sampler's buffers
    assert eng.grad.dtype == torch.int64 and eng.grad.numel() == 2 * 5000 and eng.grad.is_cuda
    t0 = eng.t_static.clone()
    drv = ShardedDimmWitted(eng, 4, 3, 0.05, 0.9)
    drv.distributed = True                      # force the collective path at world size 1
    eng.allreduce_static_counts()
    assert torch.equal(eng.t_static, t0) and int(t0.sum()) > 0
    drv.learn()
    drv.inference()
    cur = 0.05
    for _ in range(4):
        b.sample_sgd(cur); cur *= 0.9
    for _ in range(3):
        b.sample()
    b.wait()
    assert np.array_equal(a.weights, b.weights) and np.abs(a.weights).max() > 0
    assert np.array_equal(a.assignments("evid"), b.assignments("evid"))
    assert np.array_equal(a.tallies()[0], b.tallies()[0])
    assert int(eng.grad.abs().sum()) == 0       # apply cleared the accumulators
    # halo plumbing: int32 views of the assignment buffers in device order
    ev = eng.assign_tensor("evid")
    pos = eng.positions(np.arange(10, dtype=np.uint64))
    assert ev.dtype == torch.int32 and ev.numel() == 200_000
    assert np.array_equal(ev[pos].cpu().numpy().astype(np.uint64), a.assignments("evid")[:10])
    ev[pos] = 1 - ev[pos]                       # write through the view = what a halo scatter does
    torch.cuda.synchronize()
    # the halo lists of the C ABI (dwx_halo_*): gather both chains of 1000 scattered variables
    # into the list's device buffer on the sampler's stream, change them, scatter them back
    ids = np.arange(13, 200_000, 199, dtype=np.uint64)[:1000]    # (not the ten variables flipped above)
    h = eng.halo_list(ids)
    # boolean variables travel as one bit each, a chain's block padded to 8 bytes
    assert h.n == 1000 and h.message_bytes(1) == 128 and h.message_bytes(3) == 256
    assert h.tensor.numel() == 64 and h.tensor.dtype == torch.int32 and h.message(2).numel() == 32

    def bits(words):
        return np.unpackbits(words.cpu().numpy().view(np.uint8), bitorder="little")[:1000].astype(np.uint64)

    fr0, ev0 = a.assignments("free")[ids.astype(np.int64)], a.assignments("evid")[ids.astype(np.int64)]
    h.pack(3); a.wait()
    assert np.array_equal(bits(h.tensor[:32]), fr0)
    assert np.array_equal(bits(h.tensor[32:]), ev0)
    h.pack(2); a.wait()                         # one chain: its values come first
    assert np.array_equal(bits(h.tensor[:32]), ev0)
    with eng.stream_context():
        h.tensor[:32] = ~h.tensor[:32]
    h.unpack(2); a.wait()
    assert np.array_equal(a.assignments("evid")[ids.astype(np.int64)], 1 - ev0)
    assert np.array_equal(a.assignments("free")[ids.astype(np.int64)], fr0)
    assert np.array_equal(a.assignments("evid")[:10], 1 - b.assignments("evid")[:10])
    # split learning sweeps through the collective path: per-chunk static counts are sized and
    # "summed" (world size 1) once per batch count, every chunk is accumulate -> all-reduce ->
    # apply; must equal the library's own split sweep
    e2, f2 = dwx.GibbsSampler(g, seed=41), dwx.GibbsSampler(g, seed=41)
    eng2 = HipEngine(e2)
    drv2 = ShardedDimmWitted(eng2, 3, 0, 0.01, 0.9)
    drv2.distributed = True
    eng2.allreduce_static_counts()
    assert 1 < e2.sgd_plan(0.01)[0] <= 64 and e2.sgd_plan(0.01 * 0.81)[0] > 1
    drv2.learn()
    assert eng2._shared_levels and not eng2._dynamic_counts
    cur = 0.01
    for _ in range(3):
        f2.sample_sgd(cur); cur *= 0.9
    f2.wait()
    assert np.array_equal(e2.weights, f2.weights) and np.abs(e2.weights).max() > 0
    assert np.array_equal(e2.assignments("free"), f2.assignments("free"))
    # replica mode at world size 1: the f64 weight all-reduce and the int32 tally all-reduce
    # run on the sampler's stream over the raw device buffers; averaging over 1 replica
    # must leave the state of a plain sampler
    from sampler_amd.dist import ReplicatedDimmWitted
    c, d = dwx.GibbsSampler(g, seed=77), dwx.GibbsSampler(g, seed=77)
    rep = ReplicatedDimmWitted(HipEngine(c), 3, 4, 0.05, 0.9)
    rep.world = 2                                # force the collective path (world size is 1)
    rep.n_learning_rounds, rep.n_inference_rounds = 3, 4
    rep.learn()
    rep.inference()
    t, n = rep.marginals()
    cur = 0.05
    for _ in range(3):
        d.sample_sgd(cur); cur *= 0.9
    d.clear_tallies()
    for _ in range(4):
        d.sample()
    d.wait()
    assert np.array_equal(c.weights, d.weights) and np.abs(c.weights).max() > 0
    td, nd = d.tallies()
    assert np.array_equal(t, td) and np.array_equal(n, 2 * nd)
    dist.destroy_process_group()
    print("gpu dist plumbing ok")


if __name__ == "__main__":
    main()
