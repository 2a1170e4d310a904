import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hip_engine_rccl_plumbing_world_size_1():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_dist_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "gpu dist plumbing ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("stepsize,mixed,bits", [(0.0002, False, 16), (0.05, False, 16), (0.05, False, 32), (0.0002, True, 0), (0.02, True, 0)])
def test_two_ranks_share_one_gpu_over_gloo(stepsize, mixed, bits):
    """Two ranks of the sharded driver with the product engine (real kernels, raw device
    buffers, per-chunk static count tables shared across ranks, full or split sweeps) on one
    GPU over gloo.  Weights must be bit-identical on both ranks and equal to the CPU oracle
    replaying the same chunk sequence on the union graph; per-variable state must equal the
    oracle's block.  stepsize 0.0002: un-split sweeps; 0.05: starts split and walks down.
    bits: the gradient sums travel as 16-bit counts packed two per word (the default on a graph
    this small) or as 32-bit counts (DWX_NO_16BIT_ALLREDUCE); mixed graphs send int64.
    mixed: rank 1's block is all categorical, rank 0's all boolean -- the ranks' own views of
    "has categorical variables" differ, and both must still put the same [G | T] vector
    through every collective (they used to issue W against 2 W elements)."""

    import tempfile

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import shard_graph
    from oracle import binding as orc
    from test_dist_gloo import _concat, _free_port
    total, W, world = (12_000 if mixed else 40_000), 1500, 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as out:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                       DWX_TEST_MIXED="1" if mixed else "0")
            if bits == 32:
                env["DWX_NO_16BIT_ALLREDUCE"] = "1"
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gpu_gloo_worker.py"),
                                           out, str(total), str(W), str(stepsize)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=600)[0] for p in procs]
        for p, o in zip(procs, outs):
            assert p.returncode == 0, o[-3000:]
        res = [dict(np.load(os.path.join(out, "rank%d.npz" % r))) for r in range(world)]
    assert np.array_equal(res[0]["weights"], res[1]["weights"])
    assert np.array_equal(res[0]["batches"], res[1]["batches"]) and np.array_equal(res[0]["eta"], res[1]["eta"])
    if mixed:
        pass      # (eight weights tied to thousands of categorical variables: split at any step)
    elif stepsize > 0.01:
        assert res[0]["batches"][0] > 1 and res[0]["batches"][-1] < res[0]["batches"][0]
    else:
        assert np.all(res[0]["batches"] == 1)
    shards = [shard_graph(total, W, r, world, 1234, mixed=mixed)[0] for r in range(world)]
    union = _concat(shards)
    o = orc.Oracle(union, reg_param=0.01)
    o.set_fixed_point_mask(np.concatenate([res[r]["fixed_mask"] for r in range(world)]))
    sweep = 0
    for k in range(6):
        n_chunks = max(len(res[r]["chunks%d" % k]) for r in range(world))
        batches, eta = int(res[0]["batches"][k]), float(res[0]["eta"][k])
        for c in range(n_chunks):
            parts = []
            for r in range(world):
                off = res[r]["chunks%d" % k]
                if c < len(off):
                    parts.append(res[r]["order"][int(off[c, 0]):int(off[c, 1])] + np.uint64(res[r]["begin"]))
            sl = np.concatenate(parts)
            o.sched_accumulate(sl, np.array([0, len(sl)], np.uint64), 4242, sweep)
            if batches > 1 or c + 1 == n_chunks:
                o.sched_apply(eta)
        sweep += 1
    np.testing.assert_allclose(res[0]["weights"], o.weights, rtol=1e-12, atol=1e-12)
    assert np.abs(o.weights).max() > 0
    o.clear_tallies()
    order = np.concatenate([res[r]["order"] + np.uint64(res[r]["begin"]) for r in range(world)])
    for _ in range(3):
        o.sched_sample(order, np.array([0, len(order)], np.uint64), 4242, sweep); sweep += 1
    row0 = 0      # value rows: one per boolean variable, eight per categorical one
    for r in range(world):
        b = int(res[r]["begin"]); n = len(res[r]["free"]); rows = len(res[r]["tallies"])
        assert np.array_equal(res[r]["free"], o.assignments("free")[b:b + n])
        assert np.array_equal(res[r]["evid"], o.assignments("evid")[b:b + n])
        assert np.array_equal(res[r]["tallies"], o.tallies[row0:row0 + rows])
        row0 += rows
