"""bench.py's own launcher (`--gpus N` with WORLD_SIZE unset): N fresh ranks, rendezvous,
one JSON line from rank 0, and loud failures -- exercised on CPU over gloo (--dry-run skips
the sampler, which has no CPU path) and, on the GPU box, against the real device count."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [1, 2, 3])
def test_self_launcher_dry_run_over_gloo(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["rccl_ranks"] == n and d["dry_run"] is True
    assert d["launcher"] == ("self" if n > 1 else "external") or n == 1


def test_world_size_mismatch_is_an_error_not_a_silent_single_gpu_run():
    env = dict(_clean_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_gpus_than_the_node_has_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    n = max(2, have + 1)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    assert "--gpus %d but this node has %d visible GPU" % (n, have) in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_dying_rank_stops_the_run():
    """A rank that fails must not leave the others waiting in a collective: here rank 1 of 2
    cannot get its GPU (there is none / only one), the launcher reports it and exits non-zero."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: both ranks would start")
    env = dict(_clean_env(), DWX_BENCH_SKIP_DEVICE_COUNT_CHECK="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--vars-per-gpu", "20000", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2 but this node has 1 visible GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
