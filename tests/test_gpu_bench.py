"""bench.py contract on the GPU box: one JSON line with the required keys; the launch
path the driver uses for N > 1 (python -m torch.distributed.run ... bench.py) works and
the RCCL code path (forced at world size 1) gives the same kind of line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data", "config", "roofline"]


def _check(out, n_gpus):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["value"] > 0 and d["higher_is_better"] is True
    rf = d["roofline"]
    # every fraction of the line IS a fraction: the headline one comes from PMC bytes (or, off the
    # profiled size, from the layout byte model -- a lower bound of the traffic); SURVEY 8(d)'s
    # model, which the layout undercuts, is kept aside under its own name
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and 0 < rf["frac_layout"] <= 1.0
    assert rf["frac_survey_model"] > 0 and rf["layout_bytes_per_var"] < rf["survey_bytes_per_var"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert (rf["traffic"] is None) == (rf["traffic_note"] is not None)
    assert "workload" in d["config"]
    assert d["repeats"] >= 1 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert d["dtype"].startswith("f64")
    return d


@pytest.mark.gpu
def test_bench_single_process_small():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--vars-per-gpu", "200000", "--cpu-sample-vars", "20000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check(r.stdout, 1)
    # like for like: the reference timed on the very graph the GPU number is quoted on, and on the
    # bounded sample of the same generator next to it
    assert d["cpu_baseline"]["value"] and d["cpu_baseline"]["kind"] in ("reference", "port")
    assert d["cpu_baseline"]["sample_is_smaller"] is False and "V=200000" in d["cpu_baseline"]["sample"]
    assert d["cpu_baseline_small"]["value"] and d["cpu_baseline_small"]["sample_is_smaller"] is True
    # the second ceiling: the weight-sorted stream (20 000 weights: the graph has sorted super-tiles),
    # calibrated on this box by tools/sorted_bench
    sec = d["roofline"]["secondary"]
    assert d["roofline"]["kernel"].startswith("sorted_sweep_kernel")
    assert sec["bound"] == "cu_vector_memory" and sec["reference_loop_rate"] > 0 and 0 < sec["ratio_to_reference_loop"] < 1.5
    assert "this run" in sec["reference_loop_source"] and "peak" not in sec
    assert d["roofline"]["traffic"] is None      # (not the profiled size: no PMC figure, the layout model speaks)
    # (200k variables: a block of 3 steps takes 0.2 ms, so the repeat cap ends the run, not --min-time)
    assert d["rccl_ranks"] is None and (d["timed_seconds"] >= 0.4 or d["repeats"] == 200)


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_with_rccl_path():
    env = dict(os.environ, DWX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29617",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--vars-per-gpu", "200000", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _check(r.stdout, 1)
    assert d["rccl_ranks"] == 1 and d["launcher"] == "torch.distributed.run"
    assert d["allreduce_calls_per_step"] >= 1 and d["allreduce_ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_cfg5b_workload_runs_the_pairwise_kernels_and_the_collective_path():
    """--workload cfg5b at one rank (forced through the RCCL path): the 3b mix, several colours,
    sweep_kernel instead of sweep8_kernel; the halo itself needs a second GPU."""
    env = dict(os.environ, DWX_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29619")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg5b", "--steps", "3",
                        "--warmup", "1", "--vars-per-gpu", "200000", "--no-cpu-baseline", "--min-time", "0.05"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _check(r.stdout, 1)
    assert d["config"]["colours"] >= 2 and "cfg5b" in d["config"]["workload"]
    assert d["roofline"]["kernel"].startswith("sweep_kernel") and d["roofline"]["launches_per_sweep"] >= 2


@pytest.mark.gpu
def test_bench_cfg5b_two_ranks_rehearsal_on_one_gpu_over_gloo():
    """The multi-rank control flow of bench.py with the product engine -- self-launch, global
    mini-batch plan, gradient all-reduce per mini-batch, per-shard config-5b generator, halo lists
    of the C ABI, ghost refresh after every sweep, per-block MAX over ranks -- with two ranks
    stacked on the one GPU of the test box over gloo (RCCL refuses two ranks per device; the line
    names its backend, nobody will mistake it for a measurement)."""
    env = dict(os.environ, DWX_BENCH_BACKEND="gloo", DWX_BENCH_STACK_ON_GPU0="1",
               DWX_BENCH_SKIP_DEVICE_COUNT_CHECK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for wl in ("cfg5b", "cfg3"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", wl,
                            "--steps", "3", "--warmup", "1", "--vars-per-gpu", "200000", "--min-time", "0.05"],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        d = _check(r.stdout, 2)
        assert d["rccl_ranks"] == 2 and d["backend"] == "gloo" and d["launcher"] == "self"
        assert d["allreduce_calls_per_step"] >= 1
        if wl == "cfg5b":
            # the V/8 + 3 offset of a 400k-variable graph crosses the block boundary for 50k
            # variables each way, the small offsets for a handful
            assert d["config"]["ghost_variables_per_gpu"] > 50_000
            # (three chain exchanges per step, one BIT per boolean boundary value, sent + received)
            assert d["halo_ms_per_step"] > 0 and 3 * 2 * 50_000 // 8 <= d["halo_bytes_per_step"] < 4 * 3 * 50_000
        else:
            assert d["config"]["ghost_variables_per_gpu"] == 0 and d["halo_bytes_per_step"] == 0
