#!/usr/bin/env python3
"""bench.py -- variables sampled per second of the Gibbs sweep hot path on MI355X.

Workloads (BASELINE.json; `config.workload` names the one that ran):
  cfg3  (default)  10M boolean variables x 10 unary ISTRUE factors each, 1M learnable
        weights, 50 % evidence -- the graph north_star quotes its target on.  With --gpus N > 1
        it is config 5a: the graph sharded by variable block, 12.5M variables per GPU, weights
        global, no cross-shard factor (empty halo).
  cfg5b the 3b mix (6 unary + 4 pairwise EQUAL factors per variable, offsets 1, 7, 101 and
        V/8 + 3) over variable-block shards: factors cross the shard boundaries, ghost
        variables are refreshed by a halo exchange after every sweep (dense halo from the last
        offset).  At one GPU it is config 3b.
One STEP = one learning sweep (sample_sgd: two chains + weight SGD; over N GPUs one all-reduce
of the int64 gradient vector per mini-batch) followed by one inference sweep (sample), i.e.
2 V variables sampled per step and GPU, counted as the reference's own `vars/sec` print counts
them (src/dimmwitted.cc:152,232).  Inputs are resident in HBM before timing starts.

Launch: `python bench.py --gpus N --steps K --warmup W`.  Under torch.distributed.run (the
driver's way for N > 1) every rank reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; WITHOUT it
(`WORLD_SIZE` unset) and N > 1 this process starts the N ranks itself -- fresh child processes,
started before anything here touches a GPU -- and fails loudly if the node has fewer than N
devices.  Timing: W warm-up steps, then blocks of EXACTLY K steps, each bracketed by a
barrier + synchronize on both sides, repeated until 0.5 s of timed work has accumulated
(--min-time); a block's time is the MAX over ranks; `value` comes from the MEDIAN block, the
spread is reported next to it.

Prints ONE JSON line (rank 0).  See DESIGN.md §6 for every field.
"""
import argparse
import json
import os
import re
import socket
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# SURVEY.md §8(d)'s byte model per variable (inference, learning): a 16-byte factor record + 4-byte
# index + 4-byte weight per factor.  The layout that ships moves a third of that (8-byte weight-
# sorted records, weights out of L2), so this model is NOT a lower bound of the traffic any more:
# it is reported as `frac_survey_model` only.
SURVEY_BYTES = {"cfg3": (340.0, 404.0), "cfg5b": (564.0, 684.0)}
RECORDS_PER_VAR = {"cfg3": 10, "cfg5b": 14}
# the kernel sources whose sha256 stamps profiles/traffic.json (PMC bytes per launch): a kernel
# change must not silently keep an old counter figure
KERNEL_SOURCES = ("sweep_kernels.h", "tile_walk.h", "aux_kernels.h", "persist_kernels.h", "device_types.h")


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "sampler_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def layout_bytes_per_var(wl, learn, info, V, Vq, W, sorted_path):
    """Minimum HBM bytes per visited variable of the layout that ships ('layout
    byte model'): every datum the dominant sweep kernel must move once -- a LOWER bound of the
    PMC traffic (no re-fetched weight lines, no partial sectors).
    cfg3 / sorted_sweep_kernel: 8-byte weight-sorted records (only records with d != 0 exist),
      the per-variable words (v_meta, v_orig, v_row [+ v_init when learning]: 4 B each), the
      assignment store(s) (4 B per chain written), learning: two ballot words per wave of 64
      (0.25 B), inference: the tally read-modify-write (8 B per line-resident u32), and the f32
      weight table once per XCD L2 (8 x 4 W bytes per launch).
    cfg3 / sweep8_kernel (no sorted copy): the same with 8-byte variable-major records + 4-byte row pointers.
    cfg5b / sweep_kernel: 16-byte records (14 per variable), per-variable words, the useful 4 bytes
      of each neighbour assignment gathered (8 per variable and chain), stores; gradient atomics
      and ballots are left out (lower bound)."""
    units = V if learn else Vq
    if wl == "cfg3":
        n_rec = float(info.num_sorted_records) / V if (sorted_path and info.num_sorted_records) else float(RECORDS_PER_VAR[wl])
        b = 8.0 * n_rec + (16.0 if learn else 12.0) + (8.0 if learn else 4.0)
        b += 0.25 if learn else 8.0
        if not sorted_path:
            b += 4.0
        b += 8.0 * 4.0 * W / max(units, 1)
        return b
    n_rec = float(info.num_index_entries) / V
    chains = 2 if learn else 1
    return 16.0 * n_rec + (16.0 if learn else 12.0) + 4.0 * chains + 4.0 * 8 * chains + (0.0 if learn else 8.0)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["cfg3", "cfg5b"], default="cfg3")
    ap.add_argument("--min-time", type=float, default=0.5,
                    help="repeat the timed block of --steps steps until this many seconds are timed")
    ap.add_argument("--max-repeats", type=int, default=200)
    ap.add_argument("--vars-per-gpu", type=int, default=0,
                    help="override the per-GPU variable count (default 10M at 1 GPU, 12.5M else)")
    ap.add_argument("--cpu-sample-vars", type=int, default=1_000_000)
    ap.add_argument("--cpu-small-only", action="store_true",
                    help="CPU baseline on the bounded sample only (skip the run on the bench's own 10 M-variable graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repeat-infer", action="store_true",
                    help="skip the untimed repeated-inference leg (profiling runs: keeps the kernel stats clean)")
    ap.add_argument("--no-gather-ceiling", action="store_true",
                    help="skip tools/sorted_bench / gather_bench --ceiling (roofline.secondary.reference_loop_rate is then the committed figure)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / JSON plumbing only, over gloo on CPU: no GPU, no sampler")
    ap.add_argument("--weights", type=int, default=0, help="override the weight count (experiments)")
    ap.add_argument("--tile-vars", type=int, default=0, help="graph-compile knob (experiments)")
    ap.add_argument("--tile-edges", type=int, default=0, help="graph-compile knob (experiments)")
    ap.add_argument("--wide-records", action="store_true",
                    help="graph-compile knob (experiments): 16-byte records even though the graph is all-unary")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- launcher
def self_launch(args, argv):
    """--gpus N without a launcher: start N ranks as fresh children (one process per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and pass rank 0's JSON line
    through.  Nothing in this process has touched the GPU (device_count() does not initialise
    it on this image); no process that did is ever re-executed."""
    n = args.gpus
    if not args.dry_run and not os.environ.get("DWX_BENCH_SKIP_DEVICE_COUNT_CHECK"):   # (test hook)
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.exit("bench.py: --gpus %d but this node has %d visible GPU(s); refusing to measure fewer "
                     "GPUs than asked for" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DWX_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        # a rank that dies takes the others down with it (they would wait in a collective forever)
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    log("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code))
                    for q in alive:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


# --------------------------------------------------------------------------- CPU baseline
def _reference_epochs(raw, stepsize, decay, reg, n_learn, n_infer, timeout):
    """The real reference binary (oracle/_ref/dw) on `raw`, all host cores: per-epoch times as
    the reference prints them (src/dimmwitted.cc:150-154, 229-236) -> (learn[], infer[])."""
    from oracle import binding as orc
    from sampler_amd import binary_format
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        binary_format.write_graph(raw, d)
        out = orc.run_reference_dw(d, ["-l", str(n_learn), "-i", str(n_infer), "--alpha", str(stepsize),
                                       "--diminish", str(decay), "--reg_param", str(reg)],
                                   d, quiet=False, timeout=timeout)
    tl = [float(x) for x in re.findall(r"LEARNING EPOCH[^\n]*?\.\.\.\.([0-9.eE+-]+) sec\.", out)]
    ti = [float(x) for x in re.findall(r"INFERENCE EPOCH[^\n]*?\.\.\.\.([0-9.eE+-]+) sec\.", out)]
    return tl, ti


def _port_epochs(raw, stepsize, decay, reg, n_learn, n_infer):
    """The oracle's threaded restatement of the reference (the fallback when the reference
    binary did not travel with the repo)."""
    from oracle import binding as orc
    o = orc.Oracle(raw, reg_param=reg)
    o.set_workers(os.cpu_count() or 1)
    tl, ti = [], []
    cur = stepsize
    for _ in range(n_learn):
        t0 = time.perf_counter(); o.sample_sgd(cur, threaded=True); tl.append(time.perf_counter() - t0)
        cur *= decay
    for _ in range(n_infer):
        t0 = time.perf_counter(); o.sample(threaded=True); ti.append(time.perf_counter() - t0)
    return tl, ti


def cpu_quota_cores():
    """CPUs' worth of CPU time the cgroup grants this process (cpu.max), or None: the box reports its 256
    hardware threads whatever the quota, and the reference spawns a thread for each."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(raw, stepsize, decay, reg, what, smaller, timeout=420):
    """Time the CPU side on `raw` -- 3 learning + 10 inference epochs on all host cores, the
    epoch loops the step of this bench stands for (src/dimmwitted.cc:143-154, 191-236) -- with the
    real reference binary when it travelled with the repo, else the oracle's restatement.
    value = 2V / (median learning epoch + median inference epoch): the bench's own metric."""
    from oracle import binding as orc
    n_vars = raw.num_variables
    cores = os.cpu_count() or 1
    sample = what + "; 3 learning + 10 inference epochs, all host cores; value = 2V / (median learn epoch + median inference epoch)"
    extra = {"sample_is_smaller": bool(smaller)}
    if cpu_quota_cores() is not None:
        extra["cpu_quota_cores"] = cpu_quota_cores()   # (`cores` threads share this much CPU time)
    if smaller:
        extra["note"] = ("a smaller graph than the GPU's (its tables sit higher in the CPU's caches: "
                         "conservative for the CPU); the GPU/CPU ratio is not like-for-like")
    kind = "reference" if orc.have_reference() else "port"
    tl, ti = (_reference_epochs(raw, stepsize, decay, reg, 3, 10, timeout) if kind == "reference" else ([], []))
    if not (tl and ti):
        kind = "port"
        tl, ti = _port_epochs(raw, stepsize, decay, reg, 3, 10)
    t = statistics.median(tl) + statistics.median(ti)
    return dict({"value": 2.0 * n_vars / t, "unit": "variables/s", "cores": cores, "kind": kind,
                 "sample": sample, "learn_vars_per_sec": n_vars / statistics.median(tl),
                 "infer_vars_per_sec": n_vars / statistics.median(ti)}, **extra)


def sorted_ceiling():
    """tools/sorted_bench --ceiling: the sorted sweep's own access shape -- a weight-sorted
    8-byte record stream, the gathers of neighbouring weights out of a 4 MB table, a 64-bit LDS
    atomic per record, a trivial draw per variable -- on one full round of 256 super-tiles of 16 384
    variables (the product kernel's shape: one 1024-thread workgroup per CU).  -> records per second, or None."""
    exe = os.path.join(ROOT, "tools", "sorted_bench")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--ceiling"], capture_output=True, text=True, timeout=180)
        for line in r.stdout.splitlines():
            if line.startswith('{"nv"'):
                return float(json.loads(line)["records_per_s"])
    except Exception as e:                  # the ceiling must never lose the measurement
        log("bench.py: sorted_bench failed: %r" % (e,))
    return None


def gather_ceiling():
    """tools/gather_bench --ceiling (built by __graft_entry__.build): the sweep's own access
    shape -- an 8-byte record stream + one random 4-byte gather per record out of a 4 MB table,
    3 waves per SIMD -- and nothing else.  -> gathers per second, or None."""
    exe = os.path.join(ROOT, "tools", "gather_bench")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "--ceiling"], capture_output=True, text=True, timeout=120)
        for line in r.stdout.splitlines():
            if line.startswith('{"mode"'):
                return float(json.loads(line)["gathers_per_s"])
    except Exception as e:                  # the ceiling must never lose the measurement
        log("bench.py: gather_bench failed: %r" % (e,))
    return None


# --------------------------------------------------------------------------- one rank
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args, sys.argv[1:])        # never returns

    # stdout carries exactly ONE JSON line.  Libraries that print to the process's stdout
    # (RCCL's version banner: NCCL_DEBUG=VERSION is exported on the GPU boxes) are sent to
    # stderr: descriptor 1 becomes a copy of descriptor 2, the result goes to the saved one.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d: launch one rank per GPU "
                 "(torch.distributed.run --nproc-per-node %d, or no launcher at all)"
                 % (args.gpus, world, args.gpus))
    n_gpus = world
    # DWX_BENCH_FORCE_DIST=1 runs the multi-GPU code path (process group, RCCL
    # all-reduces, barriers) even at world size 1 -- a self-check for single-GPU boxes
    use_dist = world > 1 or os.environ.get("DWX_BENCH_FORCE_DIST") == "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")

    if args.dry_run:
        # the launcher, the rendezvous and the result line, nothing else (CPU, gloo)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        ranks = int(t.item())
        dist.barrier()
        if rank == 0:
            out = {"metric": "variables sampled/sec (whole node)", "value": None, "unit": "variables/s",
                   "n_gpus": n_gpus, "rccl_ranks": ranks, "steps": args.steps, "warmup": args.warmup,
                   "dry_run": True, "backend": "gloo",
                   "launcher": "self" if os.environ.get("DWX_BENCH_SELF_LAUNCHED") else "external",
                   "config": {"workload": args.workload}}
            os.write(result_fd, (json.dumps(out) + "\n").encode())
        dist.destroy_process_group()
        return

    from sampler_amd import dwx, synthetic
    from sampler_amd.dist import CommTiming, HaloExchange, HipEngine, ShardedDimmWitted, shard_range

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the dwx sampler has no CPU fallback")
    if torch.cuda.device_count() <= local_rank and os.environ.get("DWX_BENCH_STACK_ON_GPU0") != "1":
        sys.exit("bench.py: rank %d needs GPU %d but only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    # DWX_BENCH_BACKEND=gloo + DWX_BENCH_STACK_ON_GPU0=1: a rehearsal of the multi-rank control
    # flow (plan agreement, collectives, halo exchange) on a box with ONE GPU, which RCCL
    # cannot do (it refuses two ranks on one device); tests only -- the line says so
    backend = os.environ.get("DWX_BENCH_BACKEND", "nccl")
    if os.environ.get("DWX_BENCH_STACK_ON_GPU0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    rccl_ranks = 1
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.ones(1, device="cuda")
        dist.all_reduce(t)
        rccl_ranks = int(t.item())          # how many ranks RCCL actually joined
        assert rccl_ranks == dist.get_world_size() == world

    wl = args.workload
    V = args.vars_per_gpu or (10_000_000 if n_gpus == 1 else 12_500_000)
    W = args.weights or (1_000_000 if V >= 1_000_000 else max(1, V // 10))
    stepsize, decay, reg = 0.001, 0.95, 0.01       # SURVEY.md §8(d) config-3 run flags
    t0 = time.time()
    ghosts, bounds = None, None
    if wl == "cfg3":
        raw = synthetic.cfg3(V, n_weights=W, seed=1234, shard=rank)
    else:
        bounds = [shard_range(V * n_gpus, k, n_gpus) for k in range(n_gpus)]
        raw, ghosts = synthetic.cfg5b_shard(V * n_gpus, bounds[rank][0], bounds[rank][1], W, seed=1234)
    graph = dwx.Graph(raw, tile_vars=args.tile_vars, tile_edges=args.tile_edges,
                      no_compact_records=1 if args.wide_records else 0)
    # (default options: the learning sweeps' own weight-sorted layouts are built with their plan level,
    # on the device -- what a `dw gibbs` run gets too)
    sampler = dwx.GibbsSampler(graph, device=local_rank, reg_param=reg, seed=20260103,
                               var_id_offset=rank * V)
    if rank == 0:
        log("setup: %s V/GPU=%d W=%d tiles=%d colours=%d ghosts=%d device_bytes=%.2f GB (%.1f s)"
            % (wl, V, W, graph.info.num_tiles, graph.info.num_colors, raw.num_ghost_variables,
               graph.info.device_bytes / 1e9, time.time() - t0))
    # (kept for the like-for-like CPU baseline: the reference runs on this very graph)
    raw_full = raw if (n_gpus == 1 and wl == "cfg3" and not args.no_cpu_baseline) else None
    del raw
    engine = HipEngine(sampler)
    halo = None
    if ghosts is not None and use_dist and len(ghosts):
        halo = HaloExchange(engine, bounds[rank][0], bounds[rank][1], ghosts, bounds)
    drv = ShardedDimmWitted(engine, 0, 0, stepsize, decay, halo=halo)
    if use_dist and not drv.distributed:      # forced self-check at world size 1
        drv.distributed = True
        engine.agree()
        engine.allreduce_static_counts()
    if os.environ.get("DWX_BENCH_PLAN_WORLD"):
        # experiment: one rank of an N-GPU run (its mini-batch plan and collectives, minus the
        # xGMI time): plan as if N equal shards contributed to every weight
        drv.plan_world = int(os.environ["DWX_BENCH_PLAN_WORLD"])
    engine.comm_timing = CommTiming() if use_dist else None

    drv.prepare(stepsize)         # one-off planning work, whatever --warmup says

    def step(cur):
        drv.learn_epoch(cur)      # sample_sgd (+ RCCL all-reduce of the gradient vector, + halo)
        drv.sample_epoch()        # sample (+ halo refresh of the evidence chain)

    def fence():
        engine.wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    cur = stepsize
    for _ in range(args.warmup):
        step(cur); cur *= decay
    fence()
    if engine.comm_timing is not None:
        engine.comm_timing.drain()
    sampler.kernel_time_reset(True)
    # blocks of EXACTLY --steps steps, fenced on both sides; at least one, until --min-time
    # seconds are timed.  The number of blocks must be the same on every rank: rank 0 decides.
    blocks_local, total = [], 0.0
    cur_block = cur          # every block starts from the step the warm-up left: the same --steps
    while True:              # steps of a run `-l K` each time (the mini-batch plan follows the step)
        cur = cur_block
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step(cur); cur *= decay
        fence()
        dt = time.perf_counter() - t_start
        blocks_local.append(dt)
        total += dt
        go_on = 1 if (total < args.min_time and len(blocks_local) < args.max_repeats) else 0
        if use_dist:
            flag = torch.tensor([go_on], dtype=torch.int32, device="cuda")
            dist.broadcast(flag, src=0)
            go_on = int(flag.item())
            torch.cuda.synchronize()
        if not go_on:
            break
    repeats = len(blocks_local)
    ms_i, nl_i, ns_i = sampler.kernel_time("infer")
    ms_l, nl_l, ns_l = sampler.kernel_time("learn")
    ms_p, nl_p, ns_p = sampler.kernel_time("pull")
    comm = engine.comm_timing.drain() if engine.comm_timing is not None else {}
    # outside the timed region: inference sweeps that FOLLOW EACH OTHER on unchanged weights
    # (what `dw gibbs -i N` runs) stream tabulated potential terms instead of gathering
    # weights; the step above alternates learning and inference and never gets there
    ms_r = nl_r = ns_r = 0
    if not args.no_repeat_infer and halo is None:
        sampler.kernel_time_reset(True)
        for _ in range(6):
            engine.sample()
        engine.wait()
        sampler.kernel_time_reset(True)
        for _ in range(10):
            engine.sample()
        engine.wait()
        ms_r, nl_r, ns_r = sampler.kernel_time("infer")
    # ... and what a quiet `dw gibbs -i N` runs on an all-unary graph: the N sweeps in ONE launch
    # (dwx_sample_n_async); aside too, never part of `value`
    ms_m = ns_m = 0
    N_MULTI = 100
    if not args.no_repeat_infer and halo is None and hasattr(engine, "sample_n"):
        engine.sample_n(N_MULTI); engine.wait()
        sampler.kernel_time_reset(True)
        engine.sample_n(N_MULTI)
        engine.wait()
        ms_m, _, ns_m = sampler.kernel_time("infer")
    sampler.kernel_time_reset(False)
    # per block: MAX over ranks; per rank: its own median (reported as the per-rank spread)
    blocks = list(blocks_local)
    rank_ms = [1e3 * statistics.median(blocks_local) / args.steps]
    if use_dist:
        tb = torch.tensor(blocks_local, dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        blocks = [float(x) for x in tb.tolist()]
        tr = torch.zeros(world, dtype=torch.float64, device="cuda")
        tr[rank] = rank_ms[0]
        dist.all_reduce(tr)
        rank_ms = [float(x) for x in tr.tolist()]
    w = sampler.weights
    assert np.isfinite(w).all()
    # mini-batch plan at the LARGEST step of the run: 1 = un-split sweeps
    plan_batches, _, plan_min_step = sampler.sgd_plan(stepsize)
    if use_dist or drv.plan_world:
        plan_batches = drv._plan(stepsize)[0]

    if rank == 0:
        elapsed = statistics.median(blocks)            # one block of exactly --steps steps
        total_vars = 2.0 * V * n_gpus * args.steps
        # units one launch processes: a learning launch visits every variable of the
        # block (both chains); an inference launch visits the query variables only
        # (evidence is skipped, as in the reference, but still counted in vars/sec)
        Vq = int(graph.info.num_query_variables)
        S_INFER, S_LEARN = SURVEY_BYTES[wl]
        if wl == "cfg3":
            # all-unary: the sweeps run sweep8_kernel (8-byte records) unless --wide-records
            # (8-byte records; weight-sorted super-tiles when the graph has them: sorted_sweep_kernel)
            sorted_path = int(graph.info.num_super_tiles) > 0 and not args.wide_records
            kern = "sweep_kernel" if args.wide_records else ("sorted_sweep_kernel" if sorted_path else "sweep8_kernel")
        else:
            sorted_path = False
            kern = "sweep_kernel"
        L_INFER = layout_bytes_per_var(wl, False, graph.info, V, Vq, W, sorted_path)
        L_LEARN = layout_bytes_per_var(wl, True, graph.info, V, Vq, W, sorted_path)
        # a sweep is one launch per colour: per-launch figures are per colour launch, the
        # roofline is priced per SWEEP (all its colour launches), which is what moves the bytes
        sw_l, sw_i = ms_l / max(ns_l, 1), ms_i / max(ns_i, 1)
        if sw_l >= sw_i:
            kname, per_sweep_ms, sbpv, lbpv, units, nl, ns = kern + "<LEARN=true>", sw_l, S_LEARN, L_LEARN, V, nl_l, ns_l
        else:
            kname, per_sweep_ms, sbpv, lbpv, units, nl, ns = kern + "<LEARN=false>", sw_i, S_INFER, L_INFER, Vq, nl_i, ns_i
        launches_per_sweep = max(nl / max(ns, 1), 1)
        sweep_s = per_sweep_ms * 1e-3
        # PMC bytes per launch of exactly this workload, size and KERNEL SOURCES (profiles/traffic.json,
        # written by tools/summarize_prof.py from separate --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE, the
        # guide's gfx950 correction): null when the stamp does not match the tree
        traffic, traffic_note = None, None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if not (V == 10_000_000 and W == 1_000_000 and n_gpus == 1):
            traffic_note = "no PMC figure for this size"
        elif not os.path.exists(tp):
            traffic_note = "profiles/traffic.json missing"
        else:
            try:
                tj = json.load(open(tp))
                stamp = tj.get("_stamp", {}).get("kernel_sources_sha16")
                if stamp != kernel_sources_sha16():
                    traffic_note = ("profiles/traffic.json was measured on other kernel sources (stamp %s, tree %s): "
                                    "re-run tools/profile_gpu.sh" % (stamp, kernel_sources_sha16()))
                else:
                    traffic = tj.get(wl, {}).get(kname)
            except Exception as e:
                traffic_note = "profiles/traffic.json unreadable: %r" % (e,)
        layout_gbs = lbpv * units / sweep_s / 1e9
        survey_gbs = sbpv * units / sweep_s / 1e9
        if traffic:
            achieved, source = traffic * launches_per_sweep / sweep_s / 1e9, "pmc"
        else:
            achieved, source = layout_gbs, "layout_model"
        roofline = {"bound": "hbm", "kernel": kname, "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "frac_source": ("PMC bytes per launch (profiles/traffic.json, stamped with the kernel sources' sha) / "
                                    "the launch time measured in this run" if source == "pmc" else
                                    "layout byte model (no PMC figure valid for this tree / size): a lower bound of the traffic"),
                    "traffic": traffic, "traffic_note": traffic_note,
                    "frac_layout": layout_gbs / HBM_PEAK_GBS, "layout_bytes_per_var": lbpv,
                    "frac_survey_model": survey_gbs / HBM_PEAK_GBS, "survey_bytes_per_var": sbpv,
                    "survey_model_note": "SURVEY.md 8(d) prices 32 B per record + weight gathers as HBM bytes; the 8-byte "
                                         "weight-sorted layout moves a third of that, so this figure is not a fraction of a roofline",
                    "vars_per_launch": units / launches_per_sweep,
                    "avg_launch_ms": per_sweep_ms / launches_per_sweep,
                    "launches_per_sweep": launches_per_sweep, "per_gpu": True}
        # Diagnostics, not ceilings: the same access shape with a trivial draw phase, measured on this
        # box right now (tools/sorted_bench / tools/gather_bench); the product can run ahead of it
        # on denser graphs (it read 1.14 at 100 M variables).
        if wl == "cfg3" and kern == "sorted_sweep_kernel":
            recs = RECORDS_PER_VAR[wl] * units
            peak = None if args.no_gather_ceiling else sorted_ceiling()
            committed = 3.5e11      # profiles/r03/sorted_bench.jsonl: 16 384 variables, 1024 threads, 1 per CU
            r_ach = recs / sweep_s
            roofline["secondary"] = {"bound": "cu_vector_memory", "what": "weight-sorted 8-byte record stream + "
                                     "sorted 4-byte weight gathers (about 0.2 L2 requests per record) + one LDS atomic per record",
                                     "achieved": r_ach / 1e9, "reference_loop_rate": (peak or committed) / 1e9,
                                     "unit": "Grecord/s", "ratio_to_reference_loop": r_ach / (peak or committed),
                                     "reference_loop_source": "tools/sorted_bench --ceiling, this run" if peak else
                                     "profiles/r03/sorted_bench.jsonl (committed)",
                                     "note": "a micro-benchmark of the same loop, a diagnostic and not a bound"}
        elif wl == "cfg3":
            gathers = RECORDS_PER_VAR[wl] * units
            peak = None if args.no_gather_ceiling else gather_ceiling()
            committed = 1.874e11    # profiles/r02/gather_bench.jsonl: stream mode, 4 MB, 3 waves/SIMD
            g_ach = gathers / sweep_s
            roofline["secondary"] = {"bound": "l2_req", "what": "random 4-byte weight gathers (one 128-byte L2 "
                                     "request per record) next to the 8-byte record stream",
                                     "achieved": g_ach / 1e9, "reference_loop_rate": (peak or committed) / 1e9,
                                     "unit": "Ggather/s", "ratio_to_reference_loop": g_ach / (peak or committed),
                                     "reference_loop_source": "tools/gather_bench --ceiling, this run" if peak else
                                     "profiles/r02/gather_bench.jsonl (committed)",
                                     "note": "a micro-benchmark of the same loop, a diagnostic and not a bound"}
        out = {
            "metric": "variables sampled/sec (whole node)",
            "value": total_vars / elapsed,
            "unit": "variables/s",
            "n_gpus": n_gpus, "rccl_ranks": rccl_ranks if use_dist else None, "backend": backend if use_dist else None,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "repeats": repeats,
            "ms_per_step_min": 1e3 * min(blocks) / args.steps, "ms_per_step_max": 1e3 * max(blocks) / args.steps,
            "ms_per_step_stdev": 1e3 * (statistics.pstdev(blocks) if repeats > 1 else 0.0) / args.steps,
            "timed_seconds": sum(blocks),
            "ms_per_step_per_rank_min": min(rank_ms), "ms_per_step_per_rank_max": max(rank_ms),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (f32 sampling weights)", "data": "synthetic",
            "launcher": ("self" if os.environ.get("DWX_BENCH_SELF_LAUNCHED") else
                         ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none")),
            "config": {
                "workload": ("cfg3: %d boolean vars/GPU x 10 unary ISTRUE factors, W=%d learnable "
                             "weights, 50%% evidence" % (V, W) if wl == "cfg3" else
                             "cfg5b: %d boolean vars/GPU, 6 unary ISTRUE + 4 pairwise EQUAL factors per var "
                             "(offsets 1, 7, 101, V/8+3), W=%d learnable weights, 50%% evidence" % (V, W)) +
                            "; step = 1 learning sweep (sample_sgd) + 1 inference sweep (sample)",
                "vars_per_gpu": V, "factors_per_var": 10, "weights": W,
                "parallelism": ("variable-block shards x%d, int64 gradient all-reduce per learning "
                                "mini-batch%s" % (n_gpus, ", halo exchange of boundary assignments after every sweep"
                                                 if halo is not None else "")) if n_gpus > 1 else "single GPU",
                "stepsize": stepsize, "diminish": decay, "reg_param": reg,
                "sgd_batches_per_sweep": plan_batches, "min_weight_stepsize": plan_min_step,
                "plan_layouts": "default (dwx_options.plan_layouts = 0: the learning sweeps' own weight-sorted layouts are built with their plan level, on the device)",
                "colours": int(graph.info.num_colors),
                "ghost_variables_per_gpu": int(len(ghosts)) if ghosts is not None else 0,
            },
            "roofline": roofline,
            # (both on the layout byte model: lower bounds of the traffic, so fractions of the HBM peak)
            "infer_roofline_frac": (L_INFER * Vq / (sw_i * 1e-3) / 1e9 / HBM_PEAK_GBS) if ns_i else None,
            "learn_roofline_frac": (L_LEARN * V / (sw_l * 1e-3) / 1e9 / HBM_PEAK_GBS) if ns_l else None,
            "sampled_vars_per_sec": (V + Vq) * n_gpus * args.steps / elapsed,
            "infer_vars_per_sec": V * n_gpus / (sw_i * 1e-3) if ns_i else None,
            "learn_vars_per_sec": V * n_gpus / (sw_l * 1e-3) if ns_l else None,
            "infer_kernel_ms": sw_i, "learn_kernel_ms": sw_l,
            "pull_grad_kernel_ms": (ms_p / ns_p) if ns_p else None,
            "infer_repeat_kernel_ms": (ms_r / ns_r) if ns_r else None,
            "infer_repeat_vars_per_sec": V * n_gpus / (ms_r / max(ns_r, 1) * 1e-3) if ns_r else None,
            "infer_100_in_one_launch_ms_per_sweep": (ms_m / ns_m) if ns_m else None,
            "infer_100_in_one_launch_vars_per_sec": V * n_gpus / (ms_m / max(ns_m, 1) * 1e-3) if ns_m else None,
        }
        if use_dist:
            n_steps_timed = args.steps * repeats
            ar = comm.get("allreduce", (0.0, 0, 0))
            hl = comm.get("halo", (0.0, 0, 0))
            out["allreduce_ms_per_step"] = ar[0] / n_steps_timed
            out["allreduce_calls_per_step"] = ar[1] / n_steps_timed
            out["allreduce_bytes_per_step"] = ar[2] / n_steps_timed
            out["halo_ms_per_step"] = hl[0] / n_steps_timed
            out["halo_bytes_per_step"] = hl[2] / n_steps_timed
        if n_gpus == 1 and not args.no_cpu_baseline:
            # the reference on the SAME graph the GPU number is quoted on (like for like), and on
            # the bounded 1 M-variable sample of the same generator earlier rounds reported
            for key, make, smaller in (
                    ("cpu_baseline", lambda: (raw_full, "the bench's own config-3 graph: V=%d, W=%d" % (V, W)), False),
                    ("cpu_baseline_small", lambda: (synthetic.cfg3(args.cpu_sample_vars, n_weights=max(1, args.cpu_sample_vars // 10), seed=1234),
                                                    "config-3 generator at V=%d (W=V/10)" % args.cpu_sample_vars), True)):
                if key == "cpu_baseline" and (raw_full is None or args.cpu_small_only):
                    continue
                try:
                    g_, what = make()
                    out[key] = cpu_baseline(g_, stepsize, decay, reg, what, smaller)
                except Exception as e:  # the baseline must never lose the GPU measurement
                    log("bench.py: %s failed: %r" % (key, e))
                    if key == "cpu_baseline_small":
                        out[key] = {"value": None, "unit": "variables/s", "cores": os.cpu_count(),
                                    "kind": "port", "sample": "failed: %r" % (e,)}
                    else:       # (no full-size run: the sample below becomes the baseline, and says so)
                        out["cpu_baseline_full_error"] = repr(e)
            raw_full = None
            if "cpu_baseline" not in out:      # (no full-size run: the sample is the baseline)
                out["cpu_baseline"] = out.pop("cpu_baseline_small")
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    sampler.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
