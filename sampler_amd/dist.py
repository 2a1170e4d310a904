"""Multi-GPU driver: one process per GPU (torch.distributed; backend "nccl" is RCCL on
ROCm), the graph sharded by contiguous VARIABLE BLOCK across ranks.

Per learning sweep each rank samples its block and accumulates the fixed-point
weight-gradient vector [G | T]; ONE all-reduce(sum) of that int64 vector over xGMI
replaces the reference's per-epoch replica weight averaging
(src/dimmwitted.cc:209-216) and its dormant merge_gradients_from
(src/inference_result.cc:57-62); every rank then applies the identical update, so the
weights stay bit-identical on all ranks without a broadcast.  Integer sums are
order-independent, so the result does not depend on the ring order.  Inference
tallies stay sharded (no collective) and are gathered only for the dump.

The engine is abstract so that the collective logic is testable on CPU (gloo,
world_size 2) with an oracle-backed engine living in tests/.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class _CudaArray:
    """Expose a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes, typestr, itemsize):
        self.__cuda_array_interface__ = {
            "shape": (nbytes // itemsize,), "typestr": typestr, "data": (ptr, False),
            "version": 2, "strides": None,
        }


class CommTiming:
    """Device time spent in collectives, measured with events on the engine's stream (the
    stream waits for RCCL's own stream before the closing event is recorded).  Used by
    bench.py; costs two event records per collective."""

    def __init__(self):
        self.pending, self.ms, self.calls, self.bytes = [], {}, {}, {}
        self._open = {}

    def begin(self, kind, stream, nbytes=0):
        a = torch.cuda.Event(enable_timing=True)
        a.record(stream)
        self._open[kind] = a
        self.bytes[kind] = self.bytes.get(kind, 0) + nbytes

    def end(self, kind, stream):
        b = torch.cuda.Event(enable_timing=True)
        b.record(stream)
        self.pending.append((kind, self._open.pop(kind), b))

    def drain(self):
        """(call after a synchronisation) -> {kind: (total ms, calls, bytes)}; resets."""
        for kind, a, b in self.pending:
            self.ms[kind] = self.ms.get(kind, 0.0) + a.elapsed_time(b)
            self.calls[kind] = self.calls.get(kind, 0) + 1
        out = {k: (self.ms[k], self.calls[k], self.bytes.get(k, 0)) for k in self.ms}
        self.pending, self.ms, self.calls, self.bytes = [], {}, {}, {}
        return out


class HipEngine:
    """The product engine: a dwx.GibbsSampler on this rank's GPU.  Collectives run on
    the sampler's own HIP stream (made torch's current stream), so kernels and RCCL
    calls are ordered on the device without host synchronisation."""

    def __init__(self, sampler):
        from . import dwx
        self.s = sampler
        ptr, nbytes = sampler.device_buffer(dwx.BUF_GRAD)
        dev = torch.device("cuda", sampler.opts.device)
        self._holder = _CudaArray(ptr, nbytes, "<i8", 8)
        self.grad = torch.as_tensor(self._holder, device=dev)
        # [G | T]: dynamic update counts T only exist for categorical variables (boolean
        # counts are static, t_static), so an all-boolean graph reduces the G half only.  That
        # is a property of the WHOLE graph, not of this shard: agree() settles it (until then
        # the full vector travels, which is always correct)
        self.has_categorical = bool(sampler.graph.info.has_categorical)
        self.grad_reduced = self.grad
        tp, tn = sampler.device_buffer(dwx.BUF_TSTATIC)
        self._tholder = _CudaArray(tp, tn, "<i8", 8)
        self.t_static = torch.as_tensor(self._tholder, device=dev)
        self.stream = torch.cuda.ExternalStream(sampler.stream(), device=dev)
        self._plan_batches, self._dynamic_counts, self._shared_levels = 1, False, set()
        self._level_dynamic = {}      # batches -> agreed "some rank counts dynamically"
        self.comm_timing = None       # CommTiming: device time of the collectives (bench.py)
        # all-boolean all-unary graphs: the gradient sums travel as 32- or 16-bit counts (agree())
        self._narrow_shift, self._narrow_bits, self._g32 = None, 32, None

    def agree(self, group=None):
        """Once after create: what every rank must decide alike.  A shard without categorical
        variables next to one with them (DeepDive numbers variables per relation: categoricals
        are contiguous) would otherwise reduce W elements against the other's 2 W."""
        t = torch.tensor([int(self.has_categorical)], dtype=torch.int32, device=self.grad.device)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self.has_categorical = bool(int(t[0]))
        self.grad_reduced = self.grad if self.has_categorical else self.grad[:self.s.W]
        # Half the bytes of the gradient all-reduce where the graph allows it (config 3 / 5a: every
        # contribution is +-2^31, a weight's sum a small count): all ranks must send the same type,
        # so the shift is the smallest any rank knows and the bound covers the sum over all ranks.
        info = self.s.graph.info
        world = dist.get_world_size(group)
        sh = int(info.grad_shift)
        # (the largest contribution in fixed-point units, not in units of this rank's own shift:
        # ranks may know different shifts, the count bound is taken at the common one)
        t = torch.tensor([-sh if sh > 0 else 0, int(info.grad_unit_max) << sh], dtype=torch.int64, device=self.grad.device)
        u = torch.tensor([int(info.max_records_per_weight)], dtype=torch.int64, device=self.grad.device)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)      # (MAX of -shift = -MIN shift; 0 if any rank knows nothing)
            dist.all_reduce(u, op=dist.ReduceOp.SUM, group=group)      # (>= the largest record count of a weight over all ranks)
        shift, recs = -int(t[0]), int(u[0])
        unit_max = (int(t[1]) >> shift) if shift > 0 else 0
        # (the overrides are agreed on too: ranks that disagreed would send different types into one collective)
        ov = torch.tensor([int(bool(os.environ.get("DWX_NO_NARROW_ALLREDUCE"))), int(bool(os.environ.get("DWX_NO_16BIT_ALLREDUCE")))],
                          dtype=torch.int32, device=self.grad.device)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(ov, op=dist.ReduceOp.MAX, group=group)
        ok = (not self.has_categorical and shift > 0 and unit_max > 0 and recs * unit_max < 2 ** 31
              and world > 1 and not int(ov[0]))
        self._narrow_shift = shift if ok else None
        # 16-bit counts, two per 32-bit word (dwx_grad_pack_async), while even the sum over all ranks
        # stays below 2^15: a quarter of the int64 bytes (config 5a at W = 1 M: 2 MB per mini-batch)
        self._narrow_bits = 16 if (ok and recs * unit_max < 2 ** 15 and not int(ov[1])) else 32
        self._g32 = None      # tensor view of the library's count buffer, made at the first call

    def allreduce_static_counts(self, group=None):
        """Once after create: every rank only counted its own shard's boolean updates and
        curvature bounds ([T | h], both fixed point: the sum does not depend on the order)."""
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self.t_static, op=dist.ReduceOp.SUM, group=group)

    def sgd_plan(self, stepsize, force_batches=0):
        from . import dwx
        batches, n_chunks, eta = self.s.sgd_plan(stepsize, force_batches)
        # a split plan either carries per-chunk static update counts (summed across ranks once,
        # see share_plan_static_counts) or counts dynamically into the T half of the gradient
        # vector, which then has to travel with every all-reduce
        self._plan_batches = batches
        self._local_dynamic = batches > 1 and self.s.device_buffer(dwx.BUF_TSTATIC_PLAN)[1] == 0
        # (until agree_level ran for this batch count only this rank's view is known)
        self._dynamic_counts = self._level_dynamic.get(batches, self._local_dynamic)
        if batches > 1 and self._dynamic_counts and not self._local_dynamic:
            self.s.sgd_plan_force_dynamic(True)    # another rank has no tables: count like it does
        return batches, n_chunks, eta

    def agree_level(self, group=None):
        """Once per batch count of a split plan, right after sgd_plan: ranks whose chunk counts
        differ (other colourings) may disagree on whether the level has per-chunk tables; if ANY
        rank counts dynamically, all do -- the all-reduced vector must be the same [G | T]
        everywhere and the counts of dynamic ranks must reach the others."""
        b = self._plan_batches
        if b <= 1 or b in self._level_dynamic:
            return
        t = torch.tensor([int(self._local_dynamic)], dtype=torch.int32, device=self.grad.device)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        self._level_dynamic[b] = bool(int(t[0]))
        self._dynamic_counts = self._level_dynamic[b]
        if self._dynamic_counts and not self._local_dynamic:
            self.s.sgd_plan_force_dynamic(True)

    @property
    def step_cap(self):
        return self.s.opts.step_cap

    @property
    def max_batches(self):
        return int(self.s.graph.info.num_tiles)

    def curvature(self, batches):
        return self.s.sgd_curvature(batches)

    def share_plan_static_counts(self, n_chunks_all, group=None):
        """Once per batch count of a split plan: size the per-chunk static-count table for the
        slowest rank's chunk count and sum it across shards (each rank counted its own block)."""
        from . import dwx
        if self._plan_batches <= 1 or self._dynamic_counts or self._plan_batches in self._shared_levels:
            return
        self.s.sgd_plan_rows(n_chunks_all)
        ptr, nbytes = self.s.device_buffer(dwx.BUF_TSTATIC_PLAN)
        holder = _CudaArray(ptr, nbytes, "<i8", 8)
        t = torch.as_tensor(holder, device=self.grad.device)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        self.stream.synchronize()      # the wrapper may go away; the table stays in the library
        self._shared_levels.add(self._plan_batches)

    def sgd_accumulate(self, chunk):
        self.s.sgd_accumulate(chunk)

    def sgd_apply(self):
        self.s.sgd_apply()

    def sgd_finish(self):
        self.s.sgd_finish()

    def sample(self):
        self.s.sample()

    def sample_n(self, n_sweeps):
        self.s.sample_n(n_sweeps)

    def wait(self):
        self.s.wait()

    def allreduce_grad(self, group=None):
        with torch.cuda.stream(self.stream):
            t = self.grad if self._dynamic_counts else self.grad_reduced
            narrow = self._narrow_shift is not None and not self._dynamic_counts
            if narrow:
                # the library packs (and checks: dwx_wait fails on a sum that is not a multiple of
                # 2^shift or does not fit) and unpacks; only the all-reduce of its buffer is torch's
                ptr, n_words = self.s.grad_pack(self._narrow_shift, self._narrow_bits)
                if self._g32 is None or self._g32_key != (ptr, n_words):
                    self._g32_holder = _CudaArray(ptr, n_words * 4, "<i4", 4)
                    self._g32 = torch.as_tensor(self._g32_holder, device=self.grad.device)
                    self._g32_key = (ptr, n_words)
                t = self._g32
            if self.comm_timing is not None:
                self.comm_timing.begin("allreduce", self.stream, t.numel() * t.element_size())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            if self.comm_timing is not None:
                self.comm_timing.end("allreduce", self.stream)
            if narrow:
                self.s.grad_unpack(self._narrow_shift, self._narrow_bits)

    # ---- replicas (the reference's n_datacopy) ----
    def sample_sgd(self, stepsize):
        self.s.sample_sgd(stepsize)

    def clear_tallies(self):
        self.s.clear_tallies()

    def _wrap(self, which, typestr, itemsize):
        ptr, nbytes = self.s.device_buffer(which)
        holder = _CudaArray(ptr, nbytes, typestr, itemsize)
        self._keep = getattr(self, "_keep", []) + [holder]
        return torch.as_tensor(holder, device=self.grad.device)

    def average_weights(self, group=None):
        """update_weights + copy_weights_to (src/dimmwitted.cc:199,209-216): sum the f64
        master weights over the replicas in place, then divide on every replica."""
        from . import dwx
        if not hasattr(self, "_weights"):
            self._weights = self._wrap(dwx.BUF_WEIGHTS, "<f8", 8)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self._weights, op=dist.ReduceOp.SUM, group=group)
        self.s.average_weights(dist.get_world_size(group))

    def allreduce_tallies(self, group=None):
        """aggregate_marginals_from (src/inference_result.cc:113-127): sum the replicas'
        tallies in place (same graph, same device order on every replica)."""
        from . import dwx
        if not hasattr(self, "_tallies"):
            self._tallies = self._wrap(dwx.BUF_TALLIES, "<i4", 4)
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self._tallies, op=dist.ReduceOp.SUM, group=group)

    def tallies(self):
        return self.s.tallies()

    # ---- halo exchange support ----
    def assign_tensor(self, chain):
        """int32 view of a chain's assignments in DEVICE order (values < 2^31)."""
        from . import dwx
        which = dwx.BUF_ASSIGN_FREE if chain in (0, "free") else dwx.BUF_ASSIGN_EVID
        ptr, nbytes = self.s.device_buffer(which)
        holder = _CudaArray(ptr, nbytes, "<i4", 4)
        self._keep = getattr(self, "_keep", []) + [holder]
        return torch.as_tensor(holder, device=self.grad.device)

    def positions(self, local_vids):
        """device positions of local variable ids (index into assign_tensor)."""
        return torch.as_tensor(self.s.graph.positions(local_vids).astype(np.int64), device=self.grad.device)

    def halo_list(self, local_vids):
        """One side of the halo exchange with one peer: dwx_halo_* list + a tensor view of its
        device buffer for the point-to-point call."""
        from . import dwx
        h = dwx.HaloList(self.s, local_vids)
        ptr, nbytes = h.buffer()
        holder = _CudaArray(ptr, max(nbytes, 4), "<i4", 4)
        h._holder = holder
        h.tensor = torch.as_tensor(holder, device=self.grad.device)[:nbytes // 4]
        h.message = lambda mask: h.tensor[:h.message_bytes(mask) // 4]   # what travels (int32 words)
        return h

    def stream_context(self):
        return torch.cuda.stream(self.stream)


class HaloExchange:
    """Boundary-assignment exchange between sweeps (north_star: "halo-exchange of boundary
    assignments between sweeps").  Rank r owns the global variable block
    [begin_r, end_r); its ghosts are remote variables its factors read.  Built once: every
    rank learns which of its owned variables each peer ghosts, and allocates ONE send and ONE
    receive buffer per peer, sized for both chains.  Per exchange, all on the engine's stream
    and with no allocation and no host synchronisation: gather the boundary values of the
    requested chains into the send buffers (index_select into the persistent buffer), ONE
    grouped send/recv for all peers and chains (ncclGroupStart/End under batch_isend_irecv; the
    returned works are waited on by the STREAM, which RCCL's work objects do without blocking
    the host), scatter the received values into the ghost slots.  One BIT per boundary variable
    and chain between boolean blocks (a byte up to cardinality 256, else 4 B: dwx_halo_message_bytes).  Reads of ghosts between exchanges are stale by at most one sweep -- the reference's
    own Hogwild semantics across threads."""

    def __init__(self, engine, begin, end, ghost_global_ids, bounds, group=None):
        self.e = engine
        self.group = group
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ghosts = np.asarray(ghost_global_ids, np.int64)
        n_owned = end - begin
        # who owns each of my ghosts (bounds[k] = (begin_k, end_k))
        starts = np.array([b for b, _ in bounds], np.int64)
        owner = np.searchsorted(starts, ghosts, side="right") - 1
        wanted = [ghosts[owner == k] for k in range(world)]        # ascending global ids
        all_wanted = [None] * world
        dist.all_gather_object(all_wanted, wanted, group=group)     # setup only (host)
        # one halo list per peer and direction: the engine gathers / scatters through it on its
        # own stream (HipEngine: dwx_halo_pack_async / dwx_halo_unpack_async of the C ABI) and
        # owns the buffer -- [chain 0 values | chain 1 values], persistent
        self.send, self.recv = {}, {}
        for k in range(world):
            if k == rank:
                continue
            theirs = np.asarray(all_wanted[k][rank], np.int64)     # my owned vars peer k ghosts
            if len(theirs):
                self.send[k] = engine.halo_list((theirs - begin).astype(np.uint64))
            mine = wanted[k]
            if len(mine):
                local = n_owned + np.searchsorted(ghosts, mine)
                self.recv[k] = engine.halo_list(local.astype(np.uint64))
        self.n_boundary = sum(h.n for h in self.send.values())
        self.n_ghost = sum(h.n for h in self.recv.values())
        self.bytes_per_exchange = 0      # of the last exchange: sent + received by this rank

    def exchange(self, chains=("free", "evid")):
        mask = (1 if "free" in chains else 0) | (2 if "evid" in chains else 0)
        moved = 0
        with self.e.stream_context():
            ops = []
            for k, h in self.send.items():
                h.pack(mask)
                msg = h.message(mask)
                moved += msg.numel() * msg.element_size()
                ops.append(dist.P2POp(dist.isend, msg, k, group=self.group))
            for k, h in self.recv.items():
                msg = h.message(mask)
                moved += msg.numel() * msg.element_size()
                ops.append(dist.P2POp(dist.irecv, msg, k, group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for h in self.recv.values():
                h.unpack(mask)
        self.bytes_per_exchange = moved


class ShardedDimmWitted:
    """Epoch driver over variable-block shards (DimmWitted::learn / inference,
    src/dimmwitted.cc:121-207, with the replica loop replaced by ranks)."""
    MAX_BATCHES = 64               # = MAX_PLAN_BATCHES of the library (device_types.h)

    def __init__(self, engine, n_learning_epoch, n_inference_epoch, stepsize=0.01, decay=0.95,
                 group=None, halo=None):
        self.e = engine
        self.n_learning_epoch = n_learning_epoch
        self.n_inference_epoch = n_inference_epoch
        self.stepsize = stepsize
        self.decay = decay
        self.group = group
        self.halo = halo          # HaloExchange or None (no cross-shard factors)
        self._lam = {}             # batches -> global curvature (agreed once)
        self._max_batches = None   # agreed upper end of the batch-count search
        self.plan_world = None     # experiments: plan as if this many equal shards took part
        self._level_chunks = {}    # batches -> chunk count of the slowest rank (agreed once)
        self.distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if self.distributed:
            if hasattr(self.e, "agree"):
                self.e.agree(group)
            self.e.allreduce_static_counts(group)
        if self.halo is not None:
            self.halo.exchange(("free", "evid"))   # ghosts start from their owners' state

    def learn_epoch(self, stepsize):
        """One learning sweep: plan the mini-batches (every rank must cut its block into the
        same number of pieces, because every piece ends in a collective), then per chunk
        accumulate -> all-reduce -> apply (un-split sweeps apply once, after the last chunk)."""
        batches, n_chunks, eta = self._plan(stepsize)
        for c in range(n_chunks):
            self.e.sgd_accumulate(c)          # ranks with fewer chunks idle through the rest
            if batches > 1 or c + 1 == n_chunks:
                if self.distributed:
                    self.e.allreduce_grad(self.group)
                self.e.sgd_apply()
        self.e.sgd_finish()
        if self.halo is not None:
            self._halo(("free", "evid"))

    def prepare(self, stepsize):
        """Do the one-off work of the first learning sweep now (curvature estimates, plan
        levels and their tables, the agreement collectives), e.g. before a timed region."""
        self._plan(stepsize)

    def _global_curvature(self, batches, world):
        """world x max over ranks of the library's curvature estimate for `batches` pieces per
        launch: a weight's curvature adds up over shards.  One MAX all-reduce per batch count,
        ever (the estimate does not depend on the step)."""
        if batches not in self._lam:
            lam = float(self.e.curvature(batches))
            if self.distributed:
                t = torch.tensor([lam], dtype=torch.float64, device=self.e.grad.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                lam = float(t[0])
            self._lam[batches] = world * lam
        return self._lam[batches]

    def _batch_limit(self):
        """No rank can cut finer than its tile count: the search for a batch count ends at the
        largest tile count of any rank (agreed once, so that every rank ends at the same one)."""
        if self._max_batches is None:
            n = int(getattr(self.e, "max_batches", self.MAX_BATCHES))
            if self.distributed:
                t = torch.tensor([n], dtype=torch.int64, device=self.e.grad.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                n = int(t[0])
            self._max_batches = max(1, min(n, self.MAX_BATCHES))
        return self._max_batches

    def _plan(self, stepsize):
        """The rule of dwx_sgd_plan (include/dwx.h) on the GLOBAL curvature, evaluated
        identically on every rank from cached numbers: after the first sweep at a given batch
        count there is no agreement round and no host synchronisation per sweep."""
        world = dist.get_world_size(self.group) if self.distributed else 1
        world = self.plan_world or world
        cap = float(self.e.step_cap)
        batches, eta = 1, stepsize
        if cap > 0 and stepsize > 0:
            limit = self._batch_limit()
            # the rule of dwx_sgd_plan on the global estimate: cut finer while it is above the cap
            # and a finer cut still lowers it by a fifth (a hub variable sets a floor no cut beats)
            while batches < limit and stepsize * self._global_curvature(batches, world) > cap:
                lam = self._global_curvature(batches, world)
                if self._global_curvature(2 * batches, world) > 0.8 * lam and (
                        4 * batches > limit or self._global_curvature(4 * batches, world) > 0.64 * lam):
                    break           # (two doublings ahead: one cut may fall on an unlucky boundary)
                batches *= 2
        # this rank's plan with the common batch count.  The step itself is never shortened:
        # every weight's step saturates at the inverse of its own (all-reduced) curvature bound
        # (dwx_sgd_apply_async), identically on every rank
        if batches not in self._level_chunks:
            # first use of this batch count: agree on the slowest rank's chunk count and share
            # the static-count tables -- for this level AND the coarser ones the decaying step
            # will walk down through, so that all one-off work (host-side table builds, these
            # collectives) lands in the first learning sweep
            b = batches
            while b >= 1:
                if b not in self._level_chunks:
                    if b > 1 and cap > 0:
                        self._global_curvature(b, world)
                    _, n_mine, _ = self.e.sgd_plan(eta, b)
                    n_all = n_mine
                    if self.distributed:
                        t = torch.tensor([n_mine], dtype=torch.int64, device=self.e.grad.device)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                        n_all = int(t[0])
                    self._level_chunks[b] = n_all
                    if self.distributed and b > 1 and hasattr(self.e, "agree_level"):
                        self.e.agree_level(self.group)
                    if self.distributed and b > 1 and hasattr(self.e, "share_plan_static_counts"):
                        self.e.share_plan_static_counts(n_all, self.group)
                b //= 2
        self.e.sgd_plan(eta, batches)
        return batches, self._level_chunks[batches], eta

    def learn(self):
        cur = self.stepsize
        for _ in range(self.n_learning_epoch):
            self.learn_epoch(cur)
            cur *= self.decay
        self.e.wait()

    def inference(self):
        if self.halo is None and self.n_inference_epoch and hasattr(self.e, "sample_n"):
            self.e.sample_n(self.n_inference_epoch)     # (nothing to exchange between the sweeps)
        else:
            for _ in range(self.n_inference_epoch):
                self.sample_epoch()
        self.e.wait()

    def sample_epoch(self):
        """One inference sweep (+ the evidence-chain halo refresh)."""
        self.e.sample()
        if self.halo is not None:
            self._halo(("evid",))

    def _halo(self, chains):
        ct = getattr(self.e, "comm_timing", None)
        if ct is not None:
            ct.begin("halo", self.e.stream)
        self.halo.exchange(chains)
        if ct is not None:
            ct.bytes["halo"] = ct.bytes.get("halo", 0) + self.halo.bytes_per_exchange
            ct.end("halo", self.e.stream)


class ReplicatedDimmWitted:
    """The reference's own multi-copy mode (`-c n_datacopy`, src/dimmwitted.cc:97-119):
    every rank holds the WHOLE graph and runs its own chains (distinct seeds).  Learning:
    each replica runs a full sample_sgd sweep on its own weights, then the weights are
    summed over replicas and averaged (update_weights, :209-216) -- one f64 all-reduce of W
    values per epoch; n epochs requested = ceil(n / replicas) rounds (compute_n_epochs,
    :280-282).  Inference: independent sweeps, tallies and sample counts summed at the end
    (aggregate_marginals_from).  Only valid while the graph fits one GPU."""

    def __init__(self, engine, n_learning_epoch, n_inference_epoch, stepsize=0.01, decay=0.95, group=None):
        self.e = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.n_learning_rounds = -(-n_learning_epoch // self.world)
        self.n_inference_rounds = -(-n_inference_epoch // self.world)
        self.stepsize = stepsize
        self.decay = decay

    def learn(self):
        cur = self.stepsize
        for _ in range(self.n_learning_rounds):
            self.e.sample_sgd(cur)
            if self.world > 1:
                self.e.average_weights(self.group)
            cur *= self.decay
        self.e.wait()

    def inference(self):
        self.e.clear_tallies()
        self._tallies_summed = False
        if self.n_inference_rounds and hasattr(self.e, "sample_n"):
            self.e.sample_n(self.n_inference_rounds)
        else:
            for _ in range(self.n_inference_rounds):
                self.e.sample()
        self.e.wait()

    def marginals(self):
        """-> (tallies, nsamples) summed over replicas, reference numbering, on every rank."""
        if self.world > 1 and not getattr(self, "_tallies_summed", False):
            self.e.allreduce_tallies(self.group)
            self._tallies_summed = True
        self.e.wait()
        t, n = self.e.tallies()
        return t, n * np.uint64(self.world)


def shard_range(total, rank, world):
    """Contiguous variable block of a rank: [begin, end)."""
    per = (total + world - 1) // world
    b = min(total, per * rank)
    e = min(total, b + per)
    if e <= b:
        raise ValueError("rank %d of %d would own no variable (%d variables in blocks of %d): use fewer ranks"
                         % (rank, world, total, per))
    return b, e
