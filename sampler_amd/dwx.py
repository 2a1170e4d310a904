"""ctypes binding of the C ABI in include/dwx.h (sampler_amd/csrc/libdwx.so).

There is no fallback: if the HIP library is missing, `default_library()` raises, and
`GibbsSampler` raises if no GPU is usable.  (tests/hipemu builds a host emulation of
the *kernel sources* for sanitizer runs; tests inject it explicitly through
`Library(path)` -- this module never looks for it.)
"""
import ctypes as C
import os

import numpy as np

from .rawgraph import GraphDesc, RawGraph

HERE = os.path.dirname(os.path.abspath(__file__))
# DWX_LIB: another build of the same library (kernel experiments, tools/variant.sh)
LIB_PATH = os.environ.get("DWX_LIB") or os.path.join(HERE, "csrc", "libdwx.so")

DWX_OK, DWX_E_INVALID, DWX_E_LIMIT, DWX_E_DEVICE, DWX_E_NOMEM = 0, -1, -2, -3, -4
BUF_WEIGHTS, BUF_GRAD, BUF_ASSIGN_FREE, BUF_ASSIGN_EVID, BUF_TALLIES, BUF_TSTATIC, BUF_TSTATIC_PLAN, BUF_SORTED_RECORDS, BUF_SORTED_RECORDS_PLAN = range(9)

# every symbol include/dwx.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "dwx_last_error", "dwx_version", "dwx_default_options",
    "dwx_graph_create", "dwx_graph_destroy", "dwx_graph_get_info", "dwx_graph_get_schedule",
    "dwx_graph_get_values", "dwx_graph_get_fixed_point_mask", "dwx_graph_get_positions", "dwx_graph_get_index",
    "dwx_sampler_create", "dwx_device_init", "dwx_device_count", "dwx_buffer_copy", "dwx_sampler_destroy", "dwx_sample_async", "dwx_sample_n_async", "dwx_sample_sgd_async",
    "dwx_wait", "dwx_sgd_plan", "dwx_sgd_curvature", "dwx_sgd_plan_rows", "dwx_sgd_plan_force_dynamic", "dwx_sgd_get_chunks", "dwx_grad_pack32_async", "dwx_grad_unpack32_async", "dwx_grad_pack_async", "dwx_grad_unpack_async", "dwx_sgd_accumulate_async",
    "dwx_sgd_apply_async", "dwx_sgd_finish",
    "dwx_get_weights", "dwx_set_weights", "dwx_average_weights_async",
    "dwx_clear_tallies", "dwx_get_tallies",
    "dwx_get_assignments", "dwx_set_assignments", "dwx_get_sweep", "dwx_set_sweep",
    "dwx_device_buffer", "dwx_halo_create", "dwx_halo_destroy", "dwx_halo_buffer", "dwx_halo_message_bytes", "dwx_halo_pack_async",
    "dwx_halo_unpack_async", "dwx_stream", "dwx_kernel_time", "dwx_kernel_time_reset",
    "dwx_test_factor_sign", "dwx_test_philox",
]


class DwxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dwx error %d: %s" % (code, msg))
        self.code = code


class CompileOpts(C.Structure):
    _fields_ = [("tile_vars", C.c_uint32), ("tile_edges", C.c_uint32), ("tile_rows", C.c_uint32),
                ("conflict_arity_cap", C.c_uint32), ("n_threads", C.c_uint32),
                ("no_compact_records", C.c_uint32), ("no_weight_order", C.c_uint32),
                ("wide_min_records", C.c_uint32), ("no_record_vifs", C.c_uint32),
                ("no_pull_unary", C.c_uint32), ("no_sorted_records", C.c_uint32),
                ("super_tiles", C.c_uint32), ("sorted_slots", C.c_uint32), ("defer_sorted_records", C.c_uint32),
                ("no_narrow_info", C.c_uint32)]


class GraphInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "num_variables", "num_factors", "num_edges", "num_weights", "num_owned_variables", "num_values",
        "num_index_entries", "num_vif_entries", "num_colors", "num_launches", "num_tiles",
        "num_giant_tiles", "max_cardinality", "num_query_variables", "device_bytes")] + [
        ("has_categorical", C.c_uint32), ("order_is_identity", C.c_uint32), ("num_wide_tiles", C.c_uint64),
        ("num_staged_tiles", C.c_uint64), ("num_super_tiles", C.c_uint64), ("num_sorted_records", C.c_uint64),
        ("grad_shift", C.c_uint64), ("grad_unit_max", C.c_uint64), ("max_records_per_weight", C.c_uint64)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("sample_evidence", C.c_int32),
                ("learn_non_evidence", C.c_int32), ("noise_aware", C.c_int32),
                ("regularization", C.c_int32), ("plan_layouts", C.c_int32),
                ("reg_param", C.c_double), ("step_cap", C.c_double), ("seed", C.c_uint64),
                ("var_id_offset", C.c_uint64)]


class Library:
    """A loaded libdwx with prototypes set."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                "%s not found: the HIP extension is not built (run "
                "`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback"
                % path)
        self.path = path
        L = self.L = C.CDLL(path)
        vp, u64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int, C.c_double
        L.dwx_last_error.restype = C.c_char_p
        L.dwx_version.restype = i32
        L.dwx_default_options.argtypes = [vp]; L.dwx_default_options.restype = None
        L.dwx_graph_create.argtypes = [vp, vp, vp]
        L.dwx_graph_destroy.argtypes = [vp]; L.dwx_graph_destroy.restype = None
        L.dwx_graph_get_info.argtypes = [vp, vp]
        L.dwx_graph_get_schedule.argtypes = [vp, vp, vp]
        L.dwx_graph_get_values.argtypes = [vp, vp, vp]
        L.dwx_graph_get_index.argtypes = [vp, vp, vp, vp]
        L.dwx_graph_get_positions.argtypes = [vp, vp, u64, vp]
        L.dwx_graph_get_fixed_point_mask.argtypes = [vp, vp]
        L.dwx_sampler_create.argtypes = [vp, vp, vp]
        L.dwx_device_init.argtypes = [C.c_int32]
        L.dwx_device_count.argtypes = [vp]
        L.dwx_buffer_copy.argtypes = [vp, vp, vp, u64, i32]
        L.dwx_sampler_destroy.argtypes = [vp]; L.dwx_sampler_destroy.restype = None
        L.dwx_sample_async.argtypes = [vp]
        L.dwx_sample_n_async.argtypes = [vp, C.c_uint32]
        L.dwx_sample_sgd_async.argtypes = [vp, dbl]
        L.dwx_wait.argtypes = [vp]
        L.dwx_sgd_plan.argtypes = [vp, dbl, C.c_uint32, vp, vp, vp]
        L.dwx_sgd_plan_rows.argtypes = [vp, C.c_uint32]
        L.dwx_sgd_curvature.argtypes = [vp, C.c_uint32, vp]
        L.dwx_sgd_plan_force_dynamic.argtypes = [vp, i32]
        L.dwx_sgd_get_chunks.argtypes = [vp, vp]
        L.dwx_grad_pack32_async.argtypes = [vp, C.c_uint32, vp, vp]
        L.dwx_grad_unpack32_async.argtypes = [vp, C.c_uint32]
        L.dwx_grad_pack_async.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp]
        L.dwx_grad_unpack_async.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.dwx_sgd_accumulate_async.argtypes = [vp, C.c_uint32]
        L.dwx_sgd_apply_async.argtypes = [vp]
        L.dwx_sgd_finish.argtypes = [vp]
        L.dwx_get_weights.argtypes = [vp, vp]; L.dwx_set_weights.argtypes = [vp, vp]
        L.dwx_average_weights_async.argtypes = [vp, C.c_uint32]
        L.dwx_clear_tallies.argtypes = [vp]
        L.dwx_get_tallies.argtypes = [vp, vp, vp]
        L.dwx_get_assignments.argtypes = [vp, i32, vp]
        L.dwx_set_assignments.argtypes = [vp, i32, vp]
        L.dwx_get_sweep.argtypes = [vp, vp]; L.dwx_set_sweep.argtypes = [vp, u64]
        L.dwx_device_buffer.argtypes = [vp, i32, vp, vp]
        L.dwx_stream.argtypes = [vp, vp]
        L.dwx_halo_create.argtypes = [vp, vp, u64, vp]
        L.dwx_halo_destroy.argtypes = [vp]; L.dwx_halo_destroy.restype = None
        L.dwx_halo_buffer.argtypes = [vp, vp, vp]
        L.dwx_halo_message_bytes.argtypes = [vp, C.c_int, vp]
        L.dwx_halo_pack_async.argtypes = [vp, i32]
        L.dwx_halo_unpack_async.argtypes = [vp, i32]
        L.dwx_kernel_time.argtypes = [vp, i32, vp, vp, vp]
        L.dwx_kernel_time_reset.argtypes = [vp, i32]
        L.dwx_test_factor_sign.argtypes = [i32, i32, u64, vp, vp]
        L.dwx_test_philox.argtypes = [i32, vp, vp, vp, vp]

    def check(self, rc):
        if rc != DWX_OK:
            raise DwxError(rc, self.L.dwx_last_error().decode(errors="replace"))

    def test_philox(self, key, ctr, device=0):
        """-> (philox4x32-10(key, ctr) as uint32[4], the sweep kernels' two uniforms)."""
        k, c = np.ascontiguousarray(key, np.uint32), np.ascontiguousarray(ctr, np.uint32)
        out, uni = np.zeros(4, np.uint32), np.zeros(2, np.float64)
        self.check(self.L.dwx_test_philox(device, k.ctypes.data, c.ctypes.data, out.ctypes.data, uni.ctypes.data))
        return out, uni

    def test_factor_sign(self, func, sat, device=0):
        sat = np.ascontiguousarray(sat, np.uint8)
        out = C.c_double()
        self.check(self.L.dwx_test_factor_sign(device, func, len(sat), sat.ctypes.data, C.byref(out)))
        return out.value


_default = None


def default_library():
    global _default
    if _default is None:
        _default = Library(LIB_PATH)
    return _default


class Graph:
    """Compiled factor graph (host side): dwx_graph_create."""

    def __init__(self, raw: RawGraph, lib=None, **compile_opts):
        self.lib = lib or default_library()
        self.raw = raw
        desc = raw.desc()
        co = CompileOpts(**compile_opts)
        h = C.c_void_p()
        self.lib.check(self.lib.L.dwx_graph_create(C.byref(desc), C.byref(co), C.byref(h)))
        self.h = h
        info = GraphInfo()
        self.lib.check(self.lib.L.dwx_graph_get_info(self.h, C.byref(info)))
        self.info = info

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.L.dwx_graph_destroy(self.h)
            self.h = None

    def schedule(self):
        order = np.zeros(self.info.num_owned_variables, np.uint64)
        off = np.zeros(self.info.num_launches + 1, np.uint64)
        self.lib.check(self.lib.L.dwx_graph_get_schedule(self.h, order.ctypes.data, off.ctypes.data))
        return order, off

    def fixed_point_mask(self):
        """uint8[V]: 1 where the device sums a variable's potentials in fixed point (test hook:
        orc.Oracle.set_fixed_point_mask)."""
        m = np.zeros(self.info.num_variables, np.uint8)
        self.lib.check(self.lib.L.dwx_graph_get_fixed_point_mask(self.h, m.ctypes.data))
        return m

    def positions(self, vids):
        vids = np.ascontiguousarray(vids, np.uint64)
        out = np.zeros(len(vids), np.uint64)
        self.lib.check(self.lib.L.dwx_graph_get_positions(self.h, vids.ctypes.data, len(vids), out.ctypes.data))
        return out

    def values(self):
        base = np.zeros(self.info.num_variables, np.uint64)
        sparse = np.zeros(self.info.num_values, np.uint64)
        self.lib.check(self.lib.L.dwx_graph_get_values(self.h, base.ctypes.data, sparse.ctypes.data))
        return base, sparse

    def index(self):
        R, N = self.info.num_values, self.info.num_index_entries
        base, ln, fi = np.zeros(R, np.uint64), np.zeros(R, np.uint64), np.zeros(N, np.uint64)
        self.lib.check(self.lib.L.dwx_graph_get_index(self.h, base.ctypes.data, ln.ctypes.data,
                                                      fi.ctypes.data))
        return base, ln, fi


class GibbsSampler:
    """Mirror of the reference's `GibbsSampler` (src/gibbs_sampler.h:18-57): owns the
    device copy of the graph and the InferenceResult state; `sample()` /
    `sample_sgd(stepsize)` enqueue one sweep, `wait()` joins."""

    def __init__(self, graph: Graph, device=0, sample_evidence=False, learn_non_evidence=False,
                 noise_aware=False, regularization="l2", reg_param=0.01, seed=0x5eed5eed,
                 step_cap=1.5, var_id_offset=0, plan_layouts=0):
        self.lib = graph.lib
        self.graph = graph
        o = Options()
        self.lib.L.dwx_default_options(C.byref(o))
        o.device = device
        o.sample_evidence = int(sample_evidence)
        o.learn_non_evidence = int(learn_non_evidence)
        o.noise_aware = int(noise_aware)
        o.regularization = 0 if regularization == "l1" else 1
        o.reg_param = float(reg_param)
        o.seed = int(seed)
        o.step_cap = float(step_cap)
        o.plan_layouts = int(plan_layouts)
        o.var_id_offset = int(var_id_offset)
        self.opts = o
        h = C.c_void_p()
        self.lib.check(self.lib.L.dwx_sampler_create(graph.h, C.byref(o), C.byref(h)))
        self.h = h
        self.V = graph.info.num_variables
        self.W = graph.info.num_weights
        self.num_values = graph.info.num_values

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h", None):
            self.lib.L.dwx_sampler_destroy(self.h)
            self.h = None

    # ---- the reference's interface ----
    def sample(self, i_epoch=0):
        self.lib.check(self.lib.L.dwx_sample_async(self.h))

    def sample_n(self, n_sweeps):
        """n_sweeps inference sweeps (one launch on an all-unary graph; include/dwx.h)."""
        self.lib.check(self.lib.L.dwx_sample_n_async(self.h, n_sweeps))

    def sample_sgd(self, stepsize):
        self.lib.check(self.lib.L.dwx_sample_sgd_async(self.h, stepsize))

    def wait(self):
        self.lib.check(self.lib.L.dwx_wait(self.h))

    # ---- a learning sweep in pieces (multi-GPU drivers, parity tests) ----
    def sgd_plan(self, stepsize, force_batches=0):
        """-> (batches, n_chunks, effective_stepsize); see dwx_sgd_plan in include/dwx.h."""
        b, n, e = C.c_uint32(), C.c_uint32(), C.c_double()
        self.lib.check(self.lib.L.dwx_sgd_plan(self.h, stepsize, force_batches, C.byref(b), C.byref(n),
                                               C.byref(e)))
        return b.value, n.value, e.value

    def sgd_curvature(self, batches):
        out = C.c_double()
        self.lib.check(self.lib.L.dwx_sgd_curvature(self.h, int(batches), C.byref(out)))
        return out.value

    def sgd_plan_rows(self, n_rows):
        self.lib.check(self.lib.L.dwx_sgd_plan_rows(self.h, int(n_rows)))

    def sgd_plan_force_dynamic(self, on=True):
        self.lib.check(self.lib.L.dwx_sgd_plan_force_dynamic(self.h, int(bool(on))))

    def grad_pack(self, shift, bits=32):
        """dwx_grad_pack_async: the gradient sums as 32- or 16-bit counts (two per word) in a device
        buffer of the library -> (device pointer, number of 32-bit words); dwx_wait fails if a sum
        was not a multiple of 2^shift or did not fit."""
        ptr, n = C.c_void_p(), C.c_uint64()
        self.lib.check(self.lib.L.dwx_grad_pack_async(self.h, int(shift), int(bits), C.byref(ptr), C.byref(n)))
        return ptr.value, int(n.value)

    def grad_unpack(self, shift, bits=32):
        self.lib.check(self.lib.L.dwx_grad_unpack_async(self.h, int(shift), int(bits)))

    def sgd_chunks(self, n_chunks):
        """[n_chunks, 2]: chunk c covers positions [r[c, 0], r[c, 1]) of the schedule order."""
        r = np.zeros((n_chunks, 2), np.uint64)
        self.lib.check(self.lib.L.dwx_sgd_get_chunks(self.h, r.ctypes.data))
        return r

    def sgd_accumulate(self, chunk):
        self.lib.check(self.lib.L.dwx_sgd_accumulate_async(self.h, chunk))

    def sgd_apply(self):
        self.lib.check(self.lib.L.dwx_sgd_apply_async(self.h))

    def sgd_finish(self):
        self.lib.check(self.lib.L.dwx_sgd_finish(self.h))

    # ---- InferenceResult state ----
    @property
    def weights(self):
        out = np.zeros(self.W, np.float64)
        self.lib.check(self.lib.L.dwx_get_weights(self.h, out.ctypes.data))
        return out

    @weights.setter
    def weights(self, w):
        w = np.ascontiguousarray(w, np.float64)
        assert len(w) == self.W
        self.lib.check(self.lib.L.dwx_set_weights(self.h, w.ctypes.data))

    def average_weights(self, n_replicas):
        """After an in-place sum of BUF_WEIGHTS over replicas: divide, refresh the f32 copy."""
        self.lib.check(self.lib.L.dwx_average_weights_async(self.h, int(n_replicas)))

    def clear_tallies(self):
        self.lib.check(self.lib.L.dwx_clear_tallies(self.h))

    def tallies(self):
        t = np.zeros(self.num_values, np.uint64)
        n = np.zeros(self.V, np.uint64)
        self.lib.check(self.lib.L.dwx_get_tallies(self.h, t.ctypes.data, n.ctypes.data))
        return t, n

    def assignments(self, chain):
        out = np.zeros(self.V, np.uint64)
        c = 0 if chain in (0, "free") else 1
        self.lib.check(self.lib.L.dwx_get_assignments(self.h, c, out.ctypes.data))
        return out

    def set_assignments(self, chain, a):
        a = np.ascontiguousarray(a, np.uint64)
        c = 0 if chain in (0, "free") else 1
        self.lib.check(self.lib.L.dwx_set_assignments(self.h, c, a.ctypes.data))

    @property
    def sweep(self):
        out = C.c_uint64()
        self.lib.check(self.lib.L.dwx_get_sweep(self.h, C.byref(out)))
        return out.value

    @sweep.setter
    def sweep(self, v):
        self.lib.check(self.lib.L.dwx_set_sweep(self.h, int(v)))

    def device_buffer(self, which):
        p, n = C.c_void_p(), C.c_uint64()
        self.lib.check(self.lib.L.dwx_device_buffer(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_buffer(self, which, dtype=np.uint8):
        """Host copy of a device buffer (dwx_device_buffer + dwx_buffer_copy); empty when there is none."""
        p, n = self.device_buffer(which)
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype)
        if p and n:
            self.lib.check(self.lib.L.dwx_buffer_copy(self.h, out.ctypes.data, p, n, 0))
        return out

    def stream(self):
        p = C.c_void_p()
        self.lib.check(self.lib.L.dwx_stream(self.h, C.byref(p)))
        return p.value

    def kernel_time_reset(self, enable=True):
        self.lib.check(self.lib.L.dwx_kernel_time_reset(self.h, int(enable)))

    def kernel_time(self, kind):
        ms, nl, ns = C.c_double(), C.c_uint64(), C.c_uint64()
        k = {0: 0, "infer": 0, 1: 1, "learn": 1, 2: 2, "pull": 2, 3: 3, "graph": 3, 4: 4, "persist": 4, 5: 5, "merged": 5}[kind]
        self.lib.check(self.lib.L.dwx_kernel_time(self.h, k, C.byref(ms), C.byref(nl), C.byref(ns)))
        return ms.value, nl.value, ns.value

    # ---- result files (src/inference_result.cc:101-105, 211-243) ----
    def weights_text(self):
        return "".join("%d %s\n" % (j, fmt_g(w)) for j, w in enumerate(self.weights))

    def marginals(self):
        """(tallies / nsamples) per value row in the reference numbering."""
        t, n = self.tallies()
        base, _ = self.graph.values()
        return t, n, base

    def marginals_text(self):
        raw = self.graph.raw
        t, n = self.tallies()
        base, sparse = self.graph.values()
        out = []
        for v in range(self.V):
            if raw.var_role[v] >= 1 and not self.opts.sample_evidence:
                continue
            b = int(base[v])
            if raw.var_dtype[v] == 0:
                out.append("%d 1 %s\n" % (v, fmt_g(_div(t[b], n[v]))))
            else:
                for j in range(int(raw.var_cardinality[v])):
                    out.append("%d %d %s\n" % (v, int(sparse[b + j]), fmt_g(_div(t[b + j], n[v]))))
        return "".join(out)


class HaloList:
    """One side of a halo exchange with one peer (dwx_halo_*): the listed local variables'
    assignments are packed into / unpacked from a device buffer on the sampler's stream."""
    FREE, EVID = 1, 2

    def __init__(self, sampler: GibbsSampler, local_vids):
        self.lib, self.sampler = sampler.lib, sampler
        v = np.ascontiguousarray(local_vids, np.uint64)
        self.n = len(v)
        h = C.c_void_p()
        self.lib.check(self.lib.L.dwx_halo_create(sampler.h, v.ctypes.data, len(v), C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.L.dwx_halo_destroy(self.h)
            self.h = None

    def buffer(self):
        p, n = C.c_void_p(), C.c_uint64()
        self.lib.check(self.lib.L.dwx_halo_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def message_bytes(self, chains):
        """bytes at the start of the buffer that pack(chains) fills / unpack(chains) reads."""
        n = C.c_uint64()
        self.lib.check(self.lib.L.dwx_halo_message_bytes(self.h, int(chains), C.byref(n)))
        return n.value

    def pack(self, chains):
        self.lib.check(self.lib.L.dwx_halo_pack_async(self.h, int(chains)))

    def unpack(self, chains):
        self.lib.check(self.lib.L.dwx_halo_unpack_async(self.h, int(chains)))


def _div(a, b):
    a, b = float(a), float(b)
    if b == 0:
        return float("nan") if a == 0 else float("inf")
    return a / b


def fmt_g(x):
    """C++ ostream default formatting (6 significant digits, %g)."""
    s = "%g" % x
    if s == "nan":
        return "-nan"
    return s


class DimmWitted:
    """Mirror of the reference's epoch driver (src/dimmwitted.cc:97-282) over ONE
    sampler (n_datacopy = 1): learn() then inference(), same stepsize decay."""

    def __init__(self, sampler: GibbsSampler, n_learning_epoch, n_inference_epoch,
                 stepsize=0.01, decay=0.95):
        self.sampler = sampler
        self.n_learning_epoch = n_learning_epoch
        self.n_inference_epoch = n_inference_epoch
        self.stepsize = stepsize
        self.decay = decay

    def learn(self):
        cur = self.stepsize
        for _ in range(self.n_learning_epoch):
            self.sampler.sample_sgd(cur)
            self.sampler.wait()
            cur *= self.decay

    def inference(self):
        self.sampler.clear_tallies()
        if self.n_inference_epoch:
            self.sampler.sample_n(self.n_inference_epoch)
        self.sampler.wait()
