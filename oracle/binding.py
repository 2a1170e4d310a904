"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (sampler_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DW = os.path.join(HERE, "_ref", "dw")


class OrcOpts(C.Structure):
    _fields_ = [("sample_evidence", C.c_int32), ("learn_non_evidence", C.c_int32),
                ("noise_aware", C.c_int32), ("regularization", C.c_int32),
                ("reg_param", C.c_double)]


class OrcSchedule(C.Structure):
    _fields_ = [("n_order", C.c_uint64), ("order", C.c_void_p),
                ("n_launches", C.c_uint64), ("launch_off", C.c_void_p)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE, "liboracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, u16, dbl, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint16, C.c_double, C.c_int
        L.orc_create.restype = vp; L.orc_create.argtypes = [vp, vp]
        L.orc_destroy.argtypes = [vp]
        L.orc_last_error.restype = C.c_char_p
        for n in ("orc_num_values", "orc_num_index_entries"):
            getattr(L, n).restype = u64; getattr(L, n).argtypes = [vp]
        for n in ("orc_weights", "orc_tallies", "orc_nsamples", "orc_var_val_base",
                  "orc_value_sparse", "orc_value_index_base", "orc_value_index_len",
                  "orc_factor_index", "orc_var_assignment_dense"):
            getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp]
        L.orc_assignments.restype = vp; L.orc_assignments.argtypes = [vp, i32]
        L.orc_clear_tallies.argtypes = [vp]
        L.orc_ref_set_workers.argtypes = [vp, u32]
        L.orc_ref_set_seed.argtypes = [vp, u32, u16, u16, u16]
        L.orc_ref_sample_single_variable.argtypes = [vp, u32, u64]
        L.orc_ref_sample_sgd_single_variable.argtypes = [vp, u32, u64, dbl]
        L.orc_ref_sgd_on_variable.argtypes = [vp, u64, dbl]
        L.orc_ref_sample.argtypes = [vp, i32]
        L.orc_ref_sample_sgd.argtypes = [vp, dbl, i32]
        L.orc_ref_learn.argtypes = [vp, u64, dbl, dbl, i32]
        L.orc_ref_inference.argtypes = [vp, u64, i32]
        L.orc_potential.restype = dbl; L.orc_potential.argtypes = [vp, u64, u64, i32]
        L.orc_factor_sign.restype = dbl; L.orc_factor_sign.argtypes = [i32, u64, vp]
        L.orc_logadd.restype = dbl; L.orc_logadd.argtypes = [dbl, dbl]
        L.orc_erand48.restype = dbl; L.orc_erand48.argtypes = [vp]
        L.orc_sched_sample.argtypes = [vp, vp, u64, u64]
        L.orc_sched_sample_sgd.argtypes = [vp, vp, u64, u64, dbl]
        L.orc_sched_accumulate.argtypes = [vp, vp, u64, u64]
        L.orc_sched_apply.argtypes = [vp, dbl]
        L.orc_sched_apply_h.argtypes = [vp, dbl, vp]
        L.orc_sched_curvature.argtypes = [vp, vp, vp]
        L.orc_grad.restype = vp; L.orc_grad.argtypes = [vp]
        L.orc_set_var_id_offset.argtypes = [vp, u64]
        L.orc_set_sampling_weight_f32.argtypes = [vp, i32]
        L.orc_set_fixed_point_mask.argtypes = [vp, vp]; L.orc_set_fixed_point_mask.restype = None
        L.orc_sched_check_independent.restype = i32
        L.orc_sched_check_independent.argtypes = [vp, vp]
        L.orc_philox_uniforms.argtypes = [u64, u64, u64, vp]
        L.orc_philox4x32_10.argtypes = [vp, vp]; L.orc_philox4x32_10.restype = None
        _lib = L
    return _lib


def _view(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype)


class Oracle:
    """One graph + InferenceResult state, reference semantics."""

    def __init__(self, graph, sample_evidence=False, learn_non_evidence=False,
                 noise_aware=False, regularization="l2", reg_param=0.01):
        self.L = lib()
        self.graph = graph
        self._desc = graph.desc()
        self.opts = OrcOpts(int(sample_evidence), int(learn_non_evidence), int(noise_aware),
                            0 if regularization == "l1" else 1, float(reg_param))
        self.h = self.L.orc_create(C.addressof(self._desc), C.addressof(self.opts))
        if not self.h:
            raise RuntimeError("oracle: " + self.L.orc_last_error().decode())
        self.V, self.W = graph.num_variables, graph.num_weights
        self.num_values = self.L.orc_num_values(self.h)
        self._sched_keep = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    # ---- live views into oracle state ----
    @property
    def weights(self):
        return _view(self.L.orc_weights(self.h), self.W, np.float64)

    def assignments(self, chain):
        return _view(self.L.orc_assignments(self.h, 0 if chain in (0, "free") else 1), self.V, np.uint64)

    @property
    def tallies(self):
        return _view(self.L.orc_tallies(self.h), self.num_values, np.uint64)

    @property
    def nsamples(self):
        return _view(self.L.orc_nsamples(self.h), self.V, np.uint64)

    @property
    def var_val_base(self):
        return _view(self.L.orc_var_val_base(self.h), self.V, np.uint64)

    @property
    def value_sparse(self):
        return _view(self.L.orc_value_sparse(self.h), self.num_values, np.uint64)

    @property
    def value_index_base(self):
        return _view(self.L.orc_value_index_base(self.h), self.num_values, np.uint64)

    @property
    def value_index_len(self):
        return _view(self.L.orc_value_index_len(self.h), self.num_values, np.uint64)

    @property
    def factor_index(self):
        return _view(self.L.orc_factor_index(self.h), self.L.orc_num_index_entries(self.h), np.uint64)

    @property
    def assignment_dense(self):
        return _view(self.L.orc_var_assignment_dense(self.h), self.V, np.uint64)

    def clear_tallies(self):
        self.L.orc_clear_tallies(self.h)

    def set_fixed_point_mask(self, mask):
        """Schedule mode: uint8[V], 1 where the device sums a variable's potentials in fixed point
        (sampler_amd.dwx.Graph.fixed_point_mask); None clears it."""
        if mask is None:
            self.L.orc_set_fixed_point_mask(self.h, None)
        else:
            m = np.ascontiguousarray(mask, np.uint8)
            assert len(m) == self.V
            self.L.orc_set_fixed_point_mask(self.h, m.ctypes.data)

    # ---- reference mode ----
    def set_workers(self, n):
        self.L.orc_set_sampling_weight_f32(self.h, 0)
        self.L.orc_ref_set_workers(self.h, n)

    def set_seed(self, worker, s0, s1, s2):
        self.L.orc_ref_set_seed(self.h, worker, s0, s1, s2)

    def sample_single_variable(self, vid, worker=0):
        self.L.orc_ref_sample_single_variable(self.h, worker, vid)

    def sample_sgd_single_variable(self, vid, stepsize, worker=0):
        self.L.orc_ref_sample_sgd_single_variable(self.h, worker, vid, stepsize)

    def sgd_on_variable(self, vid, stepsize):
        self.L.orc_ref_sgd_on_variable(self.h, vid, stepsize)

    def sample(self, threaded=False):
        self.L.orc_ref_sample(self.h, int(threaded))

    def sample_sgd(self, stepsize, threaded=False):
        self.L.orc_ref_sample_sgd(self.h, stepsize, int(threaded))

    def learn(self, n_epoch, stepsize, decay, threaded=False):
        self.L.orc_ref_learn(self.h, n_epoch, stepsize, decay, int(threaded))

    def inference(self, n_epoch, threaded=False):
        self.L.orc_ref_inference(self.h, n_epoch, int(threaded))

    def potential(self, vid, proposal, chain="evid"):
        return self.L.orc_potential(self.h, vid, proposal, 0 if chain in (0, "free") else 1)

    # ---- schedule mode (device semantics) ----
    def _sched(self, order, launch_off):
        order = np.ascontiguousarray(order, np.uint64)
        launch_off = np.ascontiguousarray(launch_off, np.uint64)
        s = OrcSchedule(len(order), order.ctypes.data, len(launch_off) - 1, launch_off.ctypes.data)
        self._sched_keep = (order, launch_off, s)
        return s

    def sched_check_independent(self, order, launch_off):
        s = self._sched(order, launch_off)
        return bool(self.L.orc_sched_check_independent(self.h, C.addressof(s)))

    def sched_sample(self, order, launch_off, seed, sweep):
        self.L.orc_set_sampling_weight_f32(self.h, 1)
        s = self._sched(order, launch_off)
        self.L.orc_sched_sample(self.h, C.addressof(s), seed, sweep)

    def sched_sample_sgd(self, order, launch_off, seed, sweep, stepsize):
        self.L.orc_set_sampling_weight_f32(self.h, 1)
        s = self._sched(order, launch_off)
        self.L.orc_sched_sample_sgd(self.h, C.addressof(s), seed, sweep, stepsize)

    def sched_accumulate(self, order, launch_off, seed, sweep):
        self.L.orc_set_sampling_weight_f32(self.h, 1)
        s = self._sched(order, launch_off)
        self.L.orc_sched_accumulate(self.h, C.addressof(s), seed, sweep)

    def sched_apply(self, stepsize, hess=None):
        """hess: int64[W] curvature bounds to use instead of the accumulated ones."""
        if hess is None:
            self.L.orc_sched_apply(self.h, stepsize)
        else:
            hess = np.ascontiguousarray(hess, np.int64)
            self.L.orc_sched_apply_h(self.h, stepsize, hess.ctypes.data)

    def sched_curvature(self, order):
        order = np.ascontiguousarray(order, np.uint64)
        s = self._sched(order, np.array([0, len(order)], np.uint64))
        out = np.zeros(self.W, np.int64)
        self.L.orc_sched_curvature(self.h, C.addressof(s), out.ctypes.data)
        return out

    @property
    def grad(self):
        return _view(self.L.orc_grad(self.h), 3 * self.W, np.int64)   # [G | T | H]

    def set_var_id_offset(self, off):
        self.L.orc_set_var_id_offset(self.h, off)

    # ---- result files (src/inference_result.cc:101-105,211-243) ----
    def weights_text(self):
        return "".join("%d %s\n" % (j, fmt_g(w)) for j, w in enumerate(self.weights))

    def marginals_text(self):
        g = self.graph
        out = []
        t, n, base, sparse = self.tallies, self.nsamples, self.var_val_base, self.value_sparse
        for v in range(self.V):
            if g.var_role[v] >= 1 and not self.opts.sample_evidence:
                continue
            b = int(base[v])
            if g.var_dtype[v] == 0:
                out.append("%d 1 %s\n" % (v, fmt_g(_div(t[b], n[v]))))
            else:
                for j in range(int(g.var_cardinality[v])):
                    out.append("%d %d %s\n" % (v, int(sparse[b + j]), fmt_g(_div(t[b + j], n[v]))))
        return "".join(out)


def _div(a, b):
    a, b = float(a), float(b)
    if b == 0:
        return float("nan") if a == 0 else float("inf")
    return a / b


def fmt_g(x):
    """C++ ostream default formatting (precision 6, %g)."""
    s = "%g" % x
    if s == "nan":
        return "-nan" if np.signbit(x) else "nan"
    return s


def factor_sign(func, sat):
    sat = np.ascontiguousarray(sat, np.uint8)
    return lib().orc_factor_sign(func, len(sat), sat.ctypes.data)


def erand48_seq(seed, n):
    x = (C.c_uint16 * 3)(*seed)
    return [lib().orc_erand48(C.addressof(x)) for _ in range(n)]


def philox4x32_10(key, ctr):
    """Raw Philox4x32-10 block: (key[2], ctr[4]) -> out[4] (uint32)."""
    k = np.ascontiguousarray(key, np.uint32)
    c = np.array(ctr, np.uint32)
    lib().orc_philox4x32_10(k.ctypes.data, c.ctypes.data)
    return c


def philox_uniforms(seed, vid, sweep):
    out = (C.c_double * 2)()
    lib().orc_philox_uniforms(seed, vid, sweep, C.addressof(out))
    return out[0], out[1]


def run_reference_dw(graph_dir, args, out_dir, quiet=True, timeout=None):
    """Run the REAL reference binary (oracle/_ref/dw gibbs ...) on binary files."""
    p = lambda n: os.path.join(graph_dir, n)
    cmd = [REF_DW, "gibbs", "-m", p("graph.meta"), "-w", p("graph.weights"),
           "-v", p("graph.variables"), "-f", p("graph.factors"), "-o", out_dir]
    if os.path.exists(p("graph.domains")):
        cmd += ["--domains", p("graph.domains")]
    if quiet:
        cmd.append("--quiet")
    cmd += list(args)
    return subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=timeout).stdout


def have_reference():
    return os.access(REF_DW, os.X_OK)
