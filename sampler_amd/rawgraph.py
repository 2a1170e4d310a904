"""Columnar image of a DeepDive factor graph: exactly the content of the reference's
binary input files (/root/reference/doc/binary_format.md), one numpy column per field.

This is the host-side hand-off format of the drop-in boundary: `RawGraph.desc()`
yields the `dwx_graph_desc` C structure of include/dwx.h (plain pointers + sizes).
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

# /root/reference/src/common.h:35-48
FUNC_IMPLY_NATURAL = 0
FUNC_OR = 1
FUNC_AND = 2
FUNC_EQUAL = 3
FUNC_ISTRUE = 4
FUNC_LINEAR = 7
FUNC_RATIO = 8
FUNC_LOGICAL = 9
FUNC_AND_CATEGORICAL = 12
FUNC_IMPLY_MLN = 13

DTYPE_BOOLEAN = 0
DTYPE_CATEGORICAL = 1


class GraphDesc(C.Structure):
    """ctypes mirror of `dwx_graph_desc` (include/dwx.h)."""
    _fields_ = [
        ("num_variables", C.c_uint64), ("num_factors", C.c_uint64),
        ("num_edges", C.c_uint64), ("num_weights", C.c_uint64),
        ("var_role", C.c_void_p), ("var_init_value", C.c_void_p),
        ("var_dtype", C.c_void_p), ("var_cardinality", C.c_void_p),
        ("num_domains", C.c_uint64),
        ("dom_vid", C.c_void_p), ("dom_offset", C.c_void_p),
        ("dom_value", C.c_void_p), ("dom_truthiness", C.c_void_p),
        ("fac_func", C.c_void_p), ("fac_edge_offset", C.c_void_p),
        ("fac_weight_id", C.c_void_p), ("fac_feature_value", C.c_void_p),
        ("edge_vid", C.c_void_p), ("edge_equal_to", C.c_void_p),
        ("w_initial_value", C.c_void_p), ("w_is_fixed", C.c_void_p),
        ("num_ghost_variables", C.c_uint64),
    ]


def _col(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


@dataclass
class RawGraph:
    # variables, indexed by variable id
    var_role: np.ndarray            # u8  isEvidence byte (evidence iff >= 1)
    var_init_value: np.ndarray      # u64 initialValue as in the file
    var_dtype: np.ndarray           # u16 0 boolean / 1 categorical
    var_cardinality: np.ndarray     # u64
    # factors, indexed by factor id (file order)
    fac_func: np.ndarray            # u16
    fac_edge_offset: np.ndarray     # u64 [F+1]
    fac_weight_id: np.ndarray       # u64
    fac_feature_value: np.ndarray   # f64
    edge_vid: np.ndarray            # u64 [E]
    edge_equal_to: np.ndarray       # u64 [E] equalPredicate as in the file
    # weights, indexed by weight id
    w_initial_value: np.ndarray     # f64
    w_is_fixed: np.ndarray          # u8
    # categorical domains (optional)
    dom_vid: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint64))
    dom_offset: np.ndarray = field(default_factory=lambda: np.zeros(1, np.uint64))
    dom_value: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint64))
    dom_truthiness: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    # sharding: the last num_ghost_variables variables are ghosts (remote, never sampled)
    num_ghost_variables: int = 0

    def __post_init__(self):
        self.var_role = _col(self.var_role, np.uint8)
        self.var_init_value = _col(self.var_init_value, np.uint64)
        self.var_dtype = _col(self.var_dtype, np.uint16)
        self.var_cardinality = _col(self.var_cardinality, np.uint64)
        self.fac_func = _col(self.fac_func, np.uint16)
        self.fac_edge_offset = _col(self.fac_edge_offset, np.uint64)
        self.fac_weight_id = _col(self.fac_weight_id, np.uint64)
        self.fac_feature_value = _col(self.fac_feature_value, np.float64)
        self.edge_vid = _col(self.edge_vid, np.uint64)
        self.edge_equal_to = _col(self.edge_equal_to, np.uint64)
        self.w_initial_value = _col(self.w_initial_value, np.float64)
        self.w_is_fixed = _col(self.w_is_fixed, np.uint8)
        self.dom_vid = _col(self.dom_vid, np.uint64)
        self.dom_offset = _col(self.dom_offset, np.uint64)
        self.dom_value = _col(self.dom_value, np.uint64)
        self.dom_truthiness = _col(self.dom_truthiness, np.float64)
        V, F, E = self.num_variables, self.num_factors, self.num_edges
        assert len(self.var_init_value) == V and len(self.var_dtype) == V
        assert len(self.var_cardinality) == V
        assert len(self.fac_edge_offset) == F + 1 and len(self.fac_weight_id) == F
        assert len(self.fac_feature_value) == F
        assert len(self.edge_equal_to) == E
        assert F == 0 or int(self.fac_edge_offset[-1]) == E
        assert len(self.w_is_fixed) == self.num_weights
        assert len(self.dom_offset) == len(self.dom_vid) + 1

    @property
    def num_variables(self):
        return len(self.var_role)

    @property
    def num_factors(self):
        return len(self.fac_func)

    @property
    def num_edges(self):
        return len(self.edge_vid)

    @property
    def num_weights(self):
        return len(self.w_initial_value)

    @property
    def is_evid(self):
        return self.var_role >= 1

    def desc(self):
        """Build the C descriptor. The RawGraph must outlive the call that uses it."""
        d = GraphDesc()
        d.num_variables = self.num_variables
        d.num_factors = self.num_factors
        d.num_edges = self.num_edges
        d.num_weights = self.num_weights
        d.num_domains = len(self.dom_vid)
        d.num_ghost_variables = int(self.num_ghost_variables)
        for name in ("var_role", "var_init_value", "var_dtype", "var_cardinality",
                     "dom_vid", "dom_offset", "dom_value", "dom_truthiness",
                     "fac_func", "fac_edge_offset", "fac_weight_id", "fac_feature_value",
                     "edge_vid", "edge_equal_to", "w_initial_value", "w_is_fixed"):
            setattr(d, name, getattr(self, name).ctypes.data)
        return d
