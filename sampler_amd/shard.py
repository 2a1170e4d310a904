"""Variable-block sharding of a factor graph.

`make_shard(raw, begin, end)` builds the LOCAL graph of the rank that owns variables
[begin, end): its owned variables first (local id = global id - begin), then the ghost
variables -- remote variables that a local factor reads -- in ascending global id.
Factors that touch at least one owned variable are kept (so a factor spanning two
shards is replicated on both sides; each side evaluates it for its own variables and
contributes the gradient of its own visits, exactly as sgd_on_variable iterates a
variable's adjacent factors, src/factor_graph.cc:262-314).  Weights stay global."""
import numpy as np

from .rawgraph import RawGraph


def make_shard(raw: RawGraph, begin: int, end: int):
    F = raw.num_factors
    off = raw.fac_edge_offset.astype(np.int64)
    arity = np.diff(off)
    fid_of_edge = np.repeat(np.arange(F, dtype=np.int64), arity)
    ev = raw.edge_vid.astype(np.int64)
    owned_edge = (ev >= begin) & (ev < end)
    keep_f = np.zeros(F, bool)
    keep_f[fid_of_edge[owned_edge]] = True
    keep_e = keep_f[fid_of_edge]
    ghosts = np.unique(ev[keep_e & ~owned_edge])
    n_owned = end - begin
    # global -> local ids
    ev_k = ev[keep_e]
    local = np.where((ev_k >= begin) & (ev_k < end), ev_k - begin,
                     n_owned + np.searchsorted(ghosts, ev_k))
    new_off = np.zeros(int(keep_f.sum()) + 1, np.uint64)
    np.cumsum(arity[keep_f], out=new_off[1:])
    ids = np.concatenate([np.arange(begin, end, dtype=np.int64), ghosts])
    # domain blocks of the variables present
    dom_vid, dom_off, dom_val, dom_tr = [], [0], [], []
    if len(raw.dom_vid):
        where = {int(v): i for i, v in enumerate(ids)}
        for b, v in enumerate(raw.dom_vid):
            if int(v) in where:
                lo, hi = int(raw.dom_offset[b]), int(raw.dom_offset[b + 1])
                dom_vid.append(where[int(v)])
                dom_val.extend(raw.dom_value[lo:hi].tolist())
                dom_tr.extend(raw.dom_truthiness[lo:hi].tolist())
                dom_off.append(len(dom_val))
    g = RawGraph(
        var_role=raw.var_role[ids], var_init_value=raw.var_init_value[ids],
        var_dtype=raw.var_dtype[ids], var_cardinality=raw.var_cardinality[ids],
        fac_func=raw.fac_func[keep_f], fac_edge_offset=new_off,
        fac_weight_id=raw.fac_weight_id[keep_f], fac_feature_value=raw.fac_feature_value[keep_f],
        edge_vid=local.astype(np.uint64), edge_equal_to=raw.edge_equal_to[keep_e],
        w_initial_value=raw.w_initial_value, w_is_fixed=raw.w_is_fixed,
        dom_vid=np.array(dom_vid, np.uint64), dom_offset=np.array(dom_off, np.uint64),
        dom_value=np.array(dom_val, np.uint64), dom_truthiness=np.array(dom_tr, np.float64),
        num_ghost_variables=len(ghosts))
    return g, ghosts.astype(np.uint64)
