"""The synthetic golden cases of tests/golden/make_golden.py: same generator calls, so
the graphs the reference was run on can be rebuilt instead of committed."""
import hashlib
import json
import os
import tempfile

from sampler_amd import binary_format, synthetic

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = {
    "synth_cfg2": lambda: synthetic.cfg2(2000, n_weights=200, seed=1234),
    "synth_cfg3": lambda: synthetic.cfg3(2000, n_weights=200, seed=1234),
    "synth_cfg3b": lambda: synthetic.cfg3b(2000, n_weights=200, seed=1234),
    "synth_cfg4": lambda: synthetic.cfg4(1000, card=8, seed=1234, learn=False),
}


def load(name, verify=True):
    raw = CASES[name]()
    if verify:
        want = json.load(open(os.path.join(GOLDEN, "synth_graph_sha256.json")))[name]
        with tempfile.TemporaryDirectory() as d:
            binary_format.write_graph(raw, d)
            for f, h in want.items():
                got = hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()
                assert got == h, "regenerated %s/%s differs from the graph the reference ran on" % (name, f)
    return raw


def rotate_variables(g, shift):
    """The same factor graph with variable ids rotated by `shift` (an isomorphic relabelling:
    the model and its weights are unchanged, but a sampler -- the reference, whose seeds come
    from an un-seeded rand() and cannot be set, or this build -- now spends its random streams
    on other variables and visits them in another order).  tests/golden/make_golden.py `tied`
    runs the reference on these; tests/test_tied_weights.py runs this build on the same ones."""
    import numpy as np
    from sampler_amd.rawgraph import RawGraph
    V = g.num_variables
    new_of_old = (np.arange(V, dtype=np.uint64) + np.uint64(shift)) % np.uint64(V)
    old_of_new = np.empty(V, np.int64)
    old_of_new[new_of_old.astype(np.int64)] = np.arange(V)
    return RawGraph(
        var_role=g.var_role[old_of_new], var_init_value=g.var_init_value[old_of_new],
        var_dtype=g.var_dtype[old_of_new], var_cardinality=g.var_cardinality[old_of_new],
        fac_func=g.fac_func, fac_edge_offset=g.fac_edge_offset, fac_weight_id=g.fac_weight_id,
        fac_feature_value=g.fac_feature_value, edge_vid=new_of_old[g.edge_vid.astype(np.int64)],
        edge_equal_to=g.edge_equal_to, w_initial_value=g.w_initial_value, w_is_fixed=g.w_is_fixed)


def tied_shift(num_variables, j, n_rotations):
    return j * (num_variables // n_rotations) + 7919 * j
