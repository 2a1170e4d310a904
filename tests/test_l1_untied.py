"""`--regularization l1` on UNTIED weights (one factor, hence one visit per weight and sweep), against
the oracle's REFERENCE mode (byte-pinned to the real `dw`): the reference's update of one visit is
`w += reg_param * (w < 0); w -= stepsize * g` (src/inference_result.h:76-78) -- the push is a jump of the
whole reg_param and does not stop at zero.  The batched device update must reproduce it exactly when a
batch holds one visit (the gradient-flow form stopped at the zero crossing -- w0 = -0.001,
reg_param = 0.01, no gradient: 0.005 where the reference lands on 0.009).

The graph makes every draw deterministic, so the two random number generators cannot matter: each
evidence variable carries one factor on a FIXED weight of +-20 (its free chain sits at 1 or 0 with
probability 1 - 4e-18) and one factor on its own learnable weight, whose gradient per visit is then a
constant: 0 (free == evidence), -2 (free 0, evidence 1) or +2 (free 1, evidence 0)."""
import numpy as np
import pytest

from sampler_amd import dwx
from sampler_amd.rawgraph import DTYPE_BOOLEAN, FUNC_ISTRUE, RawGraph

W0 = [-0.001, -0.004, -0.02, -0.0095, 0.0, 0.003, 0.0005, -0.0105, 0.012, -1e-9]


def _graph():
    # groups: (evidence value, fixed weight) -> per-visit gradient g = sign(free) - sign(evidence)
    groups = [(1, +20.0), (1, -20.0), (0, +20.0), (0, -20.0)]      # g = 0, -2, +2, 0
    n = len(W0) * len(groups)
    role = np.ones(n, np.uint8)
    init = np.zeros(n, np.uint64)
    wid, func = [], []
    w_init, w_fixed = [+20.0, -20.0], [1, 1]
    for gi, (ev, fw) in enumerate(groups):
        for k, w0 in enumerate(W0):
            v = gi * len(W0) + k
            init[v] = ev
            wid += [0 if fw > 0 else 1, len(w_init)]      # the factor on the fixed weight first
            w_init.append(w0); w_fixed.append(0)
    F = 2 * n
    return RawGraph(
        var_role=role, var_init_value=init, var_dtype=np.full(n, DTYPE_BOOLEAN, np.uint16),
        var_cardinality=np.full(n, 2, np.uint64), fac_func=np.full(F, FUNC_ISTRUE, np.uint16),
        fac_edge_offset=np.arange(F + 1, dtype=np.uint64), fac_weight_id=np.array(wid, np.uint64),
        fac_feature_value=np.ones(F), edge_vid=np.repeat(np.arange(n, dtype=np.uint64), 2),
        edge_equal_to=np.ones(F, np.uint64), w_initial_value=np.array(w_init), w_is_fixed=np.array(w_fixed, np.uint8))


def _check(lib, reg_param, stepsize, epochs=12):
    from oracle import binding as orc
    raw = _graph()
    o = orc.Oracle(raw, regularization="l1", reg_param=reg_param)          # REFERENCE mode: sequential, per visit
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, regularization="l1", reg_param=reg_param, seed=5)
    cur = stepsize
    for e in range(epochs):
        o.sample_sgd(cur)
        s.sample_sgd(cur); s.wait()
        assert np.array_equal(s.assignments("free"), o.assignments("free")), "the draws are not deterministic"
        np.testing.assert_allclose(s.weights, o.weights, rtol=0, atol=1e-15, err_msg="epoch %d" % e)
        cur *= 0.95
    w = np.asarray(s.weights)[2:].reshape(4, len(W0))
    # the advisor's example: -0.001 + 0.01 = 0.009 and it stays (no gradient, not negative any more)
    if reg_param == 0.01:
        assert abs(w[0, 0] - 0.009) < 1e-15 and abs(w[3, 0] - 0.009) < 1e-15
    return w


@pytest.mark.parametrize("reg_param,stepsize", [(0.01, 0.001), (0.01, 0.004), (0.003, 0.001), (0.0, 0.01)])
def test_one_visit_per_batch_equals_the_reference_update_emulated(reg_param, stepsize):
    from parity import emu_library
    _check(emu_library(), reg_param, stepsize)


@pytest.mark.gpu
@pytest.mark.parametrize("reg_param,stepsize", [(0.01, 0.001), (0.01, 0.004), (0.003, 0.001)])
def test_one_visit_per_batch_equals_the_reference_update_gpu(reg_param, stepsize):
    _check(dwx.default_library(), reg_param, stepsize)
