import numpy as np
import pytest

from edge_cases import cases
from parity import emu_library, gpu_library, run_parity

CASES = cases()


def _run(lib, name, raw, kw):
    s, o = run_parity(lib, raw, n_learn=12, n_infer=12, stepsize=0.05, **kw)
    if name == "all_fixed":
        assert np.array_equal(s.weights, raw.w_initial_value)
    if name == "all_evidence":
        assert (s.tallies()[1] == 0).all()
        assert np.array_equal(s.assignments("evid"), raw.var_init_value)
    if name == "isolated_variables":
        t, n = s.tallies()
        assert n[1] == 12 and n[3] == 12 and n[2] == 0
        base, _ = s.graph.values()
        assert t[int(base[3])] == 12           # cardinality-1 variable always takes its only value


@pytest.mark.parametrize("name,raw,kw", CASES, ids=[c[0] for c in CASES])
def test_edge_case_emulated(name, raw, kw):
    _run(emu_library(), name, raw, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("name,raw,kw", CASES, ids=[c[0] for c in CASES])
def test_edge_case_gpu(name, raw, kw):
    _run(gpu_library(), name, raw, kw)
