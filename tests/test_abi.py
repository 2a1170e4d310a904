"""The C-ABI library loads without a GPU and exports every symbol include/dwx.h
declares; graph compilation (host only) works; creating a sampler without a device
fails loudly (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from sampler_amd import dwx, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dwx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dwx_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(dwx.SYMBOLS)


def test_library_exports_every_symbol():
    lib = dwx.default_library()
    for name in _header_symbols():
        assert hasattr(lib.L, name), name
    assert lib.L.dwx_version() == 1


def test_graph_compile_host_only_and_no_cpu_fallback():
    import torch
    lib = dwx.default_library()
    g = dwx.Graph(synthetic.cfg3b(500, n_weights=10, seed=1), lib=lib)
    assert g.info.num_variables == 500 and g.info.num_colors >= 2
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device failure cannot be observed")
    with pytest.raises(dwx.DwxError) as e:
        dwx.GibbsSampler(g)
    assert e.value.code == dwx.DWX_E_DEVICE


def test_malformed_graphs_are_rejected():
    lib = dwx.default_library()
    raw = synthetic.cfg2(10, n_weights=2, seed=1)
    raw.fac_weight_id[3] = 99
    with pytest.raises(dwx.DwxError) as e:
        dwx.Graph(raw, lib=lib)
    assert e.value.code == dwx.DWX_E_INVALID
    raw = synthetic.cfg2(10, n_weights=2, seed=1)
    raw.fac_func[0] = 5          # not a FACTOR_FUNCTION_TYPE
    with pytest.raises(dwx.DwxError):
        dwx.Graph(raw, lib=lib)
    raw = synthetic.cfg2(10, n_weights=2, seed=1)
    raw.edge_vid[0] = 10_000
    with pytest.raises(dwx.DwxError):
        dwx.Graph(raw, lib=lib)
    raw = synthetic.cfg2(10, n_weights=2, seed=1)
    raw.var_dtype[0] = 2
    with pytest.raises(dwx.DwxError):
        dwx.Graph(raw, lib=lib)
    # feature values: finite; on a learnable weight within the fixed-point gradient range
    raw = synthetic.cfg3(10, n_weights=2, seed=1)
    raw.fac_feature_value[5] = float("nan")
    with pytest.raises(dwx.DwxError) as e:
        dwx.Graph(raw, lib=lib)
    assert e.value.code == dwx.DWX_E_INVALID
    raw = synthetic.cfg3(10, n_weights=2, seed=1)
    raw.fac_feature_value[5] = 1e6
    with pytest.raises(dwx.DwxError) as e:
        dwx.Graph(raw, lib=lib)
    assert e.value.code == dwx.DWX_E_LIMIT and "65536" in str(e.value)
    raw = synthetic.cfg2(10, n_weights=2, seed=1)      # fixed weights: no gradient, no limit
    raw.fac_feature_value[5] = 1e6
    dwx.Graph(raw, lib=lib)


def test_empty_graph():
    from sampler_amd.rawgraph import RawGraph
    z8, zf = np.zeros(0, np.uint64), np.zeros(0)
    raw = RawGraph(np.zeros(0, np.uint8), z8, np.zeros(0, np.uint16), z8, np.zeros(0, np.uint16),
                   np.zeros(1, np.uint64), z8, zf, z8, z8, zf, np.zeros(0, np.uint8))
    g = dwx.Graph(raw)
    assert g.info.num_variables == 0 and g.info.num_tiles == 0


def test_device_init_without_gpu_fails_loudly():
    """dwx_device_init (the optional early HIP start-up) reports the missing device through the
    usual error channel instead of crashing; on a GPU box it succeeds (tests/test_gpu_parity.py
    runs everything after it)."""
    import torch
    lib = dwx.default_library()
    rc = lib.L.dwx_device_init(0)
    if torch.cuda.is_available():
        assert rc == 0
        assert lib.L.dwx_device_init(10_000) != 0
    else:
        assert rc == dwx.DWX_E_DEVICE
        assert b"HIP" in lib.L.dwx_last_error() or b"device" in lib.L.dwx_last_error()
