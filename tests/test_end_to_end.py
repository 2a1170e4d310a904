"""The reference's seven end-to-end statistical fixtures with the reference's own
acceptance tolerances (check_result), run with the fixture's own flags and epoch
counts through:  the oracle in reference mode (pin), the kernel source under host
emulation (no GPU), and -- with -m gpu -- the HIP library on the device."""
import os

import pytest

import check_result
from conftest import FIXTURES, GOLDEN, parse_dw_args
from oracle import binding as orc
from sampler_amd import binary_format, dwx


def _load(fx):
    d = os.path.join(GOLDEN, fx)
    return binary_format.read_graph_dir(d), parse_dw_args(open(os.path.join(d, "dw-args")).read())


@pytest.mark.parametrize("fx", FIXTURES)
def test_golden_reference_outputs_pass_check_result(fx):
    d = os.path.join(GOLDEN, fx)
    check_result.check(fx, open(os.path.join(d, "ref_full.weights.text")).read(),
                       open(os.path.join(d, "ref_full.text")).read())


@pytest.mark.parametrize("fx", FIXTURES)
def test_oracle_reference_mode(fx):
    raw, o = _load(fx)
    s = orc.Oracle(raw, o["sample_evidence"], o["learn_non_evidence"], o["noise_aware"],
                   o["regularization"], o["reg_param"])
    s.set_workers(1)
    s.learn(o["l"], o["alpha"], o["diminish"])
    w = s.weights_text()
    s.inference(o["i"])
    check_result.check(fx, w, s.marginals_text())


def _run_dwx(lib, fx, seed):
    raw, o = _load(fx)
    g = dwx.Graph(raw, lib=lib)
    s = dwx.GibbsSampler(g, sample_evidence=o["sample_evidence"],
                         learn_non_evidence=o["learn_non_evidence"], noise_aware=o["noise_aware"],
                         regularization=o["regularization"], reg_param=o["reg_param"], seed=seed)
    drv = dwx.DimmWitted(s, o["l"], o["i"], o["alpha"], o["diminish"])
    drv.learn()
    w = s.weights_text()
    drv.inference()
    check_result.check(fx, w, s.marginals_text())


@pytest.mark.parametrize("fx", FIXTURES)
def test_emulated_kernels_end_to_end(fx):
    from parity import emu_library
    _run_dwx(emu_library(), fx, seed=2024)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 3])
@pytest.mark.parametrize("fx", FIXTURES)
def test_gpu_end_to_end(fx, seed):
    from parity import gpu_library
    _run_dwx(gpu_library(), fx, seed=seed)
