"""Factor-function truth tables (reference test/factor_test.cc) on the oracle, on the
kernel source under emulation and -- with -m gpu -- on the device."""
import pytest

from oracle import binding as orc
from truth_tables import CASES


@pytest.mark.parametrize("func,sat,want", CASES)
def test_oracle_truth_table(func, sat, want):
    assert abs(orc.factor_sign(func, sat) - want) < 1e-12


def test_emulated_kernel_truth_table():
    from parity import emu_library
    lib = emu_library()
    for func, sat, want in CASES:
        assert abs(lib.test_factor_sign(func, sat) - want) < 1e-12, (func, sat)


@pytest.mark.gpu
def test_device_truth_table():
    from parity import gpu_library
    lib = gpu_library()
    for func, sat, want in CASES:
        assert abs(lib.test_factor_sign(func, sat) - want) < 1e-12, (func, sat)
