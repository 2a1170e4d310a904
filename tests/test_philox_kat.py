"""Known-answer test of the device RNG.  The reference draws with erand48
(/root/reference/src/gibbs_sampler.h:177,204,230); north_star replaces it by a counter-based
per-lane generator: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
easy as 1, 2, 3", SC'11).  The vectors below are the philox4x32 10-round entries of the
Random123 distribution's kat_vectors file (counter[4], key[2] -> output[4]); both the
oracle's restatement and the device code must reproduce them, and the two uniforms the
sweep kernels derive from a block must be the documented function of it."""
import numpy as np
import pytest

from oracle import binding as orc
from sampler_amd import dwx

# (counter, key, expected)
KAT = [
    ([0x00000000] * 4, [0x00000000] * 2, [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
     [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
]


def _uniforms(block):
    """Two uniforms in [0, 1) with 53 random bits each from one 128-bit block."""
    a = int(block[0]) | (int(block[1]) << 32)
    b = int(block[2]) | (int(block[3]) << 32)
    return (a >> 11) / 2.0 ** 53, (b >> 11) / 2.0 ** 53


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_oracle_philox_known_answers(ctr, key, want):
    got = orc.philox4x32_10(key, ctr)
    assert [int(x) for x in got] == want
    seed = key[0] | (key[1] << 32)
    vid, sweep = ctr[0] | (ctr[1] << 32), ctr[2] | (ctr[3] << 32)
    assert tuple(orc.philox_uniforms(seed, vid, sweep)) == _uniforms(want)


def test_emulated_kernel_source_philox_known_answers():
    """The product's kernel source compiled for the host (tests/hipemu): same text, CPU."""
    from parity import emu_library
    lib = emu_library()
    for ctr, key, want in KAT:
        out, uni = lib.test_philox(key, ctr)
        assert [int(x) for x in out] == want
        assert tuple(uni) == _uniforms(want)


@pytest.mark.gpu
@pytest.mark.parametrize("ctr,key,want", KAT)
def test_device_philox_known_answers(ctr, key, want):
    out, uni = dwx.default_library().test_philox(key, ctr, device=0)
    assert [int(x) for x in out] == want
    assert tuple(uni) == _uniforms(want)
    # and the oracle draws the very same uniforms for the same (seed, variable, sweep)
    seed = key[0] | (key[1] << 32)
    assert tuple(orc.philox_uniforms(seed, ctr[0] | (ctr[1] << 32), ctr[2] | (ctr[3] << 32))) == tuple(uni)
