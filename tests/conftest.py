import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

FIXTURES = ["biased_coin", "biased_coin_continuous", "biased_coin_with_multinomial",
            "biased_coin_truthiness", "partial_observation", "sparse_domains",
            "sparse_multinomial2"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def parse_dw_args(argstr):
    """The subset of `dw gibbs` flags the fixtures use -> dict."""
    a = argstr.split()
    o = dict(l=0, i=0, alpha=0.01, diminish=0.95, reg_param=0.01, sample_evidence=False,
             learn_non_evidence=False, noise_aware=False, regularization="l2")
    i = 0
    while i < len(a):
        k = a[i]
        if k == "-l": o["l"] = int(a[i + 1]); i += 2
        elif k == "-i": o["i"] = int(a[i + 1]); i += 2
        elif k in ("--alpha", "-a"): o["alpha"] = float(a[i + 1]); i += 2
        elif k in ("--diminish", "-d"): o["diminish"] = float(a[i + 1]); i += 2
        elif k in ("--reg_param", "-b"): o["reg_param"] = float(a[i + 1]); i += 2
        elif k == "--regularization": o["regularization"] = a[i + 1]; i += 2
        elif k in ("-c", "-t"): i += 2
        elif k == "--sample_evidence": o["sample_evidence"] = True; i += 1
        elif k == "--learn_non_evidence": o["learn_non_evidence"] = True; i += 1
        elif k == "--noise_aware": o["noise_aware"] = True; i += 1
        elif k in ("--quiet", "-q"): i += 1
        else: raise ValueError("unknown flag " + k)
    return o


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
