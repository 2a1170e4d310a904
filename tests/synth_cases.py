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
