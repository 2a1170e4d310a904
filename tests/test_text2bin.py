"""`dw text2bin` / `dw bin2text` (host utilities of the drop-in CLI): exact big-endian
bytes against the reference's codec fixtures (test/text2bin/*.bin.txt, xxd dumps) and
against the binaries the reference's own text2bin produced from the seven fixtures' TSVs."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

from conftest import FIXTURES, GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DW = os.path.join(ROOT, "sampler_amd", "csrc", "dw")


def xxd_to_bytes(path):
    out = bytearray()
    for line in open(path):
        m = re.match(r"^[0-9a-f]+:\s+((?:[0-9a-f]{2,4}\s)+)", line)
        if m:
            out += bytes.fromhex(m.group(1).replace(" ", ""))
    return bytes(out)


@pytest.mark.parametrize("kind,extra", [("variable", []), ("weight", []), ("factor", ["2", "1", "1"])])
def test_codec_golden_bytes(kind, extra):
    d = os.path.join(GOLDEN, "text2bin")
    with tempfile.TemporaryDirectory() as t:
        out, cnt = os.path.join(t, "o.bin"), os.path.join(t, "count")
        r = subprocess.run([DW, "text2bin", kind, os.path.join(d, "dd_%ss.txt" % kind), out, cnt] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == xxd_to_bytes(os.path.join(d, "dd_%ss.bin.txt" % kind))
        assert int(open(cnt).read()) > 0


@pytest.mark.parametrize("fx", FIXTURES)
def test_text2bin_equals_reference_conversion(fx):
    d = os.path.join(GOLDEN, fx)
    with tempfile.TemporaryDirectory() as t:
        for what in ("variable", "domain", "factor", "weight"):
            parts = []
            for tsv in sorted(glob.glob(os.path.join(d, "tsv", what + "s*.tsv"))):
                base = os.path.basename(tsv)[:-4]
                argf = os.path.join(d, "tsv", base + ".text2bin-args")
                extra = open(argf).read().split() if os.path.exists(argf) else []
                out = os.path.join(t, "graph." + base)
                r = subprocess.run([DW, "text2bin", what, tsv, out, os.path.join(t, "cnt")] + extra,
                                   capture_output=True, text=True)
                assert r.returncode == 0, r.stderr
                parts.append(out)
            if parts:
                got = b"".join(open(p, "rb").read() for p in sorted(parts))
                assert got == open(os.path.join(d, "graph.%ss" % what), "rb").read(), what


def test_bin2text_round_trip():
    """bin2text then text2bin reproduces the binaries (boolean fixture)."""
    d = os.path.join(GOLDEN, "partial_observation")
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([DW, "bin2text", "-m", os.path.join(d, "graph.meta"), "-v", os.path.join(d, "graph.variables"),
                            "-w", os.path.join(d, "graph.weights"), "-f", os.path.join(d, "graph.factors"), "-o", t],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(os.path.join(t, "graph.meta")).read().startswith("2,12,8,16,")
        for what, extra in (("variable", []), ("weight", []), ("factor", ["3", "2", "1", "1"])):
            out = os.path.join(t, "graph." + what + "s")
            r = subprocess.run([DW, "text2bin", what, os.path.join(t, what + "s.tsv"), out, os.path.join(t, "c")] + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert open(out, "rb").read() == open(os.path.join(d, "graph.%ss" % what), "rb").read()
    # categorical fixture with sparse domains: the dump lists the domain values in order
    d = os.path.join(GOLDEN, "sparse_domains")
    with tempfile.TemporaryDirectory() as t:
        r = subprocess.run([DW, "bin2text", "-m", os.path.join(d, "graph.meta"), "-v", os.path.join(d, "graph.variables"),
                            "-w", os.path.join(d, "graph.weights"), "-f", os.path.join(d, "graph.factors"),
                            "--domains", os.path.join(d, "graph.domains"), "-o", t], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        dom = dict(l.split("\t")[0::2] for l in open(os.path.join(t, "domains.tsv")).read().splitlines())
        assert dom["14"] == "{0,1,3}" and dom["16"] == "{1,3}" and dom["0"] == "{0,1,2,3}"
        assert len(open(os.path.join(t, "factors.tsv")).read().splitlines()) == 67


def test_text2bin_errors():
    r = subprocess.run([DW, "text2bin", "variable"], capture_output=True, text=True)
    assert r.returncode != 0
    with tempfile.TemporaryDirectory() as t:
        bad = os.path.join(t, "bad.tsv")
        open(bad, "w").write("1\tx\n")
        r = subprocess.run([DW, "text2bin", "variable", bad, os.path.join(t, "o"), os.path.join(t, "c")],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "bad variable line" in r.stderr
        r = subprocess.run([DW, "text2bin", "nonsense", bad, os.path.join(t, "o"), os.path.join(t, "c")],
                           capture_output=True, text=True)
        assert r.returncode != 0
