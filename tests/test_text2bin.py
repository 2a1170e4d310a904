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


def _factor_bytes(recs):
    """recs: [(func, [(vid, equal_to), ...], wid, fval)] -> big-endian factor records."""
    import struct
    out = bytearray()
    for func, edges, wid, fval in recs:
        out += struct.pack(">HQ", func, len(edges))
        for vid, eq in edges:
            out += struct.pack(">QQ", vid, eq)
        out += struct.pack(">Qd", wid, fval)
    return bytes(out)


def _load_through_bin2text(tmp, recs, n_vars=8, n_weights=4, n_factors=None, n_edges=None, raw=None, split=None):
    """Writes a graph whose factor file(s) hold `recs`, loads it with `dw bin2text` (the
    native loader) and returns (returncode, stderr, factors.tsv lines)."""
    import struct
    n_factors = len(recs) if n_factors is None else n_factors
    n_edges = sum(len(r[1]) for r in recs) if n_edges is None else n_edges
    open(os.path.join(tmp, "graph.meta"), "w").write("%d,%d,%d,%d" % (n_weights, n_vars, n_factors, n_edges))
    open(os.path.join(tmp, "graph.variables"), "wb").write(
        b"".join(struct.pack(">QBQHQ", v, 0, 0, 0, 2) for v in range(n_vars)))
    open(os.path.join(tmp, "graph.weights"), "wb").write(
        b"".join(struct.pack(">QBd", w, 0, 0.0) for w in range(n_weights)))
    blob = _factor_bytes(recs) if raw is None else raw
    files = []
    cuts = [0] + (split or []) + [len(recs)]
    if raw is not None or not split:
        open(os.path.join(tmp, "graph.factors"), "wb").write(blob)
        files = [os.path.join(tmp, "graph.factors")]
    else:
        for i in range(len(cuts) - 1):
            fn = os.path.join(tmp, "graph.factors.%d" % i)
            open(fn, "wb").write(_factor_bytes(recs[cuts[i]:cuts[i + 1]]))
            files.append(fn)
    out = os.path.join(tmp, "txt")
    os.makedirs(out, exist_ok=True)
    cmd = [DW, "bin2text", "-m", os.path.join(tmp, "graph.meta"), "-v", os.path.join(tmp, "graph.variables"),
           "-w", os.path.join(tmp, "graph.weights"), "-o", out]
    for f in files:
        cmd += ["-f", f]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = open(os.path.join(out, "factors.tsv")).read().splitlines() if r.returncode == 0 else None
    return r.returncode, r.stderr, lines


def _expected_lines(recs):
    out = []
    for func, edges, wid, fval in recs:
        cols = [str(v) for v, _ in edges] + ([str(e) for _, e in edges] if func == 12 else [])
        out.append("\t".join(cols + [str(wid), "%g" % fval]))
    return out


def test_factor_loader_cuts_files_at_record_boundaries():
    """The native loader decodes factor files in parallel pieces (fixed stride when every
    record has the first one's arity, otherwise a hop over the arity fields): uniform
    files, mixed arities, the look-alike case where the file size is a multiple of the
    first record's size, several files, and > 64k records (more than one piece)."""
    import random
    rnd = random.Random(5)

    def rec(arity):
        return (rnd.choice([0, 1, 2, 3, 4]), [(rnd.randrange(8), 1) for _ in range(arity)],
                rnd.randrange(4), float(rnd.randrange(-3, 4)))

    cases = {
        "uniform": [rec(2) for _ in range(1000)],
        "mixed": [rec(rnd.randint(1, 4)) for _ in range(1000)],
        # first record arity 1 (42 B); arities 3 (74 B) + ... chosen so the size is a multiple of 42
        "lookalike": [rec(1)] + [rec(3), rec(3), rec(3)] * 7 + [rec(1)] * 20,
        "many_uniform": [rec(1) for _ in range(150_000)],
        "many_mixed": [rec(1 + (i % 3 == 0)) for i in range(150_000)],
    }
    assert len(_factor_bytes(cases["lookalike"])) % 42 == 0
    for name, recs in cases.items():
        with tempfile.TemporaryDirectory() as t:
            rc, err, lines = _load_through_bin2text(t, recs)
            assert rc == 0, (name, err)
            assert lines == _expected_lines(recs), name
    with tempfile.TemporaryDirectory() as t:      # three files: uniform, mixed, uniform
        recs = [rec(2) for _ in range(500)] + [rec(rnd.randint(1, 3)) for _ in range(500)] + [rec(1) for _ in range(500)]
        rc, err, lines = _load_through_bin2text(t, recs, split=[500, 1000])
        assert rc == 0, err
        assert lines == _expected_lines(recs)


def test_factor_loader_rejects_malformed_files():
    import random
    rnd = random.Random(6)
    recs = [(4, [(rnd.randrange(8), 1)], 0, 1.0) for _ in range(100)]
    blob = _factor_bytes(recs)
    with tempfile.TemporaryDirectory() as t:      # cut in the middle of a record
        rc, err, _ = _load_through_bin2text(t, recs, raw=blob[:-5])
        assert rc != 0 and "truncated" in err
    with tempfile.TemporaryDirectory() as t:      # fewer records than graph.meta announces
        rc, err, _ = _load_through_bin2text(t, recs, n_factors=101, n_edges=101)
        assert rc != 0 and "factor count" in err
    with tempfile.TemporaryDirectory() as t:      # more records than graph.meta announces
        rc, err, _ = _load_through_bin2text(t, recs, n_factors=99, n_edges=99)
        assert rc != 0 and "count" in err
    with tempfile.TemporaryDirectory() as t:      # an arity field pointing far past the file
        bad = bytearray(blob)
        bad[42 * 50 + 2:42 * 50 + 10] = (1 << 40).to_bytes(8, "big")
        rc, err, _ = _load_through_bin2text(t, recs, raw=bytes(bad))
        assert rc != 0 and "truncated" in err


def test_variable_loader_parallel_path():
    """One variable file with exactly #V >= 65 536 records is decoded by all host threads (dw_cli.cc,
    load_variables): shuffled ids, roles, initial values and cardinalities come back exactly (through
    `dw bin2text`'s variables.tsv); an id met twice -- another one is then missing -- is rejected as on the
    serial path."""
    import struct
    import numpy as np
    rng = np.random.default_rng(5)
    n = 70_000
    ids = rng.permutation(n)
    role = rng.integers(0, 2, n)
    dtype = rng.integers(0, 2, n)
    card = np.where(dtype == 1, rng.integers(2, 7, n), 2)
    init = np.where(role == 1, rng.integers(0, 1 << 30, n) % card, 0)

    def write(tmp, id_list):
        open(os.path.join(tmp, "graph.meta"), "w").write("1,%d,1,1" % n)
        open(os.path.join(tmp, "graph.variables"), "wb").write(
            b"".join(struct.pack(">QBQHQ", int(v), int(role[v]), int(init[v]), int(dtype[v]), int(card[v])) for v in id_list))
        open(os.path.join(tmp, "graph.weights"), "wb").write(struct.pack(">QBd", 0, 0, 0.0))
        open(os.path.join(tmp, "graph.factors"), "wb").write(_factor_bytes([(4, [(0, 1)], 0, 1.0)]))
        out = os.path.join(tmp, "txt")
        os.makedirs(out, exist_ok=True)
        return subprocess.run([DW, "bin2text", "-m", os.path.join(tmp, "graph.meta"), "-v", os.path.join(tmp, "graph.variables"),
                               "-w", os.path.join(tmp, "graph.weights"), "-f", os.path.join(tmp, "graph.factors"), "-o", out],
                              capture_output=True, text=True), out

    with tempfile.TemporaryDirectory() as t:
        r, out = write(t, ids)
        assert r.returncode == 0, r.stderr
        lines = open(os.path.join(out, "variables.tsv")).read().splitlines()
        assert len(lines) == n
        for v in list(range(0, n, 997)) + [n - 1]:
            cols = lines[v].split("\t")
            assert int(cols[0]) == v and int(cols[1]) == role[v] and int(cols[3]) == dtype[v] and int(cols[4]) == card[v]
            assert int(cols[2]) == (init[v] if role[v] else 0)
    with tempfile.TemporaryDirectory() as t:
        twice = ids.copy()
        twice[123] = twice[456]          # (one id twice, one missing: still #V records)
        r, _ = write(t, twice)
        assert r.returncode != 0 and "variable count" in r.stderr
