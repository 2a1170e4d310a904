"""The `dw` drop-in binary (C++ host over the C ABI): command line, native big-endian
loader, epoch driver, result files.  CPU legs run the same host sources linked against
the emulated library (tests/hipemu/build/dw_emu); the -m gpu leg runs the product binary."""
import os
import subprocess
import tempfile

import pytest

import check_result
from conftest import FIXTURES, GOLDEN, parse_dw_args
from sampler_amd import binary_format, dwx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DW = os.path.join(ROOT, "sampler_amd", "csrc", "dw")
DW_EMU = os.path.join(ROOT, "tests", "hipemu", "build", "dw_emu")


@pytest.fixture(scope="module")
def dw_emu():
    subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "tests", "hipemu")], check=True)
    return DW_EMU


def run_dw(binary, fx, out, extra=(), args=None, env=None):
    d = os.path.join(GOLDEN, fx)
    cmd = [binary, "gibbs", "-m", os.path.join(d, "graph.meta"), "-w", os.path.join(d, "graph.weights"),
           "-v", os.path.join(d, "graph.variables"), "-f", os.path.join(d, "graph.factors"), "-o", out]
    if os.path.exists(os.path.join(d, "graph.domains")):
        cmd += ["--domains", os.path.join(d, "graph.domains")]
    cmd += (args if args is not None else open(os.path.join(d, "dw-args")).read().split())
    cmd += list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, env=env)


def outputs(out):
    w = open(os.path.join(out, "inference_result.out.weights.text")).read()
    p = os.path.join(out, "inference_result.out.text")
    return w, (open(p).read() if os.path.exists(p) else "")


@pytest.mark.parametrize("fx", FIXTURES)
def test_dw_emu_end_to_end(dw_emu, fx):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, fx, out, ["--quiet", "--seed", "3"])
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
        check_result.check(fx, w, m)


@pytest.mark.parametrize("fx", ["biased_coin", "sparse_domains", "sparse_multinomial2"])
def test_dw_emu_matches_python_driver(dw_emu, fx):
    """native loader + driver + dumps == the Python mirror, byte for byte (same seed)."""
    from parity import emu_library
    d = os.path.join(GOLDEN, fx)
    o = parse_dw_args(open(os.path.join(d, "dw-args")).read())
    n_l, n_i = min(o["l"], 40), min(o["i"], 40)
    args = ["-l", str(n_l), "-i", str(n_i), "--alpha", str(o["alpha"]), "--diminish", str(o["diminish"]),
            "--reg_param", str(o["reg_param"]), "--seed", "77", "-q"]
    if o["sample_evidence"]:
        args.append("--sample_evidence")
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, fx, out, args=args)
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
    raw = binary_format.read_graph_dir(d)
    s = dwx.GibbsSampler(dwx.Graph(raw, lib=emu_library()), sample_evidence=o["sample_evidence"],
                         reg_param=o["reg_param"], seed=77)
    drv = dwx.DimmWitted(s, n_l, n_i, o["alpha"], o["diminish"])
    drv.learn()
    assert s.weights_text() == w
    drv.inference()
    assert s.marginals_text() == m


def test_dw_progress_output_and_multi_file_flags(dw_emu):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "2", "-i", "3", "-a", "0.1", "-c", "1", "-t", "4"])
        assert r.returncode == 0, r.stderr
        assert r.stdout.count("LEARNING EPOCH") == 2 and r.stdout.count("INFERENCE EPOCH") == 3
        assert "vars/sec" in r.stdout and "lmax=" in r.stdout and "TOTAL INFERENCE TIME" in r.stdout
        assert "Factor graph loaded:\t#V=18(#Vqry=9 #Vevd=9) #F=18 #W=1 #E=18 #Val=18" in r.stdout
    # -i 0: the weights file is written, the marginals file is not (src/dimmwitted.cc:89-92)
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "0", "-q"])
        assert r.returncode == 0
        assert os.path.exists(os.path.join(out, "inference_result.out.weights.text"))
        assert not os.path.exists(os.path.join(out, "inference_result.out.text"))


def test_dw_argument_errors(dw_emu):
    r = subprocess.run([dw_emu], capture_output=True, text=True)
    assert r.returncode != 0 and "Usage" in r.stderr
    r = subprocess.run([dw_emu, "frobnicate"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unrecognized MODE" in r.stderr
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(dw_emu, "biased_coin", out, args=["-i", "3"])         # -l is required
        assert r.returncode != 0 and "n_learning_epoch" in r.stderr
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "1", "-i", "1", "--bogus"])
        assert r.returncode != 0 and "--bogus" in r.stderr
        r = run_dw(dw_emu, "biased_coin", out, args=["-l", "x", "-i", "1"])
        assert r.returncode != 0
    # malformed input: meta claims more variables than the file holds
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(GOLDEN, "biased_coin")
        for f in ("graph.variables", "graph.weights", "graph.factors"):
            open(os.path.join(d, f), "wb").write(open(os.path.join(src, f), "rb").read())
        open(os.path.join(d, "graph.meta"), "w").write("1,19,18,18")
        r = subprocess.run([dw_emu, "gibbs", "-m", os.path.join(d, "graph.meta"),
                            "-v", os.path.join(d, "graph.variables"), "-w", os.path.join(d, "graph.weights"),
                            "-f", os.path.join(d, "graph.factors"), "-o", d, "-l", "1", "-i", "1"],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "variable count" in r.stderr


@pytest.mark.parametrize("asan", [False, True])
def test_dw_loader_rejects_meta_with_too_few_edges_or_factors(dw_emu, asan):
    """graph.meta announcing fewer edges / factors than the factor file holds: the parallel
    fixed-stride parse must refuse before any piece writes past the columns (it
    used to segfault with several host threads).  Run plain and under ASan/UBSan."""
    from sampler_amd import synthetic
    raw = synthetic.cfg3(30_000, n_weights=300, seed=5)        # 300 000 records: 5 pieces
    binary = os.path.join(os.path.dirname(dw_emu), "dw_emu_asan") if asan else dw_emu
    env = dict(os.environ, DWX_HOST_THREADS="8",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    with tempfile.TemporaryDirectory() as d:
        binary_format.write_graph(raw, d)
        W, V, F, E = open(os.path.join(d, "graph.meta")).read().strip().split(",")[:4]
        base = [binary, "bin2text", "-w", os.path.join(d, "graph.weights"), "-v", os.path.join(d, "graph.variables"),
                "-f", os.path.join(d, "graph.factors"), "-o", d]
        for meta, needle in (("%s,%s,%s,10" % (W, V, F), "edge count"),
                             ("%s,%s,70000,%s" % (W, V, E), "factor count"),
                             ("%s,%s,10,10" % (W, V), "count")):
            m = os.path.join(d, "bad.meta")
            open(m, "w").write(meta)
            r = subprocess.run(base + ["-m", m], capture_output=True, text=True, env=env)
            assert r.returncode == 1, (meta, r.returncode, r.stderr[-2000:])
            assert needle in r.stderr and "!= meta" in r.stderr, r.stderr[-2000:]
        # and the untouched meta still loads
        r = subprocess.run(base + ["-m", os.path.join(d, "graph.meta")], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]


def test_product_dw_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert os.path.exists(DW), "build the product first (__graft_entry__.build)"
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(DW, "biased_coin", out, args=["-l", "1", "-i", "1", "-q"])
        assert r.returncode != 0
        assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr
        assert not os.path.exists(os.path.join(out, "inference_result.out.text"))


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIXTURES)
def test_product_dw_end_to_end_on_gpu(fx):
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(DW, fx, out, ["--quiet", "--seed", "3"])
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
        check_result.check(fx, w, m)


@pytest.mark.gpu
def test_product_dw_equals_python_driver_on_gpu():
    d = os.path.join(GOLDEN, "sparse_domains")
    with tempfile.TemporaryDirectory() as out:
        r = run_dw(DW, "sparse_domains", out, args=["-l", "50", "-i", "50", "--alpha", "0.01",
                                                     "--reg_param", "0", "--seed", "9", "-q"])
        assert r.returncode == 0, r.stderr
        w, m = outputs(out)
    raw = binary_format.read_graph_dir(d)
    s = dwx.GibbsSampler(dwx.Graph(raw), reg_param=0.0, seed=9)
    drv = dwx.DimmWitted(s, 50, 50, 0.01, 0.95)
    drv.learn()
    assert s.weights_text() == w
    drv.inference()
    assert s.marginals_text() == m


@pytest.mark.parametrize("fx", ["sparse_multinomial2", "biased_coin", "sparse_domains"])
def test_dw_inference_snippets_and_calibration_blocks_like_the_reference(dw_emu, fx):
    """Non-quiet runs end with the reference's INFERENCE SNIPPETS (first ten sampled variables,
    one EXP line per value, sparse values) and the 10-bin INFERENCE CALIBRATION histogram
    (src/inference_result.cc:129-209).  Same lines as the real reference binary up to the
    Monte-Carlo numbers: identical variable/value labels, identical bin labels, equal totals."""
    import re
    from oracle import binding as orc
    if not orc.have_reference():
        pytest.skip("oracle/_ref/dw not present")
    args = ["-l", "0", "-i", "50"] + (["--sample_evidence"] if fx != "sparse_domains" else [])

    def blocks(text):
        snip = text[text.index("INFERENCE SNIPPETS"):text.index("DUMPING... TEXT    : ", text.index("INFERENCE SNIPPETS"))]
        cal = text[text.index("INFERENCE CALIBRATION"):]
        return snip.splitlines(), cal.splitlines()[:11]

    with tempfile.TemporaryDirectory() as a, tempfile.TemporaryDirectory() as b:
        mine = run_dw(dw_emu, fx, a, args=args)
        ref = run_dw(orc.REF_DW, fx, b, args=args)
        assert mine.returncode == 0 and ref.returncode == 0, mine.stderr + ref.stderr
    ms, mc = blocks(mine.stdout)
    rs, rc = blocks(ref.stdout)
    strip = lambda lines: [re.sub(r"EXP=.*", "EXP=", l) for l in lines]
    assert strip(ms) == strip(rs)
    assert all(re.fullmatch(r"      @ \d+ -> EXP=[0-9.e+-]+|   \d+  NSAMPLE=50|   \.\.\.|INFERENCE SNIPPETS \(QUERY VARIABLES\):", l) for l in ms)
    labels = lambda lines: [l.split("-->")[0] for l in lines]
    assert labels(mc) == labels(rc) and mc[0] == "INFERENCE CALIBRATION (QUERY BINS):"
    assert mc[1].startswith("PROB BIN 0.0~0.1  -->  # ")
    total = lambda lines: sum(int(l.split("#")[1]) for l in lines[1:])
    assert total(mc) == total(rc) > 0
