#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference and oracle/_ref/dw, built by
oracle/build_ref.sh).  What it commits is DATA, never reference source:

  tests/golden/<fixture>/graph.{meta,variables,weights,factors,domains}
      the reference's own TSV test fixtures (/root/reference/test/<fixture>/*.tsv)
      converted with the reference's `dw text2bin` exactly as
      /root/reference/test/run_end_to_end.sh:6-15 does (per-kind files are
      concatenated in glob order, as its `cat graph.factors*` does).
  tests/golden/<fixture>/dw-args
      the fixture's sampler flags (test/<fixture>/dw-args).
  tests/golden/<fixture>/ref_{short,full}.{weights.text,text}
      outputs of the reference `dw gibbs -t 1 -c 1` (single worker => bit
      reproducible) with the fixture's flags ("full") and with
      the epoch counts cut to <=100 ("short").
  tests/golden/synth_*/...
      small synthetic graphs (written by sampler_amd.binary_format from
      sampler_amd.synthetic) plus the reference's multi-threaded marginals on them.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DW_REFERENCE_DIR", "/root/reference")
DW = os.path.join(ROOT, "oracle", "_ref", "dw")

FIXTURES = [
    "biased_coin",
    "biased_coin_continuous",
    "biased_coin_with_multinomial",
    "biased_coin_truthiness",
    "partial_observation",
    "sparse_domains",
    "sparse_multinomial2",
]


def short_args(args):
    """Cut -l/-i epoch counts to at most 100."""
    out = list(args)
    for i, a in enumerate(out):
        if a in ("-l", "-i"):
            out[i + 1] = str(min(int(out[i + 1]), 100))
    return out


def run_ref(gdir, args, outdir):
    cmd = [DW, "gibbs", "-m", os.path.join(gdir, "graph.meta"),
           "-w", os.path.join(gdir, "graph.weights"),
           "-v", os.path.join(gdir, "graph.variables"),
           "-f", os.path.join(gdir, "graph.factors"),
           "-o", outdir, "--quiet"]
    if os.path.exists(os.path.join(gdir, "graph.domains")):
        cmd += ["--domains", os.path.join(gdir, "graph.domains")]
    cmd += args
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)


def convert_fixture(name):
    src = os.path.join(REF, "test", name)
    dst = os.path.join(HERE, name)
    os.makedirs(dst, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for what in ("variable", "domain", "factor", "weight"):
            parts = []
            for tsv in sorted(glob.glob(os.path.join(src, what + "s*.tsv"))):
                base = os.path.basename(tsv)[:-4]
                out = os.path.join(tmp, "graph." + base)
                extra = []
                argf = os.path.join(src, base + ".text2bin-args")
                if os.path.exists(argf):
                    extra = open(argf).read().split()
                subprocess.run([DW, "text2bin", what, tsv, out, "/dev/null"] + extra,
                               check=True)
                parts.append(out)
            if parts:
                # run_end_to_end.sh feeds `cat graph.<kind>s*` (glob order)
                parts.sort()
                with open(os.path.join(dst, "graph." + what + "s"), "wb") as f:
                    for p in parts:
                        f.write(open(p, "rb").read())
    # the fixture's TSV inputs and text2bin arguments (test DATA of the reference), so the
    # `dw text2bin` of this build can be checked against the converted binaries
    tsv_dir = os.path.join(dst, "tsv")
    os.makedirs(tsv_dir, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(src, "*.tsv")) + glob.glob(os.path.join(src, "*.text2bin-args"))):
        shutil.copy(f, os.path.join(tsv_dir, os.path.basename(f)))
        os.chmod(os.path.join(tsv_dir, os.path.basename(f)), 0o644)
    shutil.copy(os.path.join(src, "graph.meta"), os.path.join(dst, "graph.meta"))
    shutil.copy(os.path.join(src, "dw-args"), os.path.join(dst, "dw-args"))
    os.chmod(os.path.join(dst, "graph.meta"), 0o644)
    os.chmod(os.path.join(dst, "dw-args"), 0o644)
    args = open(os.path.join(src, "dw-args")).read().split()
    # strip any -c/-t the fixture sets; goldens are single worker, single copy
    clean = []
    skip = False
    for a in args:
        if skip:
            skip = False
            continue
        if a in ("-c", "-t", "--n_datacopy", "--n_threads"):
            skip = True
            continue
        clean.append(a)
    for tag, a in (("full", clean), ("short", short_args(clean))):
        with tempfile.TemporaryDirectory() as out:
            run_ref(dst, a + ["-t", "1", "-c", "1"], out)
            shutil.copy(os.path.join(out, "inference_result.out.weights.text"),
                        os.path.join(dst, "ref_%s.weights.text" % tag))
            res = os.path.join(out, "inference_result.out.text")
            if os.path.exists(res):
                shutil.copy(res, os.path.join(dst, "ref_%s.text" % tag))
        with open(os.path.join(dst, "ref_%s.args" % tag), "w") as f:
            f.write(" ".join(a + ["-t", "1", "-c", "1"]) + "\n")


FLAG_GOLDENS = {
    # tag: flags appended to the fixture's "short" arguments (TCLAP keeps the last value of a
    # repeated flag, /root/reference/src/cmd_parser.cc getLastValueOrDefault)
    "l1": ["--regularization", "l1", "--reg_param", "0.01"],
    "lne": ["--learn_non_evidence"],
    "l1_lne": ["--regularization", "l1", "--reg_param", "0.02", "--learn_non_evidence"],
}


def flag_goldens():
    """`-t 1 -c 1` byte goldens of all seven fixtures with the learning flags none of the
    reference's own dw-args exercise: --regularization l1 (src/inference_result.h:76-78) and
    --learn_non_evidence (src/gibbs_sampler.h:144-146).  Pins the oracle's reference mode on
    those branches (tests/test_oracle_golden.py).  sparse_multinomial2 ships with -l 0; it
    gets 50 learning epochs here so that the flags do something."""
    for name in FIXTURES:
        dst = os.path.join(HERE, name)
        base = open(os.path.join(dst, "ref_short.args")).read().split()
        if name == "sparse_multinomial2":
            base[base.index("-l") + 1] = "50"
        for tag, extra in FLAG_GOLDENS.items():
            a = base + extra
            with tempfile.TemporaryDirectory() as out:
                run_ref(dst, a, out)
                shutil.copy(os.path.join(out, "inference_result.out.weights.text"),
                            os.path.join(dst, "ref_%s.weights.text" % tag))
                shutil.copy(os.path.join(out, "inference_result.out.text"),
                            os.path.join(dst, "ref_%s.text" % tag))
            with open(os.path.join(dst, "ref_%s.args" % tag), "w") as f:
                f.write(" ".join(a) + "\n")
        print("golden:", name, "l1 / learn_non_evidence")


def synth_goldens():
    """Reference marginals / weights on small synthetic graphs (1/500 scale of
    BASELINE.json configs 2-4).  The reference runs multi-threaded here, so these
    are statistical (KS / tolerance) pins, not bit-exact ones."""
    sys.path.insert(0, ROOT)
    from sampler_amd import synthetic, binary_format
    cases = {
        "synth_cfg2": (synthetic.cfg2(2000, n_weights=200, seed=1234),
                       ["-l", "0", "-i", "400"]),
        "synth_cfg3": (synthetic.cfg3(2000, n_weights=200, seed=1234),
                       ["-l", "60", "-i", "400", "--alpha", "0.01",
                        "--diminish", "0.95", "--reg_param", "0.01"]),
        "synth_cfg3b": (synthetic.cfg3b(2000, n_weights=200, seed=1234),
                        ["-l", "60", "-i", "400", "--alpha", "0.01",
                         "--diminish", "0.95", "--reg_param", "0.01"]),
        "synth_cfg4": (synthetic.cfg4(1000, card=8, seed=1234, learn=False),
                       ["-l", "0", "-i", "400"]),
    }
    sums = {}
    for name, (g, args) in cases.items():
        dst = os.path.join(HERE, name)
        os.makedirs(dst, exist_ok=True)
        binary_format.write_graph(g, dst)
        with open(os.path.join(dst, "dw-args"), "w") as f:
            f.write(" ".join(args) + "\n")
        with tempfile.TemporaryDirectory() as out:
            run_ref(dst, args, out)
            shutil.copy(os.path.join(out, "inference_result.out.weights.text"),
                        os.path.join(dst, "ref.weights.text"))
            shutil.copy(os.path.join(out, "inference_result.out.text"),
                        os.path.join(dst, "ref.text"))
        # the graph files themselves are not committed (0.3-1 MB each): tests rebuild
        # them from sampler_amd.synthetic with the same arguments and check the sha256
        import hashlib
        sums[name] = {f: hashlib.sha256(open(os.path.join(dst, f), "rb").read()).hexdigest()
                      for f in sorted(os.listdir(dst)) if f.startswith("graph.")}
        for f in list(os.listdir(dst)):
            if f.startswith("graph."):
                os.remove(os.path.join(dst, f))
    import json
    json.dump(sums, open(os.path.join(HERE, "synth_graph_sha256.json"), "w"), indent=1,
              sort_keys=True)


def synth_second_runs():
    """A SECOND reference run (other thread count: other shards and seeds) of the learning
    goldens, kept as ref_b.*: learned weights are noisy, so marginals produced by two
    independent learn+infer runs differ per variable by more than Monte-Carlo error; the
    distance between the reference's own two runs is the yardstick for this build's distance
    to the reference (tests/ks_golden.py)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sampler_amd import binary_format
    import synth_cases
    for name in ("synth_cfg3", "synth_cfg3b"):
        dst = os.path.join(HERE, name)
        args = open(os.path.join(dst, "dw-args")).read().split()
        g = synth_cases.load(name)
        with tempfile.TemporaryDirectory() as d, tempfile.TemporaryDirectory() as out:
            binary_format.write_graph(g, d)
            run_ref(d, args + ["-t", "3", "-c", "1"], out)
            shutil.copy(os.path.join(out, "inference_result.out.weights.text"), os.path.join(dst, "ref_b.weights.text"))
            shutil.copy(os.path.join(out, "inference_result.out.text"), os.path.join(dst, "ref_b.text"))
        print("golden:", name, "second run")


TIED_CASES = {
    # name: (generator call, sampler flags, learning-epoch counts)
    "tied_one": ("tied(100000, 1000, 1, seed=7)",
                 ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 20, 60]),
    "tied_three": ("tied(60000, 600, 3, seed=8)",
                   ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 20, 60]),
    "tied_cfg4learn": ("cfg4(20000, card=8, seed=1234, learn=True)",
                       ["--alpha", "0.001", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 10]),
    "tied_cfg4learn_bigstep": ("cfg4(20000, card=8, seed=1234, learn=True)",
                               ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"], [1, 5, 10, 40]),
    # --regularization l1 (src/inference_result.h:76-78: the push stops at the zero crossing)
    "tied_one_l1": ("tied(100000, 1000, 1, seed=7, p_one=(0.3,))",
                    ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01", "--regularization", "l1"],
                    [1, 2, 5, 20]),
    "tied_three_l1": ("tied(60000, 600, 3, seed=8)",
                      ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01", "--regularization", "l1"],
                      [1, 2, 5, 20]),
    "tied_cfg4learn_l1": ("cfg4(20000, card=8, seed=1234, learn=True)",
                          ["--alpha", "0.001", "--diminish", "0.95", "--reg_param", "0.01", "--regularization", "l1"],
                          [1, 2, 5, 20]),
    # weights tied through PAIRWISE factors (src/factor_graph.cc:243-314 visits every grounding)
    "tied_pair": ("cfg3b(30000, n_weights=4)",
                  ["--alpha", "0.001", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 20]),
    "tied_pair_bigstep": ("cfg3b(30000, n_weights=4)",
                          ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 20]),
    "tied_chain": ("chain(100000)",
                   ["--alpha", "0.01", "--diminish", "0.95", "--reg_param", "0.01"], [1, 2, 5, 20]),
}
TIED_ROTATIONS = 3
# every rotation is run with each of these thread counts: the reference's seeds come from an
# un-seeded rand() (worker i always gets the same erand48 state), so runs that differ only in the
# rotation share their random streams -- their spread understates the reference's run-to-run
# spread (measured on cfg4 with learning, 2 epochs: the means of six rotations at -t 1 / 3 / 4 / 7
# are 0.708 / 0.715 / 0.721 / 0.662 for one weight while the six of one thread count agree to
# 0.014).  Other thread counts = other shards meeting other streams: the INDEPENDENT samples are
# the thread counts, and tests/test_tied_weights.py takes its standard error from their means.
# weights[L] is rotation-major: run index = rotation * len(threads) + thread index.
TIED_THREADS = [1, 2, 3, 4, 5, 6, 7, 8]


def tied_goldens():
    """Learned weights of the reference on graphs whose weights are tied to 10^4..10^5 factors
    (one weight per rule, the DeepDive shape), at SEVERAL epoch counts, each on six rotations
    of the variable ids: the reference's seeds cannot be set (un-seeded rand()), and runs that
    differ only in the thread count share the stream of worker 0 -- relabelled graphs are the
    same model with independent noise, i.e. the reference's run-to-run spread.  The device's
    batched update is checked against that spread at equal epochs
    (tests/test_tied_weights.py).  Data only: tests/golden/tied_weights.json."""
    sys.path.insert(0, ROOT)
    import json
    from sampler_amd import synthetic, binary_format  # noqa: F401
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth_cases import rotate_variables, tied_shift
    path = os.path.join(HERE, "tied_weights.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, (call, flags, epochs) in TIED_CASES.items():
        if name in out and os.environ.get("TIED_REGENERATE") != "1":
            continue      # reference runs are reproducible only in distribution: keep what is pinned
        g0 = eval("synthetic." + call)
        runs = {str(L): [] for L in epochs}
        for j in range(TIED_ROTATIONS):
            g = rotate_variables(g0, tied_shift(g0.num_variables, j, TIED_ROTATIONS))
            with tempfile.TemporaryDirectory() as d:
                binary_format.write_graph(g, d)
                for L in epochs:
                    for t in TIED_THREADS:
                        with tempfile.TemporaryDirectory() as o:
                            run_ref(d, ["-l", str(L), "-i", "0", "-t", str(t), "-c", "1"] + flags, o)
                            runs[str(L)].append([float(l.split()[1]) for l in
                                                 open(os.path.join(o, "inference_result.out.weights.text"))])
        out[name] = {"generator": call, "flags": flags, "threads": TIED_THREADS,
                     "rotations": TIED_ROTATIONS, "weights": runs}
        print("golden:", name)
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)


def codec_goldens():
    """The reference's text2bin codec fixtures (test/text2bin/: TSV inputs and xxd dumps of
    the expected big-endian bytes) -- data files of the reference's tests."""
    src = os.path.join(REF, "test", "text2bin")
    dst = os.path.join(HERE, "text2bin")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)


if __name__ == "__main__":
    if not os.path.exists(DW):
        sys.exit("oracle/_ref/dw missing: run oracle/build_ref.sh first")
    which = sys.argv[1:] or ["fixtures", "synth"]
    if "fixtures" in which:
        for fx in FIXTURES:
            convert_fixture(fx)
            print("golden:", fx)
    if "fixtures" in which:
        codec_goldens()
        print("golden: text2bin codec fixtures")
    if "flags" in which:
        flag_goldens()
    if "synth" in which:
        synth_goldens()
        print("golden: synthetic")
    if "synth_b" in which:
        synth_second_runs()
    if "tied" in which:
        tied_goldens()
