"""Run the emulated-kernel parity tests once more with the ASan + UBSan build of the
kernel/API sources (GPU sanitizers are unavailable on the pool; this is the CPU
sanitizer leg).  Any out-of-bounds index in the kernels or the graph compiler aborts
the subprocess."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_under_asan_ubsan():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True,
                             text=True, check=True).stdout.strip()
    libubsan = subprocess.run(["g++", "-print-file-name=libubsan.so"], capture_output=True,
                              text=True, check=True).stdout.strip()
    env = dict(os.environ, DWX_EMU_ASAN="1", LD_PRELOAD=libasan + ":" + libubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:detect_stack_use_after_return=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # (the long statistical runs -- KS against the reference's marginals, 30-epoch learning --
    # execute the same kernel paths as the parity tests many more times: left to the plain build)
    # (the cases are independent: four workers when pytest-xdist is there -- the ASan build runs the
    # fibers of the emulation several times slower than the plain one)
    try:
        import xdist  # noqa: F401
        workers = ["-n", "4"] if (os.cpu_count() or 1) >= 4 else []
    except ImportError:
        workers = []
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *workers,
                        "-k", "not ks_against and not heavy_tying",
                        os.path.join(ROOT, "tests", "test_kernels_emu.py"),
                        os.path.join(ROOT, "tests", "test_multi_sweep.py"), "-m", "not gpu"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert "passed" in r.stdout
