#!/usr/bin/env bash
# tools/big_parity.sh -- the opt-in whole-config-5-on-one-GPU parity test (tests/test_gpu_parity.py),
# first at 50 M variables to measure the host memory it takes, then -- if twice that fits the box's
# limit with room to spare -- at the full 100 M.  Logs under gpurun_out/big/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
mkdir -p gpurun_out/big
( while true; do echo "$(date +%T) $(cat /sys/fs/cgroup/memory.current 2>/dev/null)" >> gpurun_out/big/mem.log; sleep 15; done ) &
LOGGER=$!
trap 'kill $LOGGER 2>/dev/null' EXIT
run() {
  DWX_BIG_TESTS=1 DWX_BIG_VARS=$1 timeout -k 10 $2 python - "$1" > gpurun_out/big/parity_$1.log 2>&1 <<'PY'
import resource, subprocess, sys, time
t0 = time.time()
import pytest
rc = pytest.main(["tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-s", "-k", "config5"])
print("V", sys.argv[1], "rc", int(rc), "wall_s %.1f" % (time.time() - t0),
      "maxrss_GB %.1f" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))
sys.exit(int(rc))
PY
}
run 50000000 600 || { tail -5 gpurun_out/big/parity_50000000.log; exit 1; }
tail -4 gpurun_out/big/parity_50000000.log
RSS=$(grep -o "maxrss_GB [0-9.]*" gpurun_out/big/parity_50000000.log | cut -d' ' -f2)
python3 -c "import sys; sys.exit(0 if 2.1 * float('$RSS') < 260 else 1)" || { echo "100 M would not fit: stopping at 50 M"; exit 0; }
run 100000000 1000 || { tail -5 gpurun_out/big/parity_100000000.log; exit 1; }
tail -4 gpurun_out/big/parity_100000000.log
