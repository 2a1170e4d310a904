#!/bin/bash
# Host-phase timing of `dw gibbs` (load, compile, upload, dumps) at several host thread
# counts, on a config-3 graph written to $TMPDIR.  Run on the GPU box:
#   tools/host_phases.sh [vars]
set -e
V=${1:-10000000}
D=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0, ".")
from sampler_amd import synthetic, binary_format
binary_format.write_graph(synthetic.cfg3($V, n_weights=max(1, $V // 10), seed=1234), "$D")
PY
mkdir -p $D/out
for T in 1 8 32 64 128; do
  echo "== DWX_HOST_THREADS=$T"
  export DWX_TIMING=1 DWX_HOST_THREADS=$T
  time sampler_amd/csrc/dw gibbs -m $D/graph.meta -v $D/graph.variables \
    -w $D/graph.weights -f $D/graph.factors -o $D/out -l 1 -i 1 -q 2>&1 | grep -v "^Factor\|loading\|initializing\|TOTAL\|DUMPING"
done
rm -rf $D
