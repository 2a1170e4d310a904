#!/bin/bash
# tools/ab_bench.sh OUTDIR VARIANT... -- bench.py once per variant library (GPU box)
out=$1; shift
mkdir -p $out
for v in "$@"; do
  DWX_LIB=$PWD/sampler_amd/csrc/variants/$v.so python bench.py --no-cpu-baseline --steps 20 ${BENCH_ARGS} > $out/$v.json 2> $out/$v.err || echo "$v failed"
done
python tools/brief.py $out/*.json
