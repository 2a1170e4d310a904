#!/bin/bash
# tools/ab_configs.sh OUTDIR CONFIGS VARIANT... -- tools/bench_configs.py --only CONFIGS once per variant library (GPU box)
out=$1; cfgs=$2; shift; shift
mkdir -p $out
for v in "$@"; do
  DWX_LIB=$PWD/sampler_amd/csrc/variants/$v.so python tools/bench_configs.py --only $cfgs > $out/$v.jsonl 2> $out/$v.err || echo "$v failed"
  python - <<PY
import json
for l in open("$out/$v.jsonl"):
    d = json.loads(l)
    print("%-8s %-10s infer %.3f ms  learn %s" % ("$v", d["config"], d["infer_ms_per_sweep"], d["learn_ms_per_sweep"]))
PY
done
