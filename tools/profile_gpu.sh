#!/usr/bin/env bash
# tools/profile_gpu.sh TAG -- run on the GPU box (via gpurun; BENCH_ARGS='--workload cfg5b' for other
# workloads): kernel-trace stats of the
# default bench command, then the PMC passes (each in its own run, as the MI355X guide
# prescribes).  Everything lands in gpurun_out/prof_TAG/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-repeat-infer --no-gather-ceiling --min-time 0 ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_EA0_ATOMIC_sum TCC_REQ_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -o pmc -- $BENCH > "$OUT/pmc_$N.json" 2> "$OUT/pmc_$N.err"
done
python3 $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
