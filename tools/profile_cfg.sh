#!/usr/bin/env bash
# tools/profile_cfg.sh CFG TAG -- PMC passes around tools/bench_configs.py --only CFG
set -u
CFG=${1:-cfg4}; TAG=${2:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${CFG}_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (PROFILE_CMD: another command under the same passes, e.g. tools/multi_sweep_bench.py --config cfg2)
CMD=${PROFILE_CMD:-"python3 $REPO/tools/bench_configs.py --only $CFG"}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_EA0_ATOMIC_sum TCC_REQ_sum" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -o pmc -- $CMD > "$OUT/pmc_$N.json" 2> "$OUT/pmc_$N.err"
done
python3 $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
