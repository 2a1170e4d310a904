#!/usr/bin/env bash
# tools/profile_cfg.sh CFG TAG -- PMC passes around tools/bench_configs.py --only CFG
set -u
CFG=${1:-cfg4}; TAG=${2:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${CFG}_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (PROFILE_CMD: another command under the same passes; it must START WITH THE INTERPRETER BINARY -- e.g.
#  PROFILE_CMD="python3 $REPO/tools/multi_sweep_bench.py --config cfg2" -- never a script with an env shebang,
#  env, bash -c or any launcher that re-execs: the profiler has initialised the GPU by then)
case "${PROFILE_CMD:-python3}" in python3\ *|python3|python\ *|/*/python3\ *|./*) ;; *) echo "PROFILE_CMD must start with an interpreter binary or a compiled program" >&2; exit 2;; esac
CMD=${PROFILE_CMD:-"python3 $REPO/tools/bench_configs.py --only $CFG"}
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.json" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "TCC_EA0_ATOMIC_sum TCC_REQ_sum" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$N" -o pmc -- $CMD > "$OUT/pmc_$N.json" 2> "$OUT/pmc_$N.err"
done
python3 $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
