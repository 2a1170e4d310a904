#!/usr/bin/env bash
# tools/profile_round.sh TAG -- the round's whole evidence set in one GPU call: the bench line, the
# rocprofv3 kernel trace + PMC passes of the bench (config 3) and of configs 2 / 3b / 4 / 5b-shard,
# the all-config timing table, the calibration tools, the kernel resource table.  Everything lands
# under gpurun_out/; tools/physical_table.py turns the profile directories into physical.json.
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
echo "bench done"
bash tools/profile_gpu.sh $TAG > gpurun_out/$TAG/profile_bench.log 2>&1
echo "profile bench done"
BENCH_ARGS="--workload cfg5b" bash tools/profile_gpu.sh ${TAG}_cfg5b > gpurun_out/$TAG/profile_cfg5b.log 2>&1
echo "profile cfg5b done"
for c in cfg2 cfg3b cfg4; do bash tools/profile_cfg.sh $c $TAG > gpurun_out/$TAG/profile_$c.log 2>&1; echo "profile $c done"; done
cd "$REPO"
python tools/bench_configs.py > gpurun_out/$TAG/configs.jsonl 2> gpurun_out/$TAG/configs.err
echo "configs done"
python bench.py --workload cfg5b --no-cpu-baseline > gpurun_out/$TAG/bench_cfg5b_1gpu.json 2> gpurun_out/$TAG/bench_cfg5b_1gpu.err
./tools/sorted_bench > gpurun_out/$TAG/sorted_bench.jsonl 2> gpurun_out/$TAG/sorted_bench.err
DWX_TIMING=1 python tools/e2e_walltime.py --vars 10000000 --skip-ref > gpurun_out/$TAG/e2e.json 2> gpurun_out/$TAG/e2e.err
# (tools/kernel_resources.py recompiles the library: run it in the build container, not on the GPU box)
python tools/physical_table.py bench=gpurun_out/prof_$TAG cfg5b=gpurun_out/prof_${TAG}_cfg5b cfg2=gpurun_out/prof_cfg2_$TAG \
  cfg3b=gpurun_out/prof_cfg3b_$TAG cfg4=gpurun_out/prof_cfg4_$TAG > gpurun_out/$TAG/physical.json
python tools/update_traffic.py cfg3=gpurun_out/prof_$TAG cfg5b=gpurun_out/prof_${TAG}_cfg5b > gpurun_out/$TAG/traffic.json 2> gpurun_out/$TAG/traffic.err
cp profiles/traffic.json gpurun_out/$TAG/traffic_stamped.json
python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_with_traffic.json 2> gpurun_out/$TAG/bench_with_traffic.err
python tools/physical_table.py --check-bench gpurun_out/$TAG/bench_with_traffic.json > gpurun_out/$TAG/check_bench.json 2>&1
echo "all done"
