set -u
mkdir -p gpurun_out/r04c
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag; nproc; free -g | head -2
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg3b or binary or terms or fixtures or truth" > gpurun_out/r04c/parity.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04c/parity.log
python tools/bench_configs.py --only cfg3b > gpurun_out/r04c/cfg3b.jsonl 2> gpurun_out/r04c/cfg3b.err; echo rc $?
python - <<'PY'
import json
for l in open("gpurun_out/r04c/cfg3b.jsonl"):
    d=json.loads(l); print(d["config"], "infer", d["infer_ms_per_sweep"], "learn", d["learn_ms_per_sweep"], "wall", d["learn_wall_ms_per_sweep"], d["infer_wall_ms_per_sweep"])
PY
