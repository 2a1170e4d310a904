set -u
mkdir -p gpurun_out/r04c/ab_lrec16
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cfg3b or binary or terms or fixtures" > gpurun_out/r04c/parity_lrec16.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04c/parity_lrec16.log
python tools/bench_configs.py --only cfg3b > gpurun_out/r04c/ab_lrec16/wg4.jsonl 2> gpurun_out/r04c/ab_lrec16/wg4.err
bash tools/ab_configs.sh gpurun_out/r04c/ab_lrec16 cfg3b wg3 wg5
python tools/bench_configs.py --only cfg3b > gpurun_out/r04c/ab_lrec16/wg4_again.jsonl 2> gpurun_out/r04c/ab_lrec16/wg4_again.err
python - <<'PY'
import json
for f in ("wg4","wg4_again"):
    for l in open("gpurun_out/r04c/ab_lrec16/%s.jsonl"%f):
        d=json.loads(l); print(f, d["config"], "infer", round(d["infer_ms_per_sweep"],4), "learn", round(d["learn_ms_per_sweep"],4), "wall", round(d["learn_wall_ms_per_sweep"],4))
PY
