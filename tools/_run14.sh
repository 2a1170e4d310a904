set -u
mkdir -p gpurun_out/r04c
DWX_TIMING=1 python tools/e2e_walltime.py --vars 100000 --skip-ref > gpurun_out/r04c/e2e_small.json 2> gpurun_out/r04c/e2e_small.err
DWX_TIMING=1 python tools/e2e_walltime.py --vars 10000000 --skip-ref > gpurun_out/r04c/e2e.json 2> gpurun_out/r04c/e2e.err
DWX_TIMING=1 timeout -k 10 400 python tools/e2e_walltime.py --vars 100000000 --skip-ref > gpurun_out/r04c/e2e_cfg5_a.json 2> gpurun_out/r04c/e2e_cfg5_a.err
DWX_TIMING=1 timeout -k 10 400 python tools/e2e_walltime.py --vars 100000000 --skip-ref > gpurun_out/r04c/e2e_cfg5_b.json 2> gpurun_out/r04c/e2e_cfg5_b.err
python - <<'PY'
import json
for f in ("e2e_small","e2e","e2e_cfg5_a","e2e_cfg5_b"):
    d=json.load(open("gpurun_out/r04c/%s.json"%f)); print(f, d["dwx_wall_s"], "startup", d.get("dwx_startup_s"), "exit", d.get("dwx_process_exit_s"))
    print("\n".join(l for l in d["dwx_phases"] if "compile" not in l and "epoch" not in l and "devb" not in l))
PY
