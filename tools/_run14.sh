set -u
mkdir -p gpurun_out/r04c
for i in 1 2; do
DWX_TIMING=1 timeout -k 10 400 python tools/e2e_walltime.py --vars 100000000 --skip-ref > gpurun_out/r04c/e2e_cfg5_drop$i.json 2> gpurun_out/r04c/e2e_cfg5_a.err
DWX_NO_EXIT_DROP=1 DWX_TIMING=1 timeout -k 10 400 python tools/e2e_walltime.py --vars 100000000 --skip-ref > gpurun_out/r04c/e2e_cfg5_nodrop$i.json 2> gpurun_out/r04c/e2e_cfg5_b.err
done
python - <<'PY'
import json
for f in ("drop1","nodrop1","drop2","nodrop2"):
    d=json.load(open("gpurun_out/r04c/e2e_cfg5_%s.json"%f)); 
    ph=[l for l in d["dwx_phases"] if "dw timing" in l and "epoch" not in l and "skipped" not in l]
    tot=sum(float(l.split(":")[-1].split()[0]) for l in ph)
    print(f, d["dwx_wall_s"], "exit", d.get("dwx_process_exit_s"), "phases", round(tot,2), "unaccounted", round(d["dwx_wall_s"]-d.get("dwx_process_exit_s")-tot,2))
PY
