"""gpurun_out/scale/*.log (tools/scale_envs.sh) -> profiles/r03_scale_envs.json"""
import glob, json, os
rows = []
for f in sorted(glob.glob("gpurun_out/scale/*.log")):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    r = json.loads(lines[-1])
    row = {"run": os.path.basename(f)[:-4], "env_steps_per_s": r.get("value"), "ms_per_step": r.get("ms_per_step")}
    cfg = r.get("config", {})
    for k in ("n_envs_per_gpu", "device_memory", "kernel_path", "mean_solver_iter"):
        if k in cfg:
            row[k] = cfg[k]
    rf = r.get("roofline", {})
    for k in ("frac", "frac_measured"):
        if k in rf:
            row[k] = rf[k]
    for k in ("newton", "pgs", "parts"):
        if k in r:
            row[k] = r[k]
    rows.append(row)
json.dump({"note": "single-GPU scale points and secondary configurations, `bash tools/scale_envs.sh` = bench.py variants with --no-cpu-baseline on one MI355X, END of round 3 (after the second-half kernel work)", "rows": rows},
          open("profiles/r03_scale_envs.json", "w"), indent=1)
for r in rows:
    print(r["run"], round((r["env_steps_per_s"] or 0) / 1e6, 2), "M", r.get("ms_per_step"), (r.get("device_memory") or {}).get("total"))
