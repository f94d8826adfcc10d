"""Developer tool: table of per-section counter differences from a tools/cuts_run.py run under rocprofv3 --pmc."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
order = json.load(open("gpurun_out/cuts_order.json"))
kern = "k_tp_chain" if order["free"] else "k_tpc_chain"
per_step = 1 if order["free"] else 3   # launches of the kernel per step (three ranges with contacts)
import os
f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)   # the newest run in the directory
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if not r["Kernel_Name"].startswith(kern):
        continue
    k = int(r["Dispatch_Id"])
    disp.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp[k]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ds = [disp[k] for k in sorted(disp)]
skip = order["warm_steps"] * per_step
ds = ds[skip:]
names = {0: "prologue", 18: "check pos/vel", 19: "joint-local pose", 12: "chain pose + anchor", 13: "comPos cinert cdof", 14: "vel rne crb M rows",
         15: "factor M", 16: "-", 1: "passive + smooth solve", 2: "contact assembly", 3: "constraint rows", 4: "warm start", 5: "constraint update 1",
         6: "Hessian assembly", 7: "Cholesky + fwd", 8: "back substitution", 9: "-", 23: "M*search", 24: "J*search", 10: "line search + update",
         25: "later iterations", 17: "final update + outputs", 26: "Euler", -1: "stores + end"}
cols = [c for c in ds[0] if c != "_ns"] + ["_ns"]
prev = {c: 0.0 for c in cols}
waves = None
print(f"{'section':28s}" + "".join(f"{c[-16:]:>18s}" for c in cols))
for i, cut in enumerate(order["cuts"]):
    grp = ds[i * order["rep"] * per_step:(i + 1) * order["rep"] * per_step]
    cur = {c: sum(g.get(c, 0.0) for g in grp) / order["rep"] for c in cols}   # per step (all ranges)
    if waves is None:
        waves = cur.get("SQ_WAVES", 0.0)
    div = waves if waves else 1.0
    print(f"{names.get(cut, str(cut)):28s}" + "".join(f"{(cur[c] - prev[c]) / (div if c not in ('_ns',) else 1e3):18.1f}" for c in cols))
    prev = cur
print("(per wavefront of the kernel; _ns column: microseconds per step summed over the kernel's launches, serialized by the profiler)")
