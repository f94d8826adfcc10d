"""Turns the rocprofv3 outputs under gpurun_out/ (kernel-trace --stats runs and separate --pmc passes
of `bench.py` and `bench.py --contacts --envs 65536`, the commands are in profiles/collect.sh, see profiles/README.md) into
the committed summaries: profiles/r01_*_kernel_stats.csv (verbatim copies) and r01_pmc_summary.json
(per-launch averages per kernel)."""
import collections
import csv
import glob
import json
import shutil


def agg(d):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    return {k: {a: b / n[(k, a)] for a, b in v.items()} for k, v in acc.items() if k.startswith("k_")}


def stats(d, dst):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(f, dst)
    return {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
            for r in csv.DictReader(open(f)) if r["Name"].startswith("k_")}


out = {"config1_kernel_stats": stats("p_tp_stats", "profiles/r01_tp_kernel_stats.csv"),
       "config3_kernel_stats": stats("p_c_stats", "profiles/r01_contacts_kernel_stats.csv")}
for tag, ds in (("config1_pmc_per_launch", ["p_tp_fetch", "p_tp_write", "p_tp_sq", "p_tp_sq2"]),
                ("config3_pmc_per_launch", ["p_c_fetch", "p_c_write", "p_c_sq"])):
    m = collections.defaultdict(dict)
    for d in ds:
        for k, v in agg(d).items():
            m[k].update(v)
    out[tag] = m
out["notes"] = ("per-launch averages over all dispatches of the kernel. config1 = `python3 bench.py --steps 500 --warmup 100 "
                "--no-cpu-baseline` (k_tp: 50 steps x 4096 envs per launch); config3 = `python3 bench.py --contacts --envs 65536 "
                "--steps 50 --warmup 50 --no-cpu-baseline` (one step per launch, two ranges of 32768 envs on two streams). "
                "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; HBM-side traffic estimate = (2 x FETCH_SIZE + "
                "WRITE_SIZE) x 1024 B (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md; strided 8-byte accesses are "
                "uncalibrated). SQ_*CYCLES and SQ_WAIT/ACTIVE counters are in quad-cycles.")
json.dump(out, open("profiles/r01_pmc_summary.json", "w"), indent=1)
k = out["config1_pmc_per_launch"]["k_tp"]
print({a: round(b) for a, b in k.items()})
print("k_tp traffic MB/launch", (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / 1e6)
print(out["config1_kernel_stats"])
print(out["config3_kernel_stats"])
print({kk: {a: round(b) for a, b in v.items()} for kk, v in out["config3_pmc_per_launch"].items() if kk in ("k_tpc", "k_c_narrow_list", "k_tp_col")})
