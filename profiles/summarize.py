"""Turns the rocprofv3 outputs under gpurun_out/ (kernel-trace --stats runs and separate --pmc passes
of `bench.py` and `bench.py --contacts --envs 65536`, the commands are in profiles/collect.sh, see profiles/README.md) into
the committed summaries: profiles/rNN_*_kernel_stats.csv (verbatim copies) and rNN_pmc_summary.json (NN = ROUND below)
(per-launch averages per kernel)."""
import collections
import csv
import glob
import json
import shutil

ROUND = "r03"


def agg(d):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    return {fam(k.replace("void ", "").split("<")[0]): {a: b / n[(k, a)] for a, b in v.items()} for k, v in acc.items() if "k_" in k}


def fam(name):
    """k_tp_chain(4) / k_tpc_chain(4) are the instantiations of k_tp / k_tpc this model runs (DPP shifts on the dof chains):
    the summary keeps the family names its readers (bench.py) use; the kernel-stats CSVs keep the real ones."""
    return name[:-7] if name.endswith("_chain4") else name[:-6] if name.endswith("_chain") else name


def stats(d, dst):
    f = glob.glob(f"gpurun_out/{d}/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(f, dst)
    return {fam(r["Name"].split("(")[0].replace("void ", "").split("<")[0]): {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
            for r in csv.DictReader(open(f)) if "k_" in r["Name"] and int(r["Calls"]) > 1}


out = {"config1_kernel_stats": stats("p_tp_stats", f"profiles/{ROUND}_tp_kernel_stats.csv"),
       "contacts_kernel_stats": stats("p_c_stats", f"profiles/{ROUND}_contacts_kernel_stats.csv")}
for tag, ds in (("config1_pmc_per_launch", ["p_tp_fetch", "p_tp_write", "p_tp_sq", "p_tp_sq2"]),
                ("contacts_pmc_per_launch", ["p_c_fetch", "p_c_write", "p_c_sq", "p_c_sq2", "p_c_mem"])):
    m = collections.defaultdict(dict)
    for d in ds:
        for k, v in agg(d).items():
            m[k].update(v)
    out[tag] = m
# one step of the contact path = one launch of each of the three kernels per range (three ranges of about
# 21 800 envs at 65 536): per-step traffic = 3 x the per-launch averages
out["contacts_pmc_per_step"] = {k: {a: 3 * b for a, b in v.items() if a in ("FETCH_SIZE", "WRITE_SIZE")}
                                for k, v in out["contacts_pmc_per_launch"].items() if k in ("k_tp_col", "k_c_narrow_list", "k_tpc")}



def hw_flop(k):
    """fp64 flop ISSUED per launch: wavefront instructions x 64 lanes (masked lanes included), FMA = 2."""
    return 64.0 * (2 * k.get("SQ_INSTS_VALU_FMA_F64", 0.0) + k.get("SQ_INSTS_VALU_ADD_F64", 0.0)
                   + k.get("SQ_INSTS_VALU_MUL_F64", 0.0) + k.get("SQ_INSTS_VALU_TRANS_F64", 0.0))


# hardware-counted fp64 work per env-step (bench.py's roofline.hw_fp64_frac = this x env-steps/s / 78.6 TFLOP/s):
# config1: one k_tp launch = 50 steps x 4096 envs; contacts: one launch of each kernel per range of 65 536 / 3 envs
cpl = out["contacts_pmc_per_launch"]
out["hw_fp64_flop_per_env_step"] = {
    "config1": hw_flop(out["config1_pmc_per_launch"]["k_tp"]) / (4096 * 50),
    "contacts": sum(hw_flop(cpl[k]) for k in ("k_tp_col", "k_c_narrow_list", "k_tpc") if k in cpl) / (65536 / 3.0),
    "contacts_per_kernel": {k: hw_flop(cpl[k]) / (65536 / 3.0) for k in ("k_tp_col", "k_c_narrow_list", "k_tpc") if k in cpl}}
out["notes"] = ("per-launch averages over all dispatches of the kernel; k_tp / k_tpc stand for the chain instantiations k_tp_chain / k_tpc_chain that this model runs. config1 = `python3 bench.py --envs 4096 --steps 500 --warmup 100 "
                "--no-cpu-baseline` (k_tp: 50 steps x 4096 envs per launch); contacts = `python3 bench.py --contacts --envs 65536 "
                "--steps 50 --warmup 50 --no-cpu-baseline` (one step per launch, three ranges of about 21 800 envs on three streams). "
                "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; HBM-side traffic estimate = (2 x FETCH_SIZE + "
                "WRITE_SIZE) x 1024 B (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md; strided 8-byte accesses are "
                "uncalibrated). SQ_*CYCLES and SQ_WAIT/ACTIVE counters are in quad-cycles.")
json.dump(out, open(f"profiles/{ROUND}_pmc_summary.json", "w"), indent=1)
k = out["config1_pmc_per_launch"]["k_tp"]
print({a: round(b) for a, b in k.items()})
print("k_tp traffic MB/launch", (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / 1e6)
print(out["config1_kernel_stats"])
print(out["contacts_kernel_stats"])
print({kk: {a: round(b) for a, b in v.items()} for kk, v in out["contacts_pmc_per_launch"].items() if kk in ("k_tpc", "k_c_narrow_list", "k_tp_col")})
