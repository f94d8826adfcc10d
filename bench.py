#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched physics step (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  From a plain shell (no
WORLD_SIZE in the environment) `--gpus N > 1` starts its own ranks: `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child process (mujoco_amd/launch.py), before anything here touches the GPU;
the child's JSON line is relayed and its exit code returned.

Headline at EVERY N — BASELINE.json configs[4]'s shape, one workload per GPU so that the 1 -> 8 curve divides
like by like: DexterousDynamos.xml with STL-hull mesh self-collision + a ground plane (the plane is this build's
augmentation), 131 072 envs per GPU, Euler + Newton, fp64, ctrl resampled every 50 steps, obs (qpos|qvel fp32) +
reward all-gathered over RCCL once per control period, overlapped with the next steps (weak scaling, no traffic
inside a step).  The line carries `ranks_joined` (an all-reduce of 1) and every rank's device name.
N = 1 adds, in the same line, `cpu_baseline` (the oracle on the same workload) and the blocks
    "config1_4096":    configs[1] — joint-space only (contacts disabled), 4096 lockstep envs (k_tp_chain4), with its
                       own `roofline` and `cpu_baseline`;
    "contacts_65536":  configs[2] — the headline's model at 65 536 envs without the collective, with `roofline`.

A "step" is one mj_step of the whole batch.  State and the ctrl bank are resident in HBM before the timed
region.  Timing: W warm-up steps, then the K-step region is timed R >= 5 times back to back (each bracketed by
barrier + synchronize, MAX over ranks) and the MEDIAN is reported with `steps` = K, so a short K (the driver's
20) does not measure one launch's jitter.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_ENV_STEP = 36.0e3      # contact-free, Euler, Newton (algorithmic fp64 flop)
BYTES_PER_ENV_STEP = 1.3e3      # compulsory HBM bytes per env-step
FP64_PEAK_TFLOPS = 78.6         # MI355X fp64 vector == fp64 matrix peak (datasheet [EXT])
CTRL_PERIOD = 50
REPS = 15                       # repetitions of the timed K-step region (median reported); 5 once they have taken REPS_BUDGET_S
REPS_BUDGET_S = 8.0
PMC_FILES = ("r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json")
FLOP_PER_MPR_ITER = 1.01e3      # 2 hill-climb supports x ~430 + 150 simplex update (12 k per pair = 12 of these)


def pmc_traffic(section, kernels):
    """HBM-side bytes per launch of `kernels` (summed) from the committed rocprofv3 PMC summary of this
    round if present, else of round 1: (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of
    MI355X_MICROARCH.md.  bench.py cannot collect counters itself; None if the summary is missing."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                sec = json.load(f)[section]
            return sum((2.0 * sec[k]["FETCH_SIZE"] + sec[k]["WRITE_SIZE"]) * 1024.0 for k in kernels), name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_hw_flop(workload):
    """fp64 flop the hardware ISSUED per env-step (SQ_INSTS_VALU_{FMA x 2, ADD, MUL, TRANS}_F64 x 64 lanes, all
    kernels of the step) from the committed PMC summary: (flop per env-step, file) or (None, None).  Counted per
    wavefront instruction, so lanes that are masked off count as issued: an upper bound of the useful work and
    the figure to hold against the 78.6 TFLOP/s the vector ALU can issue."""
    for name in PMC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return float(json.load(f)["hw_fp64_flop_per_env_step"][workload]), name
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def cpu_baseline(model, seconds_budget=12.0, n_steps=1000):
    """Oracle (CPU restatement, kind 'port') on this host, single thread, bounded sample."""
    from mujoco_amd import workload
    from oracle import OracleSim
    n_envs = 2048
    q, v = workload.config2_state(model, n_envs, seed=1234)
    c = workload.config2_ctrl(model, n_envs, seed=1234)
    t0 = time.perf_counter()
    done = 0
    s = OracleSim(model)
    for e in range(n_envs):
        s.reset()
        s.qpos[:] = q[e]
        s.ctrl[:] = c[e]
        s.step(n_steps)
        done += n_steps
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="env-steps/s", cores=1, kind="port",
                sample=f"{done // n_steps} envs x {n_steps} steps of the same workload "
                       f"(oracle/liboracle.so, gcc -O2, single thread, {dt:.1f} s)")


def _pool_worker(args):
    path, seed, n_envs, n_steps, contacts = args
    sys.path.insert(0, ROOT)
    from mujoco_amd import Model, workload
    from oracle import OracleSim
    model = Model.load(path)
    if not contacts:
        model = workload.contact_free(model)
    else:
        model = workload.with_ground_plane(model, z=-0.05)
    q, v = workload.config2_state(model, n_envs, seed=seed)
    c = workload.config2_ctrl(model, n_envs, seed=seed)
    t0 = time.perf_counter()
    for e in range(n_envs):
        s = OracleSim(model)
        s.qpos[:] = q[e]
        s.ctrl[:] = c[e]
        s.step(n_steps)
    return n_envs * n_steps, time.perf_counter() - t0


def cpu_baseline_allcores(model_path, contacts=False):
    """One oracle process per core of this job's CPU share.  The share is what the box grants the job
    (OMP_NUM_THREADS, 16 per GPU on the MI355X pool, where os.cpu_count() reports the whole machine and
    oversubscribing it gets the job killed); both numbers are reported."""
    import multiprocessing as mp
    share = os.environ.get("OMP_NUM_THREADS")
    cores = min(os.cpu_count() or 1, int(share)) if share and share.isdigit() else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    n_envs, n_steps = (400, 200) if contacts else (400, 1000)   # a few seconds per worker
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_pool_worker, [(model_path, 1234 + i, n_envs, n_steps, contacts) for i in range(cores)])
    wall = time.perf_counter() - t0
    work = sum(r[0] for r in res)
    inner = max(r[1] for r in res)
    return dict(value=work / inner, unit="env-steps/s", cores=cores, host_cpu_count=os.cpu_count(), kind="port",
                sample=f"{cores} processes (the job's CPU share; the host reports {os.cpu_count()} CPUs) x {n_envs} envs x "
                       f"{n_steps} steps (slowest worker {inner:.1f} s, pool wall {wall:.1f} s incl. spawn)")


def _hist(t, cap):
    import torch
    return torch.bincount(t.flatten().to(torch.int64).clamp_(0, cap), minlength=cap + 1).cpu().tolist()


class Workload:
    """One sim + ctrl bank + (optional) observation gather, timed the way the contract asks."""

    def __init__(self, model, n_envs, local_rank, seed, gather_world=0, dist=None):
        import torch
        from mujoco_amd import BatchSim, workload
        from mujoco_amd import dist as bdist
        self.torch, self.dist = torch, dist
        self.model, self.E = model, n_envs
        self.dev = f"cuda:{local_rank}"
        q, v = workload.config2_state(model, n_envs, seed=seed)
        self.sim = BatchSim(model, n_envs=n_envs, device_id=local_rank)
        self.sim.data.qpos.copy_(torch.as_tensor(q))
        self.sim.data.qvel.copy_(torch.as_tensor(v))
        self.n_bank = 8
        self.bank = torch.as_tensor(np.stack([workload.config2_ctrl(model, n_envs, seed=seed, resample_idx=k)
                                              for k in range(self.n_bank)])).to(self.dev)
        self.gather = bdist.ObsGather(self.sim, gather_world) if gather_world else None
        self.period = 0
        self.launch_events = []

    def run(self, nsteps, timed=False):
        torch, sim = self.torch, self.sim
        done = 0
        while done < nsteps:
            n = min(CTRL_PERIOD, nsteps - done)
            sim.data.ctrl.copy_(self.bank[self.period % self.n_bank])
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sim.step(n)
                e1.record()
                self.launch_events.append((e0, e1, n))
            else:
                sim.step(n)
            if self.gather is not None:
                self.gather.all_gather(overlap=True)  # under the next period's steps; fence() drains it
            done += n
            self.period += 1

    def close(self):
        """Destroy the sim NOW rather than when the last Python reference goes: with GPU_MAX_HW_QUEUES = 4 the three
        streams of a contact sim that is still alive share hardware queues with the next one's, whose ranges then do
        not overlap (measured 2.86 instead of 2.38 ms/step when `sim.data` still held its sim strongly)."""
        self.gather = None
        self.launch_events = []
        self.sim.close()

    def fence(self):
        if self.gather is not None:
            self.gather.wait()
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def measure(self, steps, warmup, reps=REPS):
        """-> dict(wall_s median over reps of the K-step region (MAX over ranks each), all walls, the
        event-timed duration of every bs_step call of the timed regions)."""
        torch = self.torch
        self.run(warmup)
        self.fence()
        walls = []
        t_all = time.perf_counter()
        for r in range(reps):
            # long regions stop at 5 repetitions; the decision is rank 0's clock so that every rank runs the same count
            if r >= 5:
                stop = time.perf_counter() - t_all > REPS_BUDGET_S
                if self.dist is not None:
                    flag = torch.tensor([1 if stop else 0], device=self.dev)
                    self.dist.broadcast(flag, src=0)
                    stop = bool(flag.item())
                if stop:
                    break
            t0 = time.perf_counter()
            self.run(steps, timed=True)
            self.fence()
            w = time.perf_counter() - t0
            if self.dist is not None:
                t = torch.tensor([w], dtype=torch.float64, device=self.dev)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                w = float(t.item())
            walls.append(w)
        ev = [(a.elapsed_time(b), n) for a, b, n in self.launch_events]
        self.launch_events = []
        return dict(wall=float(np.median(walls)), walls=walls, launches=ev)

    def state_stats(self):
        d = self.sim.data
        out = {"mean_solver_iter": float(d.solver_iter.double().mean().item()),
               "mean_nefc": float(d.nefc.double().mean().item()),
               "envs_with_warnings": int(((d.warning & 7) != 0).sum().item()),
               "envs_with_contact_overflow": int(((d.warning & 8) != 0).sum().item()),
               "solver_iter_hist": _hist(d.solver_iter, 12)}
        if self.sim.model.npair > 0 and int(d.ncon.numel()) == self.E:
            out["ncon_hist"] = _hist(d.ncon, 32)
        return out

    def collision_model(self, iters):
        """Measured work model of SURVEY.md §8(d): N_np, MPR iterations, ncon per env-step over one control period.
        -> (stats, flop per env-step by the contract's formula, the same with the MEASURED MPR iteration count)."""
        sim = self.sim
        sim.collision_stats(reset=True)
        sim.step(CTRL_PERIOD)
        st = sim.collision_stats()
        n = self.E * CTRL_PERIOD
        c = {"narrow_pairs_per_env_step": st["narrow_pairs"] / n,
             "mpr_iter_per_pair": st["mpr_iterations"] / max(1, st["narrow_pairs"]),
             "ncon_per_env_step": st["contacts"] / n}
        # contract: F = 36k + 12k N_np + 6k ncon n_iter (12 k per pair = 12 portal iterations of 1.01 k);
        # measured: the pair term with this run's own iteration count
        solver = 6e3 * c["ncon_per_env_step"] * max(iters, 1.0)
        flop = FLOP_PER_ENV_STEP + 12e3 * c["narrow_pairs_per_env_step"] + solver
        flop_meas = FLOP_PER_ENV_STEP + FLOP_PER_MPR_ITER * c["mpr_iter_per_pair"] * c["narrow_pairs_per_env_step"] + solver
        return c, flop, flop_meas


def roofline_block(launches, E, flop_env_step, kernel, traffic, traffic_src, note, flop_measured=None, hw=(None, None)):
    """achieved = algorithmic flop of the event-timed launches / their summed duration."""
    ms = sum(t for t, _ in launches)
    steps = sum(n for _, n in launches)
    rate = E * steps / (ms * 1e-3)  # env-steps/s inside the event-timed bs_step calls
    achieved = flop_env_step * rate / 1e12
    per_launch_steps = max(n for _, n in launches)
    full = [t for t, n in launches if n == per_launch_steps]
    out = {"bound": "mfma", "bound_actual": "fp64_valu_latency",
           "bound_detail": "fp64 vector ALU, latency-bound (dependent fp64 / cross-lane chains; no MFMA instruction is used; the "
                           "contract's two labels are hbm|mfma and this is the compute one)",
           "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
           "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
           "launch_ms": float(np.mean(full)), "steps_per_launch": per_launch_steps, "launches_timed": len(launches),
           "env_steps_per_launch": E * per_launch_steps, "flop_per_env_step": flop_env_step,
           "algorithmic_hbm_GBps": BYTES_PER_ENV_STEP * rate / 1e9, "note": note}
    if flop_measured is not None:
        out["flop_per_env_step_measured"] = flop_measured
        out["achieved_measured"] = flop_measured * rate / 1e12
        out["frac_measured"] = out["achieved_measured"] / FP64_PEAK_TFLOPS
        out["measured_note"] = ("the contract formula prices a narrowphase pair at 12 portal iterations; *_measured uses this run's "
                                "mpr_iter_per_pair x 1.01 kflop instead — the honest figure when the two differ")
    if hw[0] is not None:
        out["hw_fp64_flop_per_env_step"] = hw[0]
        out["hw_fp64_tflops"] = hw[0] * rate / 1e12
        out["hw_fp64_frac"] = out["hw_fp64_tflops"] / FP64_PEAK_TFLOPS
        out["hw_fp64_source"] = (f"profiles/{hw[1]}: SQ_INSTS_VALU_(2 x FMA + ADD + MUL + TRANS)_F64 x 64 lanes per env-step of the "
                                 "profiled run (masked lanes count as issued), times this run's env-steps/s")
    return out


CONTACT_LABEL = ("DexterousDynamos.xml with STL-hull mesh self-collision + a ground plane at z=-0.05 (the plane is this build's "
                 "augmentation, not in the reference XML), Euler + Newton, fp64, ctrl resampled every 50 steps")


def contacts_block(model_path, E, local_rank, seed, steps, warmup, world, dist, gather_world, label):
    """configs[2] / configs[4]-share: contacts + ground plane at E envs per GPU."""
    import torch
    from mujoco_amd import Model, workload
    model = workload.with_ground_plane(Model.load(model_path), z=-0.05)
    w = Workload(model, E, local_rank, seed, gather_world=gather_world, dist=dist)
    m = w.measure(steps, warmup)
    stats = w.state_stats()
    cstat, flop, flop_meas = w.collision_model(stats["mean_solver_iter"])
    traffic, src = pmc_traffic("contacts_pmc_per_step" if E == 65536 else "none", ("k_tp_col", "k_c_narrow_list", "k_tpc"))
    path = w.sim.kernel_path()[1]
    blk = {"value": E * world * steps / m["wall"], "unit": "env-steps/s", "ms_per_step": m["wall"] * 1e3 / steps,
           "steps": steps, "reps": len(m["walls"]), "wall_s_all_reps": m["walls"],
           "config": {"workload": label, "n_envs_per_gpu": E, "n_envs_total": E * world, "ctrl_period": CTRL_PERIOD,
                      "device_memory": w.sim.footprint(),
                      "collective": ("RCCL all_gather obs[E,42]+reward fp32 per ctrl period, overlapped with the next steps"
                                     if gather_world else "none"),
                      "kernel_path": path, "collision": cstat, **stats},
           "roofline": roofline_block(m["launches"], E, flop, "k_tp_col + k_c_narrow_list + "
                                      + ("k_tpc_chain" if "k_tpc_chain" in path else "k_tpc") + " (one step = 3 launches per range, "
                                      "3 ranges on 3 streams)", traffic, src,
                                      "flop model F = 36k + 12k N_np + 6k ncon n_iter with the measured N_np / ncon / n_iter of this run; "
                                      "duration = HIP events around every bs_step call of the timed regions; traffic = per step, all three kernels",
                                      flop_measured=flop_meas, hw=pmc_hw_flop("contacts"))}
    w.close()
    del w
    torch.cuda.empty_cache()
    return blk, model


def config1_block(model_path, E, local_rank, seed, steps, warmup, with_cpu):
    """configs[1]: joint-space only (contacts disabled), E lockstep envs on one GPU."""
    import torch
    from mujoco_amd import Model, workload
    model = workload.contact_free(Model.load(model_path))
    w = Workload(model, E, local_rank, seed)
    m = w.measure(steps, warmup)
    stats = w.state_stats()
    path = w.sim.kernel_path()[1]
    kern = "k_tp_chain4" if "k_tp_chain4" in path else "k_tp_chain" if "k_tp_chain" in path else "k_tp"
    traffic, src = pmc_traffic("config1_pmc_per_launch" if E == 4096 else "none", ("k_tp",))
    blk = {"value": E * steps / m["wall"], "unit": "env-steps/s", "ms_per_step": m["wall"] * 1e3 / steps,
           "steps": steps, "reps": len(m["walls"]), "wall_s_all_reps": m["walls"],
           "config": {"workload": "BASELINE.json configs[1]: DexterousDynamos.xml joint-space only (contacts disabled), lockstep "
                                  "envs, random ctrl resampled every 50 steps, Euler + Newton, fp64",
                      "n_envs_per_gpu": E, "n_envs_total": E, "ctrl_period": CTRL_PERIOD, "collective": "none",
                      "device_memory": w.sim.footprint(), "kernel_path": path, **stats},
           "roofline": roofline_block(m["launches"], E, FLOP_PER_ENV_STEP, kern, traffic, src,
                                      "duration = HIP events on the kernels' stream around every bs_step call of the timed regions "
                                      "(one call = one launch of min(50, steps) steps); traffic = bytes per 50-step launch from the "
                                      "committed PMC summary (prologue / epilogue state + outputs + spill scratch, not per-step traffic)",
                                      hw=pmc_hw_flop("config1"))}
    if with_cpu:
        blk["cpu_baseline"] = cpu_baseline(model, n_steps=1000)
        try:
            blk["cpu_baseline_allcores"] = cpu_baseline_allcores(model_path, False)
        except Exception as ex:  # the pool is informational; the single-thread figure is the contract
            blk["cpu_baseline_allcores"] = {"error": repr(ex)}
    w.close()
    del w
    torch.cuda.empty_cache()
    return blk


def main_mixed(args):
    """BASELINE.json configs[3]: the three exported hands (tests/golden/*.npz, compiled from the
    reference's fusion_export_* folders by the exporter restatement, see make_golden.py for the
    declared mesh stand-ins) in one mixed-model batch, environments split evenly and sorted by
    model, once with Newton and once with PGS.  Not the default bench line."""
    import torch
    from mujoco_amd import MixedBatchSim, Model, workload
    from mujoco_amd.model import SOL_NEWTON, SOL_PGS
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    names = ["dexterous_model.npz", "hand_1120.npz", "hand_1206.npz"]
    base = [Model.load(os.path.join(ROOT, "tests", "golden", n)) for n in names]
    if not args.contacts:
        base = [workload.contact_free(m) for m in base]
    per = args.envs // 3
    res = {}
    for label, sol in (("newton", SOL_NEWTON), ("pgs", SOL_PGS)):
        models = []
        for m in base:
            m = m.copy()
            m.opt_solver = sol
            models.append(m)
        sim = MixedBatchSim(models, counts=[per] * 3)
        banks = []
        for k, p in enumerate(sim.parts):
            q, v = workload.config2_state(models[k], per, seed=1234 + k)
            p.data.qpos.copy_(torch.as_tensor(q))
            banks.append(torch.as_tensor(np.stack([workload.config2_ctrl(models[k], per, seed=1234 + k, resample_idx=i)
                                                   for i in range(4)])).to(p.data.ctrl.device))

        def run(n, period):
            done = 0
            while done < n:
                for k, p in enumerate(sim.parts):
                    p.data.ctrl.copy_(banks[k][period % 4])
                sim.step(min(CTRL_PERIOD, n - done))
                done += CTRL_PERIOD
                period += 1
            return period

        period = run(args.warmup, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps, period)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        res[label] = {"value": 3 * per * args.steps / wall, "ms_per_step": wall * 1e3 / args.steps,
                      "mean_solver_iter": [float(p.data.solver_iter.double().mean().item()) for p in sim.parts],
                      "solver_iter_hist": [_hist(p.data.solver_iter, 60) for p in sim.parts],
                      "envs_with_warnings": int(sum(((p.data.warning & 7) != 0).sum().item() for p in sim.parts)),
                      "envs_with_contact_overflow": int(sum(((p.data.warning & 8) != 0).sum().item() for p in sim.parts))}
        sim.close()
    out = {"metric": "env-steps/sec", "value": res["newton"]["value"], "unit": "env-steps/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["newton"]["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[3]: three exported hands (24/23/24 bodies, 21/20/21 dofs) in one "
                                  "mixed-model batch, " + ("mesh self-collision on" if args.contacts else "joint-space only")
                                  + ", random ctrl resampled every 50 steps, Euler, Newton (value) and PGS",
                      "n_envs_per_model": per, "n_envs_total": 3 * per, "models": names},
           "newton": res["newton"], "pgs": res["pgs"], "pgs_over_newton_time": res["newton"]["value"] / res["pgs"]["value"]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=0, help="environments per GPU of a variant run (default headline: 131072 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-blocks", action="store_true", help="N = 1: only the headline workload (profiling runs)")
    ap.add_argument("--contacts", action="store_true",
                    help="variant run: BASELINE.json configs[2] shape (mesh self-collision + ground plane; use with --envs 65536), "
                         "no collective")
    ap.add_argument("--hull-verts", type=int, default=0,
                    help="with --contacts: OPT-IN hull simplification to at most this many vertices per hull "
                         "(changes the contact geometry; reported separately, never the parity configuration)")
    ap.add_argument("--solver", choices=("newton", "cg", "pgs"), default="newton",
                    help="constraint solver (default newton, the reference's default; not the default bench line otherwise)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE.json configs[3]: three hands in one mixed-model batch, Newton and PGS; "
                         "not the default bench line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.mixed:
        args.envs = args.envs or 12288
        return main_mixed(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain shell: start the ranks as a child process BEFORE anything here initialises the GPU (this process
        # never becomes a rank), relay rank 0's JSON line, exit with the child's code
        from mujoco_amd.launch import spawn_ranks
        sys.exit(spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    import torch
    from mujoco_amd import Model, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local_rank)

    model_path = os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")
    seed = 1234 + 1000 * rank
    plain = not (args.contacts or args.hull_verts or args.solver != "newton" or args.envs)
    import torch.distributed as dist
    if plain or world > 1:
        # one rank too goes through RCCL: the N = 1 point of the curve runs the same code as N = 8
        if world == 1:
            from mujoco_amd.launch import free_port
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist = None

    if plain:
        # ---- headline at every N: configs[4]'s shape, 131 072 envs per GPU
        E = 131072
        joined = torch.ones(1, dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(joined)
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}")
        blk, cmodel = contacts_block(model_path, E, local_rank, seed, args.steps, args.warmup, world, dist, world,
                                     "BASELINE.json configs[4] shape, one GPU's share per rank: " + CONTACT_LABEL
                                     + ", 131072 envs per GPU, obs/reward all-gathered over RCCL every 50 steps")
        out = None
        if rank == 0:
            out = {"metric": "env-steps/sec", "value": blk["value"], "unit": "env-steps/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": blk["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": blk["config"], "ranks_joined": int(joined.item()), "devices": names,
                   "roofline": blk["roofline"], "reps": blk["reps"], "wall_s_all_reps": blk["wall_s_all_reps"],
                   "note": "the same workload per GPU at every N (weak scaling): value(N) / (N x value(1)) is the scaling efficiency; "
                           "configs[1] and configs[2] are the blocks config1_4096 / contacts_65536 of the N = 1 line"}
        if world == 1:
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cmodel, seconds_budget=10.0, n_steps=200)
                try:
                    out["cpu_baseline_allcores"] = cpu_baseline_allcores(model_path, True)
                except Exception as ex:  # the pool is informational; the single-thread figure is the contract
                    out["cpu_baseline_allcores"] = {"error": repr(ex)}
            if not args.no_extra_blocks:
                # whole control periods: the first steps after a ctrl resample take twice the Newton iterations of the
                # settled ones, so a region shorter than a period (the driver's --steps 20) would time the transient only
                c1_steps = -(-max(args.steps, 2 * CTRL_PERIOD) // CTRL_PERIOD) * CTRL_PERIOD
                out["config1_4096"] = config1_block(model_path, 4096, local_rank, seed, c1_steps, max(args.warmup, CTRL_PERIOD), not args.no_cpu_baseline)
                b2, _ = contacts_block(model_path, 65536, local_rank, seed, min(args.steps, 200), min(args.warmup, 50), 1, None, 0,
                                       "BASELINE.json configs[2]: " + CONTACT_LABEL + ", 65536 envs, no collective")
                if "cpu_baseline" in out:  # the same model and workload per environment as the headline's
                    b2["cpu_baseline"] = out["cpu_baseline"]
                    b2["cpu_baseline_allcores"] = out["cpu_baseline_allcores"]
                out["contacts_65536"] = b2
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # ---- variant runs (flags): one workload, its roofline and CPU leg; not the default line
    model = Model.load(model_path)
    if not args.contacts:
        model = workload.contact_free(model)
    else:
        model = workload.with_ground_plane(model, z=-0.05)
        if args.hull_verts > 0:
            from mujoco_amd.meshproc import simplify_model_hulls
            model = simplify_model_hulls(model, args.hull_verts)
    if args.solver != "newton":
        from mujoco_amd.model import SOL_CG, SOL_PGS
        model = model.copy()
        model.opt_solver = SOL_CG if args.solver == "cg" else SOL_PGS
    E = args.envs or 4096
    multi = world > 1
    w = Workload(model, E, local_rank, seed, gather_world=world if multi else 0, dist=dist)
    m = w.measure(args.steps, args.warmup)
    stats = w.state_stats()
    cstat, flop_env_step, flop_meas = (w.collision_model(stats["mean_solver_iter"]) if args.contacts
                                       else (None, FLOP_PER_ENV_STEP, None))

    if rank == 0:
        wall = m["wall"]
        path = w.sim.kernel_path()[1]
        chain = "k_tp_chain" in path or "k_tpc_chain" in path
        kern = ("k_tp_chain4" if "chain4" in path else "k_tp_chain" if chain else "k_tp") if args.solver == "newton" else ("k_tp_cg" if args.solver == "cg" else "k_tp_pgs")
        hw = (None, None)
        if args.contacts:
            traffic, src = pmc_traffic("contacts_pmc_per_step" if E == 65536 and not args.hull_verts else "none",
                                       ("k_tp_col", "k_c_narrow_list", "k_tpc"))
            kern = "k_tp_col + k_c_narrow_list + " + {"newton": "k_tpc_chain" if chain else "k_tpc", "cg": "k_tpc_cg", "pgs": "k_tpc_pgs"}[args.solver]
            if args.solver == "newton" and not args.hull_verts:
                hw = pmc_hw_flop("contacts")
        else:
            traffic, src = pmc_traffic("config1_pmc_per_launch" if E == 4096 and args.solver == "newton" else "none", ("k_tp",))
            if args.solver == "newton":
                hw = pmc_hw_flop("config1")
        out = {
            "metric": "env-steps/sec", "value": E * world * args.steps / wall, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "reps": len(m["walls"]), "wall_s_all_reps": m["walls"],
            "config": {"workload": "VARIANT RUN (not the default line) — BASELINE.json configs[1]: DexterousDynamos.xml joint-space only "
                                   "(contacts disabled), lockstep envs, random ctrl resampled every 50 steps, "
                                   "Euler + Newton, fp64",
                       "n_envs_per_gpu": E, "n_envs_total": E * world, "ctrl_period": CTRL_PERIOD, "kernel_path": path,
                       "device_memory": w.sim.footprint(),
                       "collective": "RCCL all_gather obs[E,42]+reward fp32 per ctrl period" if multi else "none", **stats},
            "roofline": roofline_block(m["launches"], E, flop_env_step, kern, traffic, src,
                                       "duration = HIP events on the kernels' stream around every bs_step call of the timed regions "
                                       "(contact-free: one call = one launch of min(50, steps) steps); traffic from the committed PMC summary",
                                       flop_measured=flop_meas, hw=hw),
        }
        if cstat:
            out["config"]["workload"] = "VARIANT RUN (not the default line) — BASELINE.json configs[2]: " + CONTACT_LABEL
            out["config"]["collision"] = cstat
            if args.hull_verts > 0:
                out["config"]["workload"] += f"; OPT-IN simplified hulls (<= {args.hull_verts} vertices each): NOT the parity geometry"
                out["config"]["hull_verts"] = args.hull_verts
        if args.solver != "newton":
            out["config"]["workload"] += f"; solver = {args.solver.upper()} (non-default)"
            out["config"]["solver"] = args.solver
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, n_steps=200 if args.contacts else 1000)
            try:
                out["cpu_baseline_allcores"] = cpu_baseline_allcores(model_path, args.contacts)
            except Exception as ex:  # the pool is informational; the single-thread figure is the contract
                out["cpu_baseline_allcores"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    w.close()
    del w
    torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
