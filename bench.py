#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched physics step (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[1] — DexterousDynamos.xml joint-space only (contacts
disabled), 4096 lockstep envs on one MI355X, random ctrl resampled every 50 steps, fp64.
A "step" is one mj_step of the whole batch.  State and the ctrl bank are resident in HBM before
the timed region.  N>1 (launched by torch.distributed.run): each rank steps its own 4096 envs
(weak scaling, no traffic inside a step) and, once per control period (50 steps), all-gathers
obs (qpos|qvel, fp32) + reward over RCCL for a centralised learner.

Prints ONE JSON line on rank 0.
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


def pmc_traffic(kernel, section):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (profiles/r01_pmc_summary.json, separate FETCH_SIZE / WRITE_SIZE passes of this same command):
    (2 x FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md.  bench.py
    cannot collect counters itself; None if the summary is missing."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            k = json.load(f)[section][kernel]
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(model, seconds_budget=12.0):
    """Oracle (CPU restatement, kind 'port') on this host: single thread, then all cores."""
    from mujoco_amd import workload
    from oracle import OracleSim
    n_envs, n_steps = 2048, 1000  # bounded by seconds_budget: about 12 s of single-thread work
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
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16)
    n_envs, n_steps = (40 if contacts else 400), 1000  # a few seconds per worker
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_pool_worker, [(model_path, 1234 + i, n_envs, n_steps, contacts) for i in range(cores)])
    wall = time.perf_counter() - t0
    work = sum(r[0] for r in res)
    inner = max(r[1] for r in res)
    return dict(value=work / inner, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} processes x {n_envs} envs x {n_steps} steps (slowest worker {inner:.1f} s, "
                       f"pool wall {wall:.1f} s incl. spawn)")


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
                      "envs_with_warnings": int(sum((p.data.warning != 0).sum().item() for p in sim.parts))}
        sim.close()
    out = {"metric": "env-steps/sec", "value": res["newton"]["value"], "unit": "env-steps/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["newton"]["ms_per_step"],
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "BASELINE.json configs[3]: three exported hands (24/23/24 bodies, 21/20/21 dofs) in one "
                                  "mixed-model batch, " + ("mesh self-collision on" if args.contacts else "joint-space only")
                                  + ", random ctrl resampled every 50 steps, Euler, Newton (value) and PGS",
                      "n_envs_per_model": per, "n_envs_total": 3 * per, "models": names},
           "newton": res["newton"], "pgs": res["pgs"]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contacts", action="store_true",
                    help="BASELINE.json configs[2] shape instead of configs[1]: mesh self-collision enabled "
                         "(use with --envs 65536); not the default bench line")
    ap.add_argument("--hull-verts", type=int, default=0,
                    help="with --contacts: OPT-IN hull simplification to at most this many vertices per hull "
                         "(changes the contact geometry; reported separately, never the parity configuration)")
    ap.add_argument("--solver", choices=("newton", "cg", "pgs"), default="newton",
                    help="constraint solver (default newton, the reference's default; not the default bench line otherwise)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE.json configs[3]: three hands in one mixed-model batch, Newton and PGS; "
                         "not the default bench line")
    args = ap.parse_args()
    if args.mixed:
        return main_mixed(args)

    import torch
    from mujoco_amd import BatchSim, Model, workload

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = bool(os.environ.get("BENCH_FORCE_DIST"))  # rehearsal of the N>1 code path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))

    model_path = os.path.join(ROOT, "tests", "golden", "dexterous_model.npz")
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
    E = args.envs
    seed = 1234 + 1000 * rank
    q, v = workload.config2_state(model, E, seed=seed)
    sim = BatchSim(model, n_envs=E, device_id=local_rank)
    dev = f"cuda:{local_rank}"
    sim.data.qpos.copy_(torch.as_tensor(q))
    sim.data.qvel.copy_(torch.as_tensor(v))
    n_bank = 8
    bank = torch.as_tensor(np.stack([workload.config2_ctrl(model, E, seed=seed, resample_idx=k)
                                     for k in range(n_bank)])).to(dev)

    from mujoco_amd import dist as bdist
    gather = bdist.ObsGather(sim, world) if (world > 1 or force_dist) else None

    launch_events = []  # (start, end, steps) of every bs_step launch of the timed region, on the kernels' stream

    def run(nsteps, start_period, timed=False):
        done, period = 0, start_period
        while done < nsteps:
            n = min(CTRL_PERIOD, nsteps - done)
            sim.data.ctrl.copy_(bank[period % n_bank])
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sim.step(n)
                e1.record()
                launch_events.append((e0, e1, n))
            else:
                sim.step(n)
            if gather is not None:
                gather.all_gather(overlap=True)  # under the next period's steps; fence() drains it
            done += n
            period += 1
        return period

    def fence():
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    period = run(args.warmup, 0)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps, period, timed=True)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    gpu_ms = ev0.elapsed_time(ev1)  # same stream as the kernels (BatchSim uses torch's current stream)
    warn = int((sim.data.warning != 0).sum().item())
    iters = float(sim.data.solver_iter.double().mean().item())
    nefc = float(sim.data.nefc.double().mean().item())

    # kernel timing: events around every bs_step launch of the timed region (the sim runs on torch's
    # current stream, so the events sit on the kernels' stream); average per launch of CTRL_PERIOD steps
    full = [(a.elapsed_time(b), n) for a, b, n in launch_events if n == CTRL_PERIOD] or \
           [(a.elapsed_time(b) * CTRL_PERIOD / n, n) for a, b, n in launch_events]
    k_ms = float(np.mean([t for t, _ in full]))
    stage = sim.stage_times()
    cstat = None
    if args.contacts:
        sim.collision_stats(reset=True)
        sim.step(CTRL_PERIOD)
        st = sim.collision_stats()
        cstat = {"narrow_pairs_per_env_step": st["narrow_pairs"] / (E * CTRL_PERIOD),
                 "mpr_iter_per_pair": st["mpr_iterations"] / max(1, st["narrow_pairs"]),
                 "ncon_per_env_step": st["contacts"] / (E * CTRL_PERIOD)}

    if rank == 0:
        value = E * world * args.steps / wall
        flop_env_step = FLOP_PER_ENV_STEP
        if cstat:  # F = 36k + 12k N_np + 6k ncon n_iter
            flop_env_step += 12e3 * cstat["narrow_pairs_per_env_step"] + 6e3 * cstat["ncon_per_env_step"] * max(iters, 1.0)
        flops_launch = flop_env_step * E * CTRL_PERIOD
        achieved = flops_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: DexterousDynamos.xml joint-space only "
                                   "(contacts disabled), lockstep envs, random ctrl resampled every 50 steps, "
                                   "Euler + Newton, fp64",
                       "n_envs_per_gpu": E, "n_envs_total": E * world, "ctrl_period": CTRL_PERIOD,
                       "collective": "RCCL all_gather obs[E,42]+reward fp32 per ctrl period" if world > 1 else "none",
                       "mean_solver_iter": iters, "mean_nefc": nefc, "envs_with_warnings": warn},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": pmc_traffic("k_tp", "config1_pmc_per_launch") if not cstat else None,
                         "kernel": "k_tp", "launch_ms": k_ms, "env_steps_per_launch": E * CTRL_PERIOD,
                         "flop_per_env_step": flop_env_step,
                         "note": "fp64 compute-bound path ('mfma' stands for the fp64 ALU roof: the contract allows "
                                 "only hbm|mfma; no MFMA instruction is used): peak is the MI355X fp64 vector=matrix "
                                 "rate; traffic = bytes/launch from the committed PMC summary (mostly register-spill "
                                 "scratch, not state); algorithmic HBM rate would be "
                                 f"{BYTES_PER_ENV_STEP * E * CTRL_PERIOD / (k_ms * 1e-3) / 1e9:.2f} GB/s of 8000"},
            "gpu_event_ms_per_step": gpu_ms / args.steps,
            "stage_ms_one_step": stage,
        }
        if cstat:
            out["config"]["workload"] = ("BASELINE.json configs[2]: DexterousDynamos.xml with STL-hull mesh "
                                         "self-collision + a ground plane at z=-0.05 (the plane is this build's "
                                         "augmentation, not in the reference XML), Euler + Newton, fp64")
            out["config"]["collision"] = cstat
            if args.hull_verts > 0:
                out["config"]["workload"] += f"; OPT-IN simplified hulls (<= {args.hull_verts} vertices each): NOT the parity geometry"
                out["config"]["hull_verts"] = args.hull_verts
            out["roofline"]["kernel"] = "k_tp_col+k_c_narrow_list+k_tpc (one step = 3 launches)"
        if args.solver != "newton":
            out["config"]["workload"] += f"; solver = {args.solver.upper()} (non-default)"
            out["config"]["solver"] = args.solver
            out["roofline"]["traffic"] = None
            if args.solver == "cg":
                out["roofline"]["kernel"] = out["roofline"]["kernel"].replace("k_tpc", "k_tpc_cg") if cstat else "k_tp_cg"
            else:
                out["roofline"]["kernel"] = "serial kernels (k_c_dyn): PGS with contacts has no tree-parallel path" if cstat else "k_tp_pgs"
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model)
            try:
                out["cpu_baseline_allcores"] = cpu_baseline_allcores(model_path, args.contacts)
            except Exception as ex:  # the pool is informational; the single-thread figure is the contract
                out["cpu_baseline_allcores"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
