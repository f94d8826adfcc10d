"""Synthetic workloads of BASELINE.json / SURVEY.md §8d, shared by bench.py and the parity
tests so the HIP path and the oracle see identical numbers (host-side Philox streams)."""
import numpy as np


def _rng(seed, stream):
    return np.random.Generator(np.random.Philox(key=int(seed), counter=[0, 0, 0, int(stream)]))


def config2_state(model, n_envs, seed=1234):
    """qpos ~ U(lo + 0.1 w, hi - 0.1 w) per joint, equality-driven joints set consistent with
    their polynomial (reference src/Fusion_to_Mujoco.py:107-108), qvel = 0.  Returns [E, nq]."""
    r = _rng(seed, 0)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    w = hi - lo
    q = lo + 0.1 * w + r.random((n_envs, model.nq)) * 0.8 * w
    for e in range(model.neq):
        if not model.eq_active[e] or model.eq_obj2id[e] < 0:
            continue
        j1, j2 = int(model.eq_obj1id[e]), int(model.eq_obj2id[e])
        c = model.eq_data[e]
        x = q[:, j2] - model.qpos0[j2]
        q[:, j1] = model.qpos0[j1] + c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * c[4])))
    return q, np.zeros((n_envs, model.nv))


def config2_ctrl(model, n_envs, seed=1234, resample_idx=0):
    """ctrl_k ~ U(ctrlrange_k), resampled every 50 steps (stream = 1 + resample index)."""
    r = _rng(seed, 1 + resample_idx)
    lo, hi = model.actuator_ctrlrange[:, 0], model.actuator_ctrlrange[:, 1]
    return lo + r.random((n_envs, model.nu)) * (hi - lo)


def contact_free(model):
    """Copy of `model` with the collision stage disabled (BASELINE.json configs[1])."""
    from .model import DSBL_CONTACT
    m = model.copy()
    m.opt_disableflags = int(m.opt_disableflags) | DSBL_CONTACT
    return m
