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


def with_ground_plane(model, z=-0.05):
    """BASELINE.json configs[2] / SURVEY.md §8(d) config 3: the exported XML plus a ground plane
    `<geom type="plane" size="0 0 1" pos="0 0 z"/>` on the world body.  The plane is NOT in the
    reference XML; it is this build's augmentation and every report says so."""
    from .model import GEOM_PLANE
    m = model.copy()

    def app(a, v):
        return np.concatenate([a, np.asarray(v, dtype=a.dtype).reshape((1,) + a.shape[1:])])

    m.geom_type = app(m.geom_type, GEOM_PLANE); m.geom_bodyid = app(m.geom_bodyid, 0)
    m.geom_dataid = app(m.geom_dataid, -1); m.geom_contype = app(m.geom_contype, 1)
    m.geom_conaffinity = app(m.geom_conaffinity, 1); m.geom_condim = app(m.geom_condim, 3)
    m.geom_priority = app(m.geom_priority, 0); m.geom_pos = app(m.geom_pos, [0, 0, z])
    m.geom_quat = app(m.geom_quat, [1, 0, 0, 0]); m.geom_size = app(m.geom_size, [0, 0, 1])
    m.geom_friction = app(m.geom_friction, [1, 0.005, 0.0001]); m.geom_margin = app(m.geom_margin, 0.0)
    m.geom_gap = app(m.geom_gap, 0.0); m.geom_solref = app(m.geom_solref, [0.02, 1])
    m.geom_solimp = app(m.geom_solimp, [0.9, 0.95, 0.001, 0.5, 2]); m.geom_solmix = app(m.geom_solmix, 1.0)
    m.geom_rbound = app(m.geom_rbound, 0.0)
    m.geom_names = list(m.geom_names) + ["ground_plane"]
    m._build_pairs()
    return m
