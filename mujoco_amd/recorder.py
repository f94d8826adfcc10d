"""Headless trajectory recorder and `run_simulation` driver.

The reference's only run loop builds the sim and hands it to a GL viewer
(`src/Mujoco_XML.py:327-345`: `load_model_from_xml` -> `MjSim` -> `MjViewer.render()` forever; the
other path shells out to `python -m mujoco.viewer`, `src/mujoco_sim.sh:16`).  A GPU box has no
display, so the replacement of the viewer is a state recorder: frames of chosen `sim.data` fields
are copied into a preallocated device buffer (no host synchronisation while the simulation runs)
and written as `.npz` or `.csv` at the end.

`.npz`: one array per field, `[n_frames, n_selected_envs, ...]`, plus `frame_step` (physics step
index of each frame) and `env` (selected environment ids).
`.csv`: one row per (frame, env): `step, env, <field>_<i>...` — scalars and vectors only.
"""
import numpy as np
import torch

_SCALARS = ("time", "solver_iter", "warning", "nefc", "ncon")


class TrajectoryRecorder:
    def __init__(self, sim, fields=("time", "qpos", "qvel", "ctrl"), envs=None, capacity=1024):
        if capacity < 1:
            raise ValueError("capacity must be >= 1")
        self.sim, self.fields, self.capacity = sim, tuple(fields), int(capacity)
        n = sim.n_envs
        if envs is None:
            envs = np.arange(n)
        self.envs = np.atleast_1d(np.asarray(envs, dtype=np.int64))
        if self.envs.size == 0 or self.envs.min() < 0 or self.envs.max() >= n:
            raise ValueError("envs out of range")
        self._buf, self._steps, self.n_frames = {}, [], 0
        self._index = None
        for f in self.fields:
            v = self._field(f)
            if self._index is None:
                self._index = torch.as_tensor(self.envs, device=v.device)
            self._buf[f] = torch.empty((self.capacity, self.envs.size) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)

    def _field(self, name):
        v = getattr(self.sim.data, name)
        if self.sim.n_envs == 1 and name not in _SCALARS:
            v = v.unsqueeze(0)  # BatchSim drops the env axis of non-scalar fields for a single environment
        return v

    def record(self, step):
        """Copy the current value of every field (asynchronous, on the sim's stream)."""
        if self.n_frames >= self.capacity:
            raise RuntimeError(f"recorder is full ({self.capacity} frames)")
        for f in self.fields:
            torch.index_select(self._field(f), 0, self._index, out=self._buf[f][self.n_frames])
        self._steps.append(int(step))
        self.n_frames += 1

    def arrays(self):
        """Host copies of the recorded frames (synchronises)."""
        out = {f: self._buf[f][: self.n_frames].cpu().numpy() for f in self.fields}
        out["frame_step"] = np.asarray(self._steps, dtype=np.int64)
        out["env"] = self.envs.copy()
        return out

    def save(self, path):
        a = self.arrays()
        if str(path).endswith(".csv"):
            cols, mats = ["step", "env"], []
            for f in self.fields:
                v = a[f].reshape(self.n_frames, self.envs.size, -1)
                if a[f].ndim > 3:
                    raise ValueError(f"csv output takes scalar and vector fields only, not {f}")
                cols += [f] if v.shape[2] == 1 and a[f].ndim == 2 else [f"{f}_{i}" for i in range(v.shape[2])]
                mats.append(v.astype(np.float64))
            step = np.broadcast_to(a["frame_step"][:, None, None], (self.n_frames, self.envs.size, 1)).astype(np.float64)
            env = np.broadcast_to(self.envs[None, :, None], (self.n_frames, self.envs.size, 1)).astype(np.float64)
            table = np.concatenate([step, env] + mats, axis=2).reshape(self.n_frames * self.envs.size, -1)
            fmt = ["%d", "%d"] + ["%.17g"] * (table.shape[1] - 2)
            np.savetxt(path, table, delimiter=",", header=",".join(cols), comments="", fmt=fmt)
        else:
            np.savez_compressed(path, **a)
        return path


def run_simulation(model, n_steps, n_envs=1, ctrl_fn=None, record_every=1, fields=("time", "qpos", "qvel", "ctrl"),
                   envs=None, out=None, device_id=0):
    """Headless counterpart of `Mujoco_XML.run_simulation` (`src/Mujoco_XML.py:327-345`).

    `model`: MJCF path, MJCF string or a compiled model.  `ctrl_fn(step, sim)` may write
    `sim.data.ctrl` before each step (the reference leaves inputs as a TODO, `:327-330`).
    Frame 0 is the initial state; afterwards one frame every `record_every` steps.  Returns the
    recorder (and writes `out` if given)."""
    from .mjcf import load_model_from_path, load_model_from_xml
    from .sim import BatchSim
    if isinstance(model, str):
        model = load_model_from_xml(model) if model.lstrip().startswith("<") else load_model_from_path(model)
    if record_every < 1:
        raise ValueError("record_every must be >= 1")
    sim = BatchSim(model, n_envs=n_envs, device_id=device_id)
    rec = TrajectoryRecorder(sim, fields=fields, envs=envs, capacity=n_steps // record_every + 1)
    sim.forward()
    rec.record(0)
    for i in range(1, n_steps + 1):
        if ctrl_fn is not None:
            ctrl_fn(i - 1, sim)
        sim.step()
        if i % record_every == 0:
            rec.record(i)
    if out is not None:
        rec.save(out)
    sim.sync()  # the frame copies are asynchronous; the buffers below outlive the sim
    rec.sim = None
    sim.close()
    return rec
