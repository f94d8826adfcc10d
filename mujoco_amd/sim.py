"""BatchSim — the mujoco-py `MjSim` surface over libbatchsim.so.

Mirrors what the reference builds at src/Mujoco_XML.py:341-342 (`load_model_from_xml` ->
`MjSim(model)`) and what BASELINE.json's north_star names (`sim.step()`,
`sim.data.qpos/qvel/ctrl`), for `n_envs` lockstep environments on one MI355X.

`sim.data.<field>` are zero-copy torch views of the library's device arrays, shaped
`[n_envs, ...]` (for n_envs == 1 the leading axis is dropped, giving mujoco-py's shapes).
PyTorch is used only as the device-array / stream plumbing; all physics runs in the HIP
kernels behind the C-ABI (include/batchsim.h).  There is no CPU fallback.
"""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _abi
from ._abi import BatchSimError  # noqa: F401

MjSimState = namedtuple("MjSimState", "time qpos qvel act udd_state")
MjContact = namedtuple("MjContact", "geom1 geom2 dist pos frame")

# field -> trailing shape (as a function of the model) for reshaping [E, rows]
_SHAPES = {
    "xpos": lambda m: (m.nbody, 3), "xquat": lambda m: (m.nbody, 4), "xipos": lambda m: (m.nbody, 3),
    "xmat": lambda m: (m.nbody, 9), "xanchor": lambda m: (m.njnt, 3), "xaxis": lambda m: (m.njnt, 3),
    "geom_xpos": lambda m: (m.ngeom, 3), "geom_xmat": lambda m: (m.ngeom, 9),
    "cinert": lambda m: (m.nbody, 10), "crb": lambda m: (m.nbody, 10), "cdof": lambda m: (m.nv, 6),
    "cvel": lambda m: (m.nbody, 6), "rootcom": lambda m: (m.nbody, 3),
}
_SCALARS = ("time", "solver_iter", "warning", "nefc", "ncon")


class _DevArray:
    """Bare __cuda_array_interface__ carrier for a library-owned device buffer."""

    def __init__(self, ptr, shape, strides, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), strides=tuple(strides), typestr=typestr,
                                             data=(int(ptr), False), version=2)


def _option(value, names, what):
    """mjtSolver / mjtIntegrator number from MuJoCo's <option> name (any case) or the number itself."""
    if isinstance(value, str):
        if value.lower() not in names:
            raise ValueError(f"{what} must be one of {sorted(names)} (got {value!r})")
        return names[value.lower()]
    if int(value) not in names.values():
        raise ValueError(f"{what} must be one of {sorted(names.values())} (got {value!r})")
    return int(value)


class NativeModel:
    """Owner of a `bs_model*` handle."""

    def __init__(self, model):
        self.model = model
        desc, keep = _abi.make_desc(model)
        h = C.c_void_p()
        _abi.check(_abi.lib().bs_model_create(C.byref(desc), C.byref(h)))
        del keep
        self.handle = h

    def __del__(self):
        try:
            if self.handle:
                _abi.lib().bs_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class BatchData:
    """`sim.data` — device views named after mjData members.  The views are borrowed from the BatchSim that made
    them (the library owns the memory): once that sim is closed or garbage-collected every access raises
    BatchSimError, so keep the sim alive for as long as its data is used (`d = BatchSim(m).data` alone is not enough)."""

    def __init__(self, sim):
        # weak: sim.data <-> sim must not be a reference cycle, or `del sim` would leave the native sim (its
        # arenas and streams) alive until the cyclic collector runs
        import weakref
        object.__setattr__(self, "_ref", weakref.ref(sim))
        object.__setattr__(self, "_cache", {})

    @property
    def _sim(self):
        sim = self._ref()
        if sim is None or getattr(sim, "_h", None) is None:
            raise BatchSimError("the BatchSim that owns this data was destroyed (sim.data is a borrowed view: keep the sim alive)")
        return sim

    def _view(self, name):
        sim = self._sim
        fid = _abi.FIELD_ID[name]
        derived = name in ("xmat", "geom_xpos", "geom_xmat")
        if name in self._cache and not derived:
            return self._cache[name]
        import torch
        ptr = C.c_void_p()
        shape = (C.c_int64 * 2)()
        stride = (C.c_int64 * 2)()
        is_int = C.c_int()
        _abi.check(_abi.lib().bs_get_ptr(sim._h, fid, C.byref(ptr), shape, stride, C.byref(is_int)))
        if name in self._cache:
            return self._cache[name]  # derived: the call above refreshed the buffer in place
        rows, n = int(shape[0]), int(shape[1])
        esz = 4 if is_int.value else 8
        if rows == 0 or ptr.value is None:
            t = torch.zeros((n, 1 if name in _SCALARS else 0), dtype=torch.int32 if is_int.value else torch.float64,
                            device=f"cuda:{sim.device_id}")
        else:
            arr = _DevArray(ptr.value, (rows, n), (int(stride[0]) * esz, int(stride[1]) * esz),
                            "<i4" if is_int.value else "<f8")
            t = torch.as_tensor(arr, device=f"cuda:{sim.device_id}").T  # [n_envs, rows]
        m = sim.model
        if name in _SHAPES:
            t = t.unflatten(1, _SHAPES[name](m))
        elif name in _SCALARS:
            t = t[:, 0]
        if sim.n_envs == 1 and name not in _SCALARS:
            t = t[0]
        self._cache[name] = t
        return t

    def __getattr__(self, name):
        if name in _abi.FIELD_ID:
            return self._view(name)
        if name == "subtree_com":
            return self._view("rootcom")
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in _abi.FIELD_ID:
            import torch
            self._view(name).copy_(torch.as_tensor(value))
        else:
            raise AttributeError(f"cannot set {name}")

    # -- mujoco-py's by-name accessors (PyMjData.get_joint_qpos & co.); the leading axis is the
    #    environment (absent for n_envs == 1, as everywhere in sim.data)
    def _col(self, field, idx):
        v = self._view(field)
        return v[idx] if self._sim.n_envs == 1 else v[:, idx]

    def get_joint_qpos(self, name):
        return self._col("qpos", self._sim.model.get_joint_qpos_addr(name))

    def get_joint_qvel(self, name):
        return self._col("qvel", self._sim.model.get_joint_qvel_addr(name))

    def set_joint_qpos(self, name, value):
        import torch
        self._col("qpos", self._sim.model.get_joint_qpos_addr(name)).copy_(torch.as_tensor(value))

    def set_joint_qvel(self, name, value):
        import torch
        self._col("qvel", self._sim.model.get_joint_qvel_addr(name)).copy_(torch.as_tensor(value))

    def get_body_xpos(self, name):
        return self._col("xpos", self._sim.model.body_name2id(name))

    def get_body_xquat(self, name):
        return self._col("xquat", self._sim.model.body_name2id(name))

    def get_body_xmat(self, name):
        return self._col("xmat", self._sim.model.body_name2id(name)).unflatten(-1, (3, 3))

    def get_geom_xpos(self, name):
        return self._col("geom_xpos", self._sim.model.geom_name2id(name))

    def contacts(self, env=0, cap=64):
        """`sim.data.contact[i].geom1/geom2/dist` for one environment: list of (geom1, geom2, dist)."""
        g1 = (C.c_int32 * cap)()
        g2 = (C.c_int32 * cap)()
        dist = (C.c_double * cap)()
        n = C.c_int()
        _abi.check(_abi.lib().bs_get_contacts(self._sim._h, int(env), g1, g2, dist, cap, C.byref(n)))
        return [(g1[i], g2[i], dist[i]) for i in range(min(n.value, cap))]

    @property
    def contact(self):
        """mujoco-py's `sim.data.contact[i].geom1 / geom2 / dist / pos / frame` (single-environment sims;
        for a batch use `contact_list(env)`): the first `ncon` entries of mjData.contact."""
        if self._sim.n_envs != 1:
            raise AttributeError("sim.data.contact is per environment: use sim.data.contact_list(env) on a batched sim")
        return self.contact_list(0)

    def contact_list(self, env=0, cap=64):
        """List of `MjContact(geom1, geom2, dist, pos[3], frame[9])` of one environment."""
        g1 = (C.c_int32 * cap)()
        g2 = (C.c_int32 * cap)()
        dist = (C.c_double * cap)()
        pos = (C.c_double * (3 * cap))()
        frame = (C.c_double * (9 * cap))()
        n = C.c_int()
        _abi.check(_abi.lib().bs_get_contacts(self._sim._h, int(env), g1, g2, dist, cap, C.byref(n)))
        _abi.check(_abi.lib().bs_get_contact_geometry(self._sim._h, int(env), pos, frame, cap, C.byref(n)))
        k = min(n.value, cap)
        # device layout is [3 * maxcon][E]: row 3 i + q of the column is component q of contact i
        P = np.array(pos[:3 * k]).reshape(k, 3)
        F = np.array(frame[:9 * k]).reshape(k, 9)
        return [MjContact(int(g1[i]), int(g2[i]), float(dist[i]), P[i], F[i]) for i in range(k)]

    def to_host(self, name):
        """Dense NumPy copy `[rows, n_envs]` of a field via bs_copy_field (synchronises)."""
        sim = self._sim
        fid = _abi.FIELD_ID[name]
        ptr = C.c_void_p()
        shape = (C.c_int64 * 2)()
        stride = (C.c_int64 * 2)()
        is_int = C.c_int()
        _abi.check(_abi.lib().bs_get_ptr(sim._h, fid, C.byref(ptr), shape, stride, C.byref(is_int)))
        out = np.zeros((int(shape[0]), int(shape[1])), dtype=np.int32 if is_int.value else np.float64)
        _abi.check(_abi.lib().bs_copy_field(sim._h, fid, out.ctypes.data_as(C.c_void_p), 0))
        return out

    def from_host(self, name, arr):
        sim = self._sim
        fid = _abi.FIELD_ID[name]
        a = np.ascontiguousarray(arr)
        _abi.check(_abi.lib().bs_copy_field(sim._h, fid, a.ctypes.data_as(C.c_void_p), 1))


class BatchSim:
    """`MjSim(model, nsubsteps=1)` for `n_envs` environments stepped in lockstep on one GPU."""

    def __init__(self, model, n_envs=1, device_id=0, nsubsteps=1, use_torch_stream=True, max_contacts=0,
                 solver=None, integrator=None, device_ids=None):
        """`max_contacts`: cap of the contact list per environment (0 = the library's default, 32;
        further contacts of an environment are dropped in geom-pair order and set warning bit 3, value 8).
        `solver` ("newton" | "cg" | "pgs") and `integrator` ("euler" | "rk4"), MuJoCo's `<option>` names in any
        case or the mjtSolver / mjtIntegrator numbers: override `model.opt_solver` / `model.opt_integrator` for this
        sim (what a user of the reference would choose through `Mujoco_XML.add_option`, src/Mujoco_XML.py:137-145);
        the model passed in is left as it is.  `device_ids`: a one-element list is `device_id`; several devices in
        one process are `mujoco_amd.dist.ShardedBatchSim` (one sim per device), several processes `bench.py --gpus N`."""
        if n_envs < 1:
            raise ValueError("n_envs must be >= 1")
        if max_contacts < 0:
            raise ValueError("max_contacts must be >= 0")
        if nsubsteps < 1:
            raise ValueError("nsubsteps must be >= 1")
        if device_ids is not None:
            ids = [int(x) for x in device_ids]
            if len(ids) != 1:
                raise ValueError("BatchSim runs on one device: for several use mujoco_amd.dist.ShardedBatchSim(model, n_envs, "
                                 "device_ids=[...]) (one process) or one process per GPU (bench.py --gpus N)")
            device_id = ids[0]
        L = _abi.lib()
        if solver is not None or integrator is not None:
            base = model.model if isinstance(model, NativeModel) else model
            model = base.copy()
            if solver is not None:
                model.opt_solver = _option(solver, {"pgs": 0, "cg": 1, "newton": 2}, "solver")
            if integrator is not None:
                model.opt_integrator = _option(integrator, {"euler": 0, "rk4": 1}, "integrator")
        self.model = model
        self.n_envs = int(n_envs)
        self.device_id = int(device_id)
        self.nsubsteps = int(nsubsteps)
        self._native = model if isinstance(model, NativeModel) else NativeModel(model)
        if isinstance(model, NativeModel):
            self.model = model.model
        h = C.c_void_p()
        models = (C.c_void_p * 1)(self._native.handle)
        opt = _abi.Options(max_contacts_per_env=int(max_contacts))
        _abi.check(L.bs_sim_create(models, 1, None, self.n_envs, self.device_id, C.byref(opt), C.byref(h)))
        self._h = h
        self.data = BatchData(self)
        fast, path = self.kernel_path()
        if not fast:
            import warnings
            warnings.warn("mujoco_amd: this model runs on the generic one-environment-per-lane kernels, about 40x "
                          "slower than the tree-parallel ones: " + path, RuntimeWarning, stacklevel=2)
        if use_torch_stream:
            import torch
            with torch.cuda.device(self.device_id):
                st = torch.cuda.current_stream().cuda_stream
            _abi.check(L.bs_set_stream(self._h, C.c_void_p(st), 0))

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None):
            _abi.lib().bs_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- stream ordering between the caller's torch stream and the sim's stream.  With the default
    # use_torch_stream=True they are the same stream and both calls are no-ops; with the sim's own stream (or after
    # the caller switched torch streams) device tensors handed to the library were produced on ANOTHER stream than
    # the one its kernels run on, so the sim's stream waits for the producer first, and consumers wait for the sim.
    def _streams(self):
        import torch
        cur = torch.cuda.current_stream(self.device_id)
        sh = self.stream_handle()
        if sh == cur.cuda_stream:
            return None, None
        return cur, torch.cuda.ExternalStream(sh, device=f"cuda:{self.device_id}")

    def wait_for_torch(self):
        """The sim's stream waits for what torch's current stream has enqueued so far (inputs: masks, actions, state)."""
        cur, mine = self._streams()
        if cur is not None:
            mine.wait_stream(cur)

    def torch_waits(self):
        """torch's current stream waits for what the sim's stream has enqueued so far (outputs: obs, reward, views)."""
        cur, mine = self._streams()
        if cur is not None:
            cur.wait_stream(mine)

    # -- MjSim surface
    def step(self, n=None):
        _abi.check(_abi.lib().bs_step(self._h, self.nsubsteps if n is None else int(n)))

    def forward(self):
        _abi.check(_abi.lib().bs_forward(self._h))

    def reset(self, env_mask=None):
        """`MjSim.reset()`; `env_mask`: optional per-environment selection, a host array (staged by the
        library, no allocation per call) or a device uint8 / bool tensor (one kernel, no host round trip)."""
        if env_mask is None:
            _abi.check(_abi.lib().bs_reset(self._h, None))
        elif hasattr(env_mask, "is_cuda") and env_mask.is_cuda:
            import torch
            mk = env_mask.to(torch.uint8).contiguous()
            if tuple(mk.shape) != (self.n_envs,):
                raise ValueError("env_mask must have shape (n_envs,)")
            self.wait_for_torch()  # the mask was written on torch's current stream
            _abi.check(_abi.lib().bs_reset_device_mask(self._h, C.c_void_p(mk.data_ptr())))
            self._keep_mask = mk  # alive until the kernel has read it (stream order: freed memory is reused in order)
            self.torch_waits()
        else:
            mk = np.ascontiguousarray(np.asarray(env_mask).astype(np.uint8))
            if mk.shape != (self.n_envs,):
                raise ValueError("env_mask must have shape (n_envs,)")
            _abi.check(_abi.lib().bs_reset(self._h, mk.ctypes.data_as(C.POINTER(C.c_uint8))))

    def reset_noalloc(self, env_mask):
        """Masked reset through bs_reset_noalloc (the library's allocation-free entry point; at most
        BS_NOALLOC_MAX_ENVS environments).  `env_mask`: host uint8 array or a pinned CPU tensor."""
        if hasattr(env_mask, "data_ptr"):
            if env_mask.is_cuda or env_mask.numel() != self.n_envs or env_mask.element_size() != 1 or not env_mask.is_contiguous():
                raise ValueError("env_mask must be a contiguous 1-byte CPU tensor of n_envs elements")
            ptr = C.c_void_p(env_mask.data_ptr())
        else:
            mk = np.ascontiguousarray(np.asarray(env_mask).astype(np.uint8))
            if mk.shape != (self.n_envs,):
                raise ValueError("env_mask must have shape (n_envs,)")
            ptr = mk.ctypes.data_as(C.c_void_p)
        _abi.check(_abi.lib().bs_reset_noalloc(self._h, ptr))

    def sync(self):
        _abi.check(_abi.lib().bs_sync(self._h))

    def _state_rows(self):
        r = C.c_int()
        _abi.check(_abi.lib().bs_state_rows(self._h, C.byref(r)))
        return r.value

    def get_state(self):
        """`MjSim.get_state()`: MjSimState(time, qpos, qvel, act=None, udd_state) as views of ONE device
        snapshot buffer (bs_get_state); `udd_state` carries `qacc_warmstart` and `qacc`, which make a
        restored run bit-identical (the solvers warm-start from the previous acceleration)."""
        import torch
        m = self.model
        buf = torch.empty((self._state_rows(), self.n_envs), dtype=torch.float64, device=f"cuda:{self.device_id}")
        self.wait_for_torch()  # the buffer's memory may still be in use by earlier work of torch's stream
        _abi.check(_abi.lib().bs_get_state(self._h, C.c_void_p(buf.data_ptr()), self.n_envs))
        self.torch_waits()
        nq, nv = m.nq, m.nv
        cut = [0, 1, 1 + nq, 1 + nq + nv, 1 + nq + 2 * nv, 1 + nq + 3 * nv]
        t, q, v, a, w = [buf[cut[i]:cut[i + 1]].T for i in range(5)]
        if self.n_envs == 1:
            t, q, v, a, w = float(t[0, 0]), q[0], v[0], a[0], w[0]
        else:
            t = t[:, 0]
        return MjSimState(t, q, v, None, {"qacc": a, "qacc_warmstart": w, "_packed": buf})

    def set_state(self, state):
        """`MjSim.set_state(state)`; states without `udd_state["qacc_warmstart"]` (e.g. hand-made ones)
        restore time / qpos / qvel only and leave the warm start as it is (mujoco-py's behaviour)."""
        import torch
        udd = state.udd_state or {}
        buf = udd.get("_packed")
        E, dev = self.n_envs, f"cuda:{self.device_id}"
        if buf is None or tuple(buf.shape) != (self._state_rows(), E):
            buf = torch.empty((self._state_rows(), E), dtype=torch.float64, device=dev)
            self.wait_for_torch()
            _abi.check(_abi.lib().bs_get_state(self._h, C.c_void_p(buf.data_ptr()), E))
            self.torch_waits()
        else:
            buf = buf.clone()  # the caller may have edited the views
        m = self.model
        nq, nv = m.nq, m.nv
        cut = [0, 1, 1 + nq, 1 + nq + nv, 1 + nq + 2 * nv, 1 + nq + 3 * nv]
        parts = [state.time, state.qpos, state.qvel, udd.get("qacc"), udd.get("qacc_warmstart")]
        for i, val in enumerate(parts):
            if val is None:
                continue
            val = torch.as_tensor(val, dtype=torch.float64, device=dev)
            rows = cut[i + 1] - cut[i]
            val = val.reshape(E, rows) if val.numel() == E * rows else val.reshape(1, -1).expand(E, rows)
            buf[cut[i]:cut[i + 1]].copy_(val.T)
        self.wait_for_torch()  # the buffer was filled on torch's current stream
        _abi.check(_abi.lib().bs_set_state(self._h, C.c_void_p(buf.data_ptr()), E))
        self._keep_state = buf  # alive until the copies have run

    def kernel_path(self):
        """(fast, description): which kernel family bs_step runs for this sim (bs_sim_kernel_path)."""
        buf = C.create_string_buffer(1024)
        rc = _abi.lib().bs_sim_kernel_path(self._h, buf, 1024)
        if rc < 0:
            _abi.check(rc)
        return bool(rc), buf.value.decode()

    def footprint(self):
        """Device memory this sim holds (bs_sim_footprint): bytes per arena, their sum and bytes per environment."""
        out = (C.c_int64 * 5)()
        _abi.check(_abi.lib().bs_sim_footprint(self._h, out))
        d = dict(state=int(out[0]), contacts=int(out[1]), geom_frames=int(out[2]), serial_solver=int(out[3]))
        d["total"] = sum(d.values())
        d["bytes_per_env"] = d["total"] / max(1, int(out[4]))
        return d

    def stream_handle(self):
        """The hipStream_t the sim's kernels are enqueued on (bs_get_stream)."""
        h = C.c_void_p()
        _abi.check(_abi.lib().bs_get_stream(self._h, C.byref(h)))
        return h.value or 0

    def stage_times(self):
        ms = (C.c_double * 5)()
        _abi.check(_abi.lib().bs_stage_times(self._h, ms, 5))
        return dict(position=ms[0], collision=ms[1], velocity=ms[2], constraint=ms[3], integrate=ms[4])

    def collision_stats(self, reset=False):
        out = (C.c_uint64 * 3)()
        _abi.check(_abi.lib().bs_collision_stats(self._h, out, 1 if reset else 0))
        return dict(narrow_pairs=int(out[0]), mpr_iterations=int(out[1]), contacts=int(out[2]))

    def pack_obs(self, out):
        """obs = qpos | qvel (fp32) and the placeholder reward into `out` ([n_envs, >= nq + nv + 1]
        fp32 device tensor, rows contiguous): one kernel on the sim's stream (bs_pack_obs)."""
        m = self.model
        import torch
        if (out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] != self.n_envs
                or out.shape[1] < m.nq + m.nv + 1 or out.stride(1) != 1 or not out.is_cuda):
            raise ValueError("pack_obs needs a [n_envs, >= nq + nv + 1] fp32 CUDA tensor with contiguous rows")
        _abi.check(_abi.lib().bs_pack_obs(self._h, C.c_void_p(out.data_ptr()), int(out.stride(0))))
        return out

    def timed_steps(self, n):
        ms = C.c_double()
        _abi.check(_abi.lib().bs_timed_steps(self._h, int(n), C.byref(ms)))
        return ms.value


MjSim = BatchSim


class _Part:
    """Borrowed view of one model's environments inside a MixedBatchSim (dies with the parent)."""

    def __init__(self, parent, index):
        h = C.c_void_p()
        start = C.c_int64()
        n = C.c_int64()
        _abi.check(_abi.lib().bs_sim_part(parent._h, int(index), C.byref(h), C.byref(start), C.byref(n)))
        self._h = h
        self.env_start = int(start.value)
        self.n_envs = int(n.value)
        self.device_id = parent.device_id
        self.model = parent.models[parent._part_model[index]]
        self.data = BatchData(self)


class MixedBatchSim(BatchSim):
    """Mixed-model batch (BASELINE.json configs[3]): environments sorted by model, one part per
    model.  `models` is a list of compiled models, `env_model_id` a non-decreasing int array with
    one entry per environment (or `counts`, environments per model).  step/forward/reset/sync act
    on the whole batch; state and outputs are per part: `sim.parts[k].data.qpos` is
    `[n_envs_k, nq_k]`, `sim.parts[k].env_start` its first environment."""

    def __init__(self, models, env_model_id=None, counts=None, device_id=0, nsubsteps=1, use_torch_stream=True, max_contacts=0):
        """`max_contacts`: cap of the contact list per environment — one number for every model, or one per model
        (0 = the library's default, 32): a model that needs a longer list (up to 64) runs the wide tree-parallel
        kernels, the others keep the 32-lane ones."""
        if (env_model_id is None) == (counts is None):
            raise ValueError("give exactly one of env_model_id and counts")
        if counts is not None:
            if len(counts) != len(models):
                raise ValueError("counts needs one entry per model")
            env_model_id = np.repeat(np.arange(len(models)), np.asarray(counts, dtype=np.int64))
        ids = np.ascontiguousarray(np.asarray(env_model_id, dtype=np.int32))
        if ids.ndim != 1 or ids.size < 1:
            raise ValueError("env_model_id must be a non-empty 1-D array")
        if nsubsteps < 1:
            raise ValueError("nsubsteps must be >= 1")
        L = _abi.lib()
        self.models = list(models)
        self.model = None
        self.n_envs = int(ids.size)
        self.device_id = int(device_id)
        self.nsubsteps = int(nsubsteps)
        self._natives = [m if isinstance(m, NativeModel) else NativeModel(m) for m in self.models]
        self.models = [n.model for n in self._natives]
        arr = (C.c_void_p * len(self._natives))(*[n.handle for n in self._natives])
        h = C.c_void_p()
        caps = np.zeros(len(self._natives), dtype=np.int32)
        caps[:] = np.asarray(max_contacts, dtype=np.int32)  # scalar or one per model
        if (caps < 0).any():
            raise ValueError("max_contacts must be >= 0")
        opt = _abi.Options(max_contacts_per_env=0, max_contacts_per_model=caps.ctypes.data_as(C.POINTER(C.c_int32)))
        _abi.check(L.bs_sim_create(arr, len(self._natives), ids.ctypes.data_as(C.POINTER(C.c_int32)), self.n_envs,
                                   self.device_id, C.byref(opt), C.byref(h)))
        self._h = h
        self._part_model = [int(k) for k in sorted(set(ids.tolist()))]
        self.parts = [_Part(self, i) for i in range(L.bs_sim_num_parts(self._h))]
        self.data = None  # fields live in the parts
        fast, path = self.kernel_path()
        if not fast:
            import warnings
            warnings.warn("mujoco_amd: a part of this mixed-model batch runs on the generic one-environment-per-lane kernels, "
                          "about 40x slower than the tree-parallel ones: " + path, RuntimeWarning, stacklevel=2)
        if use_torch_stream:
            import torch
            with torch.cuda.device(self.device_id):
                st = torch.cuda.current_stream().cuda_stream
            _abi.check(L.bs_set_stream(self._h, C.c_void_p(st), 0))

    def contacts(self, env, cap=64):
        g1 = (C.c_int32 * cap)()
        g2 = (C.c_int32 * cap)()
        dist = (C.c_double * cap)()
        n = C.c_int()
        _abi.check(_abi.lib().bs_get_contacts(self._h, int(env), g1, g2, dist, cap, C.byref(n)))
        return [(g1[i], g2[i], dist[i]) for i in range(min(n.value, cap))]

    def get_state(self):
        return [MjSimState(p.data.time.clone(), p.data.qpos.clone(), p.data.qvel.clone(), None,
                           {"qacc": p.data.qacc.clone(), "qacc_warmstart": p.data.qacc_warmstart.clone()})
                for p in self.parts]

    def set_state(self, states):
        for p, st in zip(self.parts, states):
            p.data.time.copy_(st.time)
            p.data.qpos.copy_(st.qpos)
            p.data.qvel.copy_(st.qvel)
            udd = st.udd_state or {}
            if "qacc" in udd:
                p.data.qacc.copy_(udd["qacc"])
            if "qacc_warmstart" in udd:
                p.data.qacc_warmstart.copy_(udd["qacc_warmstart"])
