"""Multi-GPU glue: environments shard across ranks with NO traffic inside a
step; the only collective is one all-gather of observations + rewards per control period for a
centralised learner — RCCL over xGMI through torch.distributed (backend "nccl"), gloo on CPU
in the tests.  One process per GPU."""
import torch


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of environments owned by `rank` (remainder to low ranks)."""
    if world < 1 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_obs(qpos, qvel, ctrl_target, act_joint):
    """obs = qpos|qvel (fp32) and the placeholder reward -||qpos[act] - ctrl||^2 of SURVEY.md §8d
    config 5, packed as one [E, nq+nv+1] fp32 buffer so a single collective moves both."""
    err = qpos[:, act_joint] - ctrl_target
    reward = -(err * err).sum(dim=1, keepdim=True)
    return torch.cat([qpos, qvel, reward], dim=1).to(torch.float32).contiguous()


def all_gather_packed(local, world, group=None, out=None, async_op=False):
    """[E_local, W] -> [world * E_local, W] on every rank (equal shard sizes).  With `async_op` the
    collective is only enqueued (on the backend's own stream) and `(out, work)` is returned: the
    caller's stream goes on with the next physics steps and calls `work.wait()` before it reads `out`."""
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
    return (out, work) if async_op else out


class ObsGather:
    """Per-control-period all-gather for a BatchSim shard."""

    def __init__(self, sim, world, group=None):
        self.sim, self.world, self.group = sim, world, group
        m = sim.model
        self.act_joint = torch.as_tensor(m.actuator_trnid, dtype=torch.long, device=sim.data.qpos.device)
        self.out = torch.empty((world * sim.n_envs, m.nq + m.nv + 1), dtype=torch.float32,
                               device=sim.data.qpos.device)

        self.local = torch.empty((sim.n_envs, m.nq + m.nv + 1), dtype=torch.float32, device=sim.data.qpos.device)
        self._work, self._local = None, None

    def all_gather(self, overlap=False):
        """Gathers the observations of the state as it is now.  `overlap=True`: the collective runs
        beside the physics steps enqueued after this call (xGMI traffic under compute); the result
        is complete after `wait()` (or at the start of the next `all_gather`)."""
        self.wait()
        # a snapshot (later steps do not touch it), written by one kernel (bs_pack_obs); the buffer is
        # reused: wait() above has ordered this stream after the collective that read it last
        local = self.local
        self.sim.pack_obs(local)
        # the collective is ordered after torch's current stream: if the sim runs on another stream
        # (use_torch_stream=False, or the caller switched streams), make the current stream wait for it
        cur = torch.cuda.current_stream(local.device)
        sh = self.sim.stream_handle()
        if sh != cur.cuda_stream:
            cur.wait_stream(torch.cuda.ExternalStream(sh, device=local.device))
        if not overlap:
            return all_gather_packed(local, self.world, self.group, self.out)
        self._local = local  # kept alive until the collective has read it
        _, self._work = all_gather_packed(local, self.world, self.group, self.out, async_op=True)
        return self.out

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work, self._local = None, None
        return self.out


class ShardedBatchSim:
    """`BatchSim(model, n_envs, device_ids=[...])` of SURVEY.md §8b for ONE process driving several GPUs: the
    environments are split into contiguous shards (`shard_range`), one `BatchSim` per device on its own stream, no
    traffic between them.  step / forward / reset only enqueue (the devices run concurrently), `sync`
    waits for all.  State lives in the shards: `shards[k].data.qpos` is `[n_k, nq]` on device `device_ids[k]`,
    `env_range[k]` its `[lo, hi)` slice of the batch; `gather(name)` / `scatter(name, value)` move a whole-batch host
    array.  One process per GPU (`bench.py --gpus N`, torch.distributed) is the production shape; this class is the
    single-process convenience (a device may be listed twice: two shards on one GPU)."""

    def __init__(self, model, n_envs, device_ids, **kw):
        from .sim import BatchSim
        ids = [int(d) for d in device_ids]
        if not ids:
            raise ValueError("device_ids must name at least one device")
        if n_envs < len(ids):
            raise ValueError("fewer environments than devices")
        self.model, self.n_envs, self.device_ids = model, int(n_envs), ids
        self.env_range = [shard_range(self.n_envs, r, len(ids)) for r in range(len(ids))]
        kw.setdefault("use_torch_stream", False)  # torch's current stream belongs to ONE device
        self.shards = [BatchSim(model, n_envs=hi - lo, device_id=d, **kw) for d, (lo, hi) in zip(ids, self.env_range)]

    def step(self, n=None):
        for s in self.shards:
            s.step(n)

    def forward(self):
        for s in self.shards:
            s.forward()

    def reset(self, env_mask=None):
        for s, (lo, hi) in zip(self.shards, self.env_range):
            s.reset(None if env_mask is None else env_mask[lo:hi])

    def sync(self):
        for s in self.shards:
            s.sync()

    def gather(self, name):
        """Whole-batch host copy `[n_envs, ...]` of a `sim.data` field (synchronises every shard)."""
        import numpy as np
        self.sync()
        parts = []
        for s in self.shards:
            t = getattr(s.data, name)
            parts.append((t.unsqueeze(0) if s.n_envs == 1 and name not in ("time", "solver_iter", "warning", "nefc", "ncon") else t).cpu().numpy())
        return np.concatenate(parts, axis=0)

    def scatter(self, name, value):
        import numpy as np
        value = np.asarray(value)
        for s, (lo, hi) in zip(self.shards, self.env_range):
            with torch.cuda.device(s.device_id):
                getattr(s.data, name).copy_(torch.as_tensor(value[lo:hi] if s.n_envs > 1 else value[lo]))
            torch.cuda.current_stream(s.device_id).synchronize()  # the copy ran on torch's stream, the sim steps on its own

    def close(self):
        for s in self.shards:
            s.close()
