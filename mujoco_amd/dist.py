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
