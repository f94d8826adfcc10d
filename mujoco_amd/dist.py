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


def all_gather_packed(local, world, group=None, out=None):
    """[E_local, W] -> [world * E_local, W] on every rank (equal shard sizes)."""
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


class ObsGather:
    """Per-control-period all-gather for a BatchSim shard."""

    def __init__(self, sim, world, group=None):
        self.sim, self.world, self.group = sim, world, group
        m = sim.model
        self.act_joint = torch.as_tensor(m.actuator_trnid, dtype=torch.long, device=sim.data.qpos.device)
        self.out = torch.empty((world * sim.n_envs, m.nq + m.nv + 1), dtype=torch.float32,
                               device=sim.data.qpos.device)

    def all_gather(self):
        d = self.sim.data
        local = pack_obs(d.qpos, d.qvel, d.ctrl, self.act_joint)
        return all_gather_packed(local, self.world, self.group, self.out)
