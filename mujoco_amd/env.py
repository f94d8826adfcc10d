"""Vectorised RL-environment surface over `BatchSim`.

The reference has no environment, reward or reset code (`src/Mujoco_XML.py:327-330` is a TODO), so
everything here is this build's definition, kept deliberately small:

* observation = `qpos | qvel` as fp32, `[n_envs, nq + nv]`;
* action = position targets of the exported `<position>` actuators (`src/Mujoco_XML.py:72,261`),
  clamped to their ctrlrange, held for `frame_skip` physics steps (one control period);
* reward = `-||qpos[actuated joints] - action||^2` (the placeholder of SURVEY.md §8d config 5);
* an episode ends after `episode_steps` control periods or when `mj_check*` reset the environment
  (`warning & 7 != 0`); finished environments are reset in place — all on the device, through the
  zero-copy views of the library's arrays, so a rollout never synchronises with the host.

With `world > 1` (one process per GPU) `step` also returns the all-gathered `[world * n_envs, ...]`
observation / reward of every rank for a centralised learner (`mujoco_amd.dist`, RCCL).
"""
import ctypes as C

import torch

from . import _abi
from . import dist as bdist


class HandVecEnv:
    def __init__(self, sim, frame_skip=50, episode_steps=200, reset_noise=0.0, seed=0, world=1, group=None):
        if sim.n_envs < 2:
            raise ValueError("HandVecEnv needs a batched sim (n_envs >= 2)")
        self.sim, self.frame_skip, self.episode_steps, self.reset_noise = sim, int(frame_skip), int(episode_steps), float(reset_noise)
        m, d = sim.model, sim.data
        dev = d.qpos.device
        self.n_envs, self.obs_dim, self.act_dim = sim.n_envs, m.nq + m.nv, m.nu
        self.act_joint = torch.as_tensor(m.actuator_trnid, dtype=torch.long, device=dev)
        self.ctrl_lo = torch.as_tensor(m.actuator_ctrlrange[:, 0], device=dev)
        self.ctrl_hi = torch.as_tensor(m.actuator_ctrlrange[:, 1], device=dev)
        self.qpos0 = torch.as_tensor(m.qpos0, device=dev)
        self.jnt_lo = torch.as_tensor(m.jnt_range[:, 0], device=dev)
        self.jnt_hi = torch.as_tensor(m.jnt_range[:, 1], device=dev)
        self.t = torch.zeros(self.n_envs, dtype=torch.int32, device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        self.world, self.group = int(world), group
        self._gathered = (torch.empty((self.world * self.n_envs, self.obs_dim + 1), dtype=torch.float32, device=dev)
                          if self.world > 1 else None)

    def _obs(self):
        d = self.sim.data
        return torch.cat([d.qpos, d.qvel], dim=1).to(torch.float32)

    def _reset_where(self, mask):
        """mj_resetData for the environments in `mask` (device bool [n_envs]): one library kernel
        (bs_reset_device_mask: state, inputs, time and warning flags of the masked environments),
        optionally followed by uniform noise on qpos inside the joint ranges."""
        d = self.sim.data
        self.sim.reset(mask)
        if self.reset_noise > 0:
            u = torch.rand((self.n_envs, self.qpos0.numel()), generator=self.gen, device=mask.device, dtype=d.qpos.dtype)
            q = torch.minimum(torch.maximum(self.qpos0 + self.reset_noise * (u - 0.5) * (self.jnt_hi - self.jnt_lo), self.jnt_lo), self.jnt_hi)
            d.qpos.copy_(torch.where(mask.unsqueeze(1), q, d.qpos))
        self.t.mul_((~mask).to(self.t.dtype))

    def reset(self):
        self._reset_where(torch.ones(self.n_envs, dtype=torch.bool, device=self.t.device))
        return self._obs()

    def step(self, action):
        """action: `[n_envs, nu]` device tensor.  Returns obs, reward, done, info; finished
        environments are already reset in the returned observation (`info["terminal_obs"]` holds
        the last observation of the finished episodes).  Without reset noise a control period is three
        launches — bs_env_pre, bs_step, bs_env_post — instead of some twenty torch kernels (at 4096
        contact-free environments and frame_skip 10 those were half of the period)."""
        if self.reset_noise > 0:
            return self._step_torch(action)
        L, h = _abi.lib(), self.sim._h
        a = action.to(torch.float64).contiguous()
        dev = a.device  # fresh outputs every period (the caching allocator hands them out without a device call)
        obs = torch.empty((self.n_envs, self.obs_dim), dtype=torch.float32, device=dev)
        term = torch.empty((self.n_envs, self.obs_dim), dtype=torch.float32, device=dev)
        reward = torch.empty(self.n_envs, dtype=torch.float32, device=dev)
        flags = torch.empty(self.n_envs, dtype=torch.uint8, device=dev)
        self.sim.wait_for_torch()  # the action (and the fresh outputs' memory) belong to torch's current stream
        _abi.check(L.bs_env_pre(h, C.c_void_p(a.data_ptr()), a.shape[1]))
        self.sim.step(self.frame_skip)
        _abi.check(L.bs_env_post(h, C.c_void_p(a.data_ptr()), a.shape[1], C.c_void_p(self.t.data_ptr()), self.episode_steps,
                                 C.c_void_p(obs.data_ptr()), C.c_void_p(term.data_ptr()), self.obs_dim,
                                 C.c_void_p(reward.data_ptr()), C.c_void_p(flags.data_ptr())))
        self._keep = a  # alive until the kernels that read it have run
        self.sim.torch_waits()  # obs / reward / flags are read on torch's current stream
        done, failed, overflow = (flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0
        info = {"terminal_obs": term, "failed": failed, "contact_overflow": overflow}
        if self.world > 1:
            packed = torch.cat([obs, reward.unsqueeze(1)], dim=1).contiguous()
            info["gathered"] = bdist.all_gather_packed(packed, self.world, self.group, self._gathered)
        return obs, reward, done, info

    def _step_torch(self, action):
        """The same control period written with torch operations on the zero-copy views (the path with
        reset noise, and the reference the fused kernels are tested against)."""
        d = self.sim.data
        a = torch.minimum(torch.maximum(action.to(d.ctrl.dtype), self.ctrl_lo), self.ctrl_hi)
        d.ctrl.copy_(a)
        self.sim.step(self.frame_skip)
        err = d.qpos[:, self.act_joint] - a
        reward = -(err * err).sum(dim=1).to(torch.float32)
        self.t += 1
        # bits 0-2: mj_checkPos / Vel / Acc reset the environment; bit 3 (value 8) only says that the
        # contact or candidate list overflowed (the step went on with a truncated list): reported, not an episode end
        failed = (d.warning & 7) != 0
        overflow = (d.warning & 8) != 0
        d.warning.bitwise_and_(7)
        done = failed | (self.t >= self.episode_steps)
        obs = self._obs()
        info = {"terminal_obs": obs, "failed": failed, "contact_overflow": overflow}
        self._reset_where(done)
        obs = torch.where(done.unsqueeze(1), self._obs(), obs)
        if self.world > 1:
            packed = torch.cat([obs, reward.unsqueeze(1)], dim=1).contiguous()
            info["gathered"] = bdist.all_gather_packed(packed, self.world, self.group, self._gathered)
        return obs, reward, done, info
