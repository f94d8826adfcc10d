"""bs_reset_noalloc — the library's one allocation-free entry point (static __device__ scratch for
the device copy of the mask) against bs_reset, through the C-ABI.  Collected last on purpose."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_FIELDS = ("qpos", "qvel", "qacc", "qacc_warmstart", "ctrl", "qfrc_applied", "time", "warning", "solver_iter", "nefc")


def _pair(model, E, seed):
    """Two sims in the same stepped state."""
    import torch
    from mujoco_amd import BatchSim, workload
    q, v = workload.config2_state(model, E, seed=seed)
    c = workload.config2_ctrl(model, E, seed=seed)
    sims = []
    for _ in range(2):
        s = BatchSim(model, n_envs=E)
        s.data.qpos.copy_(torch.as_tensor(q)); s.data.ctrl.copy_(torch.as_tensor(c))
        s.step(7)
        sims.append(s)
    return sims


def test_reset_noalloc_matches_reset(hand_model):
    import torch
    from mujoco_amd import BatchSim, BatchSimError, _abi, workload
    m = workload.contact_free(hand_model)
    rng = np.random.default_rng(5)
    # two calls in a row on one GPU, the second with a smaller input: nothing of the first call's mask
    # (all ones beyond the second sim's size) may leak into the second
    for E, mask in ((700, np.ones(700, dtype=np.uint8)), (130, (rng.random(130) < 0.3).astype(np.uint8))):
        if E == 700:
            mask[[3, 64, 699]] = 0
        a, b = _pair(m, E, seed=E)
        a.reset(mask)            # existing entry point (mask staged in memory the sim allocated)
        b.reset_noalloc(mask)    # allocation-free entry point (static __device__ scratch)
        a.sync(); b.sync()
        for f in _FIELDS:
            assert torch.equal(getattr(a.data, f), getattr(b.data, f)), (E, f)
        keep = np.flatnonzero(mask == 0)
        assert float(b.data.qpos[int(keep[0])].abs().max()) > 0 and float(b.data.time[int(keep[0])]) > 0
        hit = np.flatnonzero(mask)
        assert float(b.data.time[int(hit[0])]) == 0.0
        # both keep stepping identically afterwards
        a.step(3); b.step(3); a.sync(); b.sync()
        assert torch.equal(a.data.qpos, b.data.qpos)
    # pinned host memory: the asynchronous path
    a, b = _pair(m, 96, seed=9)
    pm = torch.zeros(96, dtype=torch.uint8).pin_memory()
    pm[10:40] = 1
    a.reset(pm.numpy()); b.reset_noalloc(pm)
    a.sync(); b.sync()
    for f in _FIELDS:
        assert torch.equal(getattr(a.data, f), getattr(b.data, f)), f
    # a mixed-model batch shares the one scratch array
    from mujoco_amd import MixedBatchSim
    m2 = m.copy(); m2.dof_damping = m2.dof_damping * 2.0
    mixed = [MixedBatchSim([m, m2], counts=[70, 50]) for _ in range(2)]
    mk = (rng.random(120) < 0.5).astype(np.uint8)
    for s in mixed:
        for p in s.parts:
            p.data.qpos.copy_(torch.as_tensor(workload.config2_state(p.model, p.n_envs, seed=4)[0]))
        s.step(5)
    mixed[0].reset(mk); mixed[1].reset_noalloc(mk)
    for s in mixed:
        s.sync()
    for pa, pb in zip(mixed[0].parts, mixed[1].parts):
        for f in ("qpos", "qvel", "time"):
            assert torch.equal(getattr(pa.data, f), getattr(pb.data, f)), f
    # beyond the static scratch: a clear error, nothing launched
    L = _abi.lib()
    big = BatchSim(m, n_envs=(1 << 19) + 1)
    with pytest.raises(BatchSimError, match="exceed the static scratch"):
        big.reset_noalloc(np.ones(big.n_envs, dtype=np.uint8))
    big.close()
    with pytest.raises(BatchSimError):
        _abi.check(L.bs_reset_noalloc(a._h, None))


def test_reset_noalloc_two_sims_on_two_streams(hand_model):
    """Two sims on their OWN streams (use_torch_stream=False), pinned masks, calls issued back to back from one host
    thread: the second call's copy into the shared scratch array waits on the device for the first sim's reset kernel
    (per-device event), so neither mask is overwritten before it has been read."""
    import torch
    from mujoco_amd import BatchSim, workload
    m = workload.contact_free(hand_model)
    rng = np.random.default_rng(11)
    EA, EB = 200_000, 4_096
    q = {E: workload.config2_state(m, E, seed=E)[0] for E in (EA, EB)}
    masks = {E: torch.as_tensor((rng.random(E) < 0.5).astype(np.uint8)).pin_memory() for E in (EA, EB)}
    sims = {}
    for kind in ("ref", "noalloc"):
        for E in (EA, EB):
            s = BatchSim(m, n_envs=E, use_torch_stream=False)
            s.data.qpos.copy_(torch.as_tensor(q[E]))
            torch.cuda.synchronize()
            s.step(2)
            sims[kind, E] = s
    assert sims["noalloc", EA].stream_handle() != sims["noalloc", EB].stream_handle()
    for rep in range(3):
        for E in (EA, EB):
            sims["ref", E].reset(masks[E].numpy())
        sims["noalloc", EA].reset_noalloc(masks[EA])   # back to back, different streams, no host wait in between
        sims["noalloc", EB].reset_noalloc(masks[EB])
        for s in sims.values():
            s.step(1)
    for s in sims.values():
        s.sync()
    for E in (EA, EB):
        for f in ("qpos", "qvel", "time"):
            assert torch.equal(getattr(sims["ref", E].data, f), getattr(sims["noalloc", E].data, f)), (E, f)
    for s in sims.values():
        s.close()
