"""C-ABI loads and exports every symbol include/batchsim.h declares (no compute without a GPU);
host logic around it; world_size-2 gloo test of the multi-GPU glue."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_header_symbols_exported():
    from mujoco_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "batchsim.h")).read()
    declared = set(re.findall(r"\b(bs_[a-z_]+)\s*\(", hdr))
    assert declared == set(_abi.SYMBOLS)
    L = _abi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.bs_version() == 200
    # field enum order in the header == python table
    enum = re.search(r"typedef enum bs_field \{(.*?)\} bs_field;", hdr, re.S).group(1)
    enum = re.sub(r"/\*.*?\*/", "", enum, flags=re.S)
    names = [n.strip().split("=")[0].strip() for n in enum.split(",") if n.strip()]
    assert names[-1] == "BS_F_COUNT"
    assert [n[5:].lower() for n in names[:-1]] == [f.lower() for f in _abi.FIELD_NAMES]


def test_model_desc_validation_without_gpu(hand_model):
    """bs_model_create is host-only: validates and rejects bad descriptions with a message."""
    import ctypes as C
    from mujoco_amd import _abi, workload
    from mujoco_amd.sim import NativeModel
    NativeModel(workload.contact_free(hand_model))  # valid
    bad = hand_model.copy()
    bad.opt_solver = 7
    with pytest.raises(_abi.BatchSimError, match="not implemented"):
        NativeModel(bad)
    bad = hand_model.copy()
    bad.dof_parentid = bad.dof_parentid.copy()
    bad.dof_parentid[3] = 7
    with pytest.raises(_abi.BatchSimError):
        NativeModel(bad)
    bad = hand_model.copy()
    bad.opt_timestep = 0.0
    with pytest.raises(_abi.BatchSimError, match="timestep"):
        NativeModel(bad)
    # the per-pair collision records hold 12-bit geom ids: 4096 geoms and up are refused with a message
    from mujoco_amd import load_model_from_xml
    spheres = "".join(f'<geom type="sphere" size="0.001" pos="{0.01 * (i % 64)} {0.01 * (i // 64)} 0.1"/>' for i in range(4100))
    big = load_model_from_xml('<mujoco><worldbody><geom type="plane" size="5 5 0.1"/><body><joint type="hinge" axis="0 0 1"/>'
                              + spheres + "</body></worldbody></mujoco>")
    assert big.ngeom > 4095 and big.npair > 0
    with pytest.raises(_abi.BatchSimError, match="12-bit"):
        NativeModel(big)
    h = C.c_void_p()
    assert _abi.lib().bs_model_create(None, C.byref(h)) == -1
    n = C.c_int(-1)
    assert _abi.lib().bs_device_count(C.byref(n)) == 0 and n.value >= 0


def test_product_fails_loudly_without_device(hand_model):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mujoco_amd import BatchSim, BatchSimError
    with pytest.raises(BatchSimError, match="no CPU fallback"):
        BatchSim(hand_model, n_envs=4)


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (tier framing ③)."""
    pkg = os.path.join(ROOT, "mujoco_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_workload_determinism(hand_model):
    from mujoco_amd import workload
    q1, _ = workload.config2_state(hand_model, 8, seed=1234)
    q2, _ = workload.config2_state(hand_model, 8, seed=1234)
    assert np.array_equal(q1, q2)
    lo, hi = hand_model.jnt_range[:, 0], hand_model.jnt_range[:, 1]
    free = np.setdiff1d(np.arange(21), hand_model.eq_obj1id)
    assert np.all(q1[:, free] >= lo[free]) and np.all(q1[:, free] <= hi[free])
    assert np.allclose(q1[:, 4], 70 / 120 * q1[:, 3])
    c0 = workload.config2_ctrl(hand_model, 8, seed=1234, resample_idx=0)
    c1 = workload.config2_ctrl(hand_model, 8, seed=1234, resample_idx=1)
    assert not np.array_equal(c0, c1)
    assert np.all(c0 >= hand_model.actuator_ctrlrange[:, 0]) and np.all(c0 <= hand_model.actuator_ctrlrange[:, 1])


def test_shard_range():
    from mujoco_amd.dist import shard_range
    for n, w in ((1048576, 8), (10, 3), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from mujoco_amd.dist import pack_obs, all_gather_packed, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
E, nq, nu = 6, 21, 17
lo, hi = shard_range(E * world, rank, world)
g = torch.Generator().manual_seed(0)
qpos_all = torch.rand(E * world, nq, generator=g, dtype=torch.float64)
qvel_all = torch.rand(E * world, nq, generator=g, dtype=torch.float64)
ctrl_all = torch.rand(E * world, nu, generator=g, dtype=torch.float64)
act = torch.arange(nu)
local = pack_obs(qpos_all[lo:hi], qvel_all[lo:hi], ctrl_all[lo:hi], act)
out = all_gather_packed(local, world)
want = pack_obs(qpos_all, qvel_all, ctrl_all, act)
assert out.shape == (E * world, 2 * nq + 1) and out.dtype == torch.float32
assert torch.equal(out, want), "gathered observations differ from the unsharded ones"
out2, work = all_gather_packed(local, world, async_op=True)   # the overlapped form bench.py uses
work.wait()
assert torch.equal(out2, want)
dist.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, f"ok_{{rank}}"), "w").write("ok")
"""


def test_allgather_obs_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


_RANK_SCRIPT = r"""
import argparse, json, os, sys
import torch, torch.distributed as dist
ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int)
ap.add_argument("--fail-rank", type=int, default=-1)
a = ap.parse_args()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == a.gpus == int(os.environ["WORLD_SIZE"])
one = torch.ones(1, dtype=torch.int64)
dist.all_reduce(one)
if rank == a.fail_rank:
    sys.exit(3)
if rank == 0:
    print(json.dumps({"n_gpus": world, "ranks_joined": int(one.item())}), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_spawn_ranks_from_a_plain_shell_gloo(tmp_path, monkeypatch):
    """`bench.py --gpus N` without WORLD_SIZE starts its own ranks through mujoco_amd.launch.spawn_ranks: rehearsed
    here with a two-rank gloo script (rank 0's JSON line is relayed, a failing rank makes the exit code non-zero,
    rank variables of an outer launcher do not leak into the child)."""
    import io
    import json
    from mujoco_amd.launch import spawn_ranks
    script = tmp_path / "r.py"
    script.write_text(_RANK_SCRIPT)
    monkeypatch.setenv("WORLD_SIZE", "7")  # must not reach the ranks
    monkeypatch.setenv("RANK", "5")
    buf = io.StringIO()
    rc = spawn_ranks(2, script, ["--gpus", "2"], out=buf, timeout=300)
    assert rc == 0, buf.getvalue()
    line = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    assert len(line) == 1 and json.loads(line[0]) == {"n_gpus": 2, "ranks_joined": 2}
    buf = io.StringIO()
    assert spawn_ranks(2, script, ["--gpus", "2", "--fail-rank", "1"], out=buf, timeout=300) != 0


def test_bench_gpus_2_launches_two_ranks_itself(tmp_path):
    """From a plain shell `python bench.py --gpus 2` becomes a launcher (no GPU here: both ranks refuse to run
    without an MI355X, and the parent exits non-zero with their message on stderr instead of silently running one rank)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available():
        pytest.skip("rehearsal for the GPU-less container")
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]
    # under a launcher whose world size contradicts --gpus the rank refuses instead of running a different job
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
