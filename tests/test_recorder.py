"""Host logic of the headless trajectory recorder (f-4) on a stand-in sim with CPU tensors."""
import numpy as np
import pytest
import torch

from mujoco_amd.recorder import TrajectoryRecorder


class _Data:
    def __init__(self, n):
        self.time = torch.zeros(n, dtype=torch.float64)
        self.qpos = torch.zeros((n, 3), dtype=torch.float64)
        self.ncon = torch.zeros(n, dtype=torch.int32)
        self.xpos = torch.zeros((n, 2, 3), dtype=torch.float64)


class _Sim:
    def __init__(self, n):
        self.n_envs, self.data = n, _Data(n)


def test_frames_and_files(tmp_path):
    sim = _Sim(4)
    rec = TrajectoryRecorder(sim, fields=("time", "qpos", "ncon"), envs=[0, 3], capacity=3)
    for k in range(3):
        sim.data.time += 0.5
        sim.data.qpos += torch.arange(4, dtype=torch.float64)[:, None]
        sim.data.ncon += 1
        rec.record(k * 7)
    with pytest.raises(RuntimeError):
        rec.record(99)
    a = rec.arrays()
    assert a["qpos"].shape == (3, 2, 3) and a["ncon"].dtype == np.int32
    assert np.array_equal(a["qpos"][:, 1, 0], [3, 6, 9]) and np.array_equal(a["time"][:, 0], [0.5, 1.0, 1.5])
    assert list(a["frame_step"]) == [0, 7, 14]
    z = np.load(rec.save(str(tmp_path / "t.npz")))
    assert np.array_equal(z["qpos"], a["qpos"])
    rec.save(str(tmp_path / "t.csv"))
    head = open(tmp_path / "t.csv").readline().strip().split(",")
    assert head == ["step", "env", "time", "qpos_0", "qpos_1", "qpos_2", "ncon"]
    t = np.loadtxt(tmp_path / "t.csv", delimiter=",", skiprows=1)
    assert t.shape == (6, 7) and list(t[:, 1]) == [0, 3] * 3 and list(t[:, 0]) == [0, 0, 7, 7, 14, 14]


def test_rejects_bad_arguments(tmp_path):
    sim = _Sim(2)
    with pytest.raises(ValueError):
        TrajectoryRecorder(sim, envs=[2])
    with pytest.raises(ValueError):
        TrajectoryRecorder(sim, capacity=0)
    rec = TrajectoryRecorder(sim, fields=("xpos",), capacity=1)
    rec.record(0)
    with pytest.raises(ValueError):
        rec.save(str(tmp_path / "x.csv"))
