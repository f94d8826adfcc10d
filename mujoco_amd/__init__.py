"""mujoco_amd — MI355X-native batched physics step behind a mujoco-py shaped surface.

Only the hot path named by BASELINE.json's north_star lives here: the MJCF-subset
model compiler (host, NumPy) and `BatchSim`, which drives hand-written HIP kernels
through the C-ABI declared in include/batchsim.h.
"""
from .mjcf import load_model_from_path, load_model_from_xml  # noqa: F401
from .model import Model  # noqa: F401


def __getattr__(name):
    # BatchSim needs the native library; import lazily so the model compiler is
    # usable (and testable) without it.
    if name in ("BatchSim", "MixedBatchSim", "MjSim", "MjSimState", "BatchSimError"):
        from . import sim
        return getattr(sim, name)
    raise AttributeError(name)
