"""Developer tool: per-section dynamic instruction counts of k_tpc (contact step) or k_tp (contact-free).
Needs a -DTPC_CUTS build (tools/variant.sh cuts "-DTPC_CUTS" batchsim k_tpc_chain k_tp_chain) selected with BS_LIB.
Run under `rocprofv3 --pmc ... -- python3 tools/cuts_run.py [--free]`; the truncated kernel returns before it stores its
state, so every cut sees the same state.  tools/cuts_parse.py turns the counter CSV into the table."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mujoco_amd import BatchSim, Model, workload

FREE = "--free" in sys.argv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m = Model.load(os.path.join(ROOT, "tests", "golden", "dexterous_model.npz"))
m = workload.contact_free(m) if FREE else workload.with_ground_plane(m)
E = 4096 if FREE else 65536
q, v = workload.config2_state(m, E, seed=1234)
sim = BatchSim(m, n_envs=E, device_id=0)
sim.data.qpos.copy_(torch.as_tensor(q))
bank = [torch.as_tensor(workload.config2_ctrl(m, E, seed=1234, resample_idx=k)).to("cuda:0") for k in range(2)]
os.environ.pop("BS_CUT", None)
for k in range(2):   # two control periods, then 10 steps into the third (a mix of fresh and settled environments)
    sim.data.ctrl.copy_(bank[k])
    for _ in range(50):
        sim.step(1)
sim.data.ctrl.copy_(bank[0])
for _ in range(10):
    sim.step(1)
sim.sync()
# order of the cut points along the program; -1 = whole kernel
CUTS = [0, 18, 19, 12, 13, 14, 15, 16, 1, 2, 3, 4, 5, 6, 7, 8, 9, 23, 24, 10, 25, 17, 26, -1]
REP = 2
for c in CUTS:
    if c < 0:
        os.environ.pop("BS_CUT", None)
    else:
        os.environ["BS_CUT"] = str(c)
    for _ in range(REP):
        sim.step(1)
sim.sync()
json.dump({"cuts": CUTS, "rep": REP, "warm_steps": 110, "free": FREE}, open(os.path.join(ROOT, "gpurun_out", "cuts_order.json"), "w"))
