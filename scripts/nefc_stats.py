"""Distribution of constraint rows / sweeps for the random-torque and the zero-action workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
for mode in ("random", "zero"):
    eng = L.HipEngine(model, N); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    dbg = eng.enable_debug()
    ne, it = [], []
    for i in range(150):
        if mode == "random": eng.fill_random_actions(act, i)
        eng.step(act, out)
        if i >= 50 and i % 10 == 0:
            torch.cuda.synchronize(); d = dbg.cpu().numpy(); ne.append(d[:, 243].copy()); it.append(d[:, 244].copy())
    ne = np.concatenate(ne); it = np.concatenate(it)
    print("%s: nefc mean %.1f p50 %d p90 %d p99 %d max %d | frac nefc>32: %.3f, >64: %.4f | sweeps mean %.1f p90 %d"
          % (mode, ne.mean(), np.median(ne), np.percentile(ne, 90), np.percentile(ne, 99), ne.max(), (ne > 32).mean(), (ne > 64).mean(),
             it.mean(), np.percentile(it, 90)))
    eng.close()
