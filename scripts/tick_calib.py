"""Calibrate s_memtime ticks (DM_PROFILE build) against HIP-event time with a single resident wave."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_prof.so")
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
for N in (1, 64, 1024, 2048):
    eng = L.HipEngine(model, N, lpt_schedule=0); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    dbg = eng.enable_debug(); eng.enable_timing(True)
    res = []
    for i in range(30):
        eng.fill_random_actions(act, i); eng.step(act, out); torch.cuda.synchronize()
        if i >= 10:
            res.append((eng.last_step_ms(), dbg[:, 352:368].sum(1).max().item(), dbg[:, 352:368].sum(1).mean().item()))
    r = np.array(res)
    print("N=%d kernel %.4f ms | max wave ticks %.0f mean %.0f | ticks/ms (max wave) %.0f" % (N, r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), (r[:, 1] / r[:, 0]).mean()))
    eng.close()
