"""Per-wave s_memtime ticks (DM_PROFILE build) of the 2- and 3-wave kernel variants at several resident-wave counts:
separates a wave's own latency from the cost of sharing a SIMD.  DM_WAVES=2|3 selects the variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_prof.so")
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
for N in (1, 1024, 2048, 3072, 4096, 8192):
    eng = L.HipEngine(model, N); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    dbg = eng.enable_debug(); eng.enable_timing(True)
    res = []
    for i in range(40):
        eng.fill_random_actions(act, i); eng.step(act, out); torch.cuda.synchronize()
        if i >= 20:
            t = dbg[:, 352:368].sum(1)
            res.append((eng.last_step_ms(), t.max().item(), t.mean().item()))
    r = np.array(res)
    print("DM_WAVES=%s N=%d kernel %.4f ms (%.2f M env-steps/s) | wave ticks mean %.0f max %.0f" % (
        os.environ.get("DM_WAVES", "auto"), N, r[:, 0].mean(), N / r[:, 0].mean() / 1e3, r[:, 2].mean(), r[:, 1].mean()))
    eng.close()
