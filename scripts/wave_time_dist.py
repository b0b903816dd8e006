"""Distribution of per-env wave time (s_memtime ticks, DM_PROFILE build) against the kernel duration."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_prof.so")
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
eng = L.HipEngine(model, N); eng.load_clip(0, mc)
out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
dbg = eng.enable_debug()
eng.enable_timing(True)
for i in range(120):
    eng.fill_random_actions(act, i); eng.step(act, out)
    if i >= 100 and i % 4 == 0:
        torch.cuda.synchronize()
        ms = eng.last_step_ms()
        d = dbg.cpu().numpy()
        tot = d[:, 352:368].sum(1)
        nefc = d[:, 243]
        work = eng.get_work().cpu().numpy()
        o = np.argsort(-tot)
        print("step %d kernel %.3f ms | ticks mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | sum/2048 %.0f | top5 ticks %s nefc %s work %s done %s"
              % (i, ms, tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(),
                 tot.sum() / 2048, tot[o[:5]].astype(int), nefc[o[:5]].astype(int), work[o[:5]], out["done"].cpu().numpy()[o[:5]]))
