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
acc = np.zeros(16)
for i in range(60):
    eng.fill_random_actions(act, i); eng.step(act, out)
    if i >= 20:
        torch.cuda.synchronize(); acc += dbg[:, 352:368].cpu().numpy().mean(0)
acc /= 40
print("slots:", " ".join("%d:%.0f" % (i, a) for i, a in enumerate(acc)))
