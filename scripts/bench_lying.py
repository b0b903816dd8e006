"""Throughput when every env lies on the floor (getup clip from frame 0, zero torques): the 65..128-row path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmimic_mujoco_amd._lib as L
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model()
mc = MocapDM(model=model); mc.load_mocap(MotionConfig("getup_facedown").mocap_path)
N = 4096
eng = L.HipEngine(model, N, auto_reset=False, max_ep_length=0); eng.load_clip(0, mc, floor=True)
out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
eng.reset(out["obs"], idx_init=torch.zeros(N, dtype=torch.int32, device=eng.device))
dbg = eng.enable_debug()
for i in range(60): eng.step(act, out)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
print("nefc mean %.1f max %d, sweeps mean %.1f, ncon mean %.1f" % (d[:, 243].mean(), d[:, 243].max(), d[:, 244].mean(), d[:, 242].mean()))
eng.enable_debug(False)
t0 = time.perf_counter()
for i in range(100): eng.step(act, out)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("lying: %.2f M env-steps/s, %.3f ms/step" % (100 * N / dt / 1e6, dt / 100 * 1e3))
