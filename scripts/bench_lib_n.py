import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
for N in [int(x) for x in sys.argv[2:]]:
    eng = L.HipEngine(model, N); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    for i in range(50): eng.fill_random_actions(act, i); eng.step(act, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 200 if N <= 8192 else 60
    for i in range(steps): eng.fill_random_actions(act, 50 + i); eng.step(act, out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(sys.argv[1], "N=%d env-steps/s %.0f  ms/step %.3f" % (N, steps * N / dt, dt / steps * 1e3))
    eng.close()
