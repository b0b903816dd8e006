import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepmimic_mujoco_amd._lib import HipEngine
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
for mode in ["identity", "lpt_every", "lpt_every4"]:
    eng = HipEngine(model, N); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    cost = torch.zeros(N, dtype=torch.int32, device=eng.device)
    order = torch.arange(N, dtype=torch.int32, device=eng.device)
    if mode != "identity": eng.set_schedule(order, cost)
    def step(i):
        eng.fill_random_actions(act, i); eng.step(act, out)
        if mode == "lpt_every" or (mode == "lpt_every4" and i % 4 == 3):
            order.copy_(torch.argsort(cost, descending=True).to(torch.int32))
    for i in range(50): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): step(50 + i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "env-steps/s %.0f  ms/step %.3f" % (200 * N / dt, dt / 200 * 1e3), "cost mean %.0f max %d" % (cost.float().mean().item(), cost.max().item()))
    eng.close()
