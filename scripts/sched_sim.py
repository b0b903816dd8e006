"""List-scheduling simulation from measured per-env wave ticks: how much of the launch tail is due to the
work estimate (previous step) vs. inherent.  Run on the GPU box; prints makespans in ticks."""
import os, sys, heapq
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
dbg = eng.enable_debug(); eng.enable_timing(True)

def makespan(order, t, slots=2048):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for e in order:
        s = heapq.heappop(h)
        f = s + t[e]
        end = max(end, f)
        heapq.heappush(h, f)
    return end

prev_work = None
for i in range(140):
    eng.fill_random_actions(act, i)
    if i >= 100: torch.cuda.synchronize(); prev_work = eng.get_work().cpu().numpy().copy()
    eng.step(act, out)
    if i >= 100 and i % 8 == 0:
        torch.cuda.synchronize()
        ms = eng.last_step_ms()
        t = dbg[:, 352:368].sum(1).cpu().numpy()
        b = np.clip(prev_work >> 5, 0, 255)
        est_order = np.argsort(-b, kind="stable")          # what dm_schedule_kernel does (bucket sort, heaviest first)
        print("step %d kernel %.3f ms = %.0f ticks | sum/2048 %.0f | max %.0f | makespan: est-LPT %.0f  perfect-LPT %.0f  natural %.0f  corr(work,ticks) %.2f"
              % (i, ms, ms * 2.33e6, t.sum() / 2048, t.max(), makespan(est_order, t), makespan(np.argsort(-t), t),
                 makespan(np.arange(N), t), np.corrcoef(prev_work, t)[0, 1]))
