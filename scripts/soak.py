"""Soak: 4096 envs x 3000 random-torque steps (both tasks); finite outputs, plausible rates, bit-identical repeat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model()
def clip(name):
    mc = MocapDM(model=model); mc.load_mocap(MotionConfig(name).mocap_path); return mc
walk, run, getup = clip("walk"), clip("run"), clip("getup_facedown")
N, T = 4096, 3000
for task in (0, 1):
    finals = []
    for rep in range(2):
        kw = dict(task=1, max_ep_length=2000) if task else {}
        eng = L.HipEngine(model, N, seed=99, **kw)
        eng.load_clip(0, walk)
        if task:
            eng.load_clip(1, run); eng.load_clip(2, getup, floor=True, acyclic=True)
        out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
        eng.reset(out["obs"])
        dones = 0; rsum = 0.0; bad = 0; reasons = torch.zeros(8, device=eng.device)
        t0 = time.perf_counter()
        for i in range(T):
            eng.fill_random_actions(act, i); eng.step(act, out)
            if i % 50 == 0:
                bad += int((~torch.isfinite(out["obs"])).sum()) + int((~torch.isfinite(out["rew"])).sum())
                dones += int(out["done"].sum()); rsum += float(out["rew"].mean())
                reasons += torch.bincount(out["reason"].clamp(0, 7), minlength=8).float()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        q = eng.get_state()[0]
        finals.append(q.clone())
        print("task %d rep %d: %.1f M env-steps/s, non-finite %d, done rate %.4f, mean reward %.4f, reasons %s"
              % (task, rep, N * T / dt / 1e6, bad, dones / (N * (T // 50)), rsum / (T // 50), reasons.int().tolist()))
        assert bad == 0 and bool(torch.isfinite(q).all())
        eng.close()
    assert torch.equal(finals[0], finals[1]), "repeat differs"
print("soak ok")
