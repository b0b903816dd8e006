"""Soak of the G1 engine: N envs x T steps of random actions with auto-reset on both tasks; counts termination reasons, simulator
errors, contact / row overflows and checks every output stays finite."""
import sys
import time

import torch

sys.path.insert(0, ".")
from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, HipG1VecEnv, NACT  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
for name, make, scale in (("DPEnv walk", lambda: HipG1VecEnv(n, motion="walk", seed=5), 1.0), ("DPEnv run", lambda: HipG1VecEnv(n, motion="run", seed=6), 1.0),
                          ("DPEnv getup", lambda: HipG1VecEnv(n, motion="getup_facedown", seed=7), 0.3), ("DPCombinedEnv", lambda: HipG1CombinedVecEnv(n, seed=8), 0.3)):
    env = make()
    dbg = env.engine.enable_debug()
    env.reset_tensor()
    g = torch.Generator(device=env.device).manual_seed(1)
    reasons = torch.zeros(16, dtype=torch.long, device=env.device)
    ndone = 0
    over = 0
    t0 = time.time()
    for t in range(steps):
        out = env.step_tensor((torch.rand(n, NACT, device=env.device, generator=g) * 2 - 1) * scale)
        assert torch.isfinite(out["obs"]).all() and torch.isfinite(out["rew"]).all(), (name, t)
        d = out["done"].bool()
        ndone += int(d.sum())
        reasons += torch.bincount(out["reason"][d].long(), minlength=16)[:16]
        if t % 50 == 0:
            over += int((dbg[:, 207] != 0).sum())
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("%-14s %d envs x %d steps: %.0f env-steps/s, episodes %d, reasons %s, overflow flags (sampled) %d, max |obs| %.1f" %
          (name, n, steps, n * steps / dt, ndone, {i: int(c) for i, c in enumerate(reasons.tolist()) if c}, over, float(out["obs"].abs().max())))
    env.close()
