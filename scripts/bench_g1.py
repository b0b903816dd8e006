"""Unitree G1 engine: env-steps/s of dmg1_step at N envs, random actions, auto-reset on the walk clip (auxiliary figure)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from deepmimic_mujoco_amd.config import MotionConfig  # noqa: E402
from deepmimic_mujoco_amd.g1 import G1HipEngine, NACT  # noqa: E402
from deepmimic_mujoco_amd.mocap import MocapDM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
motion = sys.argv[3] if len(sys.argv) > 3 else "walk"
pipeline = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 0 auto, 1 monolithic kernel, 2 split pipeline
mc = MocapDM(robot="unitree_g1")
mc.load_mocap(MotionConfig(motion, robot="unitree_g1").mocap_path)
eng = G1HipEngine(n, auto_reset=True, seed=3, pipeline=pipeline)
eng.load_clip(mc, floor="getup" in motion, acyclic="getup" in motion)
out = eng.alloc_outputs()
eng.reset(out["obs"])
g = torch.Generator(device=eng.device).manual_seed(0)
acts = [torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1 for _ in range(8)]
for t in range(10):
    eng.step(acts[t % 8], out)
torch.cuda.synchronize()
t0 = time.time()
ks, dn = [], 0.0
for t in range(steps):
    eng.step(acts[t % 8], out)
    if t % 10 == 9:
        ks.append(eng.last_kernel_ms())
        dn += float(out["done"].float().mean())
torch.cuda.synchronize()
dt = time.time() - t0
print({"envs": n, "motion": motion, "pipeline": pipeline, "env_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3, "kernel_ms": sum(ks) / len(ks),
       "done_fraction": dn / len(ks), "mean_reward": float(out["rew"].mean())})
if pipeline != 1:
    print("pair tickets per round (support-query, analytic, pulled):", eng.queue_counters())
