"""Pair statistics of the G1 narrowphase on the bench workload (last forward evaluation of a step, per env)."""
import sys
import torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.g1 import G1HipEngine, NACT
from deepmimic_mujoco_amd.mocap import MocapDM
n = 4096
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
eng = G1HipEngine(n, auto_reset=True, seed=3); eng.load_clip(mc); out = eng.alloc_outputs(); eng.reset(out["obs"])
dbg = eng.enable_debug()
g = torch.Generator(device=eng.device).manual_seed(0)
for t in range(40):
    eng.step(torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1, out)
torch.cuda.synchronize()
d = dbg.cpu()
for name, col in (("survivors after the box filter", 1008), ("analytic pairs", 1009), ("plane-mesh pairs", 1010), ("MPR pairs", 1011), ("contacts", 203), ("rows", 204), ("PGS sweeps", 205)):
    x = d[:, col]
    print("%-32s mean %.1f  p50 %.0f  p90 %.0f  max %.0f" % (name, x.mean(), x.median(), x.quantile(0.9), x.max()))
