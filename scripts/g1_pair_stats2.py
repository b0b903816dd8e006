"""Pair-kernel statistics of the split G1 pipeline (G1_PAIRSTATS build: hipcc ... -DG1_PAIRSTATS -o libdeepmimic_hip_pstat.so):
support evaluations, convex tickets, hint-separated / contact / miss outcomes per env-step."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", "libdeepmimic_hip_pstat.so")
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.g1 import G1HipEngine, NACT
from deepmimic_mujoco_amd.mocap import MocapDM
n, steps = 4096, 20
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
eng = G1HipEngine(n, auto_reset=True, seed=3, pipeline=int(sys.argv[1]) if len(sys.argv) > 1 else 2)
eng.load_clip(mc); out = eng.alloc_outputs(); eng.reset(out["obs"])
g = torch.Generator(device=eng.device).manual_seed(0)
acts = [torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1 for _ in range(8)]
for t in range(10): eng.step(acts[t % 8], out)
L = _lib.load_library(); buf = (C.c_ulonglong * 8)()
L.dmg1_pairstats(buf, 1)
for t in range(steps): eng.step(acts[t % 8], out)
L.dmg1_pairstats(buf, 0)
k = n * steps
print({"supports_per_env_step": buf[0] / k, "convex_tickets_per_env_step": buf[1] / k, "hint_separated": buf[2] / k, "contacts": buf[3] / k, "misses": buf[4] / k,
       "supports_per_convex_ticket": buf[0] / max(1, buf[1])})
