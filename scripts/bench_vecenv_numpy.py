"""SB3 VecEnv surface with host buffers (numpy actions in, numpy obs / rewards / dones / infos out): the PCIe-inclusive rate
of HipDeepMimicVecEnv.step at 4096 envs, next to the device-resident step_tensor rate."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = HipDeepMimicVecEnv(N, motion="walk")
env.reset()
rng = np.random.default_rng(0)
acts = [rng.uniform(-2, 2, (N, 28)).astype(np.float32) for _ in range(8)]
for i in range(30):
    env.step(acts[i % 8])
t0 = time.perf_counter()
K = 300
for i in range(K):
    obs, rew, done, infos = env.step(acts[i % 8])
dt = time.perf_counter() - t0
print("numpy VecEnv.step: %.3f ms per step, %.2f M env-steps/s (PCIe-inclusive)" % (dt / K * 1e3, N * K / dt / 1e6))
dact = [torch.as_tensor(a, device=env.device) for a in acts]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    env.step_tensor(dact[i % 8])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("step_tensor      : %.3f ms per step, %.2f M env-steps/s (device-resident)" % (dt / K * 1e3, N * K / dt / 1e6))
