import sys, time, torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, NACT
n = 4096
venv = HipG1CombinedVecEnv(n, seed=3)
venv.reset_tensor()
g = torch.Generator(device=venv.device).manual_seed(0)
acts = [(torch.rand(n, NACT, device=venv.device, generator=g) * 2 - 1) * 0.25 for _ in range(8)]
for t in range(10): venv.step_tensor(acts[t % 8])
torch.cuda.synchronize(); t0 = time.time()
for t in range(40): o = venv.step_tensor(acts[t % 8])
torch.cuda.synchronize(); dt = time.time() - t0
print({"combined_env_steps_per_s": n * 40 / dt, "ms": dt / 40 * 1e3, "kernel_ms": venv.engine.last_kernel_ms()})
try:
    q = venv.engine.queue_counters()
    print("tickets per round (support-query, analytic, pulled):", q)
    print("task mix:", torch.bincount(venv.engine.get_env_tasks().flatten().long()).tolist() if hasattr(venv.engine, "get_env_tasks") else "-")
except Exception as e:
    print("no queue counters:", e)
