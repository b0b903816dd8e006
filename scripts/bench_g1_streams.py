"""Sub-batches of one G1 batch on separate HIP streams: the tail of one engine's g1_env_kernel (its heaviest envs) overlaps the
other engines' launches.  usage: bench_g1_streams.py [combined|walk] [n] [sub_batches ...]"""
import sys, time, torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, HipG1VecEnv, NACT
what = sys.argv[1] if len(sys.argv) > 1 else "combined"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
import os
WARM, STEPS = int(os.environ.get("WARM", 10)), int(os.environ.get("STEPS", 40))
for sb in [int(x) for x in sys.argv[3:]] or [1, 2, 4]:
    venv = HipG1CombinedVecEnv(n, seed=3, sub_batches=sb) if what == "combined" else HipG1VecEnv(n, motion=what, seed=3, sub_batches=sb)
    venv.reset_tensor()
    g = torch.Generator(device=venv.device).manual_seed(0)
    sc = 0.25 if what == "combined" else 1.0
    acts = [(torch.rand(n, NACT, device=venv.device, generator=g) * 2 - 1) * sc for _ in range(8)]
    for t in range(WARM): venv.step_tensor(acts[t % 8])
    torch.cuda.synchronize(); t0 = time.time()
    for t in range(STEPS): o = venv.step_tensor(acts[t % 8])
    torch.cuda.synchronize(); dt = time.time() - t0
    if getattr(venv, "_streams", None): print("streams", [hex(s_.cuda_stream) for s_ in venv._streams], "current", hex(torch.cuda.current_stream().cuda_stream))
    print({"task": what, "envs": n, "sub_batches": sb, "env_steps_per_s": n * STEPS / dt, "ms": dt / STEPS * 1e3, "window": (WARM, WARM + STEPS)}, flush=True)
    venv.close()
