"""PPO loop on the Unitree G1 DPCombinedEnv (the reference's training setup, src/sb3_ppo.py:249-278): env-steps/s end to end.
python scripts/bench_g1_ppo.py [envs] [n_steps] [iterations] [sub_batches]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from deepmimic_mujoco_amd.combined_env import HipCombinedVecEnv  # noqa: E402
from deepmimic_mujoco_amd.ppo import PPO  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sb = int(sys.argv[4]) if len(sys.argv) > 4 else 1
env = HipCombinedVecEnv(n, seed=2, sub_batches=sb)
ppo = PPO(env, net_arch=(256, 128), n_steps=T, batch_size=4096, n_epochs=20, seed=1)
buf = ppo.collect_rollouts()
ppo.train(buf)
torch.cuda.synchronize()
t0 = time.time()
tr = 0.0
for _ in range(iters):
    t1 = time.time()
    buf = ppo.collect_rollouts()
    torch.cuda.synchronize()
    tr += time.time() - t1
    ppo.train(buf)
torch.cuda.synchronize()
dt = time.time() - t0
print({"envs": n, "sub_batches": sb, "n_steps": T, "loop_env_steps_per_s": n * T * iters / dt, "rollout_env_steps_per_s": n * T * iters / tr, "obs_dim": ppo.obs_dim,
       "act_dim": ppo.act_dim, "loss": ppo.stats.get("loss")})
