"""Optimizer step of the PPO learner alone (captured hipGraph, replayed): us per minibatch step.
usage: python scripts/bench_learner.py [H1,H2] [steps] [--no-fused-mlp] [--no-epoch-graph] [--mb=4096] [--bf16] [--no-fused-wide]
(--bf16: wide nets run the fused bf16 chain of dm_ppo_wide_grad; with --no-fused-wide the bf16 library-GEMM path)"""
import sys, time
import torch
import os
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
if os.environ.get("DM_LIB_VARIANT"):      # experiment builds: libdeepmimic_hip_<variant>.so next to the shipped library
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", "libdeepmimic_hip_%s.so" % os.environ["DM_LIB_VARIANT"])
from deepmimic_mujoco_amd.ppo import PPO

arch = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 and "," in sys.argv[1] else "256,128").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 640
dev = torch.device("cuda", 0)
MB = int([a for a in sys.argv if a.startswith("--mb=")][0][5:]) if any(a.startswith("--mb=") for a in sys.argv) else 4096
T, N = 32, MB
g = torch.Generator(device="cpu").manual_seed(0)
buf = dict(obs=torch.randn(T, N, 67, generator=g), act=torch.randn(T, N, 28, generator=g), adv=torch.randn(T, N, generator=g),
           ret=torch.randn(T, N, generator=g), logp=-40 + torch.randn(T, N, generator=g))
buf = {k: v.to(dev) for k, v in buf.items()}
epochs = max(1, steps // T)
ppo = PPO(None, net_arch=arch, n_epochs=epochs, batch_size=MB, device=dev, fused_mlp="--no-fused-mlp" not in sys.argv,
          epoch_graph="--no-epoch-graph" not in sys.argv, mlp_dtype=torch.bfloat16 if "--bf16" in sys.argv else torch.float32, fused_wide="--no-fused-wide" not in sys.argv)
ppo.train(buf)                       # capture
torch.cuda.synchronize()
t0 = time.perf_counter()
ppo.train(buf)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("arch %s%s: %.1f us per optimizer step (%d steps, PPO.train of %d epochs x %d minibatches), loss %.5f" % (
    arch, " bf16 fused-wide" if getattr(ppo, "_wide_ok", False) else (" bf16 library" if "--bf16" in sys.argv else ""), dt / (epochs * T) * 1e6, epochs * T, epochs, T, ppo.stats["loss"]))
