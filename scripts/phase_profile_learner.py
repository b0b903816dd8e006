"""Per-phase s_memtime stamps of mlp_fwdbwd_kernel, workgroup (0, 0), per wave (diagnostic build:
hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -DMLP_PROFILE -shared
      -o deepmimic_mujoco_amd/libdeepmimic_hip_mlpprof.so deepmimic_mujoco_amd/csrc/dm_abi.hip)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_mlpprof.so")
from deepmimic_mujoco_amd.ppo import PPO, FusedMlpGrad

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ppo = PPO(None, net_arch=(256, 128), batch_size=B, device=dev, use_hip_graph=False)
g = torch.Generator(device="cpu").manual_seed(0)
obs, act = torch.randn(B, 67, generator=g).to(dev), torch.randn(B, 28, generator=g).to(dev)
adv, ret, lp = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev), (-40 + torch.randn(B, generator=g)).to(dev)
mg = FusedMlpGrad(ppo.policy, ppo.optimizer, B)
lib = L.load_library()
names = ["start", "obs->LDS", "barrier", "layer1", "barrier", "layer2", "barrier", "layer3", "barrier", "loss head", "barrier", "bwd3",
         "barrier", "bwd2(end)"]
acc = np.zeros((8, 14))
R = 20
for it in range(R + 3):
    mg(obs, act, adv, ret, lp, 0.2, 0.5, 0.0, True)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 128)()
    assert lib.dm_ppo_mlp_prof(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(8, 16)[:, :14]
    if it >= 3:
        acc += st - st[:, :1].min()
acc /= R
t0 = time.perf_counter()
for _ in range(200):
    mg(obs, act, adv, ret, lp, 0.2, 0.5, 0.0, True)
torch.cuda.synchronize()
print("dm_ppo_mlp_grad eager: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
print("ticks since the first wave's start (mean of %d launches), waves 0..7; phase = interval ending at the stamp" % R)
for i, n in enumerate(names):
    print("%-10s %s" % (n, " ".join("%8.0f" % acc[w, i] for w in range(8))))
print("\nper-phase duration, wave 0 / wave 7 / max over waves:")
for i in range(1, 14):
    d = acc[:, i] - acc[:, i - 1]
    print("%-10s %8.0f %8.0f %8.0f" % (names[i], d[0], d[7], d.max()))
