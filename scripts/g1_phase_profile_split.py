"""Per-phase cycle stamps of g1_env_kernel (split pipeline; -DG1_PROFILE build, libdeepmimic_hip_g1prof.so), summed over the rounds
of a step.  usage: g1_phase_profile_split.py [walk|combined] [n]"""
import sys
import torch
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", "libdeepmimic_hip_g1prof.so")
from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, HipG1VecEnv, NACT
what = sys.argv[1] if len(sys.argv) > 1 else "walk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
venv = HipG1CombinedVecEnv(n, seed=3) if what == "combined" else HipG1VecEnv(n, motion=what, seed=3)
scale = 0.25 if what == "combined" else 1.0
venv.reset_tensor()
eng = venv.engine
dbg = eng.enable_debug()
g = torch.Generator(device=eng.device).manual_seed(0)
for t in range(40):
    venv.step_tensor((torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1) * scale)
torch.cuda.synchronize()
d = dbg.cpu()[:, 900:916].double()
names = ["kinematics+com", "crb+factor", "smooth dynamics", "broadphase", "tickets + staged geoms", "-", "-", "contact bookkeeping",
         "rows (J, R, aref)", "A = J M^-1 J^T", "J^T f, M^-1, qacc", "b, warm start, A f", "PGS sweeps", "gather contacts", "task layer + RK", "between evaluations"]
tt = d.sum(1)
print("%s, %d envs, split=%s: kernel ms %.3f, mean ticks per env-step in g1_env_kernel %.2fM, max %.2fM" % (what, n, eng.split, eng.last_kernel_ms(), tt.mean() / 1e6, tt.max() / 1e6))
for i, nm in enumerate(names):
    if d[:, i].mean() > 0:
        print("%-24s %9.0f  %5.1f %%" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tt.mean()))
