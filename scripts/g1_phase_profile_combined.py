"""Per-phase cycle stamps of g1_step_kernel on the DPCombinedEnv task (-DG1_PROFILE build, libdeepmimic_hip_g1prof.so)."""
import sys
import torch
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", "libdeepmimic_hip_g1prof.so")
from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, NACT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
venv = HipG1CombinedVecEnv(n, seed=3)
venv.reset_tensor()
eng = venv.engine
dbg = eng.enable_debug()
g = torch.Generator(device=eng.device).manual_seed(0)
for t in range(40):
    venv.step_tensor((torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1) * scale)
torch.cuda.synchronize()
full = dbg.cpu()
d = full[:, 900:916].double()
names = ["kinematics+com", "crb+factor", "smooth dynamics", "broadphase", "analytic pairs", "plane-mesh pairs", "MPR pairs", "contact bookkeeping",
         "rows (J, R, aref)", "A = J M^-1 J^T", "J^T f, M^-1, qacc", "b, warm start, A f", "PGS sweeps", "(support-pair calls)", "task layer + RK", "between evaluations"]
tt = d.sum(1) - d[:, 13]
print("kernel ms", eng.last_kernel_ms(), "mean ticks per env-step %.2fM, max %.2fM, sum / 2048 slots %.2fM" % (tt.mean() / 1e6, tt.max() / 1e6, tt.sum() / 2048 / 1e6))
for i, nm in enumerate(names):
    if d[:, i].mean() > 0:
        print("%-22s %9.0f  %5.1f %%" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tt.mean()))
nefc, ncon = full[:, 204], full[:, 203]
for lo, hi in ((0, 48), (48, 64), (64, 80), (80, 128), (128, 257)):
    m = (nefc >= lo) & (nefc < hi)
    if m.any():
        print("last-stage rows %3d..%3d: %5d envs, contacts %.1f, mean ticks %.2fM, PGS %.2fM, MPR %.2fM, A %.2fM, rows %.2fM" % (lo, hi, int(m.sum()), float(ncon[m].mean()), tt[m].mean() / 1e6, d[m, 12].mean() / 1e6, d[m, 6].mean() / 1e6, d[m, 9].mean() / 1e6, d[m, 8].mean() / 1e6))
