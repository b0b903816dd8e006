"""Per-phase cycle stamps of g1_step_kernel (-DG1_PROFILE build, libdeepmimic_hip_g1prof.so): mean over envs of one batch step."""
import sys
import torch
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", "libdeepmimic_hip_g1prof.so")
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.g1 import G1HipEngine, NACT
from deepmimic_mujoco_amd.mocap import MocapDM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
eng = G1HipEngine(n, auto_reset=True, seed=3); eng.load_clip(mc); out = eng.alloc_outputs(); eng.reset(out["obs"])
dbg = eng.enable_debug()
g = torch.Generator(device=eng.device).manual_seed(0)
for t in range(30):
    eng.step(torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1, out)
torch.cuda.synchronize()
d = dbg.cpu()[:, 900:916].double()
names = ["kinematics+com", "crb+factor", "smooth dynamics", "broadphase", "analytic pairs", "plane-mesh pairs", "MPR pairs", "contact bookkeeping",
         "rows (J, R, aref)", "A = J M^-1 J^T", "J^T f, M^-1, qacc", "b, warm start, A f", "PGS sweeps", "(support-pair calls)", "task layer + RK", "between evaluations"]
tot = (d.sum(1) - d[:, 12] - d[:, 13]).mean()
print("kernel ms", eng.last_kernel_ms(), "mean stamped ticks per env-step %.0f (100 MHz ticks?)" % tot)
for i, nm in enumerate(names):
    if d[:, i].mean() > 0:
        print("%-22s %9.0f  %5.1f %%" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot))
tt = (d.sum(1) - d[:, 13])
print("per-env stamped ticks: mean %.2fM p50 %.2fM p90 %.2fM p99 %.2fM max %.2fM; sum / 2048 slots = %.2fM" % (tt.mean() / 1e6, tt.median() / 1e6, tt.quantile(0.9) / 1e6, tt.quantile(0.99) / 1e6, tt.max() / 1e6, tt.sum() / 2048 / 1e6))
full = dbg.cpu()
nefc = full[:, 204]
for lo, hi in ((0, 48), (48, 64), (64, 80), (80, 128), (128, 256)):
    m = (nefc >= lo) & (nefc < hi)
    if m.any():
        print("last-stage rows %3d..%3d: %5d envs, mean ticks %.2fM, PGS %.2fM, MPR %.2fM" % (lo, hi, int(m.sum()), tt[m].mean() / 1e6, d[m, 12].mean() / 1e6, d[m, 6].mean() / 1e6))
top = tt >= tt.quantile(0.99)
print("heaviest 1 %% of envs (%d): mean ticks %.2fM, rows %.0f, contacts %.0f" % (int(top.sum()), tt[top].mean() / 1e6, float(nefc[top].mean()), float(full[top, 203].mean())))
for i, nm in enumerate(names):
    if i != 13 and d[top, i].mean() > 0:
        print("   %-22s %9.0f  %5.1f %%" % (nm, d[top, i].mean(), 100 * d[top, i].mean() / tt[top].mean()))
