"""Which envs set the launch's critical path: phase mix of the slowest waves vs the average (DM_PROFILE build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_prof.so")
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
eng = L.HipEngine(model, N); eng.load_clip(0, mc)
out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
dbg = eng.enable_debug()
names = ["pre", "kin+cinert", "crb+M", "factor", "vel+solve", "collide", "rows", "A", "PGS", "finish", "task", "post"]
acc_top = np.zeros(12); acc_all = np.zeros(12); cnt = 0; done_top = 0; nefc_top = []; it_top = []
for i in range(140):
    eng.fill_random_actions(act, i); eng.step(act, out)
    if i >= 100 and i % 4 == 0:
        torch.cuda.synchronize()
        d = dbg.cpu().numpy(); p = d[:, 352:364]; tot = p.sum(1)
        top = np.argsort(-tot)[:40]
        acc_top += p[top].mean(0); acc_all += p.mean(0); cnt += 1
        done_top += out["done"].cpu().numpy()[top].mean(); nefc_top.append(d[top, 243].mean()); it_top.append(d[top, 244].mean())
print("slot        " + " ".join("%9s" % n for n in names))
print("all  (mean) " + " ".join("%9.0f" % v for v in acc_all / cnt), " total %.0f" % (acc_all.sum() / cnt))
print("top-40 mean " + " ".join("%9.0f" % v for v in acc_top / cnt), " total %.0f" % (acc_top.sum() / cnt))
print("top-40: done fraction %.2f, last-stage nefc %.1f, last-stage sweeps %.1f" % (done_top / cnt, np.mean(nefc_top), np.mean(it_top)))
