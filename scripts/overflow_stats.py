"""Diagnostic: how often do the 32-contact / 64-row caps bind in the bench workloads?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepmimic_mujoco_amd._lib import HipEngine
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
for mode in ["random", "zero", "small"]:
    eng = HipEngine(model, N); eng.load_clip(0, mc)
    out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
    dbg = eng.enable_debug()
    nefc_hist = np.zeros(65); ov = 0; tot = 0; it = []
    for i in range(300):
        if mode == "random": eng.fill_random_actions(act, i)
        if mode == "small": eng.fill_random_actions(act, i); act *= 0.15
        eng.step(act, out)
        d = dbg.cpu().numpy()
        packs = d[:, 247:249].copy().view(np.int32)
        for k in range(4):
            ne = (packs[:, 1] >> (8 * k)) & 0xFF
            nefc_hist += np.bincount(ne, minlength=65)[:65]
        ov += int((d[:, 246] != 0).sum()); tot += N
        it.append(d[:, 244].mean())
    c = np.cumsum(nefc_hist) / nefc_hist.sum()
    print(mode, "overflow frac %.2e" % (ov / tot), "nefc mean %.1f p50 %d p90 %d p99 %d max %d" % (
        (nefc_hist * np.arange(65)).sum() / nefc_hist.sum(), np.searchsorted(c, .5), np.searchsorted(c, .9),
        np.searchsorted(c, .99), np.nonzero(nefc_hist)[0].max()), "mean last-stage sweeps %.1f" % np.mean(it))
    eng.close()
