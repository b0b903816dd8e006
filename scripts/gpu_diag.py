"""Diagnostic (not a test): teacher-forced parity details for the worst envs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
from oracle.oracle import OracleClip
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_parity as T

model = load_model()
clips = {}
for n in ["walk"]:
    mc = MocapDM(model=model); mc.load_mocap(MotionConfig(n).mocap_path); clips[n] = mc
oclips = {k: OracleClip(*v.tables()) for k, v in clips.items()}
np.set_printoptions(precision=6, suppress=True, linewidth=200)
for scale, seed in [(0.3, 2), (2.0, 3)]:
    res = T._run_teacher_forced(model, clips, oclips, torch, scale, seed, nenv=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    order = np.argsort(-res["qpos"])
    print("=== scale", scale, "n", len(order), "qpos max", res["qpos"].max(), "frac>1e-5", (res["qpos"] > 1e-5).mean(),
          "contact mismatches", len(res["contact_mismatch"]))
    for i in order[:8]:
        r = res["recs"][i]
        con = r["contact"]
        print(" env", i, "qpos err %.2e qvel err %.2e" % (res["qpos"][i], res["qvel"][i]), "nefc o/g", r["nefc"], int(res["nefc_gpu"][i]),
              "pairs", [(int(c[13]), int(c[14]), round(c[0], 5)) for c in con], "done", r["done"])
    for m in res["contact_mismatch"][:5]:
        print(" mismatch", m)
