"""Randomised teacher-forced parity sweep (more seeds / clips than the test suite); prints every contact-list
mismatch that is not at the activation margin."""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_gpu_parity as T
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.mocap import MocapDM
from oracle.oracle import OracleClip
model = load_model()
clips = {}
for name in ["walk", "run", "dance_b", "spinkick"]:
    mc = MocapDM(model=model); mc.load_mocap(MotionConfig(name).mocap_path); clips[name] = mc
oc = {k: OracleClip(*v.tables()) for k, v in clips.items()}
worst = dict(qpos=0, qvel=0, obs=0, rew=0); hard = 0; flips = 0; total = 0
for seed in (int(x) for x in (sys.argv[1:] or range(100, 112))):
    for scale, motion in ((2.0, "walk"), (0.3, "run"), (1.0, "spinkick")):
        res = T._run_teacher_forced(model, clips, oc, torch, scale, seed, nenv=12, nsteps=50, motion=motion)
        ok = np.ones(len(res["qpos"]), bool)
        for m in res["contact_mismatch"]:
            ok[m[0]] = False
            hard += 0 if m[3] else 1
            if not m[3]:
                r = res["recs"][m[0]]
                print("HARD seed", seed, motion, scale, "gpu", m[1], "oracle", m[2], "oracle dists", [round(float(c[0]), 7) for c in r["contact"]])
        ok[res["stage_flips"]] = False
        flips += len(res["stage_flips"]); total += len(ok)
        for k in worst: worst[k] = max(worst[k], float(res[k][ok].max()))
print("states", total, "hard contact mismatches", hard, "stage flips", flips, "worst", worst)
