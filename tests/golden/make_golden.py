"""Generate the golden fixtures under tests/golden/ — run in the BUILD container only.

The only pieces of the reference that import in this image (numpy-only) are
``src/transformations.py`` and ``src/extracted_policy.py``.  This
script imports them from /root/reference (never copied) and stores numeric
input/output vectors:

  transformations_golden.npz
      quat_xyzw [K,4]  -> euler_rxyz [K,3]  = euler_from_quaternion(q, 'rxyz')
      euler_in  [M,3]  -> quat_out   [M,4]  = quaternion_from_euler(*e, 'rxyz')
    K covers every aligned ball-joint quaternion of the four clips (walk, run,
    dance_b, spinkick) plus random / near-singular / un-normalised quaternions.
  policy_kat.npz
      ExtractedPolicy weights W0,B0,W2,B2,WA,BA (trained 66->256->128->28 tanh MLP),
      the reference's own known-answer pair (extracted_policy.py:480-485) and
      extra (obs -> action) pairs produced by ExtractedPolicy.act.
"""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    tr = _load("transformations")
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.mocap import MocapDM, BODY_JOINTS, DOF_DEF

    rng = np.random.default_rng(20240813)
    quats = []
    for motion in ["walk", "run", "dance_b", "spinkick"]:
        mc = MocapDM()
        mc.load_mocap(MotionConfig(motion).mocap_path)
        for st in mc.all_states:
            for jn in BODY_JOINTS:
                if DOF_DEF[jn] == 3:
                    w, x, y, z = st[jn]
                    quats.append([x, y, z, w])
    rnd = rng.normal(size=(300, 4))
    quats += list(rnd / np.linalg.norm(rnd, axis=1, keepdims=True))
    quats += list(rng.normal(size=(50, 4)) * rng.uniform(0.2, 3.0, size=(50, 1)))  # un-normalised
    for ey in [np.pi / 2, -np.pi / 2, np.pi / 2 - 1e-9, -np.pi / 2 + 1e-7]:     # gimbal lock
        for ex, ez in [(0.3, -0.2), (-1.0, 2.0), (0.0, 0.0)]:
            quats.append(tr.quaternion_from_euler(ex, ey, ez, "rxyz"))
    quats = np.array(quats, np.float64)
    eul = np.array([tr.euler_from_quaternion(q, "rxyz") for q in quats])
    euler_in = rng.uniform(-np.pi, np.pi, size=(400, 3))
    quat_out = np.array([tr.quaternion_from_euler(e[0], e[1], e[2], "rxyz") for e in euler_in])
    np.savez_compressed(os.path.join(HERE, "transformations_golden.npz"),
                        quat_xyzw=quats, euler_rxyz=eul, euler_in=euler_in, quat_out=quat_out)
    print("transformations:", quats.shape, euler_in.shape)

    ep = _load("extracted_policy")
    pol = ep.ExtractedPolicy()
    pol.test()
    # the reference's own KAT vectors live as literals inside test(); recover them by
    # evaluating act() on the same obs is circular, so parse them out of the source text
    import re
    src = open(os.path.join(REF, "extracted_policy.py")).read()
    body = src[src.index("def test(self):"):]
    arrs = re.findall(r"np\.array\(\[\[(.*?)\]\]\)", body, flags=re.S)
    kat_obs = np.array([float(t) for t in arrs[0].replace("\n", " ").split(",")])[None, :]
    kat_exp = np.array([float(t) for t in arrs[1].replace("\n", " ").split(",")])[None, :]
    assert kat_obs.shape == (1, 66) and kat_exp.shape == (1, 28)
    extra_obs = rng.uniform(-1.0, 1.0, size=(16, 66))
    extra_act = pol.act(extra_obs)
    np.savez_compressed(os.path.join(HERE, "policy_kat.npz"),
                        W0=pol.W0.astype(np.float32), B0=pol.B0.astype(np.float32),
                        W2=pol.W2.astype(np.float32), B2=pol.B2.astype(np.float32),
                        WA=pol.WA.astype(np.float32), BA=pol.BA.astype(np.float32),
                        kat_obs=kat_obs, kat_expected=kat_exp,
                        extra_obs=extra_obs, extra_act=extra_act)
    print("policy:", pol.W0.shape, pol.W2.shape, pol.WA.shape, "KAT ok")

    # retarget tool (src/retarget.py): the reference's own outputs for the three humanoid3d clips it was run on are
    # the golden vectors (numeric frames + joint names); plus Euler helper vectors for the axis orders the tool uses
    import json
    out = {}
    for motion in ("run", "walk", "getup_facedown"):
        d = json.load(open(os.path.join(REF, "mujoco", "motions", "unitree_g1_%s.txt" % motion)))
        assert d["Format"] == "direct_qpos"
        out[motion + "_frames"] = np.array(d["Frames"], np.float64)
        out[motion + "_loop"] = np.array(d["Loop"])
        out["joint_names"] = np.array(d["JointNames"])
        out["labels"] = np.array(d["Labels"])
    eul = rng.uniform(-np.pi, np.pi, size=(300, 3))
    eul[:20, 1] = np.pi / 2 * np.sign(eul[:20, 1])                         # gimbal lock rows
    mats = np.array([tr.euler_matrix(e[0], e[1], e[2], "rxyz")[:3, :3] for e in eul])
    yxz = np.array([tr.euler_from_matrix(m, "ryxz") for m in mats])
    np.savez_compressed(os.path.join(HERE, "retarget_golden.npz"), euler_rxyz_in=eul, matrix_rxyz=mats, euler_ryxz_out=yxz, **out)
    print("retarget:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
