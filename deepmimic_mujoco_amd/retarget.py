"""Offline humanoid3d -> Unitree G1 motion retargeting.

Host-side mirror of the reference's ``src/retarget.py:42-194`` (``retarget_motion_humanoid_to_unitree_g1``): every frame of a
loaded humanoid3d clip (``MocapDM.data_config``, after interpolation to the simulation rate) is mapped joint by joint onto
the 44-number qpos of ``deepmimic_unitree_g1.xml`` and written as a ``"Format": "direct_qpos"`` clip
(``src/mujoco/mocap_v2.py:271-272`` reads those; so does ``deepmimic_mujoco_amd.mocap.MocapDM``).  The reference needs two
MuJoCo models for this only to look up joint names, qpos addresses and joint ranges; here the humanoid side comes from
``model.py`` and the G1 side from ``g1_joint_table`` (a kinematic-tree read of the G1 MJCF, no physics).

The reference's own outputs (``src/mujoco/motions/unitree_g1_{run,walk,getup_facedown}.txt``) are the golden vectors:
``tests/test_retarget.py`` regenerates all three from the humanoid3d clips to 1e-9: `run` and `walk` hold the naive mapping
(written before the tool's shoulder block existed), `getup_facedown` was written by the tool as it is now, shoulder block
included.  That file confirms the block's two quirks, which are therefore restated literally:
  * the shoulder angles fed to the Euler re-ordering are read from the G1 vector at the HUMANOID's qpos addresses
    (``g1qpos[humanoid.get_joint_qpos_addr(side + "_shoulder_x")]``, ``retarget.py:76-78``), i.e. from G1's right-hip triple
    (addresses 13-15) for the right arm and from (right_ankle_pitch, right_ankle_roll, torso) (17-19) for the left arm;
  * the candidate search scores G1 (y, x', z'') candidates with ``quaternion_from_euler(.., 'rxyz')`` (``:124``).
"""
from __future__ import annotations

import json
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from .config import MotionConfig, RobotConfig
from .mocap import MocapDM, quaternion_from_euler_rxyz
from .model import load_model

# humanoid3d joint -> (G1 joint, offset, scale) or None (dropped); retarget.py:5-37.  The root scales xyz by 0.85: the
# smaller robot's feet must reach the floor and its strides are shorter.
ROOT_SCALE = np.array([0.85, 0.85, 0.85, 1.0, 1.0, 1.0, 1.0])
JOINT_MAP = {
    "root": ("floating_base_joint", 0.0, ROOT_SCALE),
    "chest_z": ("torso_joint", 0.0, 1.0),
    "right_elbow": ("right_elbow_pitch_joint", 1.57, -1.0),
    "left_elbow": ("left_elbow_pitch_joint", 1.57, -1.0),
    "right_knee": ("right_knee_joint", 0.0, -1.0),
    "left_knee": ("left_knee_joint", 0.0, -1.0),
}
for _side in ("right", "left"):
    for _h, _g in (("shoulder_x", "shoulder_roll"), ("shoulder_y", "shoulder_pitch"), ("shoulder_z", "shoulder_yaw"),
                   ("hip_x", "hip_roll"), ("hip_y", "hip_pitch"), ("hip_z", "hip_yaw"),
                   ("ankle_x", "ankle_roll"), ("ankle_y", "ankle_pitch")):
        JOINT_MAP["%s_%s" % (_side, _h)] = ("%s_%s_joint" % (_side, _g), 0.0, 1.0)
DROPPED = ["chest_x", "chest_y", "neck_x", "neck_y", "neck_z", "right_ankle_z", "left_ankle_z"]

VMX_SHOULDER = 15.0        # rad/s continuity window of the shoulder search (retarget.py:96)
_EPS = np.finfo(float).eps * 4.0


def g1_joint_table(xml_path=None):
    """Joint names (document order = MuJoCo joint order), qpos addresses and ranges of an MJCF file."""
    if xml_path is None:
        xml_path = RobotConfig("unitree_g1").xml_path
    root = ET.parse(xml_path).getroot()
    names, adr, rng = [], {}, {}
    q = 0

    def walk(body):
        nonlocal q
        for e in body:
            if e.tag in ("joint", "freejoint"):
                free = e.tag == "freejoint" or e.get("type") == "free"
                n = e.get("name")
                names.append(n)
                adr[n] = (q, q + 7) if free else q
                rng[n] = tuple(float(t) for t in e.get("range", "0 0").split())
                q += 7 if free else 1
        for e in body:
            if e.tag == "body":
                walk(e)
    walk(root.find("worldbody"))
    return dict(names=names, qpos_addr=adr, range=rng, nq=q)


def _span(addr):
    return addr if isinstance(addr, tuple) else (addr, addr + 1)


def euler_matrix_rxyz(ai, aj, ak):
    """R = Rx(ai) Ry(aj) Rz(ak): transformations.euler_matrix(ai, aj, ak, 'rxyz')[:3, :3] (src/transformations.py:968-1029)."""
    # 'rxyz' = (firstaxis z, odd parity, no repetition, rotating frame): swap first / last angle, negate all three
    a, b, c = -ak, -aj, -ai
    si, sj, sk = math.sin(a), math.sin(b), math.sin(c)
    ci, cj, ck = math.cos(a), math.cos(b), math.cos(c)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    M = np.identity(3)
    i, j, k = 2, 1, 0
    M[i, i] = cj * ck
    M[i, j] = sj * sc - cs
    M[i, k] = sj * cc + ss
    M[j, i] = cj * sk
    M[j, j] = sj * ss + cc
    M[j, k] = sj * cs - sc
    M[k, i] = -sj
    M[k, j] = cj * si
    M[k, k] = cj * ci
    return M


def euler_from_matrix_ryxz(M):
    """Angles (about y, then x', then z'') with M = Ry(a) Rx(b) Rz(c): transformations.euler_from_matrix(M, 'ryxz')
    (src/transformations.py:1031-1087: firstaxis z, even parity, rotating frame)."""
    cy = math.sqrt(M[2, 2] * M[2, 2] + M[0, 2] * M[0, 2])
    if cy > _EPS:
        ax = math.atan2(M[1, 0], M[1, 1])
        ay = math.atan2(-M[1, 2], cy)
        az = math.atan2(M[0, 2], M[2, 2])
    else:
        ax = math.atan2(-M[0, 1], M[0, 0])
        ay = math.atan2(-M[1, 2], cy)
        az = 0.0
    return az, ay, ax


def _quat_err(cand_xyzw, tgt_xyzw):
    return min(np.linalg.norm(cand_xyzw - tgt_xyzw), np.linalg.norm(-cand_xyzw - tgt_xyzw)) ** 2


def retarget_frames(motion, humanoid_mocap=None, g1=None, shoulder_euler_conversion=True):
    """The frame loop of retarget.py:50-171.  Returns (dt, frames [L x 45] = dt + 44 qpos, loop flag, G1 joint table).

    ``shoulder_euler_conversion=False`` stops after the naive joint-by-joint mapping (:55-73): that is what the
    reference's committed ``unitree_g1_{run,walk}.txt`` contain (they predate the shoulder block, which the current tool
    runs unconditionally, ``if True:`` at :75; ``unitree_g1_getup_facedown.txt`` was written with it)."""
    hmodel = load_model()
    if humanoid_mocap is None:
        humanoid_mocap = MocapDM(model=hmodel)
        humanoid_mocap.load_mocap(MotionConfig(motion).mocap_path)
    g1 = g1 or g1_joint_table()
    h_addr = {n: ((0, 7) if i == 0 else int(hmodel.jnt_qposadr[i])) for i, n in enumerate(hmodel.jnt_names)}
    dt = humanoid_mocap.dt
    lim = lambda side, part: g1["range"]["%s_shoulder_%s_joint" % (side, part)]
    prev = {}                                   # joint -> last accepted (x, y, z) of the singularity smoothing
    frames = []
    for hqpos in humanoid_mocap.data_config:
        hqpos = np.asarray(hqpos, float)
        g = np.zeros(g1["nq"])
        for hname in hmodel.jnt_names:          # naive joint-by-joint mapping (:55-73)
            m = JOINT_MAP.get(hname)
            if m is None:
                assert hname in DROPPED, hname
                continue
            gname, offset, scale = m
            if motion == "getup_facedown" and hname == "root":
                offset = np.array([0, 0, 0.17, 0, 0, 0, 0])            # lift the pelvis off the floor (:61-62)
            gs, ge = _span(g1["qpos_addr"][gname])
            hs, he = _span(h_addr[hname])
            g[gs:ge] = hqpos[hs:he] * scale + offset
        for side in (("left", "right") if shoulder_euler_conversion else ()):   # shoulders: x y' z'' (humanoid) -> y x' z'' (G1), limit-aware smoothing (:75-167)
            # quirk (module docstring): the G1 vector indexed with the humanoid's addresses
            hr, hp, hy = (g[h_addr["%s_shoulder_%s" % (side, ax)]] for ax in "xyz")
            exo, eyo, ezo = euler_from_matrix_ryxz(euler_matrix_rxyz(hr, hp, hy))
            tgt = quaternion_from_euler_rxyz(hr, hp, hy)
            jn = side + "_shoulder"
            exp, eyp, ezp = prev.get(jn, (exo, eyo, ezo))
            (xlo, xhi), (ylo, yhi), (zlo, zhi) = lim(side, "roll"), lim(side, "pitch"), lim(side, "yaw")
            w = VMX_SHOULDER * dt
            ex_min, ex_max = max(xlo, exp - w), min(xhi, exp + w)
            ey_min, ey_max = max(ylo, eyp - w), min(yhi, eyp + w)
            ez_min, ez_max = max(zlo, ezp - w), min(zhi, ezp + w)
            ex_t, ey_t, ez_t = np.clip(exo, ex_min, ex_max), np.clip(eyo, ey_min, ey_max), np.clip(ezo, ez_min, ez_max)
            if np.allclose([exo, eyo, ezo], [ex_t, ey_t, ez_t]):
                exn, eyn, ezn = exo, eyo, ezo
            else:                               # 8 x 8 x 8 candidates, first strict minimum in loop order (:119-131)
                best = np.inf
                for exc in [ex_t, exp] + list(np.linspace(ex_min, ex_max, 6)):
                    for eyc in [ey_t, eyp] + list(np.linspace(ey_min, ey_max, 6)):
                        for ezc in [ez_t, ezp] + list(np.linspace(ez_min, ez_max, 6)):
                            err = _quat_err(quaternion_from_euler_rxyz(exc, eyc, ezc), tgt)
                            if err < best:
                                best, exn, eyn, ezn = err, exc, eyc, ezc
            g1r, g1p, g1y = exn, eyn, ezn
            if motion == "getup_facedown":      # "hack to make the motion more natural for the robot" (:133-134)
                g1p = g1p - 0.4 + hqpos[h_addr["chest_y"]]
            prev[jn] = (exn, eyn, ezn)
            g[g1["qpos_addr"][side + "_shoulder_roll_joint"]] = g1r
            g[g1["qpos_addr"][side + "_shoulder_pitch_joint"]] = g1p
            g[g1["qpos_addr"][side + "_shoulder_yaw_joint"]] = g1y
        frames.append([dt] + [float(v) for v in g])
    return dt, frames, humanoid_mocap.loop, g1


def retarget_motion_humanoid_to_unitree_g1(motion, out_path=None, overwrite=False, shoulder_euler_conversion=True):
    """src/retarget.py:42: writes the retargeted clip as a direct_qpos JSON (refuses to overwrite, as the reference does)."""
    dt, frames, loop, g1 = retarget_frames(motion, shoulder_euler_conversion=shoulder_euler_conversion)
    names = g1["names"]
    doc = {"Format": "direct_qpos", "JointNames": names,
           "Labels": ["dt"] + [names[0] + p for p in ("_x", "_y", "_z", "_qw", "_qx", "_qy", "_qz")] + list(names[1:]),
           "Loop": loop, "Frames": frames}
    if out_path is None:
        out_path = MotionConfig(motion, "unitree_g1").mocap_path
    if os.path.exists(out_path) and not overwrite:
        raise FileExistsError("File exists: %s" % out_path)
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=4)
    return out_path


if __name__ == "__main__":
    import sys
    print(retarget_motion_humanoid_to_unitree_g1(sys.argv[1] if len(sys.argv) > 1 else "run",
                                                 out_path=sys.argv[2] if len(sys.argv) > 2 else None))
