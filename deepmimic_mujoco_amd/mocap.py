"""Motion-clip loader: DeepMimic mocap JSON -> per-frame reference tables.

Host-side mirror of the reference's ``MocapDM`` (src/mujoco/mocap_v2.py:12-348)
with the same attribute names (``dt``, ``data_config``, ``data_vel``,
``data_body_xpos``, ``data_geom_xpos``) and getters (``get_qpos`` ... :338-348).
It is load-time code: the tables are uploaded once to HBM (dm_load_clip) and the
step() kernels only index them.

Restated pieces and their sources:
  * frame layout / joint order       src/mujoco/mocap_v2.py:56-77, mocap_util.py:5-16
  * y-up -> z-up alignment           src/mujoco/mocap_util.py:31-48
  * quaternion -> intrinsic-XYZ Euler src/transformations.py:1031-1098 ('rxyz', :1529)
  * Euler -> quaternion              src/transformations.py:1100-1154
  * "continuity" singularity fix     src/mujoco/mocap_v2.py:143-235
  * finite-difference velocities     src/mujoco/mocap_v2.py:274-289, 350-362
  * FK tables                        src/mujoco/mocap_v2.py:292-307 (own FK, model.py)
  * interpolation to the sim rate    src/mujoco/mocap_v2.py:309-336
pyquaternion's axis/angle convention (used by calc_rot_vel) is [EXT]: unit-normalise,
angle = wrap(2*atan2(|v|, w)) into (-pi, pi], axis = v/|v| (zero vector if |v| < 1e-17).
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

from . import model as _model

# src/mujoco/mocap_util.py:5-16
BODY_JOINTS = ["chest", "neck", "right_shoulder", "right_elbow", "left_shoulder", "left_elbow",
               "right_hip", "right_knee", "right_ankle", "left_hip", "left_knee", "left_ankle"]
BODY_JOINTS_IN_DP_ORDER = ["chest", "neck", "right_hip", "right_knee", "right_ankle",
                           "right_shoulder", "right_elbow", "left_hip", "left_knee", "left_ankle",
                           "left_shoulder", "left_elbow"]
DOF_DEF = {"root": 3, "chest": 3, "neck": 3, "right_shoulder": 3, "right_elbow": 1,
           "right_wrist": 0, "left_shoulder": 3, "left_elbow": 1, "left_wrist": 0, "right_hip": 3,
           "right_knee": 1, "right_ankle": 3, "left_hip": 3, "left_knee": 1, "left_ankle": 3}

_EPS = np.finfo(float).eps * 4.0
_S = math.sqrt(0.5)
# rotations of +90 deg / -90 deg about x as wxyz quaternions (mocap_util.py:33-38)
_Q_ALIGN_LEFT = np.array([_S, _S, 0.0, 0.0])
_Q_ALIGN_RIGHT = np.array([_S, -_S, 0.0, 0.0])


def _qmul(a, b):
    return _model.quat_mul(a, b)


def align_rotation(rot_wxyz):
    """mocap_util.py:31-40: q_left * q * q_right (wxyz in, wxyz out)."""
    return _qmul(_qmul(_Q_ALIGN_LEFT, np.asarray(rot_wxyz, float)), _Q_ALIGN_RIGHT)


def align_position(pos):
    """mocap_util.py:42-48: (x, y, z) -> (x, -z, y)."""
    x, y, z = pos
    return np.array([x, -z, y], float)


def euler_from_quaternion_rxyz(q_xyzw):
    """transformations.euler_from_quaternion(q, 'rxyz') (src/transformations.py:1031-1098,1174-1193)."""
    q = np.array(q_xyzw[:4], dtype=np.float64)
    nq = float(np.dot(q, q))
    if nq < _EPS:
        M = np.identity(3)
    else:
        q = q * math.sqrt(2.0 / nq)
        o = np.outer(q, q)
        M = np.array([
            [1.0 - o[1, 1] - o[2, 2], o[0, 1] - o[2, 3], o[0, 2] + o[1, 3]],
            [o[0, 1] + o[2, 3], 1.0 - o[0, 0] - o[2, 2], o[1, 2] - o[0, 3]],
            [o[0, 2] - o[1, 3], o[1, 2] + o[0, 3], 1.0 - o[0, 0] - o[1, 1]]])
    # 'rxyz' = (firstaxis 2, parity 1, repetition 0, frame 1) -> i, j, k = 2, 1, 0
    cy = math.sqrt(M[2, 2] * M[2, 2] + M[1, 2] * M[1, 2])
    if cy > _EPS:
        ax = math.atan2(M[0, 1], M[0, 0])
        ay = math.atan2(-M[0, 2], cy)
        az = math.atan2(M[1, 2], M[2, 2])
    else:
        ax = math.atan2(-M[1, 0], M[1, 1])
        ay = math.atan2(-M[0, 2], cy)
        az = 0.0
    ax, ay, az = -ax, -ay, -az
    return az, ay, ax


def quaternion_from_euler_rxyz(ex, ey, ez):
    """transformations.quaternion_from_euler(ex, ey, ez, 'rxyz') -> xyzw; accepts arrays."""
    cx, sx = np.cos(0.5 * ex), np.sin(0.5 * ex)
    cy, sy = np.cos(0.5 * ey), np.sin(0.5 * ey)
    cz, sz = np.cos(0.5 * ez), np.sin(0.5 * ez)
    x = cy * cz * sx + sy * sz * cx
    y = sy * cz * cx - cy * sz * sx
    z = cy * sz * cx + sy * cz * sx
    w = cy * cz * cx - sy * sz * sx
    return np.stack([x, y, z, w], axis=-1)


def calc_rot_vel(seg_0, seg_1, dura):
    """mocap_v2.py:350-362: body-frame angular velocity taking wxyz q0 to q1 in `dura`."""
    q0 = np.asarray(seg_0, float)
    q1 = np.asarray(seg_1, float)
    conj = np.array([q0[0], -q0[1], -q0[2], -q0[3]])
    qd = _qmul(conj, q1)
    n = np.linalg.norm(qd)
    if n > 0:
        qd = qd / n
    vn = np.linalg.norm(qd[1:])
    axis = np.zeros(3) if vn < 1e-17 else qd[1:] / vn
    theta = 2.0 * math.atan2(vn, qd[0])
    ang = ((theta + math.pi) % (2 * math.pi)) - math.pi
    if ang == -math.pi:
        ang = math.pi
    return list(ang / dura * axis)


# singularity-fix tables, mocap_v2.py:148-158
_BALL_JOINTS = ["left_shoulder", "right_shoulder", "left_hip", "right_hip"]
_EX_LIM = {"left_shoulder": (-0.50, 3.14), "right_shoulder": (-3.14, 0.50),
           "left_hip": (-1.2, 1.2), "right_hip": (-1.2, 1.2)}
_EY_LIM = {"left_shoulder": (-3.14, 0.70), "right_shoulder": (-3.14, 0.70),
           "left_hip": (-2.57, 1.57), "right_hip": (-2.57, 1.57)}
_EZ_LIM = {"left_shoulder": (-1.50, 1.50), "right_shoulder": (-1.50, 1.50),
           "left_hip": (-1.0, 1.0), "right_hip": (-1.0, 1.0)}


def _clip(x, lo, hi):
    return min(max(x, lo), hi)  # numpy.clip semantics, also when lo > hi


def _fix_singularity(joint, euler, prev, quat_xyzw, dt, vmx):
    """"continuity" mode of mocap_v2.py:196-222; returns (angles, fired)."""
    ex, ey, ez = euler
    exp, eyp, ezp = prev
    lim = [_EX_LIM[joint], _EY_LIM[joint], _EZ_LIM[joint]]
    mins = [max(lim[a][0], p - vmx * dt) for a, p in enumerate((exp, eyp, ezp))]
    maxs = [min(lim[a][1], p + vmx * dt) for a, p in enumerate((exp, eyp, ezp))]
    tgt = [_clip(v, mins[a], maxs[a]) for a, v in enumerate((ex, ey, ez))]
    if np.allclose([ex, ey, ez], tgt):
        return (ex, ey, ez), False
    cands = [np.array([tgt[a], (exp, eyp, ezp)[a]] + list(np.linspace(mins[a], maxs[a], 6)))
             for a in range(3)]
    gx, gy, gz = np.meshgrid(cands[0], cands[1], cands[2], indexing="ij")
    qn = quaternion_from_euler_rxyz(gx.ravel(), gy.ravel(), gz.ravel())
    e1 = np.linalg.norm(qn - quat_xyzw, axis=1)
    e2 = np.linalg.norm(-qn - quat_xyzw, axis=1)
    err = np.minimum(e1, e2) ** 2
    best = int(np.argmin(err))  # first minimum in ex -> ey -> ez loop order (strict <)
    return (float(gx.ravel()[best]), float(gy.ravel()[best]), float(gz.ravel()[best])), True


_G1 = []


def _g1_model():
    if not _G1:
        from . import mjcf as _mjcf
        _G1.append(_mjcf.compile_mjcf_general(os.path.join(_model.ASSET_DIR, "deepmimic_unitree_g1.xml"),
                                              hulls=_mjcf.load_g1_hulls()))
    return _G1[0]


class MocapDM:
    """Same public surface as the reference class (src/mujoco/mocap_v2.py:12)."""

    def __init__(self, robot="humanoid3d", model=None):
        if robot not in ("humanoid3d", "unitree_g1"):
            raise Exception("Unknown robot: %s" % robot)
        self.robot = robot
        self.model = model
        if robot == "unitree_g1" and model is None:
            # G1 clips are "direct_qpos" (44 wide); kinematic tables come from the general MJCF compiler (mjcf.py)
            self.model = _g1_model()
        self.dt = None
        self.loop = None
        self.data_config = None
        self.data_vel = None
        self.data_body_xpos = None
        self.data_geom_xpos = None
        self.singularity_fired = 0

    def get_length(self):
        return 0 if self.data_config is None else len(self.data_config)

    def load_mocap(self, filepath):
        self.read_raw_data(filepath)

    def read_raw_data(self, filepath, FIX_SINGULARITY_MODE="continuity"):
        if FIX_SINGULARITY_MODE != "continuity":
            raise NotImplementedError("only the default 'continuity' mode is restated")
        with open(filepath, "r") as fin:
            data = json.load(fin)
        motions = np.array(data["Frames"], dtype=np.float64)
        self.loop = data.get("Loop")
        self.dt = float(motions[0][0])
        self.motion_name = os.path.splitext(os.path.basename(filepath))[0]
        vmx = 5.0 if "getup" in filepath else 10.0
        if "Format" in data:
            # "direct_qpos" clips (mocap_v2.py:271-272, written by src/retarget.py:176-190): frames are
            # [dt, qpos...] already in MuJoCo order; velocities / FK tables / interpolation follow as usual.
            nq = _model.NQ if self.robot == "humanoid3d" else self.model.nq
            if data["Format"] != "direct_qpos" or motions.shape[1] != 1 + nq:
                raise NotImplementedError("unsupported mocap Format %r / width %d" % (data["Format"], motions.shape[1]))
            self.all_states = []
            self.singularity_fired = 0
            self.data_config = [row[1:].copy() for row in motions]
            self._finish_tables(self.data_config)
            return

        # per-frame aligned states (mocap_v2.py:56-77)
        all_states = []
        for fr in motions:
            st = {"root_pos": align_position(fr[1:4]), "root_rot": align_rotation(fr[4:8])}
            off = 8
            for jn in BODY_JOINTS_IN_DP_ORDER:
                if DOF_DEF[jn] == 1:
                    st[jn] = fr[off:off + 1].copy()
                    off += 1
                else:
                    st[jn] = align_rotation(fr[off:off + 4])
                    off += 4
            all_states.append(st)
        self.all_states = all_states

        # qpos assembly in MuJoCo joint order (mocap_v2.py:90-251)
        prev_fixed = {}
        self.singularity_fired = 0
        configs = []
        for k, st in enumerate(all_states):
            row = list(st["root_pos"]) + list(st["root_rot"])
            for jn in BODY_JOINTS:
                if DOF_DEF[jn] == 1:
                    row += list(st[jn])
                    continue
                qw = st[jn]
                quat = np.array([qw[1], qw[2], qw[3], qw[0]])
                eul = euler_from_quaternion_rxyz(quat)
                if jn in _BALL_JOINTS:
                    prev = eul if k == 0 else prev_fixed[jn]
                    eul, fired = _fix_singularity(jn, eul, prev, quat, self.dt, vmx)
                    self.singularity_fired += int(fired)
                    prev_fixed[jn] = eul
                row += list(eul)
            configs.append(np.array(row))
        self.data_config = configs
        self._finish_tables(configs)

    def _finish_tables(self, configs):
        # velocities (mocap_v2.py:274-289)
        vels = []
        for k in range(len(configs)):
            kp = max(k - 1, 0)
            p, n = configs[kp], configs[k]
            v_xyz = (n[:3] - p[:3]) / self.dt
            v_rot = calc_rot_vel(p[3:7], n[3:7], self.dt)
            v_rest = (n[7:] - p[7:]) / self.dt
            vels.append(np.concatenate([v_xyz, v_rot, v_rest]))
        self.data_vel = vels

        # FK tables (mocap_v2.py:292-307) with the build's own kinematics
        if self.robot == "humanoid3d":
            mdl, fk = self.model or _model.load_model(), _model.forward_kinematics
        else:
            from .mjcf import forward_kinematics_general as fk
            mdl = self.model
        self.data_body_xpos, self.data_geom_xpos = [], []
        for q in configs:
            kin = fk(mdl, q)
            self.data_body_xpos.append(kin["xpos"].copy())
            self.data_geom_xpos.append(kin["geom_xpos"].copy())

        # interpolate to the simulator rate (mocap_v2.py:309-336)
        target_dt = 0.01666
        ratio = self.dt / target_dt
        iratio = int(ratio)
        if abs(ratio - iratio) > 0.1:
            raise Exception("Invalid dt ratio, cannot interpolate mocap frames: %f" % ratio)
        if iratio > 1:
            nc, nv, nb, ng = [], [], [], []
            for ia in range(len(configs) - 1):
                ib = ia + 1
                for k in range(iratio):
                    B = k * 1.0 / iratio
                    A = 1.0 - B
                    nc.append(A * self.data_config[ia] + B * self.data_config[ib])
                    nv.append(A * self.data_vel[ia] + B * self.data_vel[ib])
                    nb.append(A * self.data_body_xpos[ia] + B * self.data_body_xpos[ib])
                    ng.append(A * self.data_geom_xpos[ia] + B * self.data_geom_xpos[ib])
            self.dt = target_dt
            self.data_config, self.data_vel = nc, nv
            self.data_body_xpos, self.data_geom_xpos = nb, ng

    # getters, mocap_v2.py:338-348
    def get_qpos(self, idx):
        return self.data_config[idx]

    def get_qvel(self, idx):
        return self.data_vel[idx]

    def get_geom_xpos(self, idx):
        return self.data_geom_xpos[idx]

    def get_body_xpos(self, idx):
        return self.data_body_xpos[idx]

    def tables(self):
        """(qpos[L,35], qvel[L,34], body_xpos[L,14,3], geom_xpos[L,16,3]) as float64 arrays."""
        return (np.array(self.data_config), np.array(self.data_vel),
                np.array(self.data_body_xpos), np.array(self.data_geom_xpos))
