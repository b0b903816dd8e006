"""ctypes wrapper around oracle/libdm_oracle_g1.so — TEST INFRASTRUCTURE ONLY (Unitree G1 build of the fp64 oracle).

dm_oracle.c compiled with -DDM_ROBOT_G1: the same restatement of mj_step at the G1 dimensions (nq 44, nv 43, 94 geoms),
plus what the G1 asset needs beyond humanoid3d — convex narrowphase (libccd MPR restated, oracle/dm_convex.h), plane-cylinder,
plane-mesh, friction-loss rows — and the G1 branch of DPEnv (src/deepmimic_env.py:204-211,244-246,348-351,426-433).
PHYSICS PARITY UNPINNED, like the humanoid3d oracle.  Only tests/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NQ, NV, NU, NBODY, NGEOM, NJNT, NM, MAXPAIR, NOBS = 44, 43, 37, 39, 94, 38, 434, 1024, 85
NMESH, NMESHVERT, NREWJ, NEE = 32, 40000, 23, 4
# src/deepmimic_env.py:206-207
REW_QPOS = [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 32, 33, 34, 35, 36]
REW_QVEL = [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 31, 32, 33, 34, 35]

_i32, _f64 = C.c_int32, C.c_double


class DmModelG1(C.Structure):
    """ctypes mirror of ``struct DmModel`` under -DDM_ROBOT_G1 (include/dm_model.h) — keep in sync (size checked)."""

    _fields_ = [
        ("nq", _i32), ("nv", _i32), ("nu", _i32), ("nbody", _i32), ("ngeom", _i32), ("njnt", _i32), ("npair", _i32),
        ("nM", _i32), ("integrator", _i32), ("iterations", _i32), ("pad0", _i32), ("pad1", _i32),
        ("timestep", _f64), ("tolerance", _f64), ("gravity", _f64 * 3), ("meaninertia", _f64), ("solref", _f64 * 2),
        ("solimp", _f64 * 5), ("qpos0", _f64 * NQ),
        ("body_parent", _i32 * NBODY), ("body_jntadr", _i32 * NBODY), ("body_jntnum", _i32 * NBODY),
        ("body_dofadr", _i32 * NBODY), ("body_dofnum", _i32 * NBODY), ("body_depth", _i32 * NBODY),
        ("body_pos", _f64 * 3 * NBODY), ("body_quat", _f64 * 4 * NBODY), ("body_ipos", _f64 * 3 * NBODY),
        ("body_inertia", _f64 * 6 * NBODY), ("body_mass", _f64 * NBODY), ("body_invweight0", _f64 * 2 * NBODY),
        ("jnt_type", _i32 * NJNT), ("jnt_body", _i32 * NJNT), ("jnt_qposadr", _i32 * NJNT), ("jnt_dofadr", _i32 * NJNT),
        ("jnt_limited", _i32 * NJNT), ("jnt_pos", _f64 * 3 * NJNT), ("jnt_axis", _f64 * 3 * NJNT),
        ("jnt_range", _f64 * 2 * NJNT),
        ("dof_body", _i32 * NV), ("dof_jnt", _i32 * NV), ("dof_parent", _i32 * NV), ("dof_Madr", _i32 * NV),
        ("dof_armature", _f64 * NV), ("dof_damping", _f64 * NV), ("dof_invweight0", _f64 * NV),
        ("geom_type", _i32 * NGEOM), ("geom_body", _i32 * NGEOM), ("geom_condim", _i32 * NGEOM),
        ("geom_pos", _f64 * 3 * NGEOM), ("geom_quat", _f64 * 4 * NGEOM), ("geom_size", _f64 * 3 * NGEOM),
        ("geom_friction", _f64 * 3 * NGEOM), ("geom_margin", _f64 * NGEOM), ("geom_rbound", _f64 * NGEOM),
        ("act_dof", _i32 * NU), ("act_gear", _f64 * NU), ("act_ctrlrange", _f64 * 2 * NU),
        ("pair_geom1", _i32 * MAXPAIR), ("pair_geom2", _i32 * MAXPAIR),
        ("ee_geom", _i32 * NEE), ("torso_body", _i32), ("rfoot_geom", _i32), ("lfoot_geom", _i32), ("floor_geom", _i32),
        # trailing G1 fields
        ("dof_frictionloss", _f64 * NV), ("geom_mesh", _i32 * NGEOM), ("mesh_vertadr", _i32 * NMESH),
        ("mesh_vertnum", _i32 * NMESH), ("mesh_center", _f64 * 3 * NMESH), ("mesh_vert", _f64 * 3 * NMESHVERT),
        ("nconmax", _i32), ("n_policy_action", _i32), ("action_scale", _f64), ("low_z", _f64),
        ("rew_qposadr", _i32 * NREWJ), ("rew_dofadr", _i32 * NREWJ), ("rew_jnt", _i32 * NREWJ), ("extra_geom", _i32 * 8),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdm_oracle_g1.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.dmo_data_new.restype = C.c_void_p
        L.dmo_data_new.argtypes = [C.c_void_p]
        L.dmo_data_free.argtypes = [C.c_void_p]
        L.dmo_data_reset.argtypes = [C.c_void_p, C.c_void_p]
        for f in (L.dmo_forward, L.dmo_step):
            f.argtypes = [C.c_void_p, C.c_void_p]
            f.restype = C.c_int
        L.dmo_set_state.argtypes = [C.c_void_p] * 4
        L.dmo_get_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dmo_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.dmo_reward.restype = C.c_double
        L.dmo_env_step.argtypes = [C.c_void_p] * 11
        L.dmo_env_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.dmo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.dmo_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.dmo_get_int.argtypes = [C.c_void_p, C.c_char_p]
        L.dmo_set_caps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dmo_set_tweak.argtypes = [C.c_char_p, C.c_double]
        assert L.dmo_model_sizeof() == C.sizeof(DmModelG1), (L.dmo_model_sizeof(), C.sizeof(DmModelG1))
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def to_cstruct(g) -> DmModelG1:
    """GModel (deepmimic_mujoco_amd.mjcf.compile_mjcf_general of the G1 asset, with hulls) -> DmModelG1."""
    from deepmimic_mujoco_amd.config import RobotConfig
    rc = RobotConfig("unitree_g1")
    assert (g.nq, g.nv, g.nu, g.nbody, g.ngeom, g.njnt, g.nM) == (NQ, NV, NU, NBODY, NGEOM, NJNT, NM)
    s = DmModelG1()
    s.nq, s.nv, s.nu, s.nbody, s.ngeom, s.njnt, s.npair, s.nM = NQ, NV, NU, NBODY, NGEOM, NJNT, g.npair, NM
    s.integrator = {"Euler": 0, "RK4": 1}[g.integrator]
    s.iterations, s.timestep, s.tolerance, s.meaninertia = g.iterations, g.timestep, g.tolerance, g.meaninertia
    assert g.solver == "PGS" and g.npair <= MAXPAIR

    def put(name, arr):
        dst = np.ctypeslib.as_array(getattr(s, name))
        dst[...] = np.ascontiguousarray(arr).reshape(dst.shape)

    for name in ["gravity", "solref", "solimp", "qpos0", "body_parent", "body_jntadr", "body_jntnum", "body_dofadr",
                 "body_dofnum", "body_depth", "body_pos", "body_quat", "body_ipos", "body_inertia", "body_mass",
                 "body_invweight0", "jnt_type", "jnt_body", "jnt_qposadr", "jnt_dofadr", "jnt_limited", "jnt_pos",
                 "jnt_axis", "jnt_range", "dof_body", "dof_jnt", "dof_parent", "dof_Madr", "dof_armature", "dof_damping",
                 "dof_invweight0", "geom_type", "geom_body", "geom_condim", "geom_pos", "geom_quat", "geom_size",
                 "geom_friction", "geom_margin", "geom_rbound", "act_dof", "act_gear", "act_ctrlrange", "dof_frictionloss"]:
        put(name, getattr(g, name))
    p1, p2 = np.full(MAXPAIR, -1, np.int32), np.full(MAXPAIR, -1, np.int32)
    p1[:g.npair], p2[:g.npair] = g.pairs[:, 0], g.pairs[:, 1]
    put("pair_geom1", p1)
    put("pair_geom2", p2)
    put("ee_geom", [g.geom_id(n) for n in rc.endeffector_geom_names])
    s.torso_body = g.body_id(rc.torso_body_name)
    s.rfoot_geom, s.lfoot_geom, s.floor_geom = (g.geom_id(rc.rfoot_geom_name), g.geom_id(rc.lfoot_geom_name),
                                                g.geom_id(rc.floor_geom_name))
    put("geom_mesh", g.geom_meshid)
    adr, num, verts = np.zeros(NMESH, np.int32), np.zeros(NMESH, np.int32), np.zeros((NMESHVERT, 3))
    cen = np.zeros((NMESH, 3))
    a = 0
    assert len(g.mesh_vert) <= NMESH
    for i, v in enumerate(g.mesh_vert):
        adr[i], num[i] = a, len(v)
        verts[a:a + len(v)] = v
        cen[i] = g.mesh_center[i]
        a += len(v)
    assert a <= NMESHVERT
    put("mesh_vertadr", adr)
    put("mesh_vertnum", num)
    put("mesh_vert", verts)
    put("mesh_center", cen)
    s.nconmax, s.n_policy_action, s.action_scale, s.low_z = g.nconmax, NU - 14, 20.0, rc.low_z
    put("rew_qposadr", REW_QPOS)
    put("rew_dofadr", REW_QVEL)
    put("rew_jnt", np.array(REW_QPOS) - 7 + 1)
    put("extra_geom", [g.geom_id(n) for n in rc.extra_contact_geom_names])
    return s


_MODEL = []


def g1_model():
    """(GModel, DmModelG1) of the packaged G1 asset, cached."""
    if not _MODEL:
        from deepmimic_mujoco_amd import mjcf, model as M
        g = mjcf.compile_mjcf_general(os.path.join(M.ASSET_DIR, "deepmimic_unitree_g1.xml"), hulls=mjcf.load_g1_hulls())
        _MODEL.append((g, to_cstruct(g)))
    return _MODEL[0]


class _Clip(C.Structure):
    _fields_ = [("L", C.c_int32), ("qpos", C.c_void_p), ("qvel", C.c_void_p), ("body_xpos", C.c_void_p),
                ("geom_xpos", C.c_void_p), ("flags", C.c_int32), ("pad", C.c_int32)]


class _Env(C.Structure):
    _fields_ = [("idx_curr", C.c_int32), ("episode_length", C.c_int32), ("episode_reward", C.c_double)]


class G1Clip:
    def __init__(self, qpos, qvel, body_xpos, geom_xpos, floor=False, acyclic=False, run_rule=False):
        self.qpos = np.ascontiguousarray(qpos, np.float64).reshape(-1, NQ)
        self.qvel = np.ascontiguousarray(qvel, np.float64).reshape(-1, NV)
        self.body_xpos = np.ascontiguousarray(body_xpos, np.float64).reshape(-1, NBODY, 3)
        self.geom_xpos = np.ascontiguousarray(geom_xpos, np.float64).reshape(-1, NGEOM, 3)
        self.L = len(self.qpos)
        self.c = _Clip(self.L, _p(self.qpos).value, _p(self.qvel).value, _p(self.body_xpos).value,
                       _p(self.geom_xpos).value, (1 if floor else 0) | (2 if acyclic else 0) | (4 if run_rule else 0), 0)


class G1Sim:
    """One fp64 G1 environment: physics state + the G1 branch of DPEnv."""

    def __init__(self, gmodel=None, cstruct=None):
        if gmodel is None:
            gmodel, cstruct = g1_model()
        self.g, self.cm = gmodel, cstruct
        self.L = lib()
        self.d = self.L.dmo_data_new(C.byref(self.cm))
        self.env = _Env(0, 0, 0.0)

    def __del__(self):
        try:
            self.L.dmo_data_free(self.d)
        except Exception:
            pass

    def get(self, name):
        buf = np.zeros(600 * 600, np.float64)
        n = self.L.dmo_get(self.d, name.encode(), _p(buf), buf.size)
        if n < 0:
            raise KeyError(name)
        return buf[:n].copy()

    def geti(self, name):
        return self.L.dmo_get_int(self.d, name.encode())

    def set(self, name, val):
        a = np.ascontiguousarray(val, np.float64)
        if self.L.dmo_set(self.d, name.encode(), _p(a), a.size) != 0:
            raise KeyError(name)

    def contacts(self):
        c = self.get("contact").reshape(-1, 17)
        return [dict(dist=r[0], pos=r[1:4], frame=r[4:13].reshape(3, 3), geom1=int(r[13]), geom2=int(r[14]), dim=int(r[15]))
                for r in c]

    def set_state(self, qpos, qvel):
        q, v = np.ascontiguousarray(qpos, np.float64), np.ascontiguousarray(qvel, np.float64)
        return self.L.dmo_set_state(C.byref(self.cm), self.d, _p(q), _p(v))

    def forward(self):
        return self.L.dmo_forward(C.byref(self.cm), self.d)

    def step(self, ctrl=None):
        if ctrl is not None:
            self.set("ctrl", ctrl)
        return self.L.dmo_step(C.byref(self.cm), self.d)

    def env_reset(self, clip, idx_init):
        obs = np.zeros(NOBS)
        err = self.L.dmo_env_reset(C.byref(self.cm), self.d, C.byref(self.env), C.byref(clip.c), int(idx_init), _p(obs))
        return obs, err

    def env_step(self, clip, action, force_state=None):
        """action: the policy's 23 values (hands are padded inside, src/deepmimic_env.py:348-351)."""
        act = np.zeros(NU)
        act[:len(action)] = action
        obs, terms = np.zeros(NOBS), np.zeros(5)
        rew, reason = C.c_double(0), C.c_int32(0)
        fq = fv = None
        if force_state is not None:
            fq = _p(np.ascontiguousarray(force_state[0], np.float64))
            fv = _p(np.ascontiguousarray(force_state[1], np.float64))
        done = self.L.dmo_env_step(C.byref(self.cm), self.d, C.byref(self.env), C.byref(clip.c), _p(act), fq, fv, _p(obs),
                                   C.byref(rew), _p(terms), C.byref(reason))
        return obs, rew.value, bool(done), terms, reason.value
