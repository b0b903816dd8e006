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

# the model struct and its builder are the product's (the HIP engine consumes the same struct): the oracle checks the layout
from deepmimic_mujoco_amd.g1 import (DmModelG1, NQ, NV, NU, NBODY, NGEOM, NJNT, NM, MAXPAIR, NOBS, NMESH, NMESHVERT,  # noqa: F401,E402
                                     NREWJ, NEE, REW_QPOS, REW_QVEL, to_cstruct, load_g1_model)


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


def g1_model():
    """(GModel, DmModelG1) of the packaged G1 asset, cached."""
    return load_g1_model()


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

    def set_caps(self, maxcon, maxrow):
        """contact / row capacity (defaults: nconmax 200 of the XML, njmax 500); the HIP engine keeps 48 / 256."""
        if self.L.dmo_set_caps(self.d, int(maxcon), int(maxrow)) != 0:
            raise ValueError("caps out of range")

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


NOBS_COMBINED = 98


class _CombEnv(C.Structure):
    _fields_ = [("motion", C.c_int32), ("n_steps", C.c_int32), ("episode_length", C.c_int32), ("pad", C.c_int32),
                ("episode_reward", C.c_double)]


class G1CombSim(G1Sim):
    """One fp64 DPCombinedEnv (src/combined_env.py) as the reference runs it: Unitree G1, clips = (walk, run,
    getup_facedown_towalk); motion ids 0 walk, 1 run, 2 getup, 3 to_getup."""

    def __init__(self, clips):
        super().__init__()
        L = self.L
        L.dmo_combined_obs.argtypes = [C.c_void_p] * 5
        L.dmo_combined_step.argtypes = [C.c_void_p] * 11
        L.dmo_combined_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.clips = clips
        self.carr = (_Clip * 3)(*[c.c for c in clips])
        self.cenv = _CombEnv(0, 0, 0, 0, 0.0)

    def comb_reset(self, motion, n_steps):
        obs = np.zeros(NOBS_COMBINED)
        err = self.L.dmo_combined_reset(C.byref(self.cm), self.d, C.byref(self.cenv), self.carr, motion, n_steps, _p(obs))
        return obs, err

    def comb_step(self, action, force_state=None):
        act = np.zeros(NU)
        act[:len(action)] = action
        obs, terms = np.zeros(NOBS_COMBINED), np.zeros(8)
        rew, reason = C.c_double(0), C.c_int32(0)
        pq = pv = None
        if force_state is not None:
            fq = np.ascontiguousarray(force_state[0], np.float64)
            fv = np.ascontiguousarray(force_state[1], np.float64)
            pq, pv = _p(fq), _p(fv)
        done = self.L.dmo_combined_step(C.byref(self.cm), self.d, C.byref(self.cenv), self.carr, _p(act), pq, pv, _p(obs),
                                        C.byref(rew), _p(terms), C.byref(reason))
        return obs, rew.value, bool(done), terms, reason.value
