"""ctypes wrapper around oracle/libdm_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (deepmimic_mujoco_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from deepmimic_mujoco_amd.model import NQ, NV, NBODY, NGEOM, NOBS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Clip(C.Structure):
    _fields_ = [("L", C.c_int32), ("qpos", C.c_void_p), ("qvel", C.c_void_p),
                ("body_xpos", C.c_void_p), ("geom_xpos", C.c_void_p), ("flags", C.c_int32), ("pad", C.c_int32)]


class _Env(C.Structure):
    _fields_ = [("idx_curr", C.c_int32), ("episode_length", C.c_int32),
                ("episode_reward", C.c_double)]


class _CombEnv(C.Structure):
    _fields_ = [("motion", C.c_int32), ("n_steps", C.c_int32), ("episode_length", C.c_int32), ("pad", C.c_int32),
                ("episode_reward", C.c_double)]


NOBS_COMBINED = 72


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libdm_oracle.so")
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
        L.dmo_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dmo_get_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dmo_reward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.dmo_reward.restype = C.c_double
        L.dmo_env_step.argtypes = [C.c_void_p] * 11
        L.dmo_env_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.dmo_combined_obs.argtypes = [C.c_void_p] * 5
        L.dmo_combined_step.argtypes = [C.c_void_p] * 11
        L.dmo_combined_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dmo_narrowphase.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_double, C.c_void_p]
        L.dmo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.dmo_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.dmo_get_int.argtypes = [C.c_void_p, C.c_char_p]
        L.dmo_set_caps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dmo_quat_to_rpy.argtypes = [C.c_void_p, C.c_void_p]
        L.dmo_bench_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64]
        L.dmo_bench_steps.restype = C.c_double
        L.dmo_set_tweak.argtypes = [C.c_char_p, C.c_double]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleClip:
    def __init__(self, qpos, qvel, body_xpos, geom_xpos, floor=False, acyclic=False):
        self.qpos = np.ascontiguousarray(qpos, np.float64).reshape(-1, NQ)
        self.qvel = np.ascontiguousarray(qvel, np.float64).reshape(-1, NV)
        self.body_xpos = np.ascontiguousarray(body_xpos, np.float64).reshape(-1, NBODY, 3)
        self.geom_xpos = np.ascontiguousarray(geom_xpos, np.float64).reshape(-1, NGEOM, 3)
        self.L = len(self.qpos)
        self.c = _Clip(self.L, _p(self.qpos).value, _p(self.qvel).value,
                       _p(self.body_xpos).value, _p(self.geom_xpos).value,
                       (1 if floor else 0) | (2 if acyclic else 0), 0)


class OracleSim:
    """One fp64 environment: physics state + DPEnv task state."""

    SHAPES = {"xpos": (NBODY, 3), "xquat": (NBODY, 4), "xmat": (NBODY, 9), "xipos": (NBODY, 3),
              "geom_xpos": (NGEOM, 3), "geom_xmat": (NGEOM, 9), "cvel": (NBODY, 6),
              "cdof": (NV, 6), "cdof_dot": (NV, 6), "cinert": (NBODY, 10)}

    def __init__(self, model):
        self.model = model
        self.cm = model.cstruct
        self.L = lib()
        self.d = self.L.dmo_data_new(C.byref(self.cm))
        self.env = _Env(0, 0, 0.0)

    def __del__(self):
        try:
            self.L.dmo_data_free(self.d)
        except Exception:
            pass

    def get(self, name):
        buf = np.zeros(500 * 500, np.float64)
        n = self.L.dmo_get(self.d, name.encode(), _p(buf), buf.size)
        if n < 0:
            raise KeyError(name)
        out = buf[:n].copy()
        if name in self.SHAPES:
            out = out.reshape(self.SHAPES[name])
        elif name == "efc_J":
            out = out.reshape(-1, NV)
        elif name == "efc_AR":
            k = self.nefc
            out = out.reshape(k, k)
        elif name == "contact":
            out = out.reshape(-1, 17)
        return out

    def set(self, name, val):
        a = np.ascontiguousarray(val, np.float64).ravel()
        if self.L.dmo_set(self.d, name.encode(), _p(a), a.size) != 0:
            raise KeyError(name)

    def geti(self, name):
        return self.L.dmo_get_int(self.d, name.encode())

    ncon = property(lambda s: s.geti("ncon"))
    nefc = property(lambda s: s.geti("nefc"))

    def set_caps(self, maxcon, maxrow):
        assert self.L.dmo_set_caps(self.d, maxcon, maxrow) == 0

    def set_flag(self, name, v):
        """per-sim behaviour switches: "stale_contact_slots" (F8, src/deepmimic_env.py:88)."""
        self.L.dmo_set_flag.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        assert self.L.dmo_set_flag(self.d, name.encode(), int(v)) == 0

    def reset_data(self):
        self.L.dmo_data_reset(C.byref(self.cm), self.d)

    def forward(self):
        return self.L.dmo_forward(C.byref(self.cm), self.d)

    def step(self):
        return self.L.dmo_step(C.byref(self.cm), self.d)

    def set_state(self, qpos, qvel):
        q = np.ascontiguousarray(qpos, np.float64)
        v = np.ascontiguousarray(qvel, np.float64)
        return self.L.dmo_set_state(C.byref(self.cm), self.d, _p(q), _p(v))

    def get_obs(self, idx_curr, L):
        obs = np.zeros(NOBS)
        self.L.dmo_get_obs(C.byref(self.cm), self.d, idx_curr, L, _p(obs))
        return obs

    def reward(self, clip, idx):
        t = np.zeros(5)
        r = self.L.dmo_reward(C.byref(self.cm), self.d, C.byref(clip.c), idx, _p(t))
        return r, t

    def env_reset(self, clip, idx_init):
        obs = np.zeros(NOBS)
        self.L.dmo_env_reset(C.byref(self.cm), self.d, C.byref(self.env), C.byref(clip.c), idx_init, _p(obs))
        return obs

    def env_step(self, clip, action, force_state=None):
        obs = np.zeros(NOBS)
        rew = C.c_double(0)
        terms = np.zeros(5)
        reason = C.c_int32(0)
        a = np.ascontiguousarray(action, np.float64)
        if force_state is not None:
            fq = np.ascontiguousarray(force_state[0], np.float64)
            fv = np.ascontiguousarray(force_state[1], np.float64)
            pq, pv = _p(fq), _p(fv)
        else:
            pq = pv = None
        done = self.L.dmo_env_step(C.byref(self.cm), self.d, C.byref(self.env), C.byref(clip.c), _p(a),
                                   pq, pv, _p(obs), C.byref(rew), _p(terms), C.byref(reason))
        return obs, rew.value, bool(done), terms, reason.value


class OracleCombined(OracleSim):
    """One fp64 DPCombinedEnv (src/combined_env.py) on humanoid3d: clips = (walk, run, getup)."""

    def __init__(self, model, clips):
        super().__init__(model)
        assert len(clips) == 3
        self.clips = clips
        self.carr = (_Clip * 3)(*[c.c for c in clips])
        self.cenv = _CombEnv(0, 0, 0, 0, 0.0)

    def comb_reset(self, motion, n_steps):
        obs = np.zeros(NOBS_COMBINED)
        self.L.dmo_combined_reset(C.byref(self.cm), self.d, C.byref(self.cenv), self.carr, motion, n_steps, _p(obs))
        return obs

    def comb_step(self, action, force_state=None):
        obs = np.zeros(NOBS_COMBINED)
        rew = C.c_double(0)
        terms = np.zeros(8)
        reason = C.c_int32(0)
        a = np.ascontiguousarray(action, np.float64)
        if force_state is not None:
            fq = np.ascontiguousarray(force_state[0], np.float64)
            fv = np.ascontiguousarray(force_state[1], np.float64)
            pq, pv = _p(fq), _p(fv)
        else:
            pq = pv = None
        done = self.L.dmo_combined_step(C.byref(self.cm), self.d, C.byref(self.cenv), self.carr, _p(a), pq, pv,
                                        _p(obs), C.byref(rew), _p(terms), C.byref(reason))
        return obs, rew.value, bool(done), terms, reason.value


def narrowphase(t1, x1, M1, z1, t2, x2, M2, z2, margin=0.001):
    """One primitive pair through the oracle's collision dispatch -> list of (dist, pos[3], normal[3], tangent[3])."""
    a = [np.ascontiguousarray(v, np.float64).ravel() for v in (x1, M1, z1, x2, M2, z2)]
    out = np.zeros(80)
    n = lib().dmo_narrowphase(t1, _p(a[0]), _p(a[1]), _p(a[2]), t2, _p(a[3]), _p(a[4]), _p(a[5]), float(margin), _p(out))
    if n < 0:
        raise ValueError("unsupported geom type pair")
    return [(out[10 * k], out[10 * k + 1:10 * k + 4].copy(), out[10 * k + 4:10 * k + 7].copy(), out[10 * k + 7:10 * k + 10].copy())
            for k in range(n)]


def set_tweak(name, value):
    """Sensitivity-study switch of the oracle (dm_oracle.c TW); "reset" restores the MuJoCo defaults."""
    if lib().dmo_set_tweak(name.encode(), float(value)) != 0:
        raise KeyError(name)


def quat_to_rpy(q):
    q = np.ascontiguousarray(q, np.float64)
    out = np.zeros(3)
    lib().dmo_quat_to_rpy(_p(q), _p(out))
    return out


def bench_steps(model, clip, nenv, nsteps, seed=1234):
    return lib().dmo_bench_steps(C.byref(model.cstruct), C.byref(clip.c), nenv, nsteps, seed)
