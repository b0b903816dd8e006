"""Unitree G1 engine — ctypes binding of the dmg1_* entry points of libdeepmimic_hip.so (include/deepmimic_g1_hip.h).

Second robot of the reference: ``DPEnv(robot="unitree_g1")`` (src/deepmimic_env.py:272-484 with the unitree_g1 branches,
model ``deepmimic_unitree_g1.xml``).  The model is compiled by :mod:`mjcf` (general MJCF compiler + hull asset) into
``struct DmModel`` at the G1 dimensions (include/dm_model.h under ``-DDM_ROBOT_G1``), mirrored here as :class:`DmModelG1`.
No CPU fallback: :class:`G1HipEngine` raises if the HIP library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import mjcf, model as _model
from .config import RobotConfig

NQ, NV, NU, NBODY, NGEOM, NJNT, NM, MAXPAIR, NOBS = 44, 43, 37, 39, 94, 38, 434, 1024, 85
NACT, NMESH, NMESHVERT, NREWJ, NEE = 23, 32, 40000, 23, 4
NOBS_COMBINED = 98
TASK_DPENV, TASK_COMBINED = 0, 1
MAXCON, MAXROW, DEBUG_STRIDE = 48, 256, 1024
# [mjmodel.get_joint_qpos_addr(n) ...] minus root and hand joints (src/deepmimic_env.py:206-207)
REW_QPOS = [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 32, 33, 34, 35, 36]
REW_QVEL = [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 31, 32, 33, 34, 35]
CLIP_FLOOR, CLIP_ACYCLIC, CLIP_RUN_RULE = 1, 2, 4
REASONS = {0: None, 1: "low_z", 2: "high_z", 3: "max_ep_len", 4: "acyclical_end", 5: "sim_error", 6: "obs_out_of_bounds",
           7: "fallen without amnesty", 8: "run roll/pitch limit"}
EXPORTS = ["dmg1_default_config", "dmg1_model_sizeof", "dmg1_create", "dmg1_destroy", "dmg1_last_error", "dmg1_load_clip",
           "dmg1_reset", "dmg1_step", "dmg1_step_forced", "dmg1_set_state", "dmg1_get_state", "dmg1_get_counters",
           "dmg1_set_counters", "dmg1_set_debug", "dmg1_last_kernel_ms", "dmg1_obs_dim", "dmg1_get_motion", "dmg1_set_motion",
           "dmg1_set_seed", "dmg1_set_env_clips", "dmg1_get_env_clips", "dmg1_queue_counters"]

_i32, _f64 = C.c_int32, C.c_double


class DmModelG1(C.Structure):
    """ctypes mirror of ``struct DmModel`` under -DDM_ROBOT_G1 (include/dm_model.h) — keep in sync (size is checked)."""

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


def to_cstruct(g) -> DmModelG1:
    """GModel (mjcf.compile_mjcf_general of the G1 asset, with hulls) -> DmModelG1."""
    rc = RobotConfig("unitree_g1")
    assert (g.nq, g.nv, g.nu, g.nbody, g.ngeom, g.njnt, g.nM) == (NQ, NV, NU, NBODY, NGEOM, NJNT, NM)
    if g.solver != "PGS" or g.npair > MAXPAIR:
        raise ValueError("unsupported G1 model options")
    s = DmModelG1()
    s.nq, s.nv, s.nu, s.nbody, s.ngeom, s.njnt, s.npair, s.nM = NQ, NV, NU, NBODY, NGEOM, NJNT, g.npair, NM
    s.integrator = {"Euler": 0, "RK4": 1}[g.integrator]
    s.iterations, s.timestep, s.tolerance, s.meaninertia = g.iterations, g.timestep, g.tolerance, g.meaninertia

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
    if len(g.mesh_vert) > NMESH:
        raise ValueError("too many meshes")
    for i, v in enumerate(g.mesh_vert):
        adr[i], num[i] = a, len(v)
        verts[a:a + len(v)] = v
        cen[i] = g.mesh_center[i]
        a += len(v)
    if a > NMESHVERT:
        raise ValueError("too many hull vertices")
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


def load_g1_model():
    """(GModel, DmModelG1) of the packaged ``deepmimic_unitree_g1.xml`` + hull asset, cached."""
    if not _MODEL:
        hulls = mjcf.load_g1_hulls()
        if hulls is None:
            raise RuntimeError("assets/unitree_g1_hulls.npz is missing (scripts/make_g1_hulls.py writes it)")
        g = mjcf.compile_mjcf_general(os.path.join(_model.ASSET_DIR, "deepmimic_unitree_g1.xml"), hulls=hulls)
        _MODEL.append((g, to_cstruct(g)))
    return _MODEL[0]


class DmG1Config(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("max_ep_length", C.c_int32), ("vel_obs_scale", C.c_float), ("high_z", C.c_float),
                ("obs_bound", C.c_float), ("seed", C.c_uint64), ("auto_reset", C.c_int32), ("device", C.c_int32),
                ("task", C.c_int32), ("amnesty_steps", C.c_int32), ("to_getup_len", C.c_int32), ("pipeline", C.c_int32)]


_BOUND = []


def _lib():
    from ._lib import load_library
    L = load_library()
    if not _BOUND:
        vp, i32 = C.c_void_p, C.c_int
        L.dmg1_default_config.argtypes = [C.POINTER(DmG1Config)]
        L.dmg1_default_config.restype = None
        L.dmg1_model_sizeof.restype = C.c_size_t
        L.dmg1_create.argtypes = [vp, C.c_size_t, C.POINTER(DmG1Config), C.POINTER(vp)]
        L.dmg1_destroy.argtypes = [vp]
        L.dmg1_last_error.argtypes = [vp]
        L.dmg1_last_error.restype = C.c_char_p
        L.dmg1_load_clip.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32]
        L.dmg1_obs_dim.argtypes = [vp]
        L.dmg1_get_motion.argtypes = [vp, vp, vp]
        L.dmg1_set_motion.argtypes = [vp, vp, vp]
        L.dmg1_reset.argtypes = [vp] * 5
        L.dmg1_step.argtypes = [vp] * 9
        L.dmg1_step_forced.argtypes = [vp] * 9
        L.dmg1_set_state.argtypes = [vp, vp, vp, vp, i32, vp]
        L.dmg1_get_state.argtypes = [vp] * 5
        L.dmg1_get_counters.argtypes = [vp] * 5
        L.dmg1_set_counters.argtypes = [vp] * 4
        L.dmg1_set_debug.argtypes = [vp, vp]
        L.dmg1_set_seed.argtypes = [vp, C.c_uint64]
        L.dmg1_set_env_clips.argtypes = [vp, vp, vp]
        L.dmg1_get_env_clips.argtypes = [vp, vp, vp]
        L.dmg1_last_kernel_ms.argtypes = [vp]
        L.dmg1_last_kernel_ms.restype = C.c_float
        if L.dmg1_model_sizeof() != C.sizeof(DmModelG1):
            raise RuntimeError("DmModelG1 layout mismatch: C %d, ctypes %d" % (L.dmg1_model_sizeof(), C.sizeof(DmModelG1)))
        _BOUND.append(True)
    return L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class G1HipEngine:
    """Batch of N Unitree G1 DeepMimic environments resident on one MI355X (tensors in, tensors out)."""

    def __init__(self, num_envs, device=0, seed=0, auto_reset=True, max_ep_length=1000, task=TASK_DPENV, pipeline=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("G1HipEngine needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        self.torch = torch
        self.L = _lib()
        self.gmodel, self.cmodel = load_g1_model()
        self.N = int(num_envs)
        self.device = torch.device("cuda", device)
        cfg = DmG1Config()
        self.L.dmg1_default_config(C.byref(cfg))
        cfg.num_envs, cfg.seed, cfg.auto_reset, cfg.device, cfg.max_ep_length = self.N, seed, int(auto_reset), device, max_ep_length
        cfg.task = int(task)
        cfg.pipeline = int(pipeline)        # 0 auto (split pipeline from 512 envs up), 1 monolithic, 2 split
        self.split = int(pipeline) == 2 or (int(pipeline) == 0 and self.N >= 512)      # the rule of dmg1_create (csrc/dm_g1.hip)
        self.task = int(task)
        self.obs_dim = NOBS_COMBINED if task else NOBS
        self.terms_dim = 8 if task else 5
        self.h = C.c_void_p()
        rc = self.L.dmg1_create(C.byref(self.cmodel), C.sizeof(DmModelG1), C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("dmg1_create failed: %d" % rc)
        self.clip_len = 0
        self._debug = None

    def close(self):
        if getattr(self, "h", None):
            self.L.dmg1_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, (self.L.dmg1_last_error(self.h) or b"").decode()))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def load_clip(self, mocap, floor=False, acyclic=False, run_rule=False, clip_id=0):
        q, v, bx, gx = [np.ascontiguousarray(a, np.float64) for a in mocap.tables()]
        if clip_id == 0:
            self.clip_len = len(q)
        flags = (CLIP_FLOOR if floor else 0) | (CLIP_ACYCLIC if acyclic else 0) | (CLIP_RUN_RULE if run_rule else 0)
        self._check(self.L.dmg1_load_clip(self.h, int(clip_id), len(q), q.ctypes.data, v.ctypes.data, bx.ctypes.data, gx.ctypes.data,
                                          flags), "dmg1_load_clip")

    def alloc_outputs(self):
        t, d, n = self.torch, self.device, self.N
        return dict(obs=t.zeros(n, self.obs_dim, device=d), rew=t.zeros(n, device=d), done=t.zeros(n, dtype=t.uint8, device=d),
                    terms=t.zeros(n, self.terms_dim, device=d), reason=t.zeros(n, dtype=t.int32, device=d),
                    terminal_obs=t.zeros(n, self.obs_dim, device=d))

    def reset(self, obs, idx_init=None, mask=None):
        self._check(self.L.dmg1_reset(self.h, _ptr(mask), _ptr(idx_init), _ptr(obs), self._stream()), "dmg1_reset")

    def step(self, actions, out):
        assert actions.shape == (self.N, NACT) and actions.dtype == self.torch.float32 and actions.is_contiguous()
        self._check(self.L.dmg1_step(self.h, _ptr(actions), _ptr(out["obs"]), _ptr(out["rew"]), _ptr(out["done"]),
                                     _ptr(out.get("terms")), _ptr(out.get("reason")), _ptr(out.get("terminal_obs")), self._stream()),
                    "dmg1_step")

    def step_forced(self, qpos, qvel, out):
        assert qpos.shape == (self.N, NQ) and qvel.shape == (self.N, NV) and qpos.is_contiguous() and qvel.is_contiguous()
        self._check(self.L.dmg1_step_forced(self.h, _ptr(qpos), _ptr(qvel), _ptr(out["obs"]), _ptr(out["rew"]), _ptr(out["done"]),
                                            _ptr(out["terms"]), _ptr(out["reason"]), self._stream()), "dmg1_step_forced")

    def set_state(self, qpos, qvel, warm=None, run_forward=True):
        assert qpos.shape == (self.N, NQ) and qvel.shape == (self.N, NV) and qpos.is_contiguous() and qvel.is_contiguous()
        self._check(self.L.dmg1_set_state(self.h, _ptr(qpos), _ptr(qvel), _ptr(warm), int(run_forward), self._stream()),
                    "dmg1_set_state")

    def get_state(self):
        t, d, n = self.torch, self.device, self.N
        q, v, w = t.zeros(n, NQ, device=d), t.zeros(n, NV, device=d), t.zeros(n, NV, device=d)
        self._check(self.L.dmg1_get_state(self.h, _ptr(q), _ptr(v), _ptr(w), self._stream()), "dmg1_get_state")
        return q, v, w

    def get_counters(self):
        t, d, n = self.torch, self.device, self.N
        i, l, r = t.zeros(n, dtype=t.int32, device=d), t.zeros(n, dtype=t.int32, device=d), t.zeros(n, device=d)
        self._check(self.L.dmg1_get_counters(self.h, _ptr(i), _ptr(l), _ptr(r), self._stream()), "dmg1_get_counters")
        return i, l, r

    def set_counters(self, idx_curr=None, episode_length=None):
        self._check(self.L.dmg1_set_counters(self.h, _ptr(idx_curr), _ptr(episode_length), self._stream()), "dmg1_set_counters")

    def get_motion(self):
        m = self.torch.zeros(self.N, dtype=self.torch.int32, device=self.device)
        self._check(self.L.dmg1_get_motion(self.h, _ptr(m), self._stream()), "dmg1_get_motion")
        return m

    def set_motion(self, motion):
        self._check(self.L.dmg1_set_motion(self.h, _ptr(motion), self._stream()), "dmg1_set_motion")

    def set_seed(self, seed):
        """gym.Env.seed(): re-keys the RSI reset generator (dmg1_set_seed)."""
        self._check(self.L.dmg1_set_seed(self.h, int(seed) & 0xFFFFFFFFFFFFFFFF), "dmg1_set_seed")

    def set_env_clips(self, clip_ids):
        """DPEnv task: per-env clip id (int32 device tensor [N]) among the loaded clip slots."""
        assert clip_ids.dtype == self.torch.int32 and clip_ids.numel() == self.N and clip_ids.is_contiguous()
        self._check(self.L.dmg1_set_env_clips(self.h, _ptr(clip_ids), self._stream()), "dmg1_set_env_clips")

    def get_env_clips(self):
        m = self.torch.zeros(self.N, dtype=self.torch.int32, device=self.device)
        self._check(self.L.dmg1_get_env_clips(self.h, _ptr(m), self._stream()), "dmg1_get_env_clips")
        return m

    def enable_debug(self):
        self._debug = self.torch.zeros(self.N, DEBUG_STRIDE, device=self.device)
        self.L.dmg1_set_debug(self.h, _ptr(self._debug))
        return self._debug

    def last_kernel_ms(self):
        return float(self.L.dmg1_last_kernel_ms(self.h))

    def queue_counters(self):
        """split pipeline: per round of the last step (support-query pair tickets, analytic pair tickets, tickets pulled)."""
        buf = (C.c_int32 * 24)()
        self.L.dmg1_queue_counters.argtypes = [C.c_void_p, C.c_void_p]
        self._check(self.L.dmg1_queue_counters(self.h, buf), "dmg1_queue_counters")
        return [tuple(buf[4 * r:4 * r + 3]) for r in range(6)]


# ------------------------------------------------------------------------------------------ Gym / VecEnv surfaces
try:  # pragma: no cover - SB3 is not installed in the build image; with it the batch classes ARE VecEnvs (isinstance checks of wrappers)
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv as _SB3VecEnv
except Exception:  # noqa: BLE001
    _SB3VecEnv = object


def _g1_mocap(motion):
    from .config import MotionConfig
    from .mocap import MocapDM
    mcfg = MotionConfig(motion, robot="unitree_g1")
    mc = MocapDM(robot="unitree_g1")
    mc.load_mocap(mcfg.mocap_path)
    return mcfg, mc


def _load(engine, mcfg, mc):
    engine.load_clip(mc, floor=mcfg.motion in mcfg.floor_motions, acyclic=mcfg.motion in mcfg.acyclical_motions,
                     run_rule=mcfg.motion == "run")                                  # src/deepmimic_env.py:426


class HipG1VecEnv(_SB3VecEnv):
    """N ``DPEnv(robot="unitree_g1")`` instances as one HIP batch with SubprocVecEnv semantics (auto-reset,
    ``terminal_observation``): the G1 counterpart of :class:`deepmimic_env.HipDeepMimicVecEnv`, which constructs this class
    when asked for ``robot="unitree_g1"``.  Actions are the policy's 23 values (src/deepmimic_env.py:303-307).
    ``motion`` may be a list (per-env clip id = env index mod len(list), as BASELINE config 5 mixes clips); ``sub_batches`` > 1
    splits the batch into independent engines over contiguous env ranges (double-buffered rollouts, see HipDeepMimicVecEnv)."""

    OBS_DIM, TERMS_DIM = NOBS, 5

    def __init__(self, num_envs, motion=None, device=0, seed=1234, auto_reset=True, sub_batches=1):
        from .deepmimic_env import DPEnvConfig
        motions = [motion] if (motion is None or isinstance(motion, str)) else list(motion)
        assert 1 <= len(motions) <= 8, "an engine holds up to 8 clips (DMG1_MAX_CLIPS)"
        pairs = [_g1_mocap(m) for m in motions]
        self.motion_config, self.mocap = pairs[0]
        self.motions = [mcfg.motion for mcfg, _ in pairs]
        self.mocaps = [mc for _, mc in pairs]

        def make(nk, k):
            e = G1HipEngine(nk, device=device, seed=seed + 104729 * k, auto_reset=auto_reset, max_ep_length=DPEnvConfig().MAX_EP_LENGTH)
            for cid, (mcfg, mc) in enumerate(pairs):
                e.load_clip(mc, floor=mcfg.motion in mcfg.floor_motions, acyclic=mcfg.motion in mcfg.acyclical_motions,
                            run_rule=mcfg.motion == "run", clip_id=cid)                 # src/deepmimic_env.py:426
            if len(pairs) > 1:
                ids = (e.torch.arange(nk, device=e.device) + k * nk) % len(pairs)
                e.set_env_clips(ids.to(e.torch.int32).contiguous())
            return e
        self._build(num_envs, sub_batches, make, scale=1.0)

    def _build(self, num_envs, sub_batches, make_engine, scale):
        """Common construction: engines over contiguous env ranges, one set of [N, ...] output tensors, spaces."""
        import torch
        from .deepmimic_env import Box
        self._torch = torch
        self.num_envs, self.sub_batches = int(num_envs), int(sub_batches)
        assert self.sub_batches >= 1 and self.num_envs % self.sub_batches == 0
        nk = self.num_envs // self.sub_batches
        self.robot_config = RobotConfig("unitree_g1")
        self.engines = [make_engine(nk, k) for k in range(self.sub_batches)]
        self.engine = self.engines[0]
        self.model = self.engine.gmodel
        self.device = self.engine.device
        if self.sub_batches == 1:
            self.out = self.engine.alloc_outputs()
        else:   # every engine writes its contiguous block of rows
            z = lambda *shape, dt=torch.float32: torch.zeros(*shape, device=self.device, dtype=dt)
            N, D, K = self.num_envs, self.OBS_DIM, self.TERMS_DIM
            self.out = dict(obs=z(N, D), rew=z(N), done=z(N, dt=torch.uint8), terms=z(N, K), reason=z(N, dt=torch.int32), terminal_obs=z(N, D))
        self.sub_slices = [slice(k * nk, (k + 1) * nk) for k in range(self.sub_batches)]
        self.sub_out = [self.out] if self.sub_batches == 1 else [{k_: v[sl] for k_, v in self.out.items()} for sl in self.sub_slices]
        lo = self.model.act_ctrlrange[:NACT, 0].astype(np.float32) * scale
        hi = self.model.act_ctrlrange[:NACT, 1].astype(np.float32) * scale
        self.action_space = Box(lo, hi, dtype=np.float32)            # the first N - 14 actuators (src/deepmimic_env.py:305-307)
        self.observation_space = Box(-np.inf, np.inf, (self.OBS_DIM,), np.float32)
        self._actions = torch.zeros(self.num_envs, NACT, device=self.device)
        self.render_mode = None
        self.reset_infos = [{} for _ in range(self.num_envs)]
        if _SB3VecEnv is not object:  # pragma: no cover
            _SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def step_sub(self, k, actions_k):
        """Step sub-batch k only (on the current stream): actions_k [N / sub_batches, 23] -> its slice of the outputs."""
        self.engines[k].step(actions_k.contiguous(), self.sub_out[k])
        return self.sub_out[k]

    def reset_tensor(self, idx_init=None):
        for e, o, sl in zip(self.engines, self.sub_out, self.sub_slices):
            e.reset(o["obs"], idx_init=None if idx_init is None else idx_init[sl].contiguous())
        return self.out["obs"]

    def step_tensor(self, actions):
        """One step of the whole batch.  With sub_batches > 1 every engine's launches go to its own HIP stream, forked from and
        joined back into the current one: while one engine's g1_env_kernel drains its heaviest envs, the CUs it has left idle
        run the other engines' kernels (the envs of a batch are independent; results do not depend on the overlap)."""
        t = self._torch
        actions = actions.contiguous()
        if self.sub_batches == 1:
            self.engine.step(actions, self.out)
            return self.out
        cur = t.cuda.current_stream(self.device)
        if getattr(self, "_streams", None) is None:
            from .streams import concurrent_streams
            self._streams = concurrent_streams(self.device, len(self.engines))      # on distinct hardware queues (probed)
        fork = cur.record_event()
        for e, o, sl, s in zip(self.engines, self.sub_out, self.sub_slices, self._streams):
            s.wait_event(fork)
            with t.cuda.stream(s):
                e.step(actions[sl], o)
            cur.wait_event(s.record_event())
        return self.out

    def reset(self):
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        t = self._torch
        self._actions.copy_(t.as_tensor(np.ascontiguousarray(actions, dtype=np.float32)).reshape(self._actions.shape))

    def _infos(self, terms, reason, done, tobs):
        from .deepmimic_env import LazyInfos
        return LazyInfos(terms, reason, done, tobs)

    def step_wait(self):
        t = self._torch
        out = self.step_tensor(self._actions)
        packed = t.cat([out["obs"], out["terminal_obs"], out["terms"], out["rew"][:, None], out["done"][:, None].float(),
                        out["reason"][:, None].float()], dim=1).cpu().numpy()
        d, k = self.OBS_DIM, self.TERMS_DIM
        obs, tobs, terms = (np.ascontiguousarray(packed[:, 0:d]), np.ascontiguousarray(packed[:, d:2 * d]),
                            np.ascontiguousarray(packed[:, 2 * d:2 * d + k]))
        rew = packed[:, 2 * d + k].copy()
        done = packed[:, 2 * d + k + 1] != 0
        reason = packed[:, 2 * d + k + 2].astype(np.int32)
        return obs, rew, done, self._infos(terms, reason, done, tobs)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()

    def seed(self, seed=None):
        """SB3 VecEnv.seed: env i gets seed + i.  Here: re-keys the engines' counter-based RSI generator (dmg1_set_seed; env index
        and reset count are part of the key already) and returns the per-env seeds SB3 expects."""
        if seed is None:
            return [None] * self.num_envs
        for k, e in enumerate(self.engines):
            e.set_seed(int(seed) + 104729 * k)
        return [int(seed) + i for i in range(self.num_envs)]

    # ---- the rest of SB3's VecEnv surface (same conventions as HipDeepMimicVecEnv: one batch object stands for all envs)
    def _n_indices(self, indices):
        return self.num_envs if indices is None else len(np.atleast_1d(indices))

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * self._n_indices(indices)

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self, method_name)(*method_args, **method_kwargs)] * self._n_indices(indices)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self._n_indices(indices)

    def getattr_depth_check(self, name, already_found):
        return None

    @property
    def unwrapped(self):
        return self

    def get_images(self):
        """One frame per env in SB3; a 4 096-tile mosaic is of no use: the frame of env 0 stands for the batch."""
        return [self.render(mode="rgb_array")]

    def render(self, mode=None):
        """Software stick figure (render.py) of env 0 of the batch, 240 x 320 x 3 uint8 — what VecVideoRecorder-style callers get.
        The forward evaluation that refreshes the body poses puts the warm start back: rendering never changes the physics."""
        from .render import stick_figure
        e = self.engine
        if e._debug is None:
            e.enable_debug()
        q, v, w = e.get_state()
        e.set_state(q, v, warm=w, run_forward=True)
        e.set_state(q, v, warm=w, run_forward=False)
        xpos = e._debug[0, :117].double().cpu().numpy().reshape(39, 3)
        return stick_figure(xpos, self.model.body_parent)


class G1DPEnv:
    """``DPEnv(motion, robot="unitree_g1")`` (src/deepmimic_env.py:272-510): one environment of the batch engine behind the
    reference's Gym surface; ``deepmimic_env.DPEnv`` constructs this class for the G1 robot."""

    version = "v1.0"

    def __init__(self, motion=None, load_mocap=True, robot="unitree_g1", _profile=False, device=0):
        import random
        import torch
        from .deepmimic_env import Box, DPEnvConfig
        if robot != "unitree_g1":
            raise ValueError("G1DPEnv is the unitree_g1 environment")
        if not load_mocap:
            raise NotImplementedError("the mocap-less probing instance (src/deepmimic_env.py:287-293) is only used by MocapDM's "
                                      "own kinematics pass, which mjcf.forward_kinematics_general replaces")
        self._torch, self._random = torch, random
        self.ENV_CFG = DPEnvConfig()
        self.robot_config = RobotConfig("unitree_g1")
        self.motion_config, self.mocap = _g1_mocap(motion)
        self._eng = G1HipEngine(1, device=device, auto_reset=False, max_ep_length=self.ENV_CFG.MAX_EP_LENGTH)
        _load(self._eng, self.motion_config, self.mocap)
        self.model = self._eng.gmodel
        self._out = self._eng.alloc_outputs()
        self.mocap_dt, self.mocap_data_len = self.mocap.dt, len(self.mocap.data_config)
        self.idx_curr, self.episode_reward, self.episode_length = -1, 0, 0
        lo, hi = self.model.act_ctrlrange[:NACT, 0].astype(np.float32), self.model.act_ctrlrange[:NACT, 1].astype(np.float32)
        self.action_space = Box(lo, hi, dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, (NOBS,), np.float64)
        self.reference_state_init()

    def reference_state_init(self, idx_init=None):
        self.idx_init = self._random.randint(0, self.mocap_data_len - 1) if idx_init is None else idx_init
        self.idx_curr = self.idx_init

    def _push(self):
        t = self._torch
        self._eng.set_counters(t.tensor([max(self.idx_curr, 0)], dtype=t.int32, device=self._eng.device),
                               t.tensor([self.episode_length], dtype=t.int32, device=self._eng.device))

    def step(self, action, force_state=None):
        from .deepmimic_env import _make_info
        t, dev = self._torch, self._eng.device
        action = np.asarray(action, np.float64)
        if action.shape == (NU,):          # the full action vector MujocoEnv.__init__ probes with (:349): hands are dropped
            action = action[:NACT]
        assert action.shape == (NACT,)
        self._push()
        if force_state is not None:
            q, v = force_state
            self._eng.step_forced(t.tensor(np.asarray(q)[None], dtype=t.float32, device=dev),
                                  t.tensor(np.asarray(v)[None], dtype=t.float32, device=dev), self._out)
        else:
            self._eng.step(t.tensor(action[None], dtype=t.float32, device=dev), self._out)
        obs = self._out["obs"][0].double().cpu().numpy()
        reason, done = int(self._out["reason"][0].item()), bool(self._out["done"][0].item())
        if reason in (5, 6):
            if reason == 6:
                self.idx_curr = (self.idx_curr + 1) % self.mocap_data_len
                self.episode_reward = float(self._eng.get_counters()[2][0].item())
                self.episode_length += 1
            return obs, 0, True, {}
        reward = float(self._out["rew"][0].item())
        info = _make_info(self._out["terms"][0].cpu().numpy(), reason)
        if reason == 8:
            info["done_reason"] = REASONS[8]
        self.idx_curr = (self.idx_curr + 1) % self.mocap_data_len
        self.episode_reward += reward
        self.episode_length += 1
        return obs, reward, done, info

    def reset(self):
        self.episode_reward, self.episode_length = 0, 0
        return self.reset_model()

    def reset_model(self, idx_init=None):
        t = self._torch
        self.reference_state_init(idx_init=idx_init)
        obs = t.zeros(1, NOBS, device=self._eng.device)
        self._eng.reset(obs, idx_init=t.tensor([self.idx_init], dtype=t.int32, device=self._eng.device))
        self._eng.set_counters(None, t.tensor([self.episode_length], dtype=t.int32, device=self._eng.device))
        return obs[0].double().cpu().numpy()

    def set_state(self, qpos, qvel):
        t = self._torch
        self._eng.set_state(t.tensor(np.asarray(qpos)[None], dtype=t.float32, device=self._eng.device),
                            t.tensor(np.asarray(qvel)[None], dtype=t.float32, device=self._eng.device))

    def render(self, mode=None):
        """Software stick figure (render.py) of the current body poses."""
        from .render import stick_figure
        if self._eng._debug is None:
            self._eng.enable_debug()
        q, v, w = self._eng.get_state()
        self._eng.set_state(q, v, warm=w, run_forward=True)     # refresh the derived arrays; the warm start is put back
        self._eng.set_state(q, v, warm=w, run_forward=False)
        xpos = self._eng._debug[0, :117].double().cpu().numpy().reshape(39, 3)
        return stick_figure(xpos, self.model.body_parent)

    def close(self):
        self._eng.close()


# ------------------------------------------------------------------------------------------ DPCombinedEnv on the G1
COMBINED_CLIPS = ("walk", "run", "getup_facedown_towalk")      # src/combined_env.py:168-170
MOTION_WALK, MOTION_RUN, MOTION_GETUP, MOTION_TO_GETUP = 0, 1, 2, 3


def _combined_engine(num_envs, device, seed, auto_reset):
    from .combined_env import DPCombinedEnvConfig
    cfg = DPCombinedEnvConfig()
    eng = G1HipEngine(num_envs, device=device, seed=seed, auto_reset=auto_reset, max_ep_length=cfg.MAX_EP_LENGTH, task=TASK_COMBINED)
    mocaps = []
    for cid, m in enumerate(COMBINED_CLIPS):
        mcfg, mc = _g1_mocap(m)
        eng.load_clip(mc, clip_id=cid)
        mocaps.append(mc)
    return eng, mocaps


class HipG1CombinedVecEnv(HipG1VecEnv):
    """N ``DPCombinedEnv()`` instances — the reference's training environment (src/sb3_ppo.py:277-278): Unitree G1, walk / run /
    getup / to_getup motion state machine, obs 98, 23 actions — as one HIP batch with SubprocVecEnv semantics."""

    OBS_DIM, TERMS_DIM = NOBS_COMBINED, 8

    def __init__(self, num_envs, device=0, seed=1234, auto_reset=True, sub_batches=1):
        self.mocaps = None

        def make(nk, k):
            e, mocaps = _combined_engine(nk, device, seed + 104729 * k, auto_reset)
            self.mocaps = self.mocaps or mocaps
            return e
        self._build(num_envs, sub_batches, make, scale=1.0 / 20.0)      # action space = ctrlrange / ACT_SCALE (src/combined_env.py:196-200)
        self.mocap = self.mocaps[0]

    def _infos(self, terms, reason, done, tobs):
        from .combined_env import _LazyCombinedInfos
        return _LazyCombinedInfos(terms, reason, done, tobs)


class G1CombinedEnv:
    """``DPCombinedEnv()`` of the reference (src/combined_env.py:102-533) — hard-wired to the Unitree G1 there — as one
    environment of the batch engine; ``combined_env.DPCombinedEnv(robot="unitree_g1")`` constructs this class."""

    version = "v0.2.up"

    def __init__(self, verbose=0, _profile=False, device=0):
        import random
        import torch
        from .combined_env import DPCombinedEnvConfig, MTToGetup, PAWalk
        from .deepmimic_env import Box
        self._torch, self._random, self.verbose = torch, random, verbose
        self.ENV_CFG = DPCombinedEnvConfig()
        self.robot = "unitree_g1"
        self.robot_config = RobotConfig(self.robot)
        self._eng, mocaps = _combined_engine(1, device, 0, False)
        self.walk_mocap, self.run_mocap, self.getup_mocap = mocaps
        self.action_mocap = None
        self.to_getup_mocap = MTToGetup(self.getup_mocap)
        self._motions = [self.walk_mocap, self.run_mocap, self.getup_mocap, self.to_getup_mocap]
        self.model = self._eng.gmodel
        self._out = self._eng.alloc_outputs()
        self.episode_reward, self.episode_length, self.debug_n_bad_angles = 0, 0, 0
        self.current_motion_n_steps, self.current_motion_mocap, self.current_player_action = None, None, PAWalk()
        lo, hi = self.model.act_ctrlrange[:NACT, 0].astype(np.float32) / 20.0, self.model.act_ctrlrange[:NACT, 1].astype(np.float32) / 20.0
        self.action_space = Box(lo, hi, dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, (NOBS_COMBINED,), np.float64)

    def _motion_id(self):
        return self._motions.index(self.current_motion_mocap)

    def get_current_motion_state(self):
        idx = self.current_motion_n_steps % self.current_motion_mocap.get_length()
        return self.current_motion_mocap.get_qpos(idx) * 1.0, self.current_motion_mocap.get_qvel(idx) * 1.0

    def change_to_motion(self, motion):
        self.current_motion_mocap = motion
        self.current_motion_n_steps = 0

    def _push(self):
        t, dev = self._torch, self._eng.device
        self._eng.set_motion(t.tensor([self._motion_id()], dtype=t.int32, device=dev))
        self._eng.set_counters(t.tensor([self.current_motion_n_steps], dtype=t.int32, device=dev),
                               t.tensor([self.episode_length], dtype=t.int32, device=dev))

    def reset(self, rsi=True):
        from .combined_env import PAWalk
        t, dev = self._torch, self._eng.device
        if rsi:   # :219-227
            if self._random.randint(0, 1) == 0:
                self.current_motion_mocap = self.walk_mocap
                self.current_motion_n_steps = self.ENV_CFG.AMNESTY_STEPS + 10 + self._random.randint(0, self.walk_mocap.get_length() - 1)
            else:
                self.current_motion_mocap = self.getup_mocap
                self.current_motion_n_steps = self._random.randint(0, self.getup_mocap.get_length() - 1)
        else:
            self.current_motion_mocap, self.current_motion_n_steps = self.getup_mocap, 0
        self.current_player_action = PAWalk()
        self.episode_reward, self.episode_length = 0, 0
        self._eng.set_motion(t.tensor([self._motion_id()], dtype=t.int32, device=dev))
        obs = t.zeros(1, NOBS_COMBINED, device=dev)
        self._eng.reset(obs, idx_init=t.tensor([self.current_motion_n_steps], dtype=t.int32, device=dev))
        return obs[0].double().cpu().numpy()

    def step(self, action, force_state=None):
        t, dev = self._torch, self._eng.device
        action = np.asarray(action, np.float64)
        if action.shape == (NU,):
            action = action[:NACT]
        assert action.shape == (NACT,)
        self._push()
        if force_state is not None:
            q, v = force_state
            self._eng.step_forced(t.tensor(np.asarray(q)[None], dtype=t.float32, device=dev),
                                  t.tensor(np.asarray(v)[None], dtype=t.float32, device=dev), self._out)
        else:
            self._eng.step(t.tensor(action[None], dtype=t.float32, device=dev), self._out)
        obs = self._out["obs"][0].double().cpu().numpy()
        reason, done = int(self._out["reason"][0].item()), bool(self._out["done"][0].item())
        if reason == 5:
            return obs, 0, True, {}
        terms = self._out["terms"][0].cpu().numpy()
        self.current_motion_mocap = self._motions[int(self._eng.get_motion()[0].item())]
        self.current_motion_n_steps = int(self._eng.get_counters()[0][0].item())
        self.debug_n_bad_angles = int(terms[7])
        self.episode_length += 1
        if reason == 6:
            self.episode_reward = float(self._eng.get_counters()[2][0].item())
            return obs, 0, True, {}
        reward = float(self._out["rew"][0].item())
        self.episode_reward += reward
        info = {"reward_config": float(terms[0]), "reward_qvel": float(terms[1]), "reward_end_eff": float(terms[2]),
                "reward_com": float(terms[3]), "reward_joint_limit": float(terms[4]), "imitation_reward": float(terms[5]),
                "task_reward": float(terms[6])}
        if REASONS.get(reason):
            info["done_reason"] = REASONS[reason]
        return obs, reward, done, info

    def render(self, mode=None):
        """Software stick figure (render.py) of the current body poses."""
        from .render import stick_figure
        if self._eng._debug is None:
            self._eng.enable_debug()
        q, v, w = self._eng.get_state()
        self._eng.set_state(q, v, warm=w, run_forward=True)     # refresh the derived arrays; the warm start is put back
        self._eng.set_state(q, v, warm=w, run_forward=False)
        xpos = self._eng._debug[0, :117].double().cpu().numpy().reshape(39, 3)
        return stick_figure(xpos, self.model.body_parent)

    def close(self):
        self._eng.close()
