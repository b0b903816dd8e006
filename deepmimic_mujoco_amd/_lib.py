"""ctypes binding of libdeepmimic_hip.so (include/deepmimic_hip.h).

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible
the constructor of :class:`HipEngine` raises.  PyTorch is used only as the owner
of device buffers (``tensor.data_ptr()``) and of the HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .model import DmModel, NQ, NV, NU, NOBS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdeepmimic_hip.so")
DEBUG_STRIDE = 416

REASONS = {0: None, 1: "low_z", 2: "high_z", 3: "max_ep_len", 4: "acyclical_end",
           5: "sim_error", 6: "obs_out_of_bounds", 7: "fallen without amnesty"}
TASK_DPENV, TASK_COMBINED = 0, 1

EXPORTS = ["dm_default_config", "dm_create", "dm_destroy", "dm_last_error", "dm_num_envs",
           "dm_load_clip", "dm_set_env_clips", "dm_reset", "dm_step", "dm_step_forced", "dm_physics_step",
           "dm_set_state", "dm_get_state", "dm_get_counters", "dm_set_counters", "dm_set_debug",
           "dm_fill_random_actions", "dm_last_step_ms", "dm_enable_timing", "dm_get_work",
           "dm_set_clip_flags", "dm_obs_dim", "dm_terms_dim", "dm_get_env_clips", "dm_mean_step_ms", "dm_ppo_loss", "dm_forward", "dm_linear_wgrad", "dm_ppo_gather", "dm_flat_adam_step", "dm_policy_sample",
           "dm_rollout_store", "dm_policy_pack", "dm_policy_forward", "dm_policy_packed_floats", "dm_ppo_mlp_grad", "dm_ppo_mlp_workspace_floats", "dm_flat_adam_update", "dm_flat_adam_step_gather", "dm_colsum", "dm_set_seed",
           "dm_linear_tanh", "dm_tanh_linear_wgrad", "dm_tanh_bwd_colsum",
           "dm_ppo_wide_grad", "dm_ppo_wide_packed_elems", "dm_ppo_wide_dp", "dm_ppo_wide_supported"]


class DmConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("max_ep_length", C.c_int32),
                ("vel_obs_scale", C.c_float), ("low_z", C.c_float), ("high_z", C.c_float),
                ("w_pose", C.c_float), ("w_vel", C.c_float), ("w_end_eff", C.c_float),
                ("w_com", C.c_float), ("w_joint_limit", C.c_float), ("obs_bound", C.c_float),
                ("seed", C.c_uint64), ("auto_reset", C.c_int32), ("device", C.c_int32),
                ("lpt_schedule", C.c_int32), ("task", C.c_int32),
                ("amnesty_steps", C.c_int32), ("to_getup_len", C.c_int32),
                ("integrator", C.c_int32), ("stale_contact_slots", C.c_int32)]


INTEGRATORS = {None: 0, "model": 0, "Euler": 1, "euler": 1, "RK4": 2, "rk4": 2}   # DM_CFG_INT_*


class DmPpoMlpStep(C.Structure):
    """include/deepmimic_hip.h: DmPpoMlpStep"""
    _fields_ = [("B", C.c_int32), ("D", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("A", C.c_int32),
                ("normalize_advantage", C.c_int32), ("clip_range", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float),
                ("reserved", C.c_int32),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p), ("old_logp", C.c_void_p),
                ("log_std", C.c_void_p),
                ("W", (C.c_void_p * 3) * 2), ("b", (C.c_void_p * 3) * 2), ("gW", (C.c_void_p * 3) * 2), ("gb", (C.c_void_p * 3) * 2),
                ("g_log_std", C.c_void_p), ("out8", C.c_void_p), ("workspace", C.c_void_p), ("workspace_floats", C.c_longlong),
                ("zero_ptr", C.c_void_p), ("zero_floats", C.c_longlong), ("adam_state2", C.c_void_p), ("loss_acc", C.c_void_p)]


class DmGatherSpec(C.Structure):
    """include/deepmimic_hip.h: DmGatherSpec"""
    _fields_ = [("idx", C.c_void_p), ("B", C.c_int32), ("D", C.c_int32), ("A", C.c_int32), ("reserved", C.c_int32),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p), ("logp", C.c_void_p),
                ("o_obs", C.c_void_p), ("o_act", C.c_void_p), ("o_adv", C.c_void_p), ("o_ret", C.c_void_p), ("o_logp", C.c_void_p)]


class DmPpoWideStep(C.Structure):
    """include/deepmimic_hip.h: DmPpoWideStep"""
    _fields_ = [("B", C.c_int32), ("D", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("A", C.c_int32),
                ("normalize_advantage", C.c_int32), ("clip_range", C.c_float), ("vf_coef", C.c_float), ("ent_coef", C.c_float),
                ("reserved", C.c_int32),
                ("obs", C.c_void_p), ("act", C.c_void_p), ("adv", C.c_void_p), ("ret", C.c_void_p), ("old_logp", C.c_void_p),
                ("log_std", C.c_void_p),
                ("W", (C.c_void_p * 3) * 2), ("b", (C.c_void_p * 3) * 2), ("gW", (C.c_void_p * 3) * 2), ("gb", (C.c_void_p * 3) * 2),
                ("g_log_std", C.c_void_p),
                ("wpk", C.c_void_p * 2), ("xbT", C.c_void_p), ("h1T", C.c_void_p * 2), ("dz1T", C.c_void_p * 2), ("h2T", C.c_void_p * 2),
                ("dz2T", C.c_void_p * 2), ("dz3T", C.c_void_p * 2), ("part", C.c_void_p), ("stats8", C.c_void_p), ("out8", C.c_void_p),
                ("zero_ptr", C.c_void_p), ("zero_floats", C.c_longlong), ("adam_state2", C.c_void_p), ("loss_acc", C.c_void_p)]


_LIB = None


def load_library():
    """dlopen the in-tree HIP library; raise loudly if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdeepmimic_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C deepmimic_mujoco_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # PyTorch bundles its own libamdhip64.so.7; import it first so this library binds to the SAME
    # HIP runtime instance (two runtimes in one process cannot share a device context).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.dm_default_config.argtypes = [C.POINTER(DmConfig)]
    L.dm_default_config.restype = None
    L.dm_create.argtypes = [C.POINTER(DmModel), C.POINTER(DmConfig), C.POINTER(vp)]
    L.dm_destroy.argtypes = [vp]
    L.dm_last_error.argtypes = [vp]
    L.dm_last_error.restype = C.c_char_p
    L.dm_num_envs.argtypes = [vp]
    L.dm_obs_dim.argtypes = [vp]
    L.dm_terms_dim.argtypes = [vp]
    L.dm_get_env_clips.argtypes = [vp, vp, vp]
    L.dm_load_clip.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.dm_set_env_clips.argtypes = [vp, vp, vp]
    L.dm_reset.argtypes = [vp, vp, vp, vp, vp]
    L.dm_step.argtypes = [vp] * 9
    L.dm_step_forced.argtypes = [vp] * 9
    L.dm_physics_step.argtypes = [vp] * 3
    L.dm_set_state.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp]
    L.dm_get_state.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.dm_forward.argtypes = [vp, vp, i32, vp]
    L.dm_linear_wgrad.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.dm_policy_sample.argtypes = [vp, vp, i32, i32, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    L.dm_rollout_store.argtypes = [i32, i32, i32] + [vp] * 16
    L.dm_policy_packed_floats.argtypes = [i32] * 4
    L.dm_ppo_mlp_workspace_floats.argtypes = [i32] * 5
    L.dm_ppo_mlp_grad.argtypes = [C.POINTER(DmPpoMlpStep), vp]
    L.dm_ppo_wide_grad.argtypes = [C.POINTER(DmPpoWideStep), vp]
    L.dm_ppo_wide_packed_elems.argtypes = [i32, i32, i32]
    L.dm_ppo_wide_dp.argtypes = [i32]
    L.dm_ppo_wide_supported.argtypes = [i32] * 5
    L.dm_policy_pack.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.dm_policy_forward.argtypes = [vp] + [i32] * 5 + [vp] * 9 + [C.c_uint64, vp, C.c_uint32, i32] + [vp] * 9
    L.dm_flat_adam_step.argtypes = [vp, vp, vp, vp, i32] + [C.c_float] * 6 + [vp, i32, vp]
    L.dm_colsum.argtypes = [vp, i32, i32, vp, vp]
    L.dm_linear_tanh.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.dm_tanh_linear_wgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.dm_tanh_bwd_colsum.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    L.dm_flat_adam_update.argtypes = [vp, vp, vp, vp, i32] + [C.c_float] * 6 + [vp, i32, vp]
    L.dm_flat_adam_step_gather.argtypes = [vp, vp, vp, vp, i32] + [C.c_float] * 6 + [vp, i32, i32, vp, vp]
    L.dm_flat_adam_step_gather.restype = i32
    L.dm_ppo_gather.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dm_get_counters.argtypes = [vp, vp, vp, vp, vp]
    L.dm_set_counters.argtypes = [vp, vp, vp, vp]
    L.dm_set_debug.argtypes = [vp, vp]
    L.dm_set_seed.argtypes = [vp, C.c_uint64]
    L.dm_get_work.argtypes = [vp, vp, vp]
    L.dm_set_clip_flags.argtypes = [vp, i32, i32]
    L.dm_fill_random_actions.argtypes = [vp, vp, C.c_uint32, vp]
    L.dm_last_step_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.dm_enable_timing.argtypes = [vp, i32]
    L.dm_ppo_loss.argtypes = [vp] * 7 + [i32, i32, C.c_float, C.c_float, C.c_float, i32] + [vp] * 6
    L.dm_mean_step_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    for name in EXPORTS:
        if name not in ("dm_default_config", "dm_last_error"):
            getattr(L, name).restype = C.c_longlong if name in ("dm_policy_packed_floats", "dm_ppo_mlp_workspace_floats", "dm_ppo_wide_packed_elems") else C.c_int
    _LIB = L
    return L


def default_config(**kw) -> DmConfig:
    cfg = DmConfig()
    load_library().dm_default_config(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipEngine:
    """Thin object wrapper over a DmHandle; all tensors are torch CUDA tensors owned by the caller."""

    def __init__(self, model, num_envs, device=0, seed=1234, auto_reset=True, integrator=None, **cfg_kw):
        import torch
        cfg_kw["integrator"] = INTEGRATORS[integrator]
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible to HIP: the DeepMimic engine has no CPU fallback")
        self.torch = torch
        self.L = load_library()
        self.model = model
        self.N = int(num_envs)
        self.device = torch.device("cuda", device)
        self.cfg = default_config(num_envs=self.N, device=device, seed=seed,
                                  auto_reset=1 if auto_reset else 0, **cfg_kw)
        h = C.c_void_p()
        rc = self.L.dm_create(C.byref(model.cstruct), C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError("dm_create failed with code %d" % rc)
        self.h = h
        self.obs_dim = self.L.dm_obs_dim(h)      # 67 (DPEnv) or 72 (DPCombinedEnv)
        self.terms_dim = self.L.dm_terms_dim(h)  # 5 or 8
        self.clip_len = {}
        self._debug = None

    def set_seed(self, seed):
        """gym's env.seed(): re-key the reset / random-action generator."""
        self._chk(self.L.dm_set_seed(self.h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)), "dm_set_seed")

    def close(self):
        if getattr(self, "h", None):
            self.L.dm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            msg = self.L.dm_last_error(self.h)
            raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # ---- clips
    def load_clip(self, clip_id, mocap, floor=False, acyclic=False):
        q, v, b, g = [np.ascontiguousarray(a, np.float64) for a in mocap.tables()]
        self._chk(self.L.dm_load_clip(self.h, clip_id, len(q), q.ctypes.data_as(C.c_void_p),
                                      v.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                      g.ctypes.data_as(C.c_void_p)), "dm_load_clip")
        self.clip_len[clip_id] = len(q)
        self._chk(self.L.dm_set_clip_flags(self.h, clip_id, (1 if floor else 0) | (2 if acyclic else 0)), "dm_set_clip_flags")

    def set_env_clips(self, clip_ids):
        t = None if clip_ids is None else clip_ids.to(self.device, self.torch.int32).contiguous()
        self._chk(self.L.dm_set_env_clips(self.h, _ptr(t), self._stream()), "dm_set_env_clips")
        self._keep = t

    def get_env_clips(self):
        """Per-env clip id (DPEnv task) or motion id 0 walk / 1 run / 2 getup / 3 to_getup (combined task)."""
        out = self.torch.zeros(self.N, dtype=self.torch.int32, device=self.device)
        self._chk(self.L.dm_get_env_clips(self.h, _ptr(out), self._stream()), "dm_get_env_clips")
        return out

    # ---- buffers
    def alloc_outputs(self):
        t, d = self.torch, self.device
        return dict(obs=t.zeros(self.N, self.obs_dim, device=d), rew=t.zeros(self.N, device=d),
                    done=t.zeros(self.N, dtype=t.uint8, device=d), terms=t.zeros(self.N, self.terms_dim, device=d),
                    reason=t.zeros(self.N, dtype=t.int32, device=d),
                    terminal_obs=t.zeros(self.N, self.obs_dim, device=d))

    def enable_debug(self, on=True):
        if on:
            self._debug = self.torch.zeros(self.N, DEBUG_STRIDE, device=self.device)
            self._chk(self.L.dm_set_debug(self.h, _ptr(self._debug)), "dm_set_debug")
        else:
            self._chk(self.L.dm_set_debug(self.h, None), "dm_set_debug")
            self._debug = None
        return self._debug

    # ---- hot path
    def reset(self, obs, mask=None, idx_init=None):
        self._chk(self.L.dm_reset(self.h, _ptr(mask), _ptr(idx_init), _ptr(obs), self._stream()), "dm_reset")

    def step(self, actions, out):
        self._chk(self.L.dm_step(self.h, _ptr(actions), _ptr(out["obs"]), _ptr(out["rew"]), _ptr(out["done"]),
                                 _ptr(out.get("terms")), _ptr(out.get("reason")), _ptr(out.get("terminal_obs")),
                                 self._stream()), "dm_step")

    def physics_step(self, actions):
        """sim.step() alone (src/deepmimic_env.py:362): the state advances, nothing is observed (dm_physics_step)."""
        self._chk(self.L.dm_physics_step(self.h, _ptr(actions), self._stream()), "dm_physics_step")

    def step_forced(self, qpos, qvel, out):
        self._chk(self.L.dm_step_forced(self.h, _ptr(qpos), _ptr(qvel), _ptr(out["obs"]), _ptr(out["rew"]),
                                        _ptr(out["done"]), _ptr(out.get("terms")), _ptr(out.get("reason")),
                                        self._stream()), "dm_step_forced")

    def set_state(self, qpos, qvel, warm=None, ctrl=None, env_ids=None, run_forward=False):
        n = qpos.shape[0]
        self._chk(self.L.dm_set_state(self.h, _ptr(env_ids), n, _ptr(qpos), _ptr(qvel), _ptr(warm), _ptr(ctrl),
                                      1 if run_forward else 0, self._stream()), "dm_set_state")

    def forward(self, env_ids=None, n=None):
        """sim.forward(): re-evaluate the derived quantities at the stored state."""
        self._chk(self.L.dm_forward(self.h, _ptr(env_ids), self.N if n is None else n, self._stream()), "dm_forward")

    def get_state(self, env_ids=None, n=None):
        t, d = self.torch, self.device
        n = self.N if n is None else n
        qpos, qvel = t.zeros(n, NQ, device=d), t.zeros(n, NV, device=d)
        warm, ctrl = t.zeros(n, NV, device=d), t.zeros(n, NU, device=d)
        self._chk(self.L.dm_get_state(self.h, _ptr(env_ids), n, _ptr(qpos), _ptr(qvel), _ptr(warm), _ptr(ctrl),
                                      self._stream()), "dm_get_state")
        return qpos, qvel, warm, ctrl

    def get_counters(self):
        t, d = self.torch, self.device
        idx = t.zeros(self.N, dtype=t.int32, device=d)
        ln = t.zeros(self.N, dtype=t.int32, device=d)
        rew = t.zeros(self.N, device=d)
        self._chk(self.L.dm_get_counters(self.h, _ptr(idx), _ptr(ln), _ptr(rew), self._stream()), "dm_get_counters")
        return idx, ln, rew

    def set_counters(self, idx=None, ep_len=None):
        self._chk(self.L.dm_set_counters(self.h, _ptr(idx), _ptr(ep_len), self._stream()), "dm_set_counters")

    def fill_random_actions(self, actions, step_index):
        self._chk(self.L.dm_fill_random_actions(self.h, _ptr(actions), int(step_index), self._stream()),
                  "dm_fill_random_actions")

    def get_work(self):
        w = self.torch.zeros(self.N, dtype=self.torch.int32, device=self.device)
        self._chk(self.L.dm_get_work(self.h, _ptr(w), self._stream()), "dm_get_work")
        return w

    def enable_timing(self, on=True, stride=1):
        """HIP event pair around every ``stride``-th step-kernel launch (ring of 512 pairs, read by mean_step_ms)."""
        self._chk(self.L.dm_enable_timing(self.h, max(1, int(stride)) if on else 0), "dm_enable_timing")

    def mean_step_ms(self):
        """(mean kernel ms, launches) over the steps since enable_timing(True); synchronises."""
        ms, n = C.c_float(0), C.c_int32(0)
        self._chk(self.L.dm_mean_step_ms(self.h, C.byref(ms), C.byref(n)), "dm_mean_step_ms")
        return ms.value, n.value

    def last_step_ms(self):
        ms = C.c_float(0)
        self._chk(self.L.dm_last_step_ms(self.h, C.byref(ms)), "dm_last_step_ms")
        return ms.value
