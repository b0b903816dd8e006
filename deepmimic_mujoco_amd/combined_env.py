"""Host-side mirror of the reference's `DPCombinedEnv` (src/combined_env.py:101-533) over the HIP engine.

The reference class is hard-wired to `unitree_g1` (:165); its walk / run / getup / to_getup motion state machine,
task reward, amnesty logic and player-action observation are model-independent, and this module runs them on the
`humanoid3d` model with the humanoid3d RobotConfig: no action scale (ACT_SCALE applies to G1 only,
:251), no extra-contact geoms (RobotConfig.extra_contact_geom_names is None), low_z 0.7, getup clip
`getup_facedown`.  All per-step arithmetic (physics, obs, both rewards, transitions, termination, RSI reset) runs in
`dm_step_combined_kernel` (DM_TASK_COMBINED); the classes below only keep the reference's public surface.
"""
from __future__ import annotations

import random

import numpy as np

from .deepmimic_env import LazyInfos as _LazyInfos

from . import _lib
from .config import MotionConfig, RobotConfig
from .deepmimic_env import Box, _SB3VecEnv, _SimView, _INFO_KEYS
from .mocap import MocapDM
from .model import load_model, NQ, NV, NU

NOBS_COMBINED = 72
MOTION_WALK, MOTION_RUN, MOTION_GETUP, MOTION_TO_GETUP = 0, 1, 2, 3
MOTION_NAMES = {0: "walk", 1: "run", 2: "getup", 3: "to_getup"}


class DPCombinedEnvConfig:
    """src/combined_env.py:20-34."""

    def __init__(self):
        self.MAX_EP_LENGTH = 2000
        self.VEL_OBS_SCALE = 0.1
        self.FRC_OBS_SCALE = 0.001
        self.ADD_FOOT_CONTACT_OBS = False
        self.ADD_EXTRA_CONTACT_OBS = True      # humanoid3d has no extra-contact geoms: contributes 0 entries
        self.ACT_SCALE = 20.                   # unitree_g1 only (:251)
        self.ADD_TORSO_OBS = True
        self.ADD_JOINT_FORCE_OBS = False
        self.ADD_ABSPOS_OBS = False
        self.ADD_PHASE_OBS = True
        self.ADD_PLAYER_ACTION_OBS = True
        self.MAX_PLAYER_ACTIONS = 3
        self.AMNESTY_STEPS = 150


class PlayerAction:
    """src/combined_env.py:37-56."""
    IDXS = {"walk": 0, "run": 1, "action": 2}

    def __init__(self, name, vx, vy):
        self.name, self.vx, self.vy = name, vx, vy
        assert name in PlayerAction.IDXS

    def onehot(self, n_max_actions):
        vec = np.zeros(n_max_actions)
        vec[PlayerAction.IDXS[self.name]] = 1.0
        return vec

    def heading_in_world(self):
        target_vel = np.linalg.norm([self.vx, self.vy])
        return np.array([self.vx, self.vy, 0]) / target_vel if target_vel != 0 else np.array([0, 0, 0])


class PAWalk(PlayerAction):
    def __init__(self):
        super().__init__("walk", 1.0, 0.0)


class PARun(PlayerAction):
    def __init__(self):
        super().__init__("run", 3.0, 0.0)


class MotionTransition:
    """Pseudo clip whose target is frame 1 of `target_mocap` for `length` steps (src/combined_env.py:67-99)."""
    motion_name = None
    length = None

    def __init__(self, target_mocap):
        self.target_mocap = target_mocap

    def get_qpos(self, index):
        return self.target_mocap.get_qpos(1)

    def get_qvel(self, index):
        return self.target_mocap.get_qvel(1)

    def get_body_xpos(self, index):
        return self.target_mocap.get_body_xpos(1)

    def get_geom_xpos(self, index):
        return self.target_mocap.get_geom_xpos(1)

    def get_length(self):
        return self.length


class MTToWalk(MotionTransition):
    motion_name, length = "to_walk", 120


class MTToRun(MotionTransition):
    motion_name, length = "to_run", 120


class MTToGetup(MotionTransition):
    motion_name, length = "to_getup", 180


class _Clip:
    """MocapDM plus the two attributes combined_env.py reads (`motion_name`, `get_length`)."""

    def __init__(self, model, robot, motion):
        self.mocap = MocapDM(robot=robot, model=model)
        self.mocap.load_mocap(MotionConfig(motion, robot).mocap_path)
        self.motion_name = motion

    def __getattr__(self, name):
        return getattr(self.mocap, name)

    def get_length(self):
        return len(self.mocap.data_config)


def _load_clips(engine, model, robot, getup_motion):
    clips = [_Clip(model, robot, "walk"), _Clip(model, robot, "run"), _Clip(model, robot, getup_motion)]
    for cid, c in enumerate(clips):
        mcfg = MotionConfig(c.motion_name, robot)
        engine.load_clip(cid, c.mocap, floor=c.motion_name in mcfg.floor_motions,
                         acyclic=c.motion_name in mcfg.acyclical_motions)
    return clips


def _combined_info(terms, reason):
    """info dict of combined_env.py:357-358,436,445 (+ the calc_imitation_reward keys); {} on the early-outs."""
    if reason in (5, 6):
        return {}
    info = {k: float(v) for k, v in zip(_INFO_KEYS, terms[:5])}
    info["imitation_reward"] = float(terms[5])
    info["task_reward"] = float(terms[6])
    r = _lib.REASONS.get(int(reason))
    if r is not None:
        info["done_reason"] = r
    return info


class DPCombinedEnv:
    """Single-env surface of src/combined_env.py:101 (ctor args, step/reset/get_current_motion_state/change_to_motion)."""

    version = "v0.2.up"
    ENV_CFG = DPCombinedEnvConfig()
    metadata = {"render.modes": []}

    def __new__(cls, verbose=0, _profile=False, robot="unitree_g1", getup_motion="getup_facedown", device=0):
        if robot == "unitree_g1" and cls is DPCombinedEnv:   # the reference's own configuration (:165): the G1 engine
            from .g1 import G1CombinedEnv
            return G1CombinedEnv(verbose=verbose, _profile=_profile, device=device)
        return super().__new__(cls)

    def __init__(self, verbose=0, _profile=False, robot="unitree_g1", getup_motion="getup_facedown", device=0):
        import torch
        self._torch = torch
        self.PROFILE = _profile
        self.verbose = verbose
        self.robot = robot
        self.robot_config = RobotConfig(robot=robot)
        self.model = load_model(self.robot_config.xml_path)
        cfg = self.ENV_CFG
        self._eng = _lib.HipEngine(self.model, 1, device=device, auto_reset=False, task=_lib.TASK_COMBINED,
                                   max_ep_length=cfg.MAX_EP_LENGTH, vel_obs_scale=cfg.VEL_OBS_SCALE,
                                   low_z=self.robot_config.low_z, amnesty_steps=cfg.AMNESTY_STEPS,
                                   to_getup_len=MTToGetup.length)
        self.walk_mocap, self.run_mocap, self.getup_mocap = _load_clips(self._eng, self.model, robot, getup_motion)
        self.action_mocap = None
        self.to_getup_mocap = MTToGetup(self.getup_mocap)
        self._motions = [self.walk_mocap, self.run_mocap, self.getup_mocap, self.to_getup_mocap]
        self._out = self._eng.alloc_outputs()
        self.episode_reward = 0
        self.episode_length = 0
        self.debug_n_bad_angles = 0
        self.current_motion_n_steps = None
        self.current_motion_mocap = None
        self.current_player_action = None
        self._time = 0.0
        self.sim = _SimView(self)
        lo = self.model.act_ctrlrange[:, 0].astype(np.float32)
        hi = self.model.act_ctrlrange[:, 1].astype(np.float32)
        self.action_space = Box(lo, hi, dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, (NOBS_COMBINED,), np.float64)

    # ---- helpers
    def _i32(self, v):
        t = self._torch
        return t.tensor([int(v)], dtype=t.int32, device=self._eng.device)

    def _push(self):
        self._eng.set_env_clips(self._i32(self._motions.index(self.current_motion_mocap)))
        self._eng.set_counters(self._i32(self.current_motion_n_steps), self._i32(self.episode_length))

    def _pull(self):
        self.current_motion_mocap = self._motions[int(self._eng.get_env_clips()[0].item())]
        idx, ln, rew = self._eng.get_counters()
        self.current_motion_n_steps = int(idx[0].item())
        self.episode_length = int(ln[0].item())          # advanced in the kernel (:454-460)
        self.episode_reward = float(rew[0].item())

    def _state(self):
        q, v, _, _ = self._eng.get_state()
        return q[0].double().cpu().numpy(), v[0].double().cpu().numpy()

    def get_current_motion_state(self):                                        # :199-203
        idx = self.current_motion_n_steps % self.current_motion_mocap.get_length()
        return self.current_motion_mocap.get_qpos(idx) * 1.0, self.current_motion_mocap.get_qvel(idx) * 1.0

    def change_to_motion(self, motion):                                        # :529-533
        if self.verbose:
            print("Changing to motion: {}".format(motion.motion_name))
        self.current_motion_mocap = motion
        self.current_motion_n_steps = 0

    # ---- gym.Env surface
    def reset(self, rsi=True):                                                 # :205-241
        if rsi:
            if random.randint(0, 1) == 0:
                self.current_motion_mocap = self.walk_mocap
                self.current_motion_n_steps = self.ENV_CFG.AMNESTY_STEPS + 10 + \
                    random.randint(0, self.current_motion_mocap.get_length() - 1)
            else:
                self.current_motion_mocap = self.getup_mocap
                self.current_motion_n_steps = random.randint(0, self.current_motion_mocap.get_length() - 1)
        else:
            self.current_motion_mocap = self.getup_mocap
            self.current_motion_n_steps = 0
        self.current_player_action = PAWalk()
        self.episode_reward = 0
        self.episode_length = 0
        t = self._torch
        self._eng.set_env_clips(self._i32(self._motions.index(self.current_motion_mocap)))
        obs = t.zeros(1, NOBS_COMBINED, device=self._eng.device)
        self._eng.reset(obs, idx_init=self._i32(self.current_motion_n_steps))
        return obs[0].double().cpu().numpy()

    def step(self, action, force_state=None):                                  # :243-493
        t = self._torch
        action = np.asarray(action, np.float64) * 1.0
        assert action.shape == (NU,)
        self._push()
        if force_state is not None:
            qpos, qvel = force_state
            self._eng.step_forced(t.tensor(np.asarray(qpos)[None], dtype=t.float32, device=self._eng.device),
                                  t.tensor(np.asarray(qvel)[None], dtype=t.float32, device=self._eng.device), self._out)
        else:
            self._eng.step(t.tensor(action[None], dtype=t.float32, device=self._eng.device), self._out)
            self._time += self.model.timestep
        obs = self._out["obs"][0].double().cpu().numpy()
        reason = int(self._out["reason"][0].item())
        done = bool(self._out["done"][0].item())
        if reason == 5:                                                        # :271-284 (no counter moves)
            return obs, 0, True, {}
        self._pull()
        if reason == 6:                                                        # :472-484 (counters already advanced)
            return obs, 0, True, {}
        terms = self._out["terms"][0].cpu().numpy()
        self.debug_n_bad_angles = int(terms[7])
        return obs, float(self._out["rew"][0].item()), done, _combined_info(terms, reason)

    def set_state(self, qpos, qvel):
        t = self._torch
        assert np.shape(qpos) == (NQ,) and np.shape(qvel) == (NV,)
        self._eng.set_state(t.tensor(np.asarray(qpos)[None], dtype=t.float32, device=self._eng.device),
                            t.tensor(np.asarray(qvel)[None], dtype=t.float32, device=self._eng.device), run_forward=True)

    def render(self, mode=None):
        """Software stick figure (render.py) of the current body poses; the warm start is put back (rendering is not physics)."""
        from .render import stick_figure
        if self._eng._debug is None:
            self._eng.enable_debug()
        q, v, w, c = self._eng.get_state()
        self._eng.forward()
        self._eng.set_state(q, v, warm=w, ctrl=c, run_forward=False)
        xpos = self._eng._debug[0, :42].double().cpu().numpy().reshape(14, 3)
        return stick_figure(xpos, self.model.body_parent)

    def seed(self, seed=None):
        random.seed(seed)
        return [seed]

    def close(self):
        self._eng.close()


class _LazyCombinedInfos(_LazyInfos):
    """`infos` of a DPCombinedEnv batch step: the list-like lazy container of deepmimic_env.LazyInfos (slices, item assignment,
    ``isinstance(infos, list)``) with this env's info dict (imitation terms, imitation_reward, task_reward, done_reason)."""

    _make = staticmethod(lambda terms, reason: _combined_info(terms, reason))


class HipCombinedVecEnv(_SB3VecEnv):
    """N DPCombinedEnv instances as one HIP batch with SubprocVecEnv semantics (auto-reset = reset(rsi=True))."""

    def __new__(cls, num_envs, robot="unitree_g1", getup_motion="getup_facedown", device=0, seed=1234, auto_reset=True, **kw):
        if robot == "unitree_g1" and cls is HipCombinedVecEnv:
            from .g1 import HipG1CombinedVecEnv
            return HipG1CombinedVecEnv(num_envs, device=device, seed=seed, auto_reset=auto_reset, sub_batches=kw.get("sub_batches", 1))
        return super().__new__(cls)

    def __init__(self, num_envs, robot="unitree_g1", getup_motion="getup_facedown", device=0, seed=1234,
                 auto_reset=True):
        import torch
        self._torch = torch
        self.robot_config = RobotConfig(robot)
        self.model = load_model(self.robot_config.xml_path)
        self.num_envs = int(num_envs)
        cfg = DPCombinedEnv.ENV_CFG
        self.engine = _lib.HipEngine(self.model, self.num_envs, device=device, seed=seed, auto_reset=auto_reset,
                                     task=_lib.TASK_COMBINED, max_ep_length=cfg.MAX_EP_LENGTH,
                                     vel_obs_scale=cfg.VEL_OBS_SCALE, low_z=self.robot_config.low_z,
                                     amnesty_steps=cfg.AMNESTY_STEPS, to_getup_len=MTToGetup.length)
        self.clips = _load_clips(self.engine, self.model, robot, getup_motion)
        self.device = self.engine.device
        self.out = self.engine.alloc_outputs()
        lo = self.model.act_ctrlrange[:, 0].astype(np.float32)
        hi = self.model.act_ctrlrange[:, 1].astype(np.float32)
        self.action_space = Box(lo, hi, dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, (NOBS_COMBINED,), np.float32)
        self._actions = torch.zeros(self.num_envs, NU, device=self.device)
        self.version, self.ENV_CFG = DPCombinedEnv.version, cfg
        if _SB3VecEnv is not object:  # pragma: no cover
            _SB3VecEnv.__init__(self, self.num_envs, self.observation_space, self.action_space)

    def reset_tensor(self):
        self.engine.reset(self.out["obs"])
        return self.out["obs"]

    def step_tensor(self, actions):
        self.engine.step(actions.contiguous(), self.out)
        return self.out

    def motion_state(self):
        """(motion id int32[N], n_steps int32[N]) — `current_motion_mocap` / `current_motion_n_steps` per env."""
        return self.engine.get_env_clips(), self.engine.get_counters()[0]

    def reset(self):
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        t = self._torch
        self._actions.copy_(t.as_tensor(np.asarray(actions), dtype=t.float32))

    def step_wait(self):
        # one packed download per step, as HipDeepMimicVecEnv.step_wait
        t = self._torch
        out = self.step_tensor(self._actions)
        packed = t.cat([out["obs"], out["terminal_obs"], out["terms"], out["rew"][:, None], out["done"][:, None].float(),
                        out["reason"][:, None].float()], dim=1).cpu().numpy()
        d, k = out["obs"].shape[1], out["terms"].shape[1]
        done = packed[:, 2 * d + k + 1] != 0
        infos = _LazyCombinedInfos(np.ascontiguousarray(packed[:, 2 * d:2 * d + k]), packed[:, 2 * d + k + 2].astype(np.int32), done,
                                   np.ascontiguousarray(packed[:, d:2 * d]))
        return np.ascontiguousarray(packed[:, 0:d]), packed[:, 2 * d + k].copy(), done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.engine.close()

    def seed(self, seed=None):
        """SB3 VecEnv.seed: re-keys the engine's counter-based reset generator; returns the per-env seeds SB3 expects."""
        if seed is None:
            return [None] * self.num_envs
        self.engine.set_seed(int(seed))
        return [int(seed) + i for i in range(self.num_envs)]

    def _n_indices(self, indices):
        return self.num_envs if indices is None else len(np.atleast_1d(indices))

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * self._n_indices(indices)

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """One batch object stands for all envs (as in HipDeepMimicVecEnv): the method runs once on it."""
        return [getattr(self, method_name)(*method_args, **method_kwargs)] * self._n_indices(indices)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self._n_indices(indices)

    def getattr_depth_check(self, name, already_found):
        return None

    @property
    def unwrapped(self):
        return self

    def get_images(self):
        return [self.render(mode="rgb_array")]

    def render(self, mode=None):
        """Software stick figure of env 0 of the batch (render.py); the warm start is put back (rendering is not physics)."""
        from .render import stick_figure
        e = self.engine
        if e._debug is None:
            e.enable_debug()
        q, v, w, c = e.get_state()
        e.forward()
        e.set_state(q, v, warm=w, ctrl=c, run_forward=False)
        xpos = e._debug[0, :42].double().cpu().numpy().reshape(14, 3)
        return stick_figure(xpos, self.model.body_parent)
