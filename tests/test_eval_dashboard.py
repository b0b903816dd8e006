"""Eval dashboard (src/sb3_ppo.py:24-190) and the software renderer — host-side logic, no GPU."""
import os

import numpy as np


class _FakeSpace:
    shape = (5,)


class _FakeEnv:
    """10-step episodes with reward = step index / 10; render returns the stick figure of a two-link chain."""
    action_space = _FakeSpace()

    def __init__(self):
        self.t = 0

    def reset(self):
        self.t = 0
        return np.zeros(7)

    def step(self, action):
        self.t += 1
        done = self.t >= 10
        return np.full(7, self.t, float), self.t / 10.0, done, ({"done_reason": "max_ep_len"} if done else {})

    def render(self, mode=None):
        from deepmimic_mujoco_amd.render import stick_figure
        xpos = np.array([[0, 0, 0], [0.1 * self.t, 0, 0.9], [0.1 * self.t, 0.1, 0.5], [0.1 * self.t + 0.2, -0.1, 0.1]])
        return stick_figure(xpos, [0, 0, 1, 2])


class _FakePolicy:
    def predict_values(self, obs):
        import torch
        return torch.ones(obs.shape[0])


class _FakeModel:
    device = "cpu"
    policy = _FakePolicy()

    def __init__(self):
        self.num_timesteps = 0
        self.saved = []

    def predict(self, obs, deterministic=True):
        import torch
        return torch.zeros(obs.shape[0], 5)

    def save(self, path):
        self.saved.append(path)


def test_stick_figure_renderer():
    from deepmimic_mujoco_amd.render import stick_figure
    from deepmimic_mujoco_amd.model import load_model, forward_kinematics
    m = load_model()
    img = stick_figure(forward_kinematics(m, m.qpos0)["xpos"], m.body_parent)
    assert img.shape == (240, 320, 3) and img.dtype == np.uint8
    assert len(np.unique(img.reshape(-1, 3), axis=0)) >= 4            # background, floor, links, joints
    lying = m.qpos0.copy()
    lying[2], lying[3:7] = 0.15, [0.7071, 0, 0.7071, 0]
    img2 = stick_figure(forward_kinematics(m, lying)["xpos"], m.body_parent)
    dark = lambda a: np.nonzero((a.sum(2) < 400).any(1))[0]
    assert dark(img).min() < dark(img2).min() - 40                     # the standing figure reaches much higher in the frame


def test_eval_dashboard_rollout_and_callback(tmp_path):
    from deepmimic_mujoco_amd.eval_dashboard import EvalDashboardCallback, eval_dashboard_rollout
    model, env = _FakeModel(), _FakeEnv()
    ep_len, ep_rew = eval_dashboard_rollout(model, env, 1000, "run", out_root=str(tmp_path))
    assert ep_len == 10 and abs(ep_rew - 5.5) < 1e-12
    vd = tmp_path / "run_videos"
    assert (vd / "global_step_1000.gif").stat().st_size > 1000 and (vd / "rew_plot.png").exists() and (vd / "len_plot.png").exists()
    assert open(vd / "log.csv").read().splitlines() == ["global_step,ep_len,ep_rew", "1000,10,5.5"]
    assert model.saved == [os.path.join(str(vd), "run_best")]           # first episode is the best so far
    cb = EvalDashboardCallback(env, "run", every_n_global_steps=500, out_root=str(tmp_path), figures=False)
    for n in (1200, 1400, 1800, 2500):                                   # evaluates at 1200, 1800 and 2500, not at 1400
        model.num_timesteps = n
        assert cb(model) is True
    assert [h[0] for h in cb.history] == [1200, 1800, 2500]
    assert len(open(vd / "log.csv").read().splitlines()) == 5
