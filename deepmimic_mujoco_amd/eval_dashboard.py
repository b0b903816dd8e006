"""Evaluation dashboard — `eval_dashboard_rollout` / `EvalDashboardCallback` of the reference (src/sb3_ppo.py:24-140, :143-190).

Every N global steps one deterministic episode is collected on an eval env; per step a four-panel figure (actions, rendered
frame, reward / return / value curves, observation) is written, the frames are assembled into an animation, the episode is
appended to `log.csv`, reward / length plots are refreshed and the model is saved when it is the best so far.  Differences
from the reference, all host-side: the rendered frame is the stick figure of `render.py` (no MuJoCo viewer), the animation is
a GIF written with PIL (no OpenCV in this image), wandb logging only if the module imports.  SURVEY §8f-4.
"""
from __future__ import annotations

import os

import numpy as np


def eval_dashboard_rollout(model, eval_env, n, run_name, log_wandb=False, out_root=None, max_steps=None, figures=True):
    """Collect one episode with the deterministic policy and write the dashboard; returns (ep_len, ep_rew)."""
    import torch as th
    buffer = []
    obs = eval_env.reset()
    ep_rew = 0.0
    while True:
        try:
            with th.no_grad():
                action = model.predict(obs.reshape(1, -1), deterministic=True)[0].cpu().numpy()
                val = float(model.policy.predict_values(th.as_tensor(obs.reshape(1, -1), dtype=th.float32, device=model.device))[0])
        except Exception:   # the reference swallows predictor errors the same way (:41-42)
            action, val = np.zeros(eval_env.action_space.shape[0], np.float32), 0.0
        try:
            frame = eval_env.render(mode="rgb_array")
        except NotImplementedError:
            frame = None
        obs, rewards, dones, info = eval_env.step(action)
        ep_rew += rewards
        buffer.append((np.array(obs), np.array(action), float(rewards), bool(dones), dict(info), ep_rew * 1.0, val, frame))
        if dones or (max_steps is not None and len(buffer) >= max_steps):
            break
    out_root = os.path.expanduser(out_root or "~/deep_mimic")
    video_dir = os.path.join(out_root, run_name + "_videos")
    os.makedirs(video_dir, exist_ok=True)
    if figures:
        _write_dashboard(buffer, os.path.join(video_dir, "global_step_%d.gif" % n))
    ep_len = len(buffer)
    log_path = os.path.join(video_dir, "log.csv")
    if not os.path.exists(log_path):
        with open(log_path, "w") as f:
            f.write("global_step,ep_len,ep_rew\n")
    with open(log_path, "a") as f:
        f.write("%d,%d,%r\n" % (n, ep_len, ep_rew))
    log = np.loadtxt(log_path, delimiter=",", skiprows=1).reshape(-1, 3)
    if figures:
        _plot_log(log, video_dir)
    if log_wandb:
        try:
            import wandb
            wandb.log({"eval_episode_length": ep_len, "eval_episode_reward": ep_rew, "eval_global_step": n,
                       "eval_best_episode_reward": float(np.max(log[:, 2])),
                       "eval_best_episode_global_step": int(log[np.argmax(log[:, 2]), 0])})
        except ImportError:
            pass
    if np.max(log[:, 2]) == log[-1, 2]:                      # save model if best (:136-137)
        model.save(os.path.join(video_dir, run_name + "_best"))
    print("Eval: LEN {}, EP_REW {}".format(ep_len, ep_rew))
    return ep_len, ep_rew


def _write_dashboard(buffer, path):
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    from PIL import Image
    frames = []
    stride = max(1, len(buffer) // 120)                       # cap the animation at ~120 figures
    for i in range(0, len(buffer), stride):
        obs, action, rewards, dones, info, ep_rew, val, frame = buffer[i]
        fig, ax = plt.subplots(2, 2, num="eval", figsize=(8, 6), dpi=60)
        ar = np.arange(len(action))
        ax[0, 0].axhline(0, color="black", lw=1)
        for di in range(-5, 0):
            if i + di >= 0:
                ax[0, 0].step(ar, buffer[i + di][1], where="mid", alpha=0.5, color="grey")
        ax[0, 0].step(ar, action, where="mid")
        if frame is not None:
            ax[0, 1].imshow(frame)
        ax[0, 1].axis("off")
        ax[1, 0].axhline(0, color="black", lw=1)
        ax[1, 0].plot([x[5] for x in buffer[:i + 1]])
        ax[1, 0].plot([x[2] for x in buffer[:i + 1]])
        ax[1, 0].plot([x[6] for x in buffer[:i + 1]])
        if "done_reason" in info:
            ax[1, 0].set_title(str(info["done_reason"]))
        ax[1, 1].axhline(0, color="black", lw=1)
        orng = np.arange(len(obs))
        for di in range(-5, 0):
            if i + di >= 0:
                ax[1, 1].step(orng, buffer[i + di][0], where="mid", alpha=0.5, color="grey")
        ax[1, 1].step(orng, obs, where="mid")
        fig.canvas.draw()
        frames.append(Image.fromarray(np.asarray(fig.canvas.buffer_rgba())[:, :, :3].copy()))
        plt.close(fig)
    frames[0].save(path, save_all=True, append_images=frames[1:], duration=42 if len(frames) > 10 else 1000, loop=0)
    print("Saved animation to", path)


def _plot_log(log, video_dir):
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    for col, name in ((2, "rew_plot.png"), (1, "len_plot.png")):
        fig, ax = plt.subplots(1, 1)
        ax.plot(log[:, 0], log[:, col])
        ax.set_xlabel("Global Step")
        fig.savefig(os.path.join(video_dir, name))
        plt.close(fig)


class EvalDashboardCallback:
    """`callback=` of `PPO.learn`: evaluate every `every_n_global_steps` (src/sb3_ppo.py:143-190; called once per PPO iteration)."""

    def __init__(self, eval_env, run_name, log_wandb=False, every_n_global_steps=1_000_000, out_root=None, figures=True, max_steps=None):
        self.eval_env, self.run_name, self.log_wandb = eval_env, run_name, log_wandb
        self.every, self.out_root, self.figures, self.max_steps = every_n_global_steps, out_root, figures, max_steps
        self.last_eval = None
        self.history = []

    def __call__(self, model):
        n = int(model.num_timesteps)
        if self.last_eval is None or n - self.last_eval >= self.every:
            self.last_eval = n
            self.history.append((n,) + eval_dashboard_rollout(model, self.eval_env, n, self.run_name, log_wandb=self.log_wandb,
                                                              out_root=self.out_root, max_steps=self.max_steps, figures=self.figures))
        return True
