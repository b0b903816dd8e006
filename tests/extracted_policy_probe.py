"""Closed-loop probe of the restated physics with the reference's MuJoCo-trained walk policy.

TEST INFRASTRUCTURE (imports oracle/).  Protocol = src/play_extracted.py:27-44 of the reference:
``reset_model(idx_init=14)``, ``obs[:66]``, ``action = clip(pi.act(obs), -0.5, 0.5)``, at most 1000 steps, stop at
``done``.  The policy weights are the only artefact in the reference tree that carries information about real
MuJoCo's dynamics (they were optimised against it), so how long and how well the policy walks on the oracle is a
behavioural pin of the restatement.

Used by tests/test_extracted_policy_rollout.py (CPU and GPU legs) and tests/sensitivity_extracted_policy.py.
"""
from __future__ import annotations

import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NumpyPolicy:
    """a = tanh(tanh(o W0 + B0) W2 + B2) WA + BA  (src/extracted_policy.py:471-478), fp64 numpy."""

    def __init__(self, path=None):
        z = np.load(path or os.path.join(G, "policy_kat.npz"))
        self.W0, self.B0, self.W2, self.B2, self.WA, self.BA = (z[k].astype(np.float64) for k in
                                                                ("W0", "B0", "W2", "B2", "WA", "BA"))

    def act(self, obs):
        f = np.tanh(obs @ self.W0 + self.B0)
        f = np.tanh(f @ self.W2 + self.B2)
        return f @ self.WA + self.BA


def rollout(sim, clip, policy, idx_init=14, max_steps=1000, clip_act=0.5, stale_foot_bits=False, substeps=1,
            obs_hook=None, record=False):
    """play_extracted.py protocol on one OracleSim.  Returns a dict of summary statistics:
    steps survived, mean forward speed (m/s, root x), foot-contact alternation count, mean reward, done reason."""
    obs = sim.env_reset(clip, idx_init)
    x0 = sim.get("qpos")[0]
    dt = sim.model.timestep * substeps
    rf_prev = lf_prev = None
    switches = 0
    both = none = 0
    rew_sum = 0.0
    stale = np.zeros(2)
    zs, traj = [], []
    reason = 0
    steps = 0
    for i in range(max_steps):
        o = obs[:66].copy()
        if stale_foot_bits:  # F8: mujoco-py hands back all nconmax slots, stale ones included (deepmimic_env.py:88)
            stale = np.maximum(stale, o[64:66])
            o[64:66] = stale
        if obs_hook is not None:
            o = obs_hook(o)
        a = policy.act(o)
        if clip_act is not None:
            a = np.clip(a, -clip_act, clip_act)
        for _ in range(substeps - 1):  # frame_skip > 1 variant: hold ctrl, plain physics steps
            sim.set("ctrl", a)
            sim.step()
        obs, rew, done, terms, reason = sim.env_step(clip, a)
        steps = i + 1
        rew_sum += rew
        rf, lf = obs[64] > 0.5, obs[65] > 0.5
        both += rf and lf
        none += (not rf) and (not lf)
        stance = None if rf == lf else ("r" if rf else "l")
        if stance is not None:
            if rf_prev is not None and stance != rf_prev:
                switches += 1
            rf_prev = stance
        q = sim.get("qpos")
        zs.append(q[2])
        if record:
            traj.append(q.copy())
        if done:
            break
    q = sim.get("qpos")
    out = dict(steps=steps, speed=(q[0] - x0) / (steps * dt), stance_switches=switches, both_frac=both / steps,
               flight_frac=none / steps, mean_reward=rew_sum / steps, reason=reason, mean_root_z=float(np.mean(zs)),
               y_drift=float(q[1]))
    if record:
        out["traj"] = np.array(traj)
    return out
