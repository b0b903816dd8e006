"""The reference's trained walk policy, closed loop (src/play_extracted.py:27-44), as a behavioural pin of the physics.

`src/extracted_policy.py` holds a policy that was optimised against real MuJoCo; it is the only artefact in the
reference tree that carries information about MuJoCo's dynamics.  CPU leg: the play_extracted.py protocol on the fp64
oracle — the policy must walk (forward speed of the walk clip's order, alternating single-stance phases, pelvis at
walking height) far longer than the passive body stands.  GPU leg: the same loop on the HIP engine with the policy
evaluated by `dm_policy_forward` (D = 66: the GPU known-answer test of SURVEY a15), teacher-forced against the oracle
per step, and free-running for the regime statistics.

Measured survival (oracle, start frames 14/0/30/50/60): 116/235/87/1000/133 steps; the one-ingredient-at-a-time
sensitivity table is in DESIGN.md §2 (tests/sensitivity_extracted_policy.py).
"""
import os

import numpy as np
import pytest

from extracted_policy_probe import G, NumpyPolicy, rollout

FRAMES = [14, 0, 30, 50, 60]


def test_numpy_policy_known_answer():
    z = np.load(os.path.join(G, "policy_kat.npz"))
    pol = NumpyPolicy()
    assert np.allclose(pol.act(z["kat_obs"]), z["kat_expected"], rtol=1e-5, atol=1e-5)   # extracted_policy.py:480-485


def test_extracted_policy_walks_on_oracle(model, oracle_clips):
    from oracle.oracle import OracleSim
    clip, pol = oracle_clips["walk"], NumpyPolicy()
    res = {i: rollout(OracleSim(model), clip, pol, i) for i in FRAMES}
    for i, r in res.items():
        print(i, {k: (round(float(v), 3) if not isinstance(v, int) else v) for k, v in r.items()})
    # passive control: zero torques collapse within half a second
    class Zero:
        def act(self, o):
            return np.zeros(28)
    passive = rollout(OracleSim(model), clip, Zero(), 14)
    assert passive["steps"] < 40
    surv = np.array([r["steps"] for r in res.values()])
    assert surv.min() >= 2 * passive["steps"], surv            # every start: the policy balances far beyond passive
    assert np.median(surv) >= 100, surv                          # ~2 s of walking (measured 116/235/87/1000/133)
    r14 = res[14]                                                # the start frame of play_extracted.py:31
    assert 0.6 < r14["speed"] < 1.3                              # walk clip: 0.97 m/s
    assert r14["stance_switches"] >= 5                           # alternating single-stance phases
    assert r14["flight_frac"] < 0.1 and r14["both_frac"] < 0.5
    assert 0.82 < r14["mean_root_z"] < 0.9                       # pelvis at walking height (mocap 0.85..0.88)


@pytest.mark.gpu
def test_policy_forward_known_answer_on_gpu():
    """SURVEY a15 on the HIP path: dm_policy_forward with D = 66 fed the reference's weights reproduces the reference's
    own known-answer pair (src/extracted_policy.py:480-485) and 16 more (obs, action) pairs; ragged N = 17."""
    import torch
    pol, fwd, dev = _hip_policy(torch)
    z = np.load(os.path.join(G, "policy_kat.npz"))
    obs = torch.tensor(np.concatenate([z["kat_obs"], z["extra_obs"]]), dtype=torch.float32, device=dev)
    want = np.concatenate([z["kat_expected"], z["extra_act"]])
    got = _hip_act(torch, fwd, obs, clip=None).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-4, atol=2e-5), np.abs(got - want).max()


def _hip_policy(torch):
    from deepmimic_mujoco_amd.ppo import FusedPolicyForward, MlpPolicy
    dev = torch.device("cuda", 0)
    z = np.load(os.path.join(G, "policy_kat.npz"))
    pol = MlpPolicy(obs_dim=66, net_arch=(256, 128)).to(dev)
    with torch.no_grad():
        for lin, w, b in ((pol.pi[0], "W0", "B0"), (pol.pi[2], "W2", "B2"), (pol.action_net, "WA", "BA")):
            lin.weight.copy_(torch.tensor(z[w].T.copy()))
            lin.bias.copy_(torch.tensor(z[b]))
    assert FusedPolicyForward.supported(pol, dev)
    fwd = FusedPolicyForward(pol, dev)
    fwd.pack()
    return pol, fwd, dev


def _hip_act(torch, fwd, obs66, clip=0.5):
    """Deterministic head of dm_policy_forward: act = mean, act_env = clamp(mean, lo, hi)."""
    dev, n = obs66.device, obs66.shape[0]
    big = 1e30 if clip is None else clip
    lo, hi = torch.full((28,), -big, device=dev), torch.full((28,), big, device=dev)
    act, env_act = torch.zeros(n, 28, device=dev), torch.zeros(n, 28, device=dev)
    logp, val = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    fwd(obs66.contiguous(), 0, ctr, 0, lo, hi, act, env_act, logp, val, deterministic=True)
    return env_act


@pytest.mark.gpu
def test_extracted_policy_closed_loop_on_hip_teacher_forced(model, clips, oracle_clips):
    """play_extracted.py's loop with every piece on the GPU (dm_policy_forward -> clip +-0.5 -> dm_step), one env per
    start frame, teacher-forced: before each step the HIP env is set to the oracle's state, both take the action the
    HIP policy computed from the oracle's observation, results compared (qpos 1e-4, obs, reward, done).  300 steps per
    start frame, following the oracle through its falls (the oracle env restarts at the next start frame)."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    from oracle.oracle import OracleSim
    pol, fwd, dev = _hip_policy(torch)
    npol = NumpyPolicy()
    oc = oracle_clips["walk"]
    n = len(FRAMES)
    eng = HipEngine(model, n, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    sims = [OracleSim(model) for _ in range(n)]
    for s in sims:
        s.set_caps(32, 128)
    obs = np.array([s.env_reset(oc, f) for s, f in zip(sims, FRAMES)])
    eng.reset(out["obs"], idx_init=torch.tensor(FRAMES, dtype=torch.int32, device=dev))
    nxt = list(FRAMES)
    worst = dict(act=0.0, qpos=0.0, qvel=0.0, obs=0.0, rew=0.0)
    falls = flips = 0
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    for t in range(300):
        a_hip = _hip_act(torch, fwd, f32(obs[:, :66]))
        a_np = np.clip(npol.act(obs[:, :66].astype(np.float32).astype(np.float64)), -0.5, 0.5)
        worst["act"] = max(worst["act"], float(np.abs(a_hip.cpu().numpy() - a_np).max()))
        eng.set_state(f32([s.get("qpos") for s in sims]), f32([s.get("qvel") for s in sims]),
                      f32([s.get("qacc_warmstart") for s in sims]), f32([s.get("ctrl") for s in sims]))
        eng.set_counters(torch.tensor([s.env.idx_curr for s in sims], dtype=torch.int32, device=dev),
                         torch.tensor([s.env.episode_length for s in sims], dtype=torch.int32, device=dev))
        eng.step(a_hip, out)
        torch.cuda.synchronize()
        a_used = a_hip.cpu().numpy().astype(np.float64)
        qpos, qvel, _, _ = [x.cpu().numpy() for x in eng.get_state()]
        gobs, grew, gdone = out["obs"].cpu().numpy(), out["rew"].cpu().numpy(), out["done"].cpu().numpy()
        for i, s in enumerate(sims):
            o, r, d, terms, reason = s.env_step(oc, a_used[i])
            stage = [s.geti("stage_nefc%d" % k) for k in range(4)]
            e_q = float(np.abs(qpos[i] - s.get("qpos")).max())
            if e_q > 1e-4 and len(set(stage)) > 1:
                flips += 1      # a contact at its activation margin to fp32 rounding inside an RK stage
            else:
                worst["qpos"] = max(worst["qpos"], e_q)
                worst["qvel"] = max(worst["qvel"], float(np.abs(qvel[i] - s.get("qvel")).max()))
                worst["obs"] = max(worst["obs"], float(np.abs(gobs[i] - o).max()))
                worst["rew"] = max(worst["rew"], abs(float(grew[i]) - r))
                assert bool(gdone[i]) == d, (t, i)
            obs[i] = o
            if d:
                falls += 1
                nxt[i] = (nxt[i] + 7) % oc.L
                obs[i] = s.env_reset(oc, nxt[i])
    print("teacher-forced closed loop:", worst, "falls", falls, "activation flips", flips)
    assert worst["act"] < 2e-5 and worst["qpos"] < 1e-4 and worst["qvel"] < 5e-3
    assert worst["obs"] < 2e-3 and worst["rew"] < 1e-4
    assert flips <= 0.01 * 300 * n
    eng.close()


@pytest.mark.gpu
def test_extracted_policy_free_running_on_hip(model, clips):
    """The same protocol free-running on the HIP engine alone (no oracle in the loop), all 76 start frames as one
    batch: the fp32 path shows the regime the fp64 oracle shows (median survival >= 100 steps, walking speed)."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    pol, fwd, dev = _hip_policy(torch)
    L = clips["walk"].tables()[0].shape[0]
    eng = HipEngine(model, L, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    eng.reset(out["obs"], idx_init=torch.arange(L, dtype=torch.int32, device=dev))
    x0 = eng.get_state()[0][:, 0].clone()
    alive = torch.ones(L, dtype=torch.bool, device=dev)
    steps = torch.zeros(L, dtype=torch.int32, device=dev)
    x_end = x0.clone()
    for t in range(1000):
        a = _hip_act(torch, fwd, out["obs"][:, :66])
        eng.step(a, out)
        steps += alive.int()
        x = eng.get_state()[0][:, 0]
        x_end = torch.where(alive, x, x_end)
        alive &= out["done"] == 0
        if not bool(alive.any()):
            break
    steps = steps.cpu().numpy()
    speed = ((x_end - x0).cpu().numpy() / (steps * model.timestep))
    print("HIP free-running: survival quantiles", np.percentile(steps, [0, 25, 50, 75, 100]), "cap frac",
          float((steps >= 1000).mean()), "median speed", float(np.median(speed)))
    assert np.median(steps) >= 100
    assert 0.3 < np.median(speed) < 1.3
    eng.close()
