"""DPCombinedEnv (src/combined_env.py) on humanoid3d: oracle state machine (CPU) and HIP parity (GPU).

The reference holds no test for this class; the oracle restates combined_env.py:205-505 line by line and the GPU
path (dm_step_combined_kernel through the C-ABI) is compared with it teacher-forced: every step the GPU state,
motion id, n_steps and episode length are set from the oracle, both take the same action, and observation, reward,
the eight info terms, done, done_reason and the NEXT (motion, n_steps) must agree.
"""
import numpy as np
import pytest

WALK, RUN, GETUP, TO_GETUP = 0, 1, 2, 3


@pytest.fixture(scope="module")
def comb_mocaps(model):
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.mocap import MocapDM
    out = []
    for name in ["walk", "run", "getup_facedown"]:
        mc = MocapDM(model=model)
        mc.load_mocap(MotionConfig(name).mocap_path)
        out.append(mc)
    return out


@pytest.fixture(scope="module")
def comb_oracle_clips(comb_mocaps):
    from oracle.oracle import OracleClip
    return [OracleClip(*comb_mocaps[0].tables()), OracleClip(*comb_mocaps[1].tables()),
            OracleClip(*comb_mocaps[2].tables(), floor=True, acyclic=True)]


def _rollout(model, clips, seed, nsteps, scale, start=(WALK, 165)):
    """Free-running oracle episode(s); returns one record per step."""
    from oracle.oracle import OracleCombined
    rng = np.random.default_rng(seed)
    o = OracleCombined(model, clips)
    o.set_caps(32, 128)
    o.comb_reset(*start)
    recs = []
    for t in range(nsteps):
        a = rng.uniform(-scale, scale, 28)
        before = dict(qpos=o.get("qpos"), qvel=o.get("qvel"), warm=o.get("qacc_warmstart"), ctrl=o.get("ctrl"),
                      motion=o.cenv.motion, n_steps=o.cenv.n_steps, eplen=o.cenv.episode_length)
        obs, rew, done, terms, reason = o.comb_step(a)
        recs.append(dict(before=before, action=a, obs=obs, rew=rew, done=done, terms=terms, reason=reason,
                         motion=o.cenv.motion, n_steps=o.cenv.n_steps, qpos=o.get("qpos")))
        if done:
            if rng.integers(0, 2) == 0:
                o.comb_reset(WALK, 160 + int(rng.integers(0, clips[0].L)))
            else:
                o.comb_reset(GETUP, int(rng.integers(0, clips[2].L)))
    return recs


# ------------------------------------------------------------------------------------------ CPU: oracle semantics
def test_oracle_state_machine_walk_fall_getup_run(model, comb_oracle_clips):
    recs = _rollout(model, comb_oracle_clips, seed=0, nsteps=600, scale=2.0)
    motions = [r["before"]["motion"] for r in recs]
    # walk (with amnesty) -> falls -> to_getup -> getup -> run -> falls without amnesty -> done
    order = [m for i, m in enumerate(motions) if i == 0 or m != motions[i - 1]]
    assert order[:3] == [WALK, TO_GETUP, GETUP], order
    first_done = next(i for i, r in enumerate(recs) if r["done"])
    assert not any(r["done"] for r in recs[:first_done])
    # the getup clip times out with the body still on the floor: getup -> run (n_steps 0) -> fallen without
    # amnesty -> done and to_getup, all inside one step (:394-440)
    r = recs[first_done]
    assert r["reason"] == 7 and r["before"]["motion"] == GETUP and (r["motion"], r["n_steps"]) == (TO_GETUP, 1)
    # the fall out of walk happened with amnesty (n_steps > 150): no done, motion -> to_getup with n_steps 1
    i_fall = motions.index(TO_GETUP) - 1
    assert recs[i_fall]["before"]["n_steps"] > 150 and recs[i_fall]["motion"] == TO_GETUP and recs[i_fall]["n_steps"] == 1
    # to_getup: imitation reward forced to 0, task reward exp(-err/5)/3 in (0, 1/3]
    r = recs[motions.index(TO_GETUP) + 3]
    assert r["terms"][5] == 0.0 and 0.0 < r["terms"][6] <= 1.0 / 3.0
    assert abs(r["rew"] - 0.3 * r["terms"][6]) < 1e-15
    # to_getup times out after 180 steps -> getup ; getup times out at L-1 -> run (identity-compare quirk, :396)
    i_g = motions.index(GETUP)
    assert i_g - motions.index(TO_GETUP) <= 180
    assert first_done - i_g == comb_oracle_clips[2].L - 2    # enters with n_steps 1, times out at n_steps >= L-1
    # observation layout: phase, heading, one-hot walk, getup flags
    for r in recs[:first_done]:
        o = r["obs"]
        m = r["before"]["motion"]
        assert o.shape == (72,) and list(o[67:70]) == [1.0, 0.0, 0.0]
        assert (o[70], o[71]) == (float(m == TO_GETUP), float(m == GETUP))
        assert abs(o[65] ** 2 + o[66] ** 2 - 1.0) < 1e-12
        L = 180 if m == TO_GETUP else comb_oracle_clips[m].L
        assert abs(o[64] - (r["before"]["n_steps"] % L) / L) < 1e-15


def test_oracle_combined_playback_rewards(model, comb_mocaps, comb_oracle_clips):
    """force_state playback of the walk clip: imitation terms ~1 and task reward exactly 1 (root velocity matches)."""
    from oracle.oracle import OracleCombined
    o = OracleCombined(model, comb_oracle_clips)
    o.comb_reset(WALK, 160)
    q, v = comb_mocaps[0].tables()[:2]
    L = len(q)
    for t in range(40):
        n = o.cenv.n_steps
        obs, rew, done, terms, reason = o.comb_step(np.zeros(28), force_state=(q[n % L], v[n % L]))
        assert not done and reason == 0
        assert terms[6] == 1.0                      # exp(-10 * |0|)
        assert terms[1] == 1.0 and terms[0] > 1.0 - 2e-3
        assert abs(rew - (0.7 * terms[5] + 0.3)) < 1e-12
        assert o.cenv.motion == WALK and o.cenv.n_steps == n + 1


def _playback_rollout(model, mocaps, clips, nsteps=60):
    """getup played back through its end (upright) -> run; then the run clip played back: stays in run."""
    from oracle.oracle import OracleCombined
    o = OracleCombined(model, clips)
    o.comb_reset(GETUP, clips[2].L - 6)
    tabs = [m.tables()[:2] for m in mocaps]
    recs = []
    for t in range(nsteps):
        m, n = o.cenv.motion, o.cenv.n_steps
        q, v = tabs[m][0][n % clips[m].L], tabs[m][1][n % clips[m].L]
        before = dict(qpos=q, qvel=v, warm=o.get("qacc_warmstart"), ctrl=o.get("ctrl"), motion=m, n_steps=n,
                      eplen=o.cenv.episode_length)
        obs, rew, done, terms, reason = o.comb_step(np.zeros(28), force_state=(q, v))
        recs.append(dict(before=before, action=np.zeros(28), obs=obs, rew=rew, done=done, terms=terms, reason=reason,
                         motion=o.cenv.motion, n_steps=o.cenv.n_steps, qpos=o.get("qpos"), forced=True))
    return recs


def test_oracle_getup_hands_over_to_run(model, comb_mocaps, comb_oracle_clips):
    recs = _playback_rollout(model, comb_mocaps, comb_oracle_clips)
    motions = [r["before"]["motion"] for r in recs]
    assert motions[:6] == [GETUP] * 6 and set(motions[6:]) == {RUN}      # :396 quirk: always run, never walk
    assert recs[5]["motion"] == RUN and recs[5]["n_steps"] == 1 and not recs[5]["done"]
    assert not any(r["done"] for r in recs)
    for r in recs[7:]:
        assert r["terms"][6] == 1.0 and r["terms"][1] == 1.0             # run clip on itself


def test_oracle_max_episode_length(model, comb_mocaps, comb_oracle_clips):
    from oracle.oracle import OracleCombined
    o = OracleCombined(model, comb_oracle_clips)
    o.comb_reset(WALK, 160)
    q, v = comb_mocaps[0].tables()[:2]
    L = len(q)
    dones = []
    for t in range(2002):
        n = o.cenv.n_steps
        _, _, done, _, reason = o.comb_step(np.zeros(28), force_state=(q[n % L], v[n % L]))
        dones.append((done, reason))
        if done:
            break
    assert len(dones) == 2001 and dones[-1] == (True, 3)   # episode_length >= 2000 checked before the increment


# ------------------------------------------------------------------------------------------ GPU parity
def _gpu_engine(model, comb_mocaps, n, **kw):
    from deepmimic_mujoco_amd import _lib
    eng = _lib.HipEngine(model, n, task=_lib.TASK_COMBINED, max_ep_length=2000, **kw)
    eng.load_clip(0, comb_mocaps[0])
    eng.load_clip(1, comb_mocaps[1])
    eng.load_clip(2, comb_mocaps[2], floor=True, acyclic=True)
    return eng


@pytest.mark.gpu
@pytest.mark.parametrize("seed,scale,start,tile", [(0, 2.0, (WALK, 165), 1), (1, 0.5, (GETUP, 20), 1), (2, 2.0, (WALK, 200), 1),
                                                   (0, 2.0, (WALK, 165), 6)])
def test_gpu_combined_teacher_forced_parity(model, comb_mocaps, comb_oracle_clips, seed, scale, start, tile):
    """tile = 6: the 520 oracle states six times = 3 120 envs, which dm_step runs on dm_step_combined_kernel_w3 (three waves
    per SIMD from 3 072 envs): same gates as the two-wave kernel."""
    import torch
    recs = _rollout(model, comb_oracle_clips, seed=seed, nsteps=520, scale=scale, start=start) * tile
    n = len(recs)
    eng = _gpu_engine(model, comb_mocaps, n, auto_reset=False)
    assert eng.obs_dim == 72 and eng.terms_dim == 8
    dev = eng.device
    f32 = lambda k: torch.tensor(np.array([r["before"][k] for r in recs]), dtype=torch.float32, device=dev)
    i32 = lambda k: torch.tensor([r["before"][k] for r in recs], dtype=torch.int32, device=dev)
    eng.set_state(f32("qpos"), f32("qvel"), f32("warm"), f32("ctrl"))
    eng.set_env_clips(i32("motion"))
    eng.set_counters(i32("n_steps"), i32("eplen"))
    out = eng.alloc_outputs()
    eng.step(torch.tensor(np.array([r["action"] for r in recs]), dtype=torch.float32, device=dev), out)
    torch.cuda.synchronize()
    obs = out["obs"].cpu().numpy()
    rew = out["rew"].cpu().numpy()
    done = out["done"].cpu().numpy().astype(bool)
    reason = out["reason"].cpu().numpy()
    terms = out["terms"].cpu().numpy()
    motion = eng.get_env_clips().cpu().numpy()
    n_steps = eng.get_counters()[0].cpu().numpy()
    qpos = eng.get_state()[0].cpu().numpy()
    e_obs = np.abs(obs - np.array([r["obs"] for r in recs])).max(axis=1)
    e_rew = np.abs(rew - np.array([r["rew"] for r in recs]))
    e_terms = np.abs(terms[:, :7] - np.array([r["terms"][:7] for r in recs])).max(axis=1)
    e_qpos = np.abs(qpos - np.array([r["qpos"] for r in recs])).max(axis=1)
    o_motion = np.array([r["motion"] for r in recs])
    o_nsteps = np.array([r["n_steps"] for r in recs])
    o_done = np.array([r["done"] for r in recs])
    o_reason = np.array([r["reason"] for r in recs])
    bad = (motion != o_motion) | (n_steps != o_nsteps) | (done != o_done) | (reason != o_reason)
    seen = sorted(set(int(r["before"]["motion"]) for r in recs))
    trans = sorted(set((int(r["before"]["motion"]), int(r["motion"])) for r in recs if r["before"]["motion"] != r["motion"]))
    print("motions seen", seen, "transitions", trans, "decision mismatches", int(bad.sum()), "obs", e_obs.max(),
          "rew", e_rew.max(), "terms", e_terms.max(), "qpos", e_qpos.max())
    # discrete decisions (15/60 degree and z thresholds) are evaluated in fp32: allow boundary flips on <1% of steps
    assert bad.sum() <= 0.01 * n, np.nonzero(bad)[0][:10]
    ok = ~bad
    assert e_qpos[ok].max() < 1e-4
    assert e_obs[ok].max() < 2e-3
    assert e_rew[ok].max() < 1e-4
    assert e_terms[ok].max() < 2e-4
    nb = np.abs(terms[:, 7] - np.array([r["terms"][7] for r in recs]))
    assert (nb > 0).mean() < 0.02                      # n_bad_angles: integer count, boundary flips only
    if seed == 0:
        assert seen == [WALK, GETUP, TO_GETUP]
        assert (WALK, TO_GETUP) in trans and (TO_GETUP, GETUP) in trans and (GETUP, TO_GETUP) in trans
    if tile > 1:                                       # identical inputs in every tile -> identical outputs
        q = qpos.reshape(tile, -1)
        assert np.array_equal(q[0], q[tile // 2]) and np.array_equal(q[0], q[-1])
    eng.close()


@pytest.mark.gpu
def test_gpu_combined_forced_playback_parity(model, comb_mocaps, comb_oracle_clips):
    """dm_step_forced under the combined task: getup played to its end hands over to run (no done), run stays run."""
    import torch
    recs = _playback_rollout(model, comb_mocaps, comb_oracle_clips)
    n = len(recs)
    eng = _gpu_engine(model, comb_mocaps, n, auto_reset=False)
    dev = eng.device
    f32 = lambda k: torch.tensor(np.array([r["before"][k] for r in recs]), dtype=torch.float32, device=dev)
    i32 = lambda k: torch.tensor([r["before"][k] for r in recs], dtype=torch.int32, device=dev)
    eng.set_state(f32("qpos"), f32("qvel"), f32("warm"), f32("ctrl"))
    eng.set_env_clips(i32("motion"))
    eng.set_counters(i32("n_steps"), i32("eplen"))
    out = eng.alloc_outputs()
    eng.step_forced(f32("qpos"), f32("qvel"), out)
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_env_clips().cpu().numpy(), [r["motion"] for r in recs])
    assert np.array_equal(eng.get_counters()[0].cpu().numpy(), [r["n_steps"] for r in recs])
    assert not out["done"].cpu().numpy().any() and not out["reason"].cpu().numpy().any()
    assert np.abs(out["obs"].cpu().numpy() - np.array([r["obs"] for r in recs])).max() < 2e-3
    assert np.abs(out["rew"].cpu().numpy() - np.array([r["rew"] for r in recs])).max() < 1e-4
    assert np.abs(out["terms"].cpu().numpy()[:, :7] - np.array([r["terms"][:7] for r in recs])).max() < 2e-4
    eng.close()


@pytest.mark.gpu
def test_gpu_combined_vecenv_autoreset_and_mirror(model):
    """Batched env: RSI auto-reset draws walk(+amnesty) or getup; single-env mirror keeps the reference surface."""
    import torch
    from deepmimic_mujoco_amd.combined_env import DPCombinedEnv, HipCombinedVecEnv, NOBS_COMBINED
    N = 512
    env = HipCombinedVecEnv(N, robot="humanoid3d", seed=7)
    obs = env.reset()
    assert obs.shape == (N, NOBS_COMBINED)
    motion, n_steps = [t.cpu().numpy() for t in env.motion_state()]
    Lw, Lg = env.clips[0].get_length(), env.clips[2].get_length()
    assert set(np.unique(motion)) == {WALK, GETUP}
    w = motion == WALK
    assert 0.35 < w.mean() < 0.65
    assert n_steps[w].min() >= 160 and n_steps[w].max() < 160 + Lw and n_steps[~w].max() < Lg
    # phase obs of the reset state
    L = np.where(w, Lw, Lg)
    assert np.allclose(obs[:, 64], (n_steps % L) / L, atol=1e-6)
    assert np.all(obs[:, 71] == (~w).astype(np.float32)) and np.all(obs[:, 70] == 0)
    rng = np.random.default_rng(0)
    n_done = 0
    reasons = set()
    for t in range(260):
        obs, rew, done, infos = env.step(rng.uniform(-2, 2, (N, 28)).astype(np.float32))
        assert np.isfinite(obs).all() and np.isfinite(rew).all()
        for i in np.nonzero(done)[0][:4]:
            info = infos[int(i)]
            assert info["terminal_observation"].shape == (NOBS_COMBINED,)
            reasons.add(info.get("done_reason"))
            assert "imitation_reward" in info and "task_reward" in info
        n_done += int(done.sum())
        if done.any():
            m2, n2 = [t_.cpu().numpy() for t_ in env.motion_state()]
            assert set(np.unique(m2[done])) <= {WALK, GETUP}          # freshly reset envs
            assert np.all(n2[done & (m2 == WALK)] >= 160)
    assert n_done > 0 and "fallen without amnesty" in reasons
    m2, _ = [t_.cpu().numpy() for t_ in env.motion_state()]
    assert TO_GETUP in set(np.unique(m2))                              # amnesty falls land in to_getup
    env.close()

    e1 = DPCombinedEnv(robot="humanoid3d")
    o = e1.reset(rsi=False)
    assert o.shape == (NOBS_COMBINED,) and e1.current_motion_mocap is e1.getup_mocap and e1.current_motion_n_steps == 0
    assert o[71] == 1.0 and o[64] == 0.0
    q, v = e1.get_current_motion_state()
    o2, r, d, info = e1.step(np.zeros(28), force_state=(q, v))
    assert e1.current_motion_n_steps == 1 and e1.episode_length == 1 and not d
    assert set(info) >= {"reward_config", "imitation_reward", "task_reward"} and info["task_reward"] == 0.0
    e1.change_to_motion(e1.walk_mocap)
    e1.current_motion_n_steps = 200
    q, v = e1.get_current_motion_state()
    o3, r, d, info = e1.step(np.zeros(28), force_state=(q, v))
    assert abs(info["task_reward"] - 1.0) < 1e-6 and not d and e1.current_motion_mocap is e1.walk_mocap
    assert e1.current_motion_n_steps == 201
    e1.close()


@pytest.mark.gpu
def test_gpu_ppo_on_combined_env():
    """src/sb3_ppo.py trains `dp_combined_env` by default (:247-278): the learner takes the 72-d observation."""
    from deepmimic_mujoco_amd.combined_env import HipCombinedVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    venv = HipCombinedVecEnv(128, robot="humanoid3d", seed=3)
    ppo = PPO(venv, net_arch=(64, 32), n_steps=16, batch_size=512, n_epochs=2, learning_rate=3e-4)
    assert ppo.obs_dim == 72
    ppo.learn(2 * 16 * 128, log_interval=0)
    assert ppo.num_timesteps == 2 * 16 * 128 and np.isfinite(ppo.stats["loss"])
    venv.close()
