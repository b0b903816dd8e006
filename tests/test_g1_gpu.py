"""Unitree G1 engine on the GPU (csrc/dm_g1.hip through the dmg1_* C-ABI) against the fp64 G1 oracle."""
import numpy as np
import pytest

from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.mocap import MocapDM

pytestmark = pytest.mark.gpu

MOTIONS = {}


def _clip(motion):
    if motion not in MOTIONS:
        mc = MocapDM(robot="unitree_g1")
        mc.load_mocap(MotionConfig(motion, robot="unitree_g1").mocap_path)
        MOTIONS[motion] = mc
    return MOTIONS[motion]


def _states(n, seed=0):
    """n test states: frames of the walk / getup clips pressed a little into the floor, random joint velocities, and a few
    free-flight poses with self contacts."""
    from oracle import oracle_g1 as og
    g, _ = og.g1_model()
    rng = np.random.default_rng(seed)
    qs, vs = [], []
    for i in range(n):
        kind = i % 4
        if kind in (0, 1):
            mc = _clip("walk")
            fr = int(rng.integers(len(mc.data_config)))
            q, v = np.array(mc.data_config[fr]), np.array(mc.data_vel[fr])
            q[2] -= rng.uniform(0.0, 0.02)
        elif kind == 2:
            mc = _clip("getup_facedown")
            fr = int(rng.integers(0, 90))
            q, v = np.array(mc.data_config[fr]), np.array(mc.data_vel[fr])
        else:
            q = g.qpos0.copy()
            q[2] = 1.5
            q[3:7] = rng.normal(size=4)
            q[7:] = rng.uniform(-0.6, 0.6, 37)
            v = rng.normal(size=43)
        q[3:7] /= np.linalg.norm(q[3:7])
        v = v + rng.normal(size=43) * 0.2
        qs.append(q)
        vs.append(v)
    return np.array(qs), np.array(vs)


def test_g1_forward_evaluation_matches_the_oracle():
    """set_state + forward on 32 states: body poses, unconstrained acceleration, the contact list (geoms, distance, position,
    normal, in order), the row count and the constrained acceleration."""
    q, v = _states(32)
    ncons, nefcs = _forward_parity(q, v, "G1 forward parity:")
    assert max(ncons) >= 8 and min(ncons) == 0 or max(ncons) >= 8


def test_g1_forward_evaluation_with_more_than_128_rows():
    """Robots pressed face-down into the floor with folded limbs: 20+ contacts, 129..256 constraint rows — the four-rows-per-lane
    PGS and the four-chunk A = J M^-1 J^T (the DPCombinedEnv getup phases live here)."""
    rng = np.random.default_rng(11)
    mc = _clip("getup_facedown")
    qs, vs = [], []
    for i in range(24):
        fr = int(rng.integers(0, 40))
        q, v = np.array(mc.data_config[fr]), np.array(mc.data_vel[fr]) * 0.0
        q[2] -= rng.uniform(0.015, 0.05)
        q[7:] += rng.uniform(-0.35, 0.35, 37)
        v = rng.normal(size=43) * 0.3
        qs.append(q)
        vs.append(v)
    ncons, nefcs = _forward_parity(np.array(qs), np.array(vs), "G1 forward parity, many rows:")
    print("   rows per state", nefcs)
    assert sum(1 for x in nefcs if x > 128) >= 4 and max(nefcs) <= 256


def _forward_parity(q, v, label, cnrm_tol=1e-5):
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine
    from oracle import oracle_g1 as og
    n = len(q)
    eng = G1HipEngine(n, auto_reset=False)
    eng.load_clip(_clip("walk"))
    dbg = eng.enable_debug()
    eng.set_state(torch.tensor(q, dtype=torch.float32, device=eng.device).contiguous(),
                  torch.tensor(v, dtype=torch.float32, device=eng.device).contiguous())
    torch.cuda.synchronize()
    dbg = dbg.cpu().numpy()
    worst = dict(xpos=0.0, qas=0.0, qacc=0.0, cdist=0.0, cpos=0.0, cnrm=0.0)
    ncons, nefcs, flips = [], [], 0
    for i in range(n):
        s = og.G1Sim()
        s.set_caps(48, 256)
        assert s.set_state(q[i].astype(np.float32).astype(np.float64), v[i].astype(np.float32).astype(np.float64)) == 0
        d = dbg[i]
        worst["xpos"] = max(worst["xpos"], np.abs(d[:117] - s.get("xpos")).max())
        qas = s.get("qacc_smooth")
        worst["qas"] = max(worst["qas"], np.abs(d[117:160] - qas).max() / max(1.0, np.abs(qas).max()))
        cons = s.contacts()
        ncon = int(d[203])
        ncons.append(ncon)
        gpu_c = d[208:208 + 9 * ncon].reshape(-1, 9)
        same = ncon == len(cons) and all(int(r[1]) == c["geom1"] and int(r[2]) == c["geom2"] for r, c in zip(gpu_c, cons))
        if not same:
            flips += 1          # (r2: a contact at the edge of detection could appear on one side only; the narrowphase now reads fp64 poses)
            print("env", i, "contact sets differ:", [(int(r[1]), int(r[2]), round(float(r[0]), 5)) for r in gpu_c],
                  [(c["geom1"], c["geom2"], round(c["dist"], 5)) for c in cons])
            continue
        for r, c in zip(gpu_c, cons):
            worst["cdist"] = max(worst["cdist"], abs(r[0] - c["dist"]))
            worst["cpos"] = max(worst["cpos"], np.abs(r[3:6] - c["pos"]).max())
            worst["cnrm"] = max(worst["cnrm"], np.abs(r[6:9] - c["frame"][0]).max())
        assert int(d[204]) == s.geti("nefc"), (i, d[204], s.geti("nefc"))
        nefcs.append(int(d[204]))
        qa = s.get("qacc")
        worst["qacc"] = max(worst["qacc"], np.abs(d[160:203] - qa).max() / max(1.0, np.abs(qa).max()))
    print(label, {k: float(v) for k, v in worst.items()}, "ncon", ncons, "contact-set flips", flips)
    assert flips == 0, "contact (geom1, geom2) lists must be identical (north_star: bit-exact contact-pair index sets)"
    assert worst["xpos"] < 2e-6 and worst["qas"] < 2e-4
    # (r2 held cdist 2e-5, cpos 2e-4 and normals to 2e-3 / 0.1: fp32 geom poses and the support tie; measured now 3e-9, 6e-8, 6e-8)
    assert worst["cdist"] < 1e-6 and worst["cpos"] < 1e-5 and worst["cnrm"] < cnrm_tol
    assert worst["qacc"] < 5e-3
    eng.close()
    return ncons, nefcs


def _teacher_forced(n, steps, act_scale, seed, stride=4, pipeline=0):
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine
    from oracle import oracle_g1 as og
    g, _ = og.g1_model()
    mc = _clip("walk")
    clip = og.G1Clip(*mc.tables())
    eng = G1HipEngine(n, auto_reset=False, pipeline=pipeline)
    eng.load_clip(mc)
    out = eng.alloc_outputs()
    idx = torch.arange(n, dtype=torch.int32, device=eng.device) * stride
    eng.reset(out["obs"], idx_init=idx)
    sims = [og.G1Sim() for _ in range(n)]
    for i, s in enumerate(sims):
        s.set_caps(48, 256)
        s.env_reset(clip, int(idx[i]))
    rng = np.random.default_rng(seed)
    alive = np.ones(n, bool)
    dbg = eng.enable_debug()
    recs = []   # (qpos err, qvel rel err, obs err, reward err, step used a non-analytic (MPR) contact in its last stage)
    for t in range(steps):
        q, v, w = [x.cpu().numpy().astype(np.float64) for x in eng.get_state()]
        act = (rng.uniform(-1, 1, (n, 23)) * act_scale).astype(np.float32)
        eng.step(torch.tensor(act, device=eng.device), out)
        torch.cuda.synchronize()
        q2, v2, _ = [x.cpu().numpy() for x in eng.get_state()]
        obs, rew, done = out["obs"].cpu().numpy(), out["rew"].cpu().numpy(), out["done"].cpu().numpy()
        d = dbg.cpu().numpy()
        for i, s in enumerate(sims):
            if not alive[i]:
                continue
            s.set("qpos", q[i]); s.set("qvel", v[i]); s.set("qacc_warmstart", w[i])
            o, r, dn, terms, reason = s.env_step(clip, act[i].astype(np.float64))
            mpr = any(not (g.geom_type[c["geom1"]] == 0 and g.geom_type[c["geom2"]] in (2, 5, 6))
                      and not (g.geom_type[c["geom1"]] == 2 and g.geom_type[c["geom2"]] in (2, 6))
                      and not (g.geom_type[c["geom1"]] == 6 and g.geom_type[c["geom2"]] == 6) for c in s.contacts())
            recs.append((np.abs(q2[i] - s.get("qpos")).max(),
                         np.abs(v2[i] - s.get("qvel")).max() / max(1.0, np.abs(s.get("qvel")).max()),
                         np.abs(obs[i] - o).max(), abs(rew[i] - r), mpr))
            # north_star: bit-exact contact-pair index sets — the (geom1, geom2) list of EVERY RK stage, by count and by hash
            for k in range(4):
                assert (int(d[i][1000 + k]), int(d[i][1012 + k])) == (s.geti("stage_ncon%d" % k), s.geti("stage_chash%d" % k)), \
                    ("contact list of RK stage %d differs" % k, t, i)
                assert int(d[i][1004 + k]) == (s.geti("stage_nefc%d" % k) & 0xFF), ("row count of RK stage %d differs" % k, t, i)
            assert bool(done[i]) == dn, ("termination differs", t, i, done[i], dn, reason)
            if dn:
                alive[i] = False
    eng.close()
    return np.array(recs, float)


def test_g1_teacher_forced_steps_small_actions():
    """16 envs on the walk clip, 25 steps of small torques (feet on the floor, hands on the hips: mesh-mesh contacts through MPR
    in most steps): before every step the oracle takes the engine's state (qpos, qvel, warm start), both step: state, observation
    and reward agree to fp32 round-off on EVERY step, per-stage contact lists and termination identical (asserted inside)."""
    r = _teacher_forced(16, 25, 0.05, 1)
    print("G1 teacher-forced, small actions: %d env-steps (%d with MPR contacts); max qpos %.2e qvel %.2e obs %.2e rew %.2e" %
          (len(r), int(r[:, 4].sum()), r[:, 0].max(), r[:, 1].max(), r[:, 2].max(), r[:, 3].max()))
    assert len(r) > 250 and r[:, 4].sum() > 50
    assert r[:, 0].max() < 2e-5 and r[:, 1].max() < 5e-4 and r[:, 2].max() < 5e-4 and r[:, 3].max() < 5e-4
    assert np.median(r[:, 0]) < 1e-6


@pytest.mark.parametrize("pipeline", [1, 2])
def test_g1_teacher_forced_steps_large_actions(pipeline):
    """The same with full-scale random torques: the robots thrash, fall and self-collide (mesh-mesh contacts through MPR).
    north_star's gate on every env-step, no exclusion set: qpos L-inf < 1e-4 (measured: max 7e-7), contact lists of all four RK
    stages and termination identical.  (r2 saw ~3 % of these steps off by up to 1e-2: not fp32 poses but a support TIE that
    libccd's MPR constructs at edge-edge contacts and decides by the last bit — oracle/dm_convex.h "ties")"""
    r = _teacher_forced(24, 80, 1.0, 1, stride=3, pipeline=pipeline)      # 1: monolithic kernel, 2: split pipeline (pair kernel)
    e = r[:, 0]
    print("G1 teacher-forced, full-scale actions: %d env-steps (%d with MPR contacts); qpos err median %.2e p99 %.2e max %.2e; qvel rel max %.2e" %
          (len(e), int(r[:, 4].sum()), np.median(e), np.percentile(e, 99), e.max(), r[:, 1].max()))
    assert len(e) > 300 and r[:, 4].sum() > 100
    assert e.max() < 1e-4, "north_star: per-step qpos L-inf error < 1e-4"
    assert np.median(e) < 1e-6 and np.percentile(e, 99) < 1e-5
    assert r[:, 1].max() < 2e-3 and r[:, 2].max() < 2e-3 and r[:, 3].max() < 1e-3


def test_g1_gym_and_vecenv_surfaces():
    """DPEnv(robot="unitree_g1") and HipDeepMimicVecEnv(robot="unitree_g1") construct the G1 classes; reset / step shapes,
    the 23-dim action space, force_state teacher-forcing (imitation terms exactly 1) and auto-reset with terminal_observation."""
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv, HipDeepMimicVecEnv
    from deepmimic_mujoco_amd import g1
    env = DPEnv("walk", robot="unitree_g1")
    assert isinstance(env, g1.G1DPEnv) and env.action_space.shape == (23,) and env.observation_space.shape == (85,)
    obs = env.reset_model(idx_init=5)
    assert obs.shape == (85,) and abs(obs[84] - 5 / env.mocap_data_len) < 1e-6
    mc = env.mocap
    o, r, d, info = env.step(np.zeros(23), force_state=(np.array(mc.get_qpos(5)), np.array(mc.get_qvel(5))))
    assert not d and abs(info["reward_config"] - 1) < 1e-5 and abs(info["reward_end_eff"] - 1) < 1e-5 and abs(r - 1.0) < 1e-3
    assert env.idx_curr == 6 and env.episode_length == 1
    steps = 0
    while True:
        o, r, d, info = env.step(np.zeros(23))
        steps += 1
        if d:
            break
    assert info["done_reason"] == "low_z" and 5 < steps < 80
    env.close()
    venv = HipDeepMimicVecEnv(8, motion="walk", robot="unitree_g1", seed=3)
    assert isinstance(venv, g1.HipG1VecEnv)
    obs = venv.reset()
    assert obs.shape == (8, 85)
    ndone = 0
    for t in range(60):
        obs, rew, done, infos = venv.step(np.zeros((8, 23), np.float32))
        assert obs.shape == (8, 85) and rew.shape == (8,) and len(infos) == 8 and np.isfinite(obs).all()
        for i in np.nonzero(done)[0]:
            ndone += 1
            assert infos[i]["terminal_observation"].shape == (85,) and infos[i]["done_reason"] in ("low_z", "high_z")
    assert ndone >= 8
    venv.close()


def test_g1_combined_env_matches_the_oracle():
    """DPCombinedEnv on the G1 engine (DmG1Config.task = 1): 12 envs started in walk (with amnesty), getup and to_getup,
    100 teacher-forced steps of random actions: every decision (motion, n_steps, done, reason) equals the oracle's; state, obs,
    reward and the eight info terms agree on every step."""
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine, TASK_COMBINED, COMBINED_CLIPS
    from oracle import oracle_g1 as og
    g, _ = og.g1_model()
    n, steps = 12, 100
    mocaps = [_clip(m) for m in COMBINED_CLIPS]
    clips = [og.G1Clip(*mc.tables()) for mc in mocaps]
    eng = G1HipEngine(n, auto_reset=False, task=TASK_COMBINED, max_ep_length=2000)
    for cid, mc in enumerate(mocaps):
        eng.load_clip(mc, clip_id=cid)
    out = eng.alloc_outputs()
    assert out["obs"].shape == (n, 98) and out["terms"].shape == (n, 8)
    motion0 = np.array([0, 0, 0, 0, 2, 2, 2, 2, 3, 3, 0, 2], np.int32)
    nst0 = np.array([170, 300, 10, 161, 0, 100, 250, 340, 0, 170, 200, 345], np.int32)
    eng.set_motion(torch.tensor(motion0, device=eng.device))
    eng.reset(out["obs"], idx_init=torch.tensor(nst0, device=eng.device))
    torch.cuda.synchronize()
    sims = [og.G1CombSim(clips) for _ in range(n)]
    obs0 = out["obs"].cpu().numpy()
    for i, s in enumerate(sims):
        s.set_caps(48, 256)
        o, err = s.comb_reset(int(motion0[i]), int(nst0[i]))
        assert err == 0 and np.abs(o - obs0[i]).max() < 1e-4, (i, np.abs(o - obs0[i]).max())
    rng = np.random.default_rng(5)
    alive = np.ones(n, bool)
    worst = dict(obs=0.0, rew=0.0, terms=0.0, state=0.0, vel=0.0)
    transitions = compared = 0
    for t in range(steps):
        q, v, w = [x.cpu().numpy().astype(np.float64) for x in eng.get_state()]
        act = (rng.uniform(-1, 1, (n, 23)) * 0.2).astype(np.float32)
        eng.step(torch.tensor(act, device=eng.device), out)
        torch.cuda.synchronize()
        obs, rew, done = out["obs"].cpu().numpy(), out["rew"].cpu().numpy(), out["done"].cpu().numpy()
        terms, reason = out["terms"].cpu().numpy(), out["reason"].cpu().numpy()
        mot, nst = eng.get_motion().cpu().numpy(), eng.get_counters()[0].cpu().numpy()
        q2, v2 = [x.cpu().numpy() for x in eng.get_state()[:2]]
        for i, s in enumerate(sims):
            if not alive[i]:
                continue
            m_before = s.cenv.motion
            s.set("qpos", q[i]); s.set("qvel", v[i]); s.set("qacc_warmstart", w[i])
            o, r, d, tr, rs = s.comb_step(act[i].astype(np.float64))
            state_err = np.abs(q2[i] - s.get("qpos")).max()
            assert state_err < 1e-4, (t, i, state_err)        # north_star's gate, every step, no exclusions
            assert (int(mot[i]), int(nst[i]), bool(done[i]), int(reason[i])) == (s.cenv.motion, s.cenv.n_steps, d, rs), (t, i)
            transitions += int(s.cenv.motion != m_before)
            compared += 1
            vel_err = np.abs(v2[i] - s.get("qvel")).max() if not d else 0.0
            worst["state"] = max(worst["state"], state_err)
            worst["vel"] = max(worst["vel"], vel_err)
            worst["obs"] = max(worst["obs"], np.abs(obs[i] - o).max())
            worst["rew"] = max(worst["rew"], abs(rew[i] - r))
            worst["terms"] = max(worst["terms"], np.abs(terms[i, :7] - tr[:7]).max())
            assert int(terms[i, 7]) == int(tr[7]) or state_err > 1e-7
            if d:
                alive[i] = False
    print("G1 DPCombinedEnv parity:", {k: float(x) for k, x in worst.items()}, "motion transitions", transitions, "alive", int(alive.sum()))
    assert transitions >= 3
    print("   compared env-steps", compared)
    assert compared > 600 and worst["obs"] < 3e-4 and worst["rew"] < 5e-4 and worst["terms"] < 5e-4 and worst["vel"] < 5e-3
    eng.close()


def test_g1_combined_surfaces():
    from deepmimic_mujoco_amd.combined_env import DPCombinedEnv, HipCombinedVecEnv
    from deepmimic_mujoco_amd import g1
    env = DPCombinedEnv()                       # the reference's constructor: Unitree G1
    assert isinstance(env, g1.G1CombinedEnv) and env.action_space.shape == (23,) and env.observation_space.shape == (98,)
    assert abs(float(env.action_space.high[0]) - 88 / 20) < 1e-6
    obs = env.reset(rsi=False)
    assert obs.shape == (98,) and env.current_motion_mocap is env.getup_mocap and env.current_motion_n_steps == 0
    o, r, d, info = env.step(np.zeros(23))
    assert o.shape == (98,) and env.current_motion_n_steps == 1 and "task_reward" in info and "imitation_reward" in info
    q, v = env.get_current_motion_state()
    assert q.shape == (44,) and v.shape == (43,)
    env.close()
    venv = HipCombinedVecEnv(16, seed=2)
    assert isinstance(venv, g1.HipG1CombinedVecEnv)
    obs = venv.reset()
    assert obs.shape == (16, 98)
    for t in range(30):
        obs, rew, done, infos = venv.step(np.zeros((16, 23), np.float32))
        assert obs.shape == (16, 98) and np.isfinite(obs).all() and len(infos) == 16
    venv.close()


@pytest.mark.parametrize("task", ["dpenv", "combined"])
def test_g1_split_pipeline_is_bit_identical_to_the_monolithic_kernel(task):
    """DmG1Config.pipeline: dmg1_step as ONE launch (a wave runs its env's whole step) or as the split pipeline (6 per-env launches
    around 5 batch-wide narrowphase launches, one wave per colliding pair).  Same arithmetic in the same order: observations,
    rewards, terminations, info terms and states are bit-identical over a rollout with auto-resets — robots standing, thrashing,
    falling, lying on the floor (mesh-floor and mesh-mesh contacts), for DPEnv and for DPCombinedEnv()."""
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine, TASK_COMBINED, TASK_DPENV, COMBINED_CLIPS
    n, steps = 96, 60
    engs = []
    for pl in (1, 2):
        e = G1HipEngine(n, auto_reset=True, seed=9, pipeline=pl, task=TASK_COMBINED if task == "combined" else TASK_DPENV,
                        max_ep_length=2000 if task == "combined" else 1000)
        if task == "combined":
            for cid, m in enumerate(COMBINED_CLIPS):
                e.load_clip(_clip(m), clip_id=cid)
        else:
            e.load_clip(_clip("walk"))
        engs.append((e, e.alloc_outputs()))
    for e, o in engs:
        e.reset(o["obs"])
    assert torch.equal(engs[0][1]["obs"], engs[1][1]["obs"])
    g = torch.Generator(device=engs[0][0].device).manual_seed(4)
    ndone = 0
    for t in range(steps):
        act = (torch.rand(n, 23, device=engs[0][0].device, generator=g) * 2 - 1) * (0.25 if task == "combined" else 1.0)
        for e, o in engs:
            e.step(act, o)
        torch.cuda.synchronize()
        a, b = engs[0][1], engs[1][1]
        for k in ("obs", "rew", "done", "terms", "reason", "terminal_obs"):
            assert torch.equal(a[k], b[k]), (task, t, k, int((a[k] != b[k]).sum()))
        for x, y in zip(engs[0][0].get_state(), engs[1][0].get_state()):
            assert torch.equal(x, y), (task, t)
        for x, y in zip(engs[0][0].get_counters(), engs[1][0].get_counters()):
            assert torch.equal(x, y), (task, t)
        ndone += int(a["done"].sum())
    assert ndone >= (3 if task == "combined" else 20)
    for e, _ in engs:
        e.close()


def test_g1_multi_clip_batches_and_sub_batches_replay_bit_equal():
    """`HipDeepMimicVecEnv(robot="unitree_g1", motion=[...], sub_batches=2)` — per-env clip ids
    (dmg1_set_env_clips, env i follows clip i mod 3) and two engines over contiguous halves.  Twin-env replay: the same actions
    through a one-engine batch and a two-engine batch, and through per-clip single-clip batches, give bit-equal trajectories."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd import g1
    motions = ["walk", "run", "getup_facedown"]
    n = 24
    a = HipDeepMimicVecEnv(n, motion=motions, robot="unitree_g1", seed=5, auto_reset=False)
    b = HipDeepMimicVecEnv(n, motion=motions, robot="unitree_g1", seed=5, auto_reset=False, sub_batches=2)
    assert isinstance(b, g1.HipG1VecEnv) and len(b.engines) == 2 and b.sub_slices == [slice(0, 12), slice(12, 24)]
    assert a.engine.get_env_clips().tolist() == [i % 3 for i in range(n)]
    assert b.engines[1].get_env_clips().tolist() == [(12 + i) % 3 for i in range(12)]
    idx = (torch.arange(n, device=a.device, dtype=torch.int32) * 2) % 40
    oa, ob = a.reset_tensor(idx).clone(), b.reset_tensor(idx).clone()
    assert torch.equal(oa, ob)
    singles = []
    for cid, m in enumerate(motions):           # the envs of clip cid as their own single-clip batch
        env = HipDeepMimicVecEnv(n // 3, motion=m, robot="unitree_g1", seed=5, auto_reset=False)
        o = env.reset_tensor(idx[cid::3].contiguous())
        assert torch.equal(o, oa[cid::3])
        singles.append(env)
    g = torch.Generator(device=a.device).manual_seed(3)
    for t in range(6):
        act = (torch.rand(n, 23, device=a.device, generator=g) * 2 - 1) * 0.3
        ra = {k: v.clone() for k, v in a.step_tensor(act).items()}
        rb = b.step_tensor(act)
        for k in ("obs", "rew", "done", "terms", "reason"):
            assert torch.equal(ra[k], rb[k]), (t, k)
        for cid, env in enumerate(singles):
            rs = env.step_tensor(act[cid::3].contiguous())
            assert torch.equal(rs["obs"], ra["obs"][cid::3]) and torch.equal(rs["rew"], ra["rew"][cid::3]), (t, cid)
    # getup_facedown is a floor motion: no low_z termination for its envs, while walk / run envs pressed to the floor would end
    assert int(ra["reason"][2::3].max()) in (0, 3, 4)
    sub = b.step_sub(1, act[12:].contiguous())
    assert sub["obs"].shape == (12, 85)
    for e in (a, b, *singles):
        e.close()


def test_g1_vecenv_seed_rekeys_rsi_and_batch_render_returns_a_frame():
    """`seed()` on the G1 batch envs re-keys the RSI generator (dmg1_set_seed); `render()` / `get_images()` of a batch
    return the frame of env 0 instead of raising or returning None, without disturbing the physics."""
    import torch
    from deepmimic_mujoco_amd.combined_env import HipCombinedVecEnv
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv

    def frames(env, seed):
        assert env.seed(seed) == [seed + i for i in range(env.num_envs)]
        env.reset_tensor()
        return env.engine.get_counters()[0].clone()
    for make in (lambda: HipDeepMimicVecEnv(32, motion="walk", robot="unitree_g1", seed=1),
                 lambda: HipCombinedVecEnv(32, seed=1, sub_batches=2),
                 lambda: HipDeepMimicVecEnv(32, motion="walk", seed=1),
                 lambda: HipCombinedVecEnv(32, robot="humanoid3d", seed=1)):
        env, other, twin = make(), make(), make()      # (the RSI key is (seed, env, reset count): compare first resets of fresh batches)
        f1, f2, f1t = frames(env, 11), frames(other, 12), frames(twin, 11)
        assert torch.equal(f1, f1t) and not torch.equal(f1, f2), type(env).__name__
        other.close()
        img = env.render(mode="rgb_array")
        assert img.shape == (240, 320, 3) and img.dtype == np.uint8 and len(np.unique(img.reshape(-1, 3), axis=0)) >= 4
        assert len(env.get_images()) == 1
        act = torch.zeros(32, env.action_space.shape[0], device=env.device)
        o1, o2 = env.step_tensor(act)["obs"].clone(), twin.step_tensor(act)["obs"].clone()
        assert torch.equal(o1, o2), "render changed the physics"
        env.close(); twin.close()


def test_ppo_iteration_on_the_g1_combined_env():
    """The reference's training setup (src/sb3_ppo.py:249-313: PPO on DPCombinedEnv(), Unitree G1, MLP [256,128]) on the device-
    resident loop: one-launch policy forward (D = 98, A = 23), dmg1_step, fused learner; one iteration changes the weights,
    keeps everything finite and the stored actions inside the action space (ctrlrange / 20)."""
    import torch
    from deepmimic_mujoco_amd.combined_env import HipCombinedVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    env = HipCombinedVecEnv(256, seed=11)
    ppo = PPO(env, n_steps=8, batch_size=512, n_epochs=2, seed=3)
    assert ppo.obs_dim == 98 and ppo.act_dim == 23
    w0 = [p.detach().clone() for p in ppo.policy.parameters()]
    ppo.learn(256 * 8 * 2, log_interval=0)
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in ppo.policy.parameters())
    assert any((p.detach() - q).abs().max() > 0 for p, q in zip(ppo.policy.parameters(), w0))
    assert np.isfinite(ppo.stats["mean_reward"]) and np.isfinite(ppo.stats.get("loss", 0.0))
    env.close()


def test_render_and_eval_dashboard_on_the_real_envs(tmp_path):
    """render(mode="rgb_array") (software stick figure) on DPEnv humanoid3d / G1 and DPCombinedEnv(); rendering must not disturb
    the physics (same next step with and without a render call); one dashboard episode with a PPO policy on the G1 env."""
    import torch
    from deepmimic_mujoco_amd.combined_env import DPCombinedEnv
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv, HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.eval_dashboard import eval_dashboard_rollout
    from deepmimic_mujoco_amd.ppo import PPO
    for make in (lambda: DPEnv("walk"), lambda: DPEnv("walk", robot="unitree_g1"), lambda: DPCombinedEnv()):
        e1, e2 = make(), make()
        o1 = e1.reset_model(idx_init=3) if hasattr(e1, "reset_model") else e1.reset(rsi=False)
        o2 = e2.reset_model(idx_init=3) if hasattr(e2, "reset_model") else e2.reset(rsi=False)
        assert np.array_equal(o1, o2)
        act = np.full(e1.action_space.shape[0], 0.1)
        for t in range(3):
            img = e1.render(mode="rgb_array")
            assert img.shape == (240, 320, 3) and len(np.unique(img.reshape(-1, 3), axis=0)) >= 4
            a, b = e1.step(act), e2.step(act)
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], "render changed the physics"
        e1.close(); e2.close()
    venv = HipDeepMimicVecEnv(64, motion="walk", robot="unitree_g1", seed=1)
    ppo = PPO(venv, n_steps=4, batch_size=128, n_epochs=1, seed=0)
    ev = DPEnv("walk", robot="unitree_g1")
    ep_len, ep_rew = eval_dashboard_rollout(ppo, ev, 123, "g1test", out_root=str(tmp_path), max_steps=30)
    assert 1 <= ep_len <= 30 and np.isfinite(ep_rew)
    assert (tmp_path / "g1test_videos" / "global_step_123.gif").exists() and (tmp_path / "g1test_videos" / "log.csv").exists()
    ev.close(); venv.close()
