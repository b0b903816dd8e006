"""GPU tests of the reference-facing surfaces: DPEnv (gym.Env), HipDeepMimicVecEnv (SB3 VecEnv),
auto-reset semantics, multi-clip batches, free-running rollouts and a short PPO run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hash32(seed, env, step, j):
    M = (1 << 64) - 1
    seed, env, step, j = int(seed), int(env), int(step), int(j)
    x = (seed ^ (env * 0x9E3779B97F4A7C15) ^ (step * 0xBF58476D1CE4E5B9) ^ (j * 0x94D049BB133111EB)) & M
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M
    x ^= x >> 31
    return x >> 32


def test_dpenv_surface_matches_oracle(model, clips, oracle_clips):
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv
    from oracle.oracle import OracleSim
    env = DPEnv(motion="walk")
    assert env.action_space.shape == (28,) and env.action_space.high.max() == 2.0
    assert env.observation_space.shape == (67,) and env.mocap_data_len == 76 and env.version == "v1.0"
    oc = oracle_clips["walk"]
    s = OracleSim(model)
    s.set_caps(32, 128)
    obs = env.reset_model(idx_init=5)
    eobs = s.env_reset(oc, 5)
    assert obs.dtype == np.float64 and np.abs(obs - eobs).max() < 1e-5
    rng = np.random.default_rng(0)
    for t in range(40):
        a = rng.uniform(-0.4, 0.4, 28)
        o, r, d, info = env.step(a)
        eo, er, ed, et, ereason = s.env_step(oc, a)
        assert np.abs(o - eo).max() < 5e-3 and abs(r - er) < 1e-3 and d == ed
        assert set(info) == {"reward_config", "reward_qvel", "reward_end_eff", "reward_com",
                             "reward_joint_limit", "done_reason"}
        assert info["done_reason"] in ("low_z", "high_z")
        assert env.idx_curr == s.env.idx_curr and env.episode_length == s.env.episode_length
        # teacher-force the oracle onto the device state so fp32 drift does not accumulate
        q, v, w, c = [x[0].double().cpu().numpy() for x in env._eng.get_state()]
        s.set("qpos", q); s.set("qvel", v); s.set("qacc_warmstart", w); s.set("ctrl", c)
        if d:
            break
    i = env.idx_curr                      # the reward compares against clip row idx_curr
    q = clips["walk"].data_config[i]
    v = clips["walk"].data_vel[i]
    o, r, d, info = env.step(np.zeros(28), force_state=(q, v))
    assert abs(info["reward_qvel"] - 1.0) < 1e-6 and env.get_time() > 0
    assert np.abs(env.sim.data.qpos[7:] - q[7:]).max() < 1e-6
    with pytest.raises(AssertionError):
        env.step(np.zeros(27))
    env.close()


def test_vecenv_autoreset_semantics(model, clips, oracle_clips):
    """SubprocVecEnv worker: on done, infos[i]['terminal_observation'] = last obs and obs = reset obs."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from oracle.oracle import OracleSim
    N, seed = 64, 4242
    venv = HipDeepMimicVecEnv(N, motion="walk", seed=seed)
    obs = venv.reset()
    assert obs.shape == (N, 67) and venv.num_envs == N
    oc = oracle_clips["walk"]
    L = oc.L
    # reset #0 of env i picks frame hash(seed, i, 0) % L
    s = OracleSim(model)
    s.set_caps(32, 128)
    for i in (0, 17, 63):
        fi = _hash32(seed, i, 0, 0x5EED) % L
        assert np.abs(obs[i] - s.env_reset(oc, fi)).max() < 1e-5
    rng = np.random.default_rng(1)
    seen_done = 0
    resets = np.ones(N, int)
    for t in range(60):
        act = rng.uniform(-2, 2, (N, 28)).astype(np.float32)
        obs, rew, done, infos = venv.step(act)
        assert len(infos) == N
        for i in np.nonzero(done)[0]:
            info = infos[i]
            assert "terminal_observation" in info and info["terminal_observation"].shape == (67,)
            fi = _hash32(seed, i, int(resets[i]), 0x5EED) % L
            resets[i] += 1
            assert abs(obs[i, 66] - fi / L) < 1e-6            # phase of the reset frame
            assert np.abs(obs[i, :28] - oc.qpos[fi, 7:]).max() < 1e-5
            assert info["done_reason"] in ("low_z", "high_z", "max_ep_len")
            seen_done += 1
        for i in np.nonzero(~done)[0][:3]:
            assert "terminal_observation" not in infos[i]
    assert seen_done > 10
    idx, eplen, eprew = [x.cpu().numpy() for x in venv.engine.get_counters()]
    assert (eplen >= 0).all() and (eplen <= 60).all()
    venv.close()


def test_multi_clip_batch(model, clips):
    """BASELINE config 5 shape: clip id = env mod 4 over (walk, run, dance_b, spinkick)."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    names = ["walk", "run", "dance_b", "spinkick"]
    venv = HipDeepMimicVecEnv(32, motion=names, auto_reset=False)
    idx = torch.full((32,), 3, dtype=torch.int32, device=venv.device)
    obs = venv.reset_tensor(idx_init=idx).cpu().numpy()
    for i in range(32):
        mc = clips[names[i % 4]]
        assert np.abs(obs[i, :28] - mc.data_config[3][7:]).max() < 1e-5
        assert abs(obs[i, 66] - 3 / len(mc.data_config)) < 1e-6
    venv.close()


def test_free_running_rollout_stays_close_to_oracle(model, clips, oracle_clips):
    """Free-running fp32 vs fp64 diverges chaotically in contact-rich motion; over a short horizon from a
    clip frame with small torques the trajectories must still agree (reported, gated loosely)."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    from oracle.oracle import OracleSim
    N, T = 16, 25
    eng = HipEngine(model, N, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    idx = (torch.arange(N, dtype=torch.int32, device=eng.device) * 4) % 76
    eng.reset(out["obs"], idx_init=idx)
    oc = oracle_clips["walk"]
    sims = []
    for i in range(N):
        s = OracleSim(model)
        s.set_caps(32, 128)
        s.env_reset(oc, int(idx[i]))
        sims.append(s)
    rng = np.random.default_rng(5)
    worst = 0.0
    for t in range(T):
        a = rng.uniform(-0.2, 0.2, (N, 28))
        eng.step(torch.tensor(a, dtype=torch.float32, device=eng.device), out)
        q = eng.get_state()[0].double().cpu().numpy()
        for i, s in enumerate(sims):
            s.env_step(oc, a[i])
            worst = max(worst, np.abs(q[i] - s.get("qpos")).max())
    print("free-running %d steps: max |qpos - oracle| = %.3g" % (T, worst))
    assert worst < 5e-3
    eng.close()


def test_sim_error_and_obs_bound_paths(model, clips):
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    eng = HipEngine(model, 4, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    eng.reset(out["obs"], idx_init=torch.zeros(4, dtype=torch.int32, device=eng.device))
    q, v, w, c = eng.get_state()
    q[1, 10] = float("nan")          # -> MujocoException path (deepmimic_env.py:366-378)
    v[2, 8] = 5000.0                 # -> |obs| > 100 guard (:465-476): 0.1 * 5000 = 500
    eng.set_state(q, v, w, c)
    eng.step(torch.zeros(4, 28, device=eng.device), out)
    torch.cuda.synchronize()
    done, reason = out["done"].cpu().numpy(), out["reason"].cpu().numpy()
    assert done[1] == 1 and reason[1] == 5 and not out["obs"][1].any() and out["rew"][1] == 0
    assert done[2] == 1 and reason[2] == 6 and not out["obs"][2].any() and out["rew"][2] == 0
    assert done[0] == 0 and reason[0] in (1, 2) and out["obs"][0].abs().max() > 0
    q2 = eng.get_state()[0].cpu().numpy()
    assert np.allclose(q2[1], model.qpos0, atol=1e-6)       # mjData reset after the warning
    eng.close()


def test_short_ppo_run_on_device():
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    venv = HipDeepMimicVecEnv(128, motion="walk")
    ppo = PPO(venv, net_arch=(64, 32), n_steps=16, batch_size=512, n_epochs=2, learning_rate=3e-4)
    ppo.learn(2 * 16 * 128, log_interval=0)
    assert ppo.num_timesteps == 2 * 16 * 128 and np.isfinite(ppo.stats["loss"])
    a = ppo.predict(venv.reset_tensor())
    assert a.shape == (128, 28) and a.abs().max() <= 2.0
    venv.close()


def test_hip_graph_update_matches_eager():
    """The captured (hipGraph) optimizer step must produce the same parameters as the eager step."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    T, N = 8, 64
    buf = dict(obs=torch.randn(T, N, 67, generator=g), act=torch.randn(T, N, 28, generator=g),
               adv=torch.randn(T, N, generator=g), ret=torch.randn(T, N, generator=g),
               logp=-40 + torch.randn(T, N, generator=g), rew=torch.zeros(T, N), done=torch.zeros(T, N),
               val=torch.zeros(T, N))
    buf = {k: v.to(dev) for k, v in buf.items()}
    outs = []
    for use_graph in (False, True):
        ppo = PPO(None, net_arch=(64, 32), n_epochs=2, batch_size=128, device=dev, use_hip_graph=use_graph, seed=3)
        gen = torch.Generator(device=dev).manual_seed(5)
        ppo.train(buf, generator=gen)
        ppo.train(buf, generator=gen)
        outs.append(torch.cat([p.detach().reshape(-1) for p in ppo.policy.parameters()]).cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.allclose(outs[0], outs[1], rtol=1e-4, atol=1e-6), (outs[0] - outs[1]).abs().max()


def test_max_size_batch_and_tiny_batch(model, clips):
    """BASELINE config 5 size (65 536 envs on one GPU's share = 8 192; here the full 65 536 on one GPU) and N = 1."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    for n in (1, 3, 65536):
        eng = HipEngine(model, n)
        eng.load_clip(0, clips["walk"])
        out = eng.alloc_outputs()
        eng.reset(out["obs"])
        act = torch.zeros(n, 28, device=eng.device)
        for i in range(3):
            eng.fill_random_actions(act, i)
            eng.step(act, out)
        torch.cuda.synchronize()
        assert torch.isfinite(out["obs"]).all() and torch.isfinite(out["rew"]).all()
        assert out["obs"].abs().max() <= 100.0
        w = eng.get_work()
        assert w.shape == (n,) and int(w.min()) > 0
        eng.close()


def test_floor_and_acyclic_motion_semantics(model):
    """getup_facedown is a floor + acyclical motion (src/config.py:36-37): no low/high COM termination
    (deepmimic_env.py:420), episode ends with 'acyclical_end' on the last frame (:440-442)."""
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv
    from deepmimic_mujoco_amd.mocap import MocapDM
    from oracle.oracle import OracleClip, OracleSim
    env = DPEnv(motion="getup_facedown")
    mc = MocapDM(model=model)
    mc.load_mocap(MotionConfig("getup_facedown").mocap_path)
    oc = OracleClip(*mc.tables(), floor=True, acyclic=True)
    s = OracleSim(model)
    s.set_caps(32, 128)
    L = env.mocap_data_len
    env.reset_model(idx_init=L - 6)
    s.env_reset(oc, L - 6)
    for t in range(6):
        i = env.idx_curr
        fs = (mc.data_config[i], mc.data_vel[i])
        o, r, d, info = env.step(np.zeros(28), force_state=fs)
        eo, er, ed, et, ereason = s.env_step(oc, np.zeros(28), force_state=fs)
        assert d == ed and abs(r - er) < 1e-4 and np.abs(o - eo).max() < 2e-3
        if t < 5:
            assert not d and "done_reason" not in info        # lying on the floor (COM z < 0.7) does not terminate
        else:
            assert d and info["done_reason"] == "acyclical_end" and ereason == 4
    env.close()


def test_eval_env_in_thread_next_to_batched_env(model, clips, oracle_clips):
    """SURVEY §8b threading: the eval env lives in a daemon thread of the learner process (sb3_ppo.py:173-174) and
    steps concurrently with the batched training env.  Handles share no mutable state: results of the threaded
    single env must equal those of the same rollout run alone."""
    import threading
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import DPEnv, HipDeepMimicVecEnv

    def eval_rollout(out):
        env = DPEnv(motion="walk")
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            obs = [env.reset_model(idx_init=5)]
            rews = []
            rng = np.random.default_rng(11)
            for t in range(60):
                o, r, d, info = env.step(rng.uniform(-1, 1, 28))
                obs.append(o)
                rews.append(r)
                if d:
                    break
        env.close()
        out.append((np.array(obs), np.array(rews)))

    alone = []
    eval_rollout(alone)
    venv = HipDeepMimicVecEnv(2048, motion="run")
    venv.reset_tensor()
    threaded = []
    th = threading.Thread(target=eval_rollout, args=(threaded,), daemon=True)
    th.start()
    act = torch.zeros(2048, 28, device=venv.device)
    ref = None
    for i in range(200):
        venv.engine.fill_random_actions(act, i)
        venv.step_tensor(act)
    th.join(timeout=120)
    assert not th.is_alive() and len(threaded) == 1
    torch.cuda.synchronize()
    assert np.array_equal(alone[0][0], threaded[0][0]) and np.array_equal(alone[0][1], threaded[0][1])
    # and the batched env was not disturbed: same 200 steps again from the same start give the same state
    q1 = venv.engine.get_state()[0].clone()
    venv2 = HipDeepMimicVecEnv(2048, motion="run")
    venv2.reset_tensor()
    for i in range(200):
        venv2.engine.fill_random_actions(act, i)
        venv2.step_tensor(act)
    assert torch.equal(q1, venv2.engine.get_state()[0])
    venv.close()
    venv2.close()


def test_fused_ppo_loss_matches_autograd():
    """dm_ppo_loss (HIP, forward + backward) against the PyTorch-op loss of SB3's PPO.train: loss terms and all
    parameter gradients, with and without advantage normalisation, incl. samples outside the clip range."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    for B, normalize in ((4096, True), (1000, False), (257, True)):
        grads = {}
        for fused in (False, True):
            torch.manual_seed(5)
            pol = MlpPolicy(net_arch=(64, 32)).to(dev)
            with torch.no_grad():
                pol.log_std.copy_(torch.linspace(-0.5, 0.3, 28))
            ppo = PPO(None, policy=pol, device=dev, batch_size=B, normalize_advantage=normalize, use_hip_graph=False,
                      fused_loss=fused, flat_adam=False, ent_coef=0.01)
            g = torch.Generator(device=dev); g.manual_seed(11)
            obs = torch.randn(B, 67, device=dev, generator=g)
            act = torch.randn(B, 28, device=dev, generator=g) * 0.7
            adv = torch.randn(B, device=dev, generator=g) * 2 + 0.3
            ret = torch.randn(B, device=dev, generator=g)
            with torch.no_grad():
                _, logp, _ = pol.evaluate_actions(obs, act)
            old_logp = logp + torch.randn(B, device=dev, generator=g) * 0.3     # ratios well outside [0.8, 1.2] too
            loss = (ppo._loss_fused if fused else ppo._loss_torch)(obs, act, adv, ret, old_logp)
            pol.zero_grad()
            loss.backward()
            grads[fused] = (float(loss.detach()), {n: p.grad.clone() for n, p in pol.named_parameters()})
        l0, g0 = grads[False]
        l1, g1 = grads[True]
        assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0)), (l0, l1)
        for n in g0:
            scale = float(g0[n].abs().max()) + 1e-12
            assert float((g0[n] - g1[n]).abs().max()) < 2e-5 * scale + 1e-9, (n, B, normalize)


def test_dm_forward_recomputes_without_changing_the_state(model, clips, oracle_clips):
    """dm_forward = sim.forward() (src/deepmimic_env.py:491): derived arrays and warm start at the stored state."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    from oracle.oracle import OracleSim
    rng = np.random.default_rng(5)
    n = 32
    eng = HipEngine(model, n, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    q, v = clips["walk"].tables()[:2]
    idx = rng.integers(0, len(q), n)
    qpos = torch.tensor(q[idx] + rng.normal(0, 0.02, (n, 35)), dtype=torch.float32, device=eng.device)
    qvel = torch.tensor(v[idx] + rng.normal(0, 0.2, (n, 34)), dtype=torch.float32, device=eng.device)
    eng.set_state(qpos, qvel, run_forward=False)          # store only
    dbg = eng.enable_debug()
    eng.forward()
    torch.cuda.synchronize()
    q2, v2, warm, _ = eng.get_state()
    assert torch.equal(q2[:, :3], qpos[:, :3]) and torch.equal(q2[:, 7:], qpos[:, 7:]) and torch.equal(v2, qvel)
    d = dbg.cpu().numpy()
    for i in range(n):
        o = OracleSim(model)
        o.set_state(qpos[i].double().cpu().numpy(), qvel[i].double().cpu().numpy())
        assert np.abs(d[i, 0:42] - o.get("xpos").ravel()).max() < 1e-5
        assert np.abs(d[i, 174:208] - o.get("qacc")).max() < 2e-3 * max(1.0, np.abs(o.get("qacc")).max())
        assert np.abs(warm[i].cpu().numpy() - o.get("qacc_warmstart")).max() < 2e-3 * max(1.0, np.abs(o.get("qacc")).max())
    eng.close()


def test_physics_step_is_sim_step_alone(model, clips):
    """dm_physics_step = `self.sim.step()` alone (src/deepmimic_env.py:362; the physics-only leg of SURVEY 8d): from the same
    state and action it leaves exactly the qpos / qvel / warm start / ctrl the full dm_step leaves (same kernel, task layer
    skipped), matches the oracle's dmo_step, and touches neither the counters nor any output buffer."""
    import torch
    from deepmimic_mujoco_amd._lib import HipEngine
    from oracle.oracle import OracleSim
    n = 64
    eng = HipEngine(model, n, auto_reset=False)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    eng.reset(out["obs"], idx_init=(torch.arange(n, device=eng.device) % 70).to(torch.int32))
    act = torch.empty(n, 28, device=eng.device)
    for t in range(3):
        eng.fill_random_actions(act, t)
        eng.step(act, out)
    q, v, w, c = eng.get_state()
    idx0, len0, rew0 = [x.clone() for x in eng.get_counters()]
    obs0 = out["obs"].clone()
    eng.fill_random_actions(act, 7)
    eng.physics_step(act)
    torch.cuda.synchronize()
    qp, vp, wp, cp = eng.get_state()
    idx1, len1, rew1 = eng.get_counters()
    assert torch.equal(idx0, idx1) and torch.equal(len0, len1) and torch.equal(rew0, rew1) and torch.equal(obs0, out["obs"])
    assert torch.equal(cp, act)
    eng.set_state(q, v, warm=w, ctrl=c, run_forward=False)
    eng.step(act, out)
    torch.cuda.synchronize()
    qf, vf, wf, cf = eng.get_state()
    assert torch.equal(qp, qf) and torch.equal(vp, vf) and torch.equal(wp, wf) and torch.equal(cp, cf)
    assert int(eng.get_counters()[1][0]) == int(len0[0]) + 1
    worst = 0.0
    for i in range(0, n, 8):
        o = OracleSim(model)
        o.set_caps(32, 128)
        o.set("qpos", q[i].double().cpu().numpy()); o.set("qvel", v[i].double().cpu().numpy())
        o.set("qacc_warmstart", w[i].double().cpu().numpy()); o.set("ctrl", act[i].double().cpu().numpy())
        assert o.step() == 0
        worst = max(worst, float(np.abs(qp[i].cpu().numpy() - o.get("qpos")).max()))
    assert worst < 1e-4, worst
    eng.close()


def test_hip_linear_wgrad_matches_torch():
    """dm_linear_wgrad (MFMA split-K) against torch's weight / bias gradients for every layer shape of both nets."""
    import torch
    from deepmimic_mujoco_amd.ppo import HipLinear
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for B in (4096, 2048, 1024):
        # the last three shapes are beyond 256 units: library GEMM for dW, dm_colsum for db
        for (I, O) in ((67, 256), (256, 128), (128, 28), (128, 1), (72, 256), (200, 96), (67, 1024), (1024, 512), (512, 28)):
            lin = HipLinear(I, O).to(dev)
            x = torch.randn(B, I, device=dev, requires_grad=True)
            gy = torch.randn(B, O, device=dev)
            y = lin(x)                                   # hand-written backward
            y.backward(gy)
            gw, gb, gx = lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone()
            ref_w, ref_b, ref_x = gy.t() @ x.detach(), gy.sum(0), gy @ lin.weight.detach()
            sw = float(ref_w.abs().max())
            assert float((gw - ref_w).abs().max()) < 2e-4 * sw, (B, I, O)
            assert float((gb - ref_b).abs().max()) < 2e-4 * float(ref_b.abs().max() + 1), (B, I, O)
            assert torch.allclose(gx, ref_x, rtol=1e-4, atol=1e-4)


def test_fused_tanh_layer_kernels_match_torch():
    """dm_linear_tanh / dm_tanh_linear_wgrad / dm_tanh_bwd_colsum (the activation fused around the GEMMs of the wide trunks)
    against torch, for every observation width of the reference's envs and ragged sizes."""
    import ctypes as C
    import torch
    from deepmimic_mujoco_amd import _lib
    L = _lib.load_library()
    dev = torch.device("cuda", 0)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(3)
    for B, O, I in ((4096, 1024, 67), (2048, 1024, 72), (1024, 512, 85), (4096, 1024, 98), (1024, 300, 128), (1000, 77, 5)):
        x = torch.randn(B, I, generator=g).to(dev)
        w = (torch.randn(O, I, generator=g) / I ** 0.5).to(dev)
        b = torch.randn(O, generator=g).to(dev)
        y = torch.full((B, O), 7.0, device=dev)
        assert L.dm_linear_tanh(p(x), p(w), p(b), p(y), B, O, I, st) == 0
        ref = torch.tanh(torch.addmm(b, x, w.t()).double()).float()
        assert float((y - ref).abs().max()) < 2e-5, (B, O, I, float((y - ref).abs().max()))
        gy = torch.randn(B, O, generator=g).to(dev)
        gz_ref = gy.double() * (1 - ref.double() ** 2)
        gz, db = torch.empty_like(gy), torch.zeros(O, device=dev)
        assert L.dm_tanh_bwd_colsum(p(gy), p(ref), p(gz), p(db), B, O, st) == 0
        assert float((gz - gz_ref.float()).abs().max()) < 1e-6
        assert float((db - gz_ref.sum(0).float()).abs().max()) < 3e-4 * float(gz_ref.sum(0).abs().max() + 1)
        gy2 = gy.clone()                                              # in place
        db2 = torch.zeros(O, device=dev)
        assert L.dm_tanh_bwd_colsum(p(gy2), p(ref), p(gy2), p(db2), B, O, st) == 0 and torch.equal(gy2, gz)
        dw, db3 = torch.zeros(O, I, device=dev), torch.zeros(O, device=dev)
        assert L.dm_tanh_linear_wgrad(p(gy), p(ref), p(x), p(dw), p(db3), B, O, I, st) == 0
        dw_ref = (gz_ref.t() @ x.double()).float()
        assert float((dw - dw_ref).abs().max()) < 3e-4 * float(dw_ref.abs().max()), (B, O, I)
        assert float((db3 - gz_ref.sum(0).float()).abs().max()) < 3e-4 * float(gz_ref.sum(0).abs().max() + 1)
    assert L.dm_linear_tanh(p(x), p(w), p(b), p(y), 8, 8, 129, st) == -22
    assert L.dm_tanh_linear_wgrad(p(gy), p(ref), p(x), p(dw), p(db3), 8, 8, 129, st) == -22


def test_big_net_learner_step_matches_plain_autograd():
    """[1024,512]: the arena path (library GEMM written into the flat gradient buffer + dm_colsum, fused loss, flat Adam,
    value trunk on a second stream) gives the same parameters after three optimizer steps as plain autograd + torch Adam."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(1)
    B = 4096
    obs, act = torch.randn(B, 67, generator=g).to(dev), torch.randn(B, 28, generator=g).to(dev)
    adv, ret, lp = torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev), (-40 + torch.randn(B, generator=g)).to(dev)
    res = []
    for fast in (True, False):
        q = PPO(None, net_arch=(1024, 512), batch_size=B, device=dev, use_hip_graph=False, seed=4, fused_loss=fast, flat_adam=fast,
                two_stream=fast)
        for _ in range(3):
            q._minibatch_step(obs, act, adv, ret, lp)
        torch.cuda.synchronize()
        res.append(torch.cat([p.detach().reshape(-1) for p in q.policy.parameters()]))
    assert torch.isfinite(res[0]).all()
    assert torch.allclose(res[0], res[1], rtol=2e-4, atol=5e-6), float((res[0] - res[1]).abs().max())


def test_flat_adam_matches_clip_grad_norm_plus_torch_adam():
    """dm_flat_adam_step on the flat buffers against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (fused) over
    several optimizer steps of the same minibatches, with every custom piece on (fused loss, HipLinear) on one side
    and plain PyTorch on the other."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy
    dev = torch.device("cuda", 0)
    B = 2048
    results = []
    for custom in (False, True):
        torch.manual_seed(21)
        pol = MlpPolicy(net_arch=(256, 128)).to(dev)
        ppo = PPO(None, policy=pol, device=dev, batch_size=B, use_hip_graph=False, fused_loss=custom, flat_adam=custom,
                  learning_rate=3e-3, max_grad_norm=0.5)
        g = torch.Generator(device=dev); g.manual_seed(4)
        for it in range(6):
            obs = torch.randn(B, 67, device=dev, generator=g)
            act = torch.randn(B, 28, device=dev, generator=g) * 0.5
            adv = torch.randn(B, device=dev, generator=g) * (10.0 if it % 2 else 0.1)   # clipped and unclipped norms
            ret = torch.randn(B, device=dev, generator=g)
            with torch.no_grad():
                _, logp, _ = pol.evaluate_actions(obs, act)
            old_logp = logp + torch.randn(B, device=dev, generator=g) * 0.2
            ppo._minibatch_step(obs, act, adv, ret, old_logp)
        results.append({n: p.detach().clone() for n, p in pol.named_parameters()})
    for n in results[0]:
        a, b = results[0][n], results[1][n]
        assert float((a - b).abs().max()) < 2e-4 * (float(a.abs().max()) + 1e-3), n


def test_flat_adam_abi_checks_state2_and_loads_r2_checkpoints(tmp_path):
    """the partial sums of the gradient norm live behind the two scalars of state2 — the entry points take the buffer
    length and refuse a short one (r2's 2-float buffer would be overrun); checkpoints written with the 2-float state2 load."""
    import ctypes as C
    import torch
    from deepmimic_mujoco_amd import _lib
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    L = _lib.load_library()
    dev = torch.device("cuda", 0)
    n = 1000
    p_, g_, m_, v_ = (torch.zeros(n, device=dev) for _ in range(4))
    st = torch.zeros(2, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())
    s0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    assert L.dm_flat_adam_step(vp(p_), vp(g_), vp(m_), vp(v_), n, 1e-3, 0.9, 0.999, 1e-5, 0.5, 1.0, vp(st), 2, s0) == -22
    assert L.dm_flat_adam_step(vp(p_), vp(g_), vp(m_), vp(v_), n, 1e-3, 0.9, 0.999, 1e-5, 0.5, 0.0, vp(st), 2 + 1024, s0) == -22
    env = HipDeepMimicVecEnv(32, motion="walk", seed=1)
    ppo = PPO(env, n_steps=4, batch_size=64, n_epochs=1, seed=0)
    ppo.learn(32 * 4, log_interval=0)
    sd = ppo.optimizer.state_dict()
    steps = float(sd["state2"][1])
    assert steps >= 1 and sd["state2"].numel() == 2 + 1024
    old = dict(sd, state2=sd["state2"][:2].clone())                   # what r1 / early r2 wrote
    path = str(tmp_path / "old.pt")
    torch.save({"policy": ppo.policy.state_dict(), "optimizer": old, "num_timesteps": ppo.num_timesteps}, path)
    ppo2 = PPO(env, n_steps=4, batch_size=64, n_epochs=1, seed=1).load(path)
    assert float(ppo2.optimizer.state2[1]) == steps and float(ppo2.optimizer.state2[2:].abs().max()) == 0.0
    assert torch.equal(ppo2.optimizer.m, ppo.optimizer.m)
    ppo2.learn(2 * 32 * 4, log_interval=0)                           # and keeps training from there (num_timesteps was restored)
    assert float(ppo2.optimizer.state2[1]) > steps
    env.close()


def test_sub_batched_vecenv_and_captured_rollout(model):
    """sub_batches = 2: the VecEnv surface is unchanged (one [N, ...] output set, SubprocVecEnv semantics) and PPO captures
    the whole rollout as one graph with one chain per sub-batch; buffers are complete, finite and consistent."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    N = 256
    venv = HipDeepMimicVecEnv(N, motion="walk", sub_batches=2, seed=5)
    obs = venv.reset()
    assert obs.shape == (N, 67) and np.isfinite(obs).all()
    o2, r2, d2, infos = venv.step(np.zeros((N, 28), np.float32))
    assert o2.shape == (N, 67) and r2.shape == (N,) and (r2[:N // 2] != 0).any() and (r2[N // 2:] != 0).any()   # both halves stepped
    ppo = PPO(venv, net_arch=(64, 32), n_steps=8, batch_size=512, n_epochs=1, fused_policy=False)
    buf = ppo.collect_rollouts()
    assert buf["obs"].shape == (8, N, 67) and torch.isfinite(buf["obs"]).all() and torch.isfinite(buf["adv"]).all()
    assert torch.isfinite(buf["rew"]).all() and ppo.num_timesteps == 8 * N
    assert float(buf["rew"][:, :N // 2].abs().sum()) > 0 and float(buf["rew"][:, N // 2:].abs().sum()) > 0
    # the observation stored at step t+1 is the one the env returned at step t (per half), unless the env was reset
    cont = buf["done"][0] == 0
    assert cont.any()
    buf2 = {k: v.clone() for k, v in buf.items()}
    ppo.collect_rollouts()                                                       # replay continues the same episodes
    assert not torch.equal(buf2["obs"], ppo._rollout[1]["obs"])
    ppo.train(ppo._rollout[1])
    assert np.isfinite(ppo.stats["loss"])
    venv.close()


def test_fused_rollout_kernels(model):
    """dm_policy_sample: standard-normal noise (moments, independence across draws), logp equal to the policy's own
    formula for the sampled action, clamping; dm_rollout_store through a short rollout: buffers equal to what the
    unfused loop stores (obs seen by the policy, its value, rewards / dones of the env)."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    N = 4096
    venv = HipDeepMimicVecEnv(N, motion="walk", seed=9)
    ppo = PPO(venv, net_arch=(64, 32), n_steps=4, batch_size=1024, n_epochs=1, fused_policy=False)
    with torch.no_grad():
        ppo.policy.log_std.copy_(torch.linspace(-1.0, 0.5, 28))
        obs = venv.reset_tensor().clone()
        sc, val = ppo._policy_step_fused(obs)
        act1 = sc["act"].clone()
        mean = ppo.policy.action_net(ppo.policy.pi(obs))
        eps = (act1 - mean) / ppo.policy.log_std.exp()
        assert abs(float(eps.mean())) < 0.01 and abs(float(eps.std()) - 1.0) < 0.01
        assert abs(float((eps ** 3).mean())) < 0.03 and abs(float((eps ** 4).mean()) - 3.0) < 0.1
        assert abs(float(torch.corrcoef(eps[:, :2].T)[0, 1])) < 0.05                 # pairs come from one Box-Muller draw
        assert torch.allclose(sc["logp"], ppo.policy._logp(act1, mean), atol=2e-4)
        assert torch.equal(sc["act_env"], torch.clamp(act1, ppo.act_lo, ppo.act_hi))
        assert torch.allclose(val, ppo.policy.predict_values(obs), atol=1e-6)
        sc2, _ = ppo._policy_step_fused(obs)                                          # same counter -> same draw
        assert torch.equal(sc2["act"], act1)
    buf = ppo.collect_rollouts()                                                       # counter advances once per step
    with torch.no_grad():
        for t in range(4):
            m = ppo.policy.action_net(ppo.policy.pi(buf["obs"][t]))
            assert torch.allclose(buf["logp"][t], ppo.policy._logp(buf["act"][t], m), atol=2e-4)
            assert torch.allclose(buf["val"][t], ppo.policy.predict_values(buf["obs"][t]), atol=1e-5)
        e0 = (buf["act"][0] - ppo.policy.action_net(ppo.policy.pi(buf["obs"][0]))) / ppo.policy.log_std.exp()
        e1 = (buf["act"][1] - ppo.policy.action_net(ppo.policy.pi(buf["obs"][1]))) / ppo.policy.log_std.exp()
        assert abs(float((e0 * e1).mean())) < 0.01                                     # fresh noise every step
        cont = buf["done"][0] == 0
        assert cont.float().mean() > 0.5 and torch.isfinite(buf["rew"]).all()
    venv.close()


@pytest.mark.parametrize("arch,N,D", [((256, 128), 4096, 67), ((64, 32), 1000, 72), ((1024, 512), 2048, 67), ((96, 160), 33, 67)])
def test_policy_forward_kernel_matches_torch(arch, N, D):
    """dm_policy_forward (one launch: both MLP trunks on fp32 MFMA, sampling head, buffer writes) against the PyTorch
    fp32 modules: mean / value within fp32 GEMM reordering error (atol 2e-5 on O(1) outputs), the sampled action equal
    to dm_policy_sample's for the same (seed, env, counter, index), logp of the policy's own formula, clamped env
    action, verbatim observation copy; ragged batch (N not a multiple of 32), D = 72 (DPCombinedEnv), [1024,512]."""
    import ctypes as C
    import torch
    from deepmimic_mujoco_amd import _lib
    from deepmimic_mujoco_amd.ppo import MlpPolicy, FusedPolicyForward
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    pol = MlpPolicy(obs_dim=D, net_arch=arch).to(dev)
    with torch.no_grad():
        pol.log_std.copy_(torch.linspace(-1.0, 0.5, 28))
        pol.action_net.weight.mul_(30.0)                  # O(1) means, so that the clamp at +-2 is exercised
        for m in list(pol.pi) + list(pol.vf) + [pol.action_net, pol.value_net]:
            if hasattr(m, "bias") and m.bias is not None:
                m.bias.normal_(0, 0.1)
    assert FusedPolicyForward.supported(pol, dev)
    fwd = FusedPolicyForward(pol, dev)
    fwd.pack()
    obs = torch.randn(N, D, device=dev) * 0.7
    z = lambda *s: torch.full(s, float("nan"), device=dev)
    mean, act, act_env, logp, val, ocopy = z(N, 28), z(N, 28), z(N, 28), z(N), z(N), z(N, D)
    ctr = torch.tensor([5], dtype=torch.int32, device=dev)
    lo, hi = torch.full((28,), -2.0, device=dev), torch.full((28,), 2.0, device=dev)
    fwd(obs, 1234, ctr, 3, lo, hi, act, act_env, logp, val, obs_copy=ocopy, mean_out=mean)
    with torch.no_grad():
        mean_t = pol.action_net(pol.pi(obs))
        val_t = pol.value_net(pol.vf(obs)).squeeze(-1)
    assert torch.isfinite(mean).all() and torch.isfinite(val).all()
    assert float((mean - mean_t).abs().max()) < 2e-5 * max(1.0, float(mean_t.abs().max()))
    assert float((val - val_t).abs().max()) < 2e-5 * max(1.0, float(val_t.abs().max()))
    assert torch.equal(ocopy, obs)
    assert torch.equal(act_env, torch.clamp(act, lo, hi)) and float((act_env != act).float().mean()) > 0.001
    with torch.no_grad():
        assert torch.allclose(logp, pol._logp(act, mean), atol=3e-4)
    # same draws as the two-kernel path: counter[0] + draw_offset = 8
    a2, e2, l2 = z(N, 28), z(N, 28), z(N)
    ctr8 = torch.tensor([8], dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = _lib.load_library().dm_policy_sample(p(mean), p(pol.log_std), N, 28, C.c_uint64(1234), p(ctr8), p(lo), p(hi), p(a2), p(e2), p(l2),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    assert rc == 0
    assert torch.allclose(a2, act, atol=1e-6) and torch.allclose(l2, logp, atol=3e-4)
    # deterministic head: act = mean
    fwd(obs, 1234, ctr, 0, lo, hi, act, act_env, logp, val, deterministic=True)
    assert torch.allclose(act, mean_t, atol=2e-5 * max(1.0, float(mean_t.abs().max())))
    # weights change -> pack() again
    with torch.no_grad():
        pol.pi[0].weight.mul_(0.5)
    fwd.pack()
    fwd(obs, 1234, ctr, 0, lo, hi, act, act_env, logp, val, mean_out=mean)
    with torch.no_grad():
        assert float((mean - pol.action_net(pol.pi(obs))).abs().max()) < 2e-5 * max(1.0, float(mean_t.abs().max()))


@pytest.mark.parametrize("sub_batches,graph", [(1, False), (2, False), (2, True)])
def test_one_launch_policy_rollout(model, sub_batches, graph):
    """PPO.collect_rollouts on the dm_policy_forward path (two host calls per sub-batch step): the buffers are what the
    unfused loop would store — obs the policy saw, its value / logp for the stored action, the env's reward / done, and
    obs[t + 1] equal to the env's output for (state_t, clamp(act_t)) — checked by replaying the stored actions on a second
    env with the same seed; then one PPO update on them."""
    import torch
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    N, T = 512, 6
    venv = HipDeepMimicVecEnv(N, motion="walk", sub_batches=sub_batches, seed=11)
    twin = HipDeepMimicVecEnv(N, motion="walk", sub_batches=sub_batches, seed=11)
    ppo = PPO(venv, net_arch=(64, 32), n_steps=T, batch_size=1024, n_epochs=1, rollout_graph=graph)
    assert ppo._fused_policy_ok()
    buf = ppo.collect_rollouts()
    assert set(("obs", "act", "rew", "done", "val", "logp", "adv", "ret")) <= set(buf)
    with torch.no_grad():
        o = twin.reset_tensor().clone()
        skip = 1 if graph else 0                     # graph mode: one uncounted warm-up step of every sub-batch precedes the capture
        if skip == 0:
            assert torch.equal(buf["obs"][0], o)
        for t in range(T):
            m = ppo.policy.action_net(ppo.policy.pi(buf["obs"][t]))
            assert torch.allclose(buf["logp"][t], ppo.policy._logp(buf["act"][t], m), atol=3e-4)
            assert torch.allclose(buf["val"][t], ppo.policy.predict_values(buf["obs"][t]), atol=2e-5)
        if skip == 0:
            for t in range(T):
                out = twin.step_tensor(torch.clamp(buf["act"][t], ppo.act_lo, ppo.act_hi))
                assert torch.equal(out["rew"], buf["rew"][t]) and torch.equal(out["done"].float(), buf["done"][t])
                if t + 1 < T:
                    assert torch.equal(out["obs"], buf["obs"][t + 1])
            assert torch.equal(out["obs"], ppo._last_obs)
        e0 = (buf["act"][0] - ppo.policy.action_net(ppo.policy.pi(buf["obs"][0]))) / ppo.policy.log_std.exp()
        e1 = (buf["act"][1] - ppo.policy.action_net(ppo.policy.pi(buf["obs"][1]))) / ppo.policy.log_std.exp()
        assert abs(float(e0.mean())) < 0.03 and abs(float(e0.std()) - 1) < 0.03 and abs(float((e0 * e1).mean())) < 0.03
    first = buf["obs"].clone()
    ppo.train(buf)
    assert np.isfinite(ppo.stats["loss"])
    buf = ppo.collect_rollouts()                     # second rollout: continues the episodes with the updated (re-packed) weights
    assert not torch.equal(first, buf["obs"]) and ppo.num_timesteps == 2 * T * N
    with torch.no_grad():
        assert torch.allclose(buf["val"][2], ppo.policy.predict_values(buf["obs"][2]), atol=2e-5)
    venv.close(); twin.close()


@pytest.mark.parametrize("arch,B,D,ent", [((256, 128), 4096, 67, 0.0), ((64, 32), 256, 72, 0.01), ((96, 160), 128, 67, 0.0)])
def test_fused_mlp_grad_matches_autograd(arch, B, D, ent):
    """dm_ppo_mlp_grad (pack + fused forward / loss / input-gradient kernel + six-layer weight-gradient kernel) against
    autograd on the PyTorch-op loss of SB3's PPO.train: loss terms and every parameter gradient, on a minibatch whose
    ratios straddle the clip range; tolerance 3e-4 of the largest gradient entry (fp32, different summation order)."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy, FusedMlpGrad
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    pol = MlpPolicy(obs_dim=D, net_arch=arch)
    ppo = PPO(None, net_arch=arch, batch_size=B, device=dev, ent_coef=ent, policy=pol, use_hip_graph=False)
    pol = ppo.policy
    with torch.no_grad():
        pol.log_std.copy_(torch.linspace(-0.5, 0.3, 28))
        pol.action_net.weight.mul_(20.0)
        for m in pol.modules():
            if isinstance(m, torch.nn.Linear):
                m.bias.normal_(0, 0.1)
        obs = torch.randn(B, D, device=dev) * 0.7
        mean = pol.action_net(pol.pi(obs))
        act = mean + pol.log_std.exp() * torch.randn(B, 28, device=dev)
        old_logp = pol._logp(act, mean) + 0.15 * torch.randn(B, device=dev)
        adv = torch.randn(B, device=dev) * 2 + 0.3
        ret = torch.randn(B, device=dev)
    assert FusedMlpGrad.supported(pol, B)
    F = torch.nn.functional
    trunk = lambda seq, x: torch.tanh(F.linear(torch.tanh(F.linear(x, seq[0].weight, seq[0].bias)), seq[2].weight, seq[2].bias))
    mean_r = F.linear(trunk(pol.pi, obs), pol.action_net.weight, pol.action_net.bias)           # plain PyTorch ops only
    value_r = F.linear(trunk(pol.vf, obs), pol.value_net.weight, pol.value_net.bias).squeeze(-1)
    a_n = (adv - adv.mean()) / (adv.std() + 1e-8)
    rat = torch.exp(pol._logp(act, mean_r) - old_logp)
    ent_r = (0.5 + 0.5 * np.log(2 * np.pi) + pol.log_std).sum()
    loss_t = (-torch.min(a_n * rat, a_n * torch.clamp(rat, 1 - ppo.clip_range, 1 + ppo.clip_range)).mean()
              + ppo.vf_coef * F.mse_loss(value_r, ret) - ppo.ent_coef * ent_r)
    for p in pol.parameters():
        p.grad = None
    loss_t.backward()
    ref = {id(p): p.grad.clone() for p in pol.parameters()}
    opt = ppo.optimizer
    opt.zero_grad()
    mg = FusedMlpGrad(pol, opt, B)
    loss_f = mg(obs, act, adv, ret, old_logp, ppo.clip_range, ppo.vf_coef, ppo.ent_coef, True)
    torch.cuda.synchronize()
    assert abs(float(loss_f) - float(loss_t.detach())) < 2e-5 * max(1.0, abs(float(loss_t.detach())))
    with torch.no_grad():
        ratio = torch.exp(pol._logp(act, mean) - old_logp)
        clipfrac = float(((ratio - 1).abs() > ppo.clip_range).float().mean())
    assert 0.05 < clipfrac < 0.95 and abs(float(mg.out8[5]) - clipfrac) < 1e-6
    names = {id(p): n for n, p in pol.named_parameters()}
    for p, g in zip(opt.params, opt.slices):
        r = ref[id(p)]
        scale = float(r.abs().max())
        assert scale > 0, names[id(p)]
        assert float((g - r).abs().max()) < 3e-4 * scale + 1e-8, (names[id(p)], float((g - r).abs().max()), scale)
    # the optimizer step on top of it: same parameters as the autograd path after three minibatch steps
    res = []
    for fused in (True, False):
        torch.manual_seed(7)
        q = PPO(None, net_arch=arch, batch_size=B, device=dev, ent_coef=ent, policy=MlpPolicy(obs_dim=D, net_arch=arch),
                use_hip_graph=False, fused_mlp=fused)
        for _ in range(3):
            q._minibatch_step(obs, act, adv, ret, old_logp)
        res.append(torch.cat([p.detach().reshape(-1) for p in q.policy.parameters()]))
    assert torch.allclose(res[0], res[1], rtol=1e-4, atol=2e-6), float((res[0] - res[1]).abs().max())


def test_bf16_learner_option_tracks_the_fp32_learner():
    """PPO(mlp_dtype=torch.bfloat16): the library-path learner with bf16 MFMA GEMMs / activations against bf16 shadows of the
    fp32 master weights (fp32 loss kernel, gradient arena and Adam).  On the same minibatch its gradient points where the
    fp32 learner's does (cosine > 0.995, relative L2 difference < 8 %), five optimizer steps reduce the loss like the fp32
    run's, and the shadow equals the rounded master weights after every update."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy
    dev = torch.device("cuda", 0)
    B = 2048
    g = torch.Generator(device=dev); g.manual_seed(7)
    obs = torch.randn(B, 67, device=dev, generator=g)
    act = torch.randn(B, 28, device=dev, generator=g) * 0.5
    adv = torch.randn(B, device=dev, generator=g)
    ret = torch.randn(B, device=dev, generator=g)
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(11)
        pol = MlpPolicy(net_arch=(1024, 512)).to(dev)
        ppo = PPO(None, policy=pol, device=dev, batch_size=B, use_hip_graph=False, learning_rate=1e-3, mlp_dtype=dt, fused_wide=False)
        with torch.no_grad():
            _, logp, _ = pol.evaluate_actions(obs, act)
        old_logp = (logp + 0.1 * torch.randn(B, device=dev, generator=torch.Generator(device=dev).manual_seed(3))).contiguous()
        losses = []
        for it in range(5):
            losses.append(float(ppo._minibatch_step(obs, act, adv, ret, old_logp)))
            if it == 0:
                grad0 = ppo.optimizer.flat_g.clone()
            if dt == torch.bfloat16:
                assert torch.equal(ppo.optimizer.flat_pb, ppo.optimizer.flat_p.to(torch.bfloat16))
        res[dt] = (grad0, losses)
    g32, l32 = res[torch.float32]
    g16, l16 = res[torch.bfloat16]
    cos = float(torch.dot(g32, g16) / (g32.norm() * g16.norm()))
    rel = float((g32 - g16).norm() / g32.norm())
    print("bf16 learner: gradient cosine %.5f, relative L2 diff %.4f, losses fp32 %s bf16 %s" % (cos, rel, l32, l16))
    assert cos > 0.995 and rel < 0.08
    assert abs(l16[0] - l32[0]) < 0.02 * max(1.0, abs(l32[0]))
    assert l16[-1] < l16[0] and abs(l16[-1] - l32[-1]) < 0.05 * max(1.0, abs(l32[-1]))


@pytest.mark.parametrize("arch,D,A,B", [((1024, 512), 67, 28, 4096), ((1024, 512), 98, 23, 1024), ((512, 256), 72, 28, 512), ((256, 128), 85, 23, 256),
                                       ((512, 128), 112, 32, 192), ((256, 256), 1, 1, 64)])     # the last two: row tiles not a multiple of 4 (plain block order), widest / narrowest D and A
def test_wide_fused_learner_gradient_matches_fp32_autograd(arch, D, A, B):
    """dm_ppo_wide_grad (csrc/dm_ppo_wide.hip): the fused bf16 matrix-pipe forward / loss / backward chain of the [1024,512]-class
    net (BASELINE configs 3-5) + the split-K bf16 weight-gradient kernel, against the fp32 loss of SB3's PPO.train written with plain
    PyTorch ops and autograd.  bf16 operands (8-bit mantissa), fp32 accumulation: per parameter tensor the gradient points where
    the fp32 gradient does (cosine > 0.999) with a relative L2 difference < 3 %, the loss agrees to 1e-3 relative; five optimizer
    steps through the product path reduce the loss like the fp32 learner's."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy, WideMlpGrad
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    obs = torch.randn(B, D, device=dev, generator=g)
    act = torch.randn(B, A, device=dev, generator=g) * 0.5
    adv = torch.randn(B, device=dev, generator=g)
    ret = torch.randn(B, device=dev, generator=g)
    torch.manual_seed(13)
    pol = MlpPolicy(obs_dim=D, act_dim=A, net_arch=arch).to(dev)
    assert WideMlpGrad.supported(pol, B)
    with torch.no_grad():
        pol.log_std.add_(0.1 * torch.randn(A, device=dev, generator=g))
        _, logp, _ = pol.evaluate_actions(obs, act)
    old_logp = (logp + 0.1 * torch.randn(B, device=dev, generator=g)).contiguous()
    ppo = PPO(None, policy=pol, device=dev, batch_size=B, use_hip_graph=False, learning_rate=3e-4, mlp_dtype=torch.bfloat16, ent_coef=0.01)
    assert ppo._wide_ok
    loss_w, begin = ppo._minibatch_grad(obs, act, adv, ret, old_logp)
    torch.cuda.synchronize()
    assert begin is False
    gw = {id(p): s_.clone() for p, s_ in zip(ppo.optimizer.params, ppo.optimizer.slices)}
    # the fp32 reference: the loss of SB3's PPO.train in PyTorch ops; the layers' fp32 backward files its gradients in the same arena
    ppo.optimizer.zero_grad()
    loss_t = ppo._loss_torch(obs, act, adv, ret, old_logp)
    loss_t.backward()
    ppo.optimizer.gather_grads()
    torch.cuda.synchronize()
    loss_t = loss_t.detach()
    assert abs(float(loss_w) - float(loss_t)) < 2e-3 * max(1.0, abs(float(loss_t))), (float(loss_w), float(loss_t))
    ref = {id(p): s_.clone() for p, s_ in zip(ppo.optimizer.params, ppo.optimizer.slices)}
    worst_cos, worst_rel, bad = 1.0, 0.0, []
    for name, p_ in pol.named_parameters():
        a, b = gw[id(p_)].reshape(-1), ref[id(p_)].reshape(-1)
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        worst_cos, worst_rel = min(worst_cos, cos), max(worst_rel, rel)
        bad.append((name, round(cos, 5), round(rel, 4)))
    assert worst_cos > 0.999 and worst_rel < 0.03, bad
    print("wide fused learner %s D %d A %d B %d: loss %.6f / %.6f, worst cosine %.5f, worst relative L2 %.4f" % (arch, D, A, B, float(loss_w), float(loss_t), worst_cos, worst_rel))
    l0 = float(ppo._minibatch_step(obs, act, adv, ret, old_logp))
    for _ in range(4):
        l1 = float(ppo._minibatch_step(obs, act, adv, ret, old_logp))
    assert l1 < l0


def test_concurrent_streams_are_probed_and_shared():
    """streams.concurrent_streams: k streams that the probe itself finds pairwise concurrent (two torch streams out of the pool share
    a hardware queue about one time in six, and sub-batch pipelines on such a pair run slower than one batch), found once per
    process and device: every caller gets the same set."""
    import torch
    from deepmimic_mujoco_amd import streams
    dev = torch.device("cuda", 0)
    a = streams.concurrent_streams(dev, 2)
    assert len(a) == 2 and a[0].cuda_stream != a[1].cuda_stream
    assert not streams._serialized(torch, dev, a[0], a[1])
    b = streams.concurrent_streams(dev, 2)
    assert [s.cuda_stream for s in a] == [s.cuda_stream for s in b]
    c = streams.concurrent_streams(dev, 3)
    assert [s.cuda_stream for s in c[:2]] == [s.cuda_stream for s in a] and len({s.cuda_stream for s in c}) == 3


def test_flat_adam_step_with_the_next_gather_riding_on_it():
    """dm_flat_adam_step_gather: the gather of the next minibatch as extra blocks of Adam's norm launch — parameters, moments and
    step count bit-identical to dm_flat_adam_step, gathered rows bit-identical to dm_ppo_gather."""
    import torch
    from deepmimic_mujoco_amd.ppo import PPO, MlpPolicy
    dev = torch.device("cuda", 0)
    n, B, D, A = 8192, 1024, 67, 28
    g = torch.Generator(device=dev); g.manual_seed(23)
    flat = dict(obs=torch.randn(n, D, device=dev, generator=g), act=torch.randn(n, A, device=dev, generator=g),
                adv=torch.randn(n, device=dev, generator=g), ret=torch.randn(n, device=dev, generator=g), logp=torch.randn(n, device=dev, generator=g))
    idx = torch.randperm(n, device=dev, generator=g)[:B].contiguous()
    res = []
    for ride in (False, True):
        torch.manual_seed(5)
        pol = MlpPolicy(obs_dim=D, act_dim=A, net_arch=(256, 128)).to(dev)
        ppo = PPO(None, policy=pol, device=dev, batch_size=B, use_hip_graph=False)
        opt = ppo.optimizer
        gg = torch.Generator(device=dev); gg.manual_seed(99)
        out = ppo._static_minibatch()
        for _ in range(3):
            opt.flat_g.copy_(torch.randn(opt.n, device=dev, generator=gg) * 0.05)
            if ride:
                opt.step(begin=True, gather_next=(flat, idx, out))
            else:
                opt.step(begin=True)
                ppo._gather_minibatch(flat, idx, out)
        torch.cuda.synchronize()
        res.append((opt.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.state2[:2].clone(), {k: v.clone() for k, v in out.items()}))
    a, b = res
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]) and torch.equal(a[4][k], flat[k][idx]), k
