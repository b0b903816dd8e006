"""GPU parity: HIP path (through the C-ABI) vs the fp64 CPU oracle on identical inputs.

Tolerances (fp32 device arithmetic vs fp64 oracle), stated per BASELINE.json's
north_star: per-step qpos L-inf error < 1e-4 teacher-forced; contact (geom1, geom2)
index sets bit-exact away from the activation boundary (|dist - margin| > 1e-5).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_QPOS = 1e-4
TOL_QVEL = 5e-3      # qvel carries qacc * h; qacc magnitudes reach 1e3 under +-400 N m torques
TOL_OBS = 2e-3
TOL_REW = 1e-4


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _engine(model, clips, n, torch, auto_reset=False, motion="walk", **kw):
    from deepmimic_mujoco_amd._lib import HipEngine
    eng = HipEngine(model, n, auto_reset=auto_reset, **kw)
    eng.load_clip(0, clips[motion])
    return eng


def _oracle_states(model, clip, nenv, nsteps, scale, seed, caps=(32, 128)):
    """Roll the oracle with random actions; record (state before, action, state after, contacts)."""
    from oracle.oracle import OracleSim
    rng = np.random.default_rng(seed)
    recs = []
    for e in range(nenv):
        s = OracleSim(model)
        s.set_caps(*caps)
        s.env_reset(clip, int(rng.integers(0, clip.L)))
        for t in range(nsteps):
            a = rng.uniform(-scale, scale, 28)
            before = dict(qpos=s.get("qpos"), qvel=s.get("qvel"), warm=s.get("qacc_warmstart"),
                          ctrl=s.get("ctrl"), idx=s.env.idx_curr, eplen=s.env.episode_length)
            obs, rew, done, terms, reason = s.env_step(clip, a)
            con = s.get("contact")
            recs.append(dict(before=before, action=a, qpos=s.get("qpos"), qvel=s.get("qvel"),
                             warm=s.get("qacc_warmstart"), obs=obs, rew=rew, done=done, terms=terms,
                             reason=reason, contact=con, nefc=s.nefc, xpos=s.get("xpos"),
                             stage_ncon=[s.geti("stage_ncon%d" % k) for k in range(4)],
                             stage_nefc=[s.geti("stage_nefc%d" % k) for k in range(4)],
                             geom_xpos=s.get("geom_xpos"), cvel=s.get("cvel"), qacc=s.get("qacc")))
            if done:
                s.env_reset(clip, int(rng.integers(0, clip.L)))
    return recs


def _run_teacher_forced(model, clips, oracle_clips, torch, scale, seed, nenv=16, nsteps=60, motion="walk", tile=1,
                        oracle_model=None, **engine_kw):
    recs = _oracle_states(oracle_model or model, oracle_clips[motion], nenv, nsteps, scale, seed) * tile
    n = len(recs)
    eng = _engine(model, clips, n, torch, motion=motion, **engine_kw)
    dev = eng.device
    f32 = lambda key: torch.tensor(np.array([r["before"][key] for r in recs]), dtype=torch.float32, device=dev)
    eng.set_state(f32("qpos"), f32("qvel"), f32("warm"), f32("ctrl"))
    eng.set_counters(torch.tensor([r["before"]["idx"] for r in recs], dtype=torch.int32, device=dev),
                     torch.tensor([r["before"]["eplen"] for r in recs], dtype=torch.int32, device=dev))
    dbg = eng.enable_debug()
    out = eng.alloc_outputs()
    act = torch.tensor(np.array([r["action"] for r in recs]), dtype=torch.float32, device=dev)
    eng.step(act, out)
    torch.cuda.synchronize()
    qpos, qvel, warm, _ = [t.cpu().numpy().astype(np.float64) for t in eng.get_state()]
    dbg = dbg.cpu().numpy()
    res = dict(
        qpos=np.abs(qpos - np.array([r["qpos"] for r in recs])).max(axis=1),
        qvel=np.abs(qvel - np.array([r["qvel"] for r in recs])).max(axis=1),
        obs=np.abs(out["obs"].cpu().numpy() - np.array([r["obs"] for r in recs])).max(axis=1),
        rew=np.abs(out["rew"].cpu().numpy() - np.array([r["rew"] for r in recs])),
        done=(out["done"].cpu().numpy() != np.array([r["done"] for r in recs])),
        xpos=np.abs(dbg[:, 0:42] - np.array([r["xpos"].ravel() for r in recs])).max(axis=1),
        cvel=np.abs(dbg[:, 90:174] - np.array([r["cvel"].ravel() for r in recs])).max(axis=1),
    )
    # contact index sets
    mism = []
    for i, r in enumerate(recs):
        ncon = int(dbg[i, 242])
        gpu = [(int(dbg[i, 256 + 3 * c]), int(dbg[i, 257 + 3 * c])) for c in range(ncon)]
        ora = [(int(c[13]), int(c[14])) for c in r["contact"]]
        if gpu != ora:
            # tolerate flips only where the oracle distance is within 1e-5 of the activation margin
            near = [abs(c[0] - 0.001) < 1e-5 for c in r["contact"]]
            mism.append((i, gpu, ora, any(near)))
    res["contact_mismatch"] = mism
    # activation flips inside an RK stage (a contact / limit whose distance sits at the fp32
    # rounding of its activation threshold) show up as a per-stage count difference
    packs = dbg[:, 247:249].copy().view(np.int32)
    flips = []
    for i, r in enumerate(recs):
        gn = [(int(packs[i, 0]) >> (8 * k)) & 0xFF for k in range(4)]
        ge = [(int(packs[i, 1]) >> (8 * k)) & 0xFF for k in range(4)]
        if gn != r["stage_ncon"] or ge != r["stage_nefc"]:
            flips.append(i)
    res["stage_flips"] = flips
    res["nefc_gpu"] = dbg[:, 243]
    res["nefc_oracle"] = np.array([r["nefc"] for r in recs])
    res["recs"] = recs
    eng.close()
    return res


@pytest.mark.parametrize("scale,seed", [(0.0, 1), (0.3, 2), (2.0, 3)])
def test_teacher_forced_step_parity(model, clips, oracle_clips, torch_mod, scale, seed):
    res = _run_teacher_forced(model, clips, oracle_clips, torch_mod, scale, seed)
    hard = [m for m in res["contact_mismatch"] if not m[3]]
    print("scale", scale, "qpos", res["qpos"].max(), "qvel", res["qvel"].max(), "obs", res["obs"].max(),
          "rew", res["rew"].max(), "xpos", res["xpos"].max(), "cvel", res["cvel"].max(),
          "contact mismatches", len(res["contact_mismatch"]), "hard", len(hard), "stage flips", len(res["stage_flips"]))
    assert len(hard) == 0, hard[:3]
    ok = np.ones(len(res["qpos"]), bool)
    for m in res["contact_mismatch"]:
        ok[m[0]] = False  # boundary flips are excluded from the numeric comparison
    ok[res["stage_flips"]] = False
    assert len(res["stage_flips"]) <= 0.01 * len(ok), "activation flips must stay rare"
    assert res["qpos"][ok].max() < TOL_QPOS
    assert res["qvel"][ok].max() < TOL_QVEL
    assert res["obs"][ok].max() < TOL_OBS
    assert res["rew"][ok].max() < TOL_REW
    assert not res["done"][ok].any()


@pytest.mark.parametrize("motion", ["walk", "run", "dance_b", "spinkick"])
def test_force_state_clip_playback(model, clips, oracle_clips, torch_mod, motion):
    """Playing a clip onto itself (deepmimic_env.py:561-568 loop_motion / :570 check_rewards)."""
    from oracle.oracle import OracleSim
    torch = torch_mod
    mc, oc = clips[motion], oracle_clips[motion]
    q, v, _, _ = mc.tables()
    L = len(q)
    eng = _engine(model, clips, L, torch, motion=motion)
    dev = eng.device
    obs0 = torch.zeros(L, 67, device=dev)
    eng.reset(obs0, idx_init=torch.arange(L, dtype=torch.int32, device=dev))
    out = eng.alloc_outputs()
    eng.step_forced(torch.tensor(q, dtype=torch.float32, device=dev), torch.tensor(v, dtype=torch.float32, device=dev), out)
    torch.cuda.synchronize()
    s = OracleSim(model)
    s.set_caps(32, 128)
    eobs, erew, eterms = [], [], []
    for i in range(L):
        s.env_reset(oc, i)
        o, r, d, t, _ = s.env_step(oc, np.zeros(28), force_state=(q[i], v[i]))
        eobs.append(o); erew.append(r); eterms.append(t)
    gobs, grew, gterms = out["obs"].cpu().numpy(), out["rew"].cpu().numpy(), out["terms"].cpu().numpy()
    print(motion, "obs", np.abs(gobs - np.array(eobs)).max(), "rew", np.abs(grew - np.array(erew)).max(),
          "terms", np.abs(gterms - np.array(eterms)).max(0))
    assert np.abs(gobs - np.array(eobs)).max() < TOL_OBS
    assert np.abs(grew - np.array(erew)).max() < TOL_REW
    assert np.abs(gterms - np.array(eterms)).max() < 1e-4
    # analytic identity: reward_qvel == 1 exactly, reward_config ~ 1
    assert np.abs(gterms[:, 1] - 1.0).max() < 1e-6
    # reward_config: exact on raw frames; on lerped frames the target root quaternion is the
    # un-normalised lerp while qpos[3:7] is normalised, so the pitch term leaves a small residue
    assert np.abs(gterms[:, 0] - 1.0).max() < 2e-3
    eng.close()


@pytest.mark.parametrize("motion,scale,seed", [("run", 1.0, 11), ("spinkick", 0.5, 12), ("dance_b", 2.0, 13)])
def test_teacher_forced_other_clips(model, clips, oracle_clips, torch_mod, motion, scale, seed):
    res = _run_teacher_forced(model, clips, oracle_clips, torch_mod, scale, seed, nenv=8, nsteps=40, motion=motion)
    ok = np.ones(len(res["qpos"]), bool)
    for m in res["contact_mismatch"]:
        assert m[3], m
        ok[m[0]] = False
    ok[res["stage_flips"]] = False
    assert ok.mean() > 0.98
    print(motion, "qpos", res["qpos"][ok].max(), "obs", res["obs"][ok].max())
    assert res["qpos"][ok].max() < TOL_QPOS and res["obs"][ok].max() < TOL_OBS and res["rew"][ok].max() < TOL_REW


def test_full_size_batch_properties(model, clips, torch_mod):
    """BASELINE size (4096 envs): determinism and env-permutation invariance — shuffling the env order
    must permute every output bit-exactly (no cross-env leakage), and a repeated run is bit-identical."""
    torch = torch_mod
    N = 4096
    mc = clips["walk"]
    L = len(mc.data_config)

    def run(perm):
        eng = _engine(model, clips, N, torch)
        dev = eng.device
        out = eng.alloc_outputs()
        idx = (torch.arange(N, device=dev) * 7 % L).to(torch.int32)[perm]
        eng.reset(out["obs"], idx_init=idx)
        g = torch.Generator(device="cpu").manual_seed(3)
        acts = [(torch.rand(N, 28, generator=g) * 3 - 1.5) for _ in range(6)]
        res = []
        for a in acts:
            eng.step(a.to(dev)[perm].contiguous(), out)
            res.append((out["obs"].clone(), out["rew"].clone(), out["done"].clone()))
        q = eng.get_state()[0].clone()
        eng.close()
        return res, q

    ident = torch.arange(N, device="cuda")
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(9)).cuda()
    r0, q0 = run(ident)
    r1, q1 = run(ident)
    r2, q2 = run(perm)
    assert torch.equal(q0, q1) and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(r0, r1))
    assert torch.equal(q0[perm], q2)
    for a, b in zip(r0, r2):
        assert torch.equal(a[0][perm], b[0]) and torch.equal(a[1][perm], b[1]) and torch.equal(a[2][perm], b[2])
    assert torch.isfinite(q0).all() and float(r0[-1][1].mean()) > 0.0


def test_wide_constraint_path_and_caps(model, clips, oracle_clips, torch_mod):
    """Lying poses: ~19 floor contacts x 4 pyramid rows = 76+ rows take the two-rows-per-lane path
    (65..128 rows).  Contact lists, row counts and qacc must match the oracle (caps 32 contacts / 128 rows);
    a dynamic step from such a pose must agree too."""
    from oracle.oracle import OracleSim
    torch = torch_mod
    rng = np.random.default_rng(11)
    poses = []
    s = OracleSim(model)
    s.set_caps(32, 128)
    for pitch in (np.pi / 2, -np.pi / 2, np.pi / 2 + 0.05):
        for _ in range(4):
            q = model.qpos0.copy()
            q[2] = 0.05 + rng.uniform(0, 0.02)
            q[3:7] = [np.cos(pitch / 2), 0, np.sin(pitch / 2), 0]
            q[7:] = rng.uniform(-0.15, 0.15, 28)
            v = rng.normal(size=34) * 0.3
            assert s.set_state(q, v) == 0
            if 64 < s.nefc <= 128 and s.ncon <= 32 and not np.any(np.abs(s.get("contact")[:, 0] - 0.001) < 2e-5):
                poses.append((q, v, s.get("contact").copy(), s.get("qacc").copy(), s.nefc, s.geti("solver_iter")))
    assert len(poses) >= 4, len(poses)
    n = len(poses)
    eng = _engine(model, clips, n, torch)
    dbg = eng.enable_debug()
    dev = eng.device
    Q = torch.tensor(np.array([p[0] for p in poses]), dtype=torch.float32, device=dev)
    V = torch.tensor(np.array([p[1] for p in poses]), dtype=torch.float32, device=dev)
    out = eng.alloc_outputs()
    eng.step_forced(Q, V, out)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    for i, (q, v, con, qacc, nefc, it) in enumerate(poses):
        ncon = int(d[i, 242])
        gpu = [(int(d[i, 256 + 3 * c]), int(d[i, 257 + 3 * c])) for c in range(ncon)]
        assert gpu == [(int(c[13]), int(c[14])) for c in con]
        assert int(d[i, 243]) == nefc and d[i, 246] == 0
        rel = np.abs(d[i, 174:208] - qacc).max() / max(1.0, np.abs(qacc).max())
        print("wide pose", i, "nefc", nefc, "sweeps gpu/oracle", int(d[i, 244]), it, "qacc rel err %.2e" % rel)
        assert rel < 5e-3
    # one dynamic step from the same states, teacher-forced against the oracle
    eng.set_state(Q, V, run_forward=False)
    w = torch.zeros(n, 34, device=dev)
    eng.set_state(Q, V, w, torch.zeros(n, 28, device=dev))
    eng.step(torch.zeros(n, 28, device=dev), out)
    qg = eng.get_state()[0].double().cpu().numpy()
    worst = 0.0
    for i, (q, v, *_rest) in enumerate(poses):
        s2 = OracleSim(model)
        s2.set_caps(32, 128)
        s2.set("qpos", q); s2.set("qvel", v)
        assert s2.step() == 0
        worst = max(worst, np.abs(qg[i] - s2.get("qpos")).max())
    print("wide path dynamic step: qpos max err %.3g" % worst)
    assert worst < TOL_QPOS
    eng.close()


def test_thousand_step_teacher_forced_trajectory(model, clips, oracle_clips, torch_mod):
    """BASELINE target: per-step qpos L-inf error < 1e-4 vs the CPU path over 1000 steps on identical
    (state, action) sequences.  The oracle runs 1000 consecutive steps (small random torques, resets on
    done); the HIP path is reset to the oracle's (qpos, qvel, qacc_warmstart, ctrl) before every step."""
    res = _run_teacher_forced(model, clips, oracle_clips, torch_mod, 0.25, 77, nenv=1, nsteps=1000)
    ok = np.ones(len(res["qpos"]), bool)
    for m in res["contact_mismatch"]:
        assert m[3], m
        ok[m[0]] = False
    ok[res["stage_flips"]] = False
    print("1000 steps: qpos max %.3g (flips excluded: %d), qvel max %.3g" % (res["qpos"][ok].max(), (~ok).sum(), res["qvel"][ok].max()))
    assert len(res["qpos"]) == 1000 and ok.sum() >= 990
    assert res["qpos"][ok].max() < TOL_QPOS


def test_narrowphase_coverage_all_pair_types(model, clips, oracle_clips, torch_mod):
    """Random self-colliding poses (joint angles over their full ranges, root in the air or near the floor)
    so that every pair-type routine fires, including capsule-box and box-box; contact lists (geom ids in
    canonical order), distances and the resulting qacc must agree with the oracle."""
    from oracle.oracle import OracleSim
    torch = torch_mod
    rng = np.random.default_rng(2024)
    lo, hi = model.jnt_range[1:, 0], model.jnt_range[1:, 1]
    s = OracleSim(model)
    s.set_caps(32, 128)
    want = {(0, 2): 8, (0, 3): 8, (0, 6): 8, (2, 2): 8, (2, 3): 12, (2, 6): 12, (3, 3): 12, (3, 6): 16, (6, 6): 16}
    got = {k: 0 for k in want}
    samples = []
    tries = 0
    while any(got[k] < want[k] for k in want) and tries < 60000:
        tries += 1
        q = model.qpos0.copy()
        q[7:] = rng.uniform(lo, hi)
        if rng.random() < 0.5:                       # bias towards feet/legs meeting
            q[7 + 13:7 + 28] = rng.uniform(lo[13:], hi[13:]) * rng.uniform(0.2, 1.0)
        quat = rng.normal(size=4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[2] = rng.choice([2.0, 2.0, rng.uniform(0.2, 1.0)])
        v = rng.normal(size=34) * 0.5
        if s.set_state(q, v) != 0:
            continue
        con = s.get("contact")
        if len(con) == 0 or len(con) > 30 or s.nefc > 60:
            continue
        if np.any(np.abs(con[:, 0] - 0.001) < 2e-5):   # keep away from the activation threshold
            continue
        types = {(int(model.geom_type[int(c[13])]), int(model.geom_type[int(c[14])])) for c in con}
        need = [t for t in types if t in want and got[t] < want[t]]
        if not need:
            continue
        for t in types:
            if t in got:
                got[t] += 1
        samples.append((q, v, con.copy(), s.get("qacc").copy(), s.nefc))
    print("narrowphase samples", len(samples), "tries", tries, got)
    assert all(got[k] >= want[k] for k in want), got
    n = len(samples)
    eng = _engine(model, clips, n, torch)
    dbg = eng.enable_debug()
    dev = eng.device
    Q = torch.tensor(np.array([x[0] for x in samples]), dtype=torch.float32, device=dev)
    V = torch.tensor(np.array([x[1] for x in samples]), dtype=torch.float32, device=dev)
    out = eng.alloc_outputs()
    eng.step_forced(Q, V, out)        # set_state + forward from a zero warm start, as the oracle did
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    worst_dist, worst_acc = 0.0, 0.0
    for i, (q, v, con, qacc, nefc) in enumerate(samples):
        ncon = int(d[i, 242])
        gpu = [(int(d[i, 256 + 3 * c]), int(d[i, 257 + 3 * c])) for c in range(ncon)]
        ora = [(int(c[13]), int(c[14])) for c in con]
        assert gpu == ora, (i, gpu, ora)
        assert int(d[i, 243]) == nefc
        gd = np.array([d[i, 258 + 3 * c] for c in range(ncon)])
        worst_dist = max(worst_dist, np.abs(gd - con[:, 0]).max())
        scale = max(1.0, np.abs(qacc).max())
        worst_acc = max(worst_acc, np.abs(d[i, 174:208] - qacc).max() / scale)
    print("contact dist max err %.3g, qacc max rel err %.3g" % (worst_dist, worst_acc))
    assert worst_dist < 2e-6
    assert worst_acc < 2e-3
    eng.close()


def _gates(res):
    hard = [m for m in res["contact_mismatch"] if not m[3]]
    assert len(hard) == 0, hard[:3]
    ok = np.ones(len(res["qpos"]), bool)
    for m in res["contact_mismatch"]:
        ok[m[0]] = False
    ok[res["stage_flips"]] = False
    assert len(res["stage_flips"]) <= 0.01 * len(ok)
    assert res["qpos"][ok].max() < TOL_QPOS and res["qvel"][ok].max() < TOL_QVEL
    assert res["obs"][ok].max() < TOL_OBS and res["rew"][ok].max() < TOL_REW
    assert not res["done"][ok].any()
    return ok


@pytest.mark.parametrize("how", ["xml", "config"])
def test_euler_integrator_parity(model, clips, oracle_clips, torch_mod, how):
    """north_star's semi-implicit Euler option (MuJoCo's mj_Euler: one forward evaluation, joint damping integrated
    implicitly through a second factorisation of M + h B): HIP vs the oracle's Euler on 960 teacher-forced states at
    full torque scale, selected either by the model (<option integrator="Euler">) or by DmConfig.integrator on the RK4
    model; and the Euler step differs from the RK4 step of the same state (the option really switches)."""
    from deepmimic_mujoco_amd.model import INT_EULER, compile_mjcf
    me = compile_mjcf()
    me.integrator = me.cstruct.integrator = INT_EULER
    if how == "xml":
        res = _run_teacher_forced(me, clips, oracle_clips, torch_mod, 2.0, 5)
    else:
        res = _run_teacher_forced(model, clips, oracle_clips, torch_mod, 2.0, 5, oracle_model=me, integrator="Euler")
    ok = _gates(res)
    print("euler", how, "qpos", res["qpos"][ok].max(), "qvel", res["qvel"][ok].max(), "obs", res["obs"][ok].max())
    rk = _run_teacher_forced(model, clips, oracle_clips, torch_mod, 2.0, 5, oracle_model=me, integrator="RK4")
    assert np.median(rk["qvel"]) > 10 * np.median(res["qvel"])     # RK4 engine vs Euler oracle: visibly different


def test_three_wave_kernel_variant_parity(model, clips, oracle_clips, torch_mod):
    """Batches of >= 3072 envs run dm_step_kernel_w3 (the same body compiled for 3 waves/SIMD, 168 VGPRs): same parity
    gates as the two-wave kernel, on 960 oracle states tiled to 6720 envs."""
    res = _run_teacher_forced(model, clips, oracle_clips, torch_mod, 2.0, 3, tile=7)
    assert len(res["qpos"]) == 6720
    hard = [m for m in res["contact_mismatch"] if not m[3]]
    assert len(hard) == 0
    ok = np.ones(len(res["qpos"]), bool)
    for m in res["contact_mismatch"]:
        ok[m[0]] = False
    ok[res["stage_flips"]] = False
    assert len(res["stage_flips"]) <= 0.01 * len(ok)
    assert res["qpos"][ok].max() < TOL_QPOS and res["qvel"][ok].max() < TOL_QVEL
    assert res["obs"][ok].max() < TOL_OBS and res["rew"][ok].max() < TOL_REW
    # identical inputs in every tile -> identical outputs (the variant is deterministic across slots)
    q = res["qpos"].reshape(7, -1)
    assert np.array_equal(q[0], q[3]) and np.array_equal(q[0], q[6])


def test_f8_stale_contact_slots_option(model, clips, oracle_clips, torch_mod):
    """SURVEY F8 as a selectable behaviour (DmConfig.stale_contact_slots, default off): the reference's foot-contact observation
    (src/deepmimic_env.py:78-105) scans ALL of mujoco-py's `mjdata.contact`, so a foot bit stays set while a stale slot >= ncon
    still holds the foot-floor pair.  Sequential teacher-forced rollouts, kernel against oracle with the option on both sides:
    the two foot bits agree on every step, and the option is exercised (steps where the stale scan differs from the active one)."""
    from deepmimic_mujoco_amd._lib import HipEngine
    from oracle.oracle import OracleSim
    torch = torch_mod
    n, T = 12, 150
    clip = oracle_clips["walk"]
    eng = HipEngine(model, n, auto_reset=False, stale_contact_slots=1)
    eng.load_clip(0, clips["walk"])
    out = eng.alloc_outputs()
    idx = (np.arange(n) * 5) % clip.L
    eng.reset(out["obs"], idx_init=torch.tensor(idx, dtype=torch.int32, device=eng.device))
    sims = []
    for i in range(n):
        s = OracleSim(model)
        s.set_caps(32, 128)
        s.set_flag("stale_contact_slots", 1)
        s.env_reset(clip, int(idx[i]))
        sims.append(s)
    rng = np.random.default_rng(8)
    differs = compared = 0
    floor, rfoot, lfoot = model.floor_geom, model.rfoot_geom, model.lfoot_geom
    for t in range(T):
        st = [np.array([s.get(k) for s in sims]) for k in ("qpos", "qvel", "qacc_warmstart", "ctrl")]
        eng.set_state(*[torch.tensor(x, dtype=torch.float32, device=eng.device) for x in st], run_forward=False)
        act = rng.uniform(-0.6, 0.6, (n, 28)).astype(np.float32)
        eng.step(torch.tensor(act, device=eng.device), out)
        torch.cuda.synchronize()
        obs, done = out["obs"].cpu().numpy(), out["done"].cpu().numpy()
        redo = np.zeros(n, np.uint8)
        for i, s in enumerate(sims):
            o, r, d, terms, reason = s.env_step(clip, act[i].astype(np.float64))
            assert tuple(obs[i, 64:66]) == tuple(o[64:66]), (t, i, obs[i, 64:66], o[64:66])
            assert np.abs(obs[i] - o).max() < 1e-4 and bool(done[i]) == d
            con = s.get("contact")
            active = (float(any({int(c[13]), int(c[14])} == {floor, rfoot} for c in con)), float(any({int(c[13]), int(c[14])} == {floor, lfoot} for c in con)))
            differs += active != tuple(o[64:66])
            compared += 1
            if d:
                redo[i] = 1
                s.env_reset(clip, int((idx[i] + t) % clip.L))
        if redo.any():
            eng.reset(out["obs"], mask=torch.tensor(redo, device=eng.device),
                      idx_init=torch.tensor((idx + t) % clip.L, dtype=torch.int32, device=eng.device))
    print("F8: %d env-steps compared, stale scan differs from the active-contact scan on %d" % (compared, differs))
    assert compared == n * T and differs >= 10
    eng.close()
