"""Where a teacher-forced G1 outlier comes from: the RK stage states of the step are rebuilt on the oracle (python RK4 over
single evaluations) and every stage is evaluated on both sides from the SAME fp32 state: contact lists, normals, qacc.
python scripts/g1_stage_debug.py [n] [steps] [max cases]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.g1 import G1HipEngine
from oracle import oracle_g1 as og
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
maxcases = int(sys.argv[3]) if len(sys.argv) > 3 else 6
g, cm = og.g1_model()
H = float(cm.timestep)
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
clip = og.G1Clip(*mc.tables())
eng = G1HipEngine(n, auto_reset=False); eng.load_clip(mc); out = eng.alloc_outputs()
eng2 = G1HipEngine(1, auto_reset=False); eng2.load_clip(mc); out2 = eng2.alloc_outputs(); dbg2 = eng2.enable_debug()
idx = (torch.arange(n, dtype=torch.int32, device=eng.device) * 2) % 70
eng.reset(out["obs"], idx_init=idx)
sims = [og.G1Sim() for _ in range(n)]
for i, s in enumerate(sims):
    s.set_caps(48, 256); s.env_reset(clip, int(idx[i]))


def qmul(a, b):
    return np.array([a[0]*b[0]-a[1]*b[1]-a[2]*b[2]-a[3]*b[3], a[0]*b[1]+a[1]*b[0]+a[2]*b[3]-a[3]*b[2],
                     a[0]*b[2]-a[1]*b[3]+a[2]*b[0]+a[3]*b[1], a[0]*b[3]+a[1]*b[2]-a[2]*b[1]+a[3]*b[0]])


def integ(q0, vel, h):
    q = q0.copy()
    q[:3] += h * vel[:3]
    w = vel[3:6]; nw = np.linalg.norm(w)
    ax = w / nw if nw > 1e-15 else np.array([1.0, 0, 0]); ang = h * nw if nw > 1e-15 else 0.0
    qr = np.concatenate([[np.cos(ang / 2)], ax * np.sin(ang / 2)])
    qo = q0[3:7] / np.linalg.norm(q0[3:7])
    qn = qmul(qo, qr); q[3:7] = qn / np.linalg.norm(qn)
    q[7:] += h * vel[6:]
    return q


def both_eval(q, v, w, ctrl):
    """one evaluation on both sides from the same fp32 state"""
    q32, v32, w32 = [np.asarray(x, np.float32) for x in (q, v, w)]
    eng2.set_state(torch.tensor(q32[None], device=eng2.device), torch.tensor(v32[None], device=eng2.device), torch.tensor(w32[None], device=eng2.device))
    torch.cuda.synchronize()
    d = dbg2.cpu().numpy()[0].copy()
    s = og.G1Sim(); s.set_caps(48, 256)
    s.set("ctrl", ctrl); s.set("qacc_warmstart", w32.astype(np.float64))
    s.set_state(q32.astype(np.float64), v32.astype(np.float64))
    return d, s


rng = np.random.default_rng(1)
alive = np.ones(n, bool); cases = 0
for t in range(steps):
    q, v, w = [x.cpu().numpy().astype(np.float64) for x in eng.get_state()]
    act = rng.uniform(-1, 1, (n, 23)).astype(np.float32)
    eng.step(torch.tensor(act, device=eng.device), out); torch.cuda.synchronize()
    q2, v2, _ = [x.cpu().numpy() for x in eng.get_state()]
    for i, s in enumerate(sims):
        if not alive[i]: continue
        s.set("qpos", q[i]); s.set("qvel", v[i]); s.set("qacc_warmstart", w[i])
        o, r, dn, terms, reason = s.env_step(clip, act[i].astype(np.float64))
        eq = np.abs(q2[i] - s.get("qpos")).max()
        if dn: alive[i] = False
        if eq > 1e-4 and cases < maxcases:
            cases += 1
            print("==== t %d env %d: qpos err %.2e" % (t, i, eq))
            ctrl = s.get("ctrl")
            # engine 2: take the pre-step state, apply the action once (stores ctrl), then evaluate the stage states
            eng2.set_state(torch.tensor(q[i][None], dtype=torch.float32, device=eng2.device), torch.tensor(v[i][None], dtype=torch.float32, device=eng2.device),
                           torch.tensor(w[i][None], dtype=torch.float32, device=eng2.device), run_forward=False)
            eng2.step(torch.tensor(act[i][None], device=eng2.device), out2); torch.cuda.synchronize()
            X0q, X0v = q[i].copy(), v[i].copy()
            A = [0.5, 0.5, 1.0]
            qs, vs, ws = X0q, X0v, w[i].copy()
            for st in range(4):
                d, so = both_eval(qs, vs, ws, ctrl)
                cons = so.contacts(); nc = int(d[203])
                gc = d[208:208 + 9 * nc].reshape(-1, 9)
                same = nc == len(cons) and all(int(a[1]) == c["geom1"] and int(a[2]) == c["geom2"] for a, c in zip(gc, cons))
                qa = so.get("qacc")
                print("  stage %d: ncon %d/%d list-equal %s nefc %d/%d iter %d/%d qacc err %.2e (max %.1f) qas err %.2e" % (
                    st, nc, len(cons), same, int(d[204]), so.geti("nefc"), int(d[205]), so.geti("solver_iter"),
                    np.abs(d[160:203] - qa).max(), np.abs(qa).max(), np.abs(d[117:160] - so.get("qacc_smooth")).max()))
                if same:
                    for a, c in zip(gc, cons):
                        ne, pe, de = np.abs(a[6:9] - c["frame"][0]).max(), np.abs(a[3:6] - c["pos"]).max(), abs(a[0] - c["dist"])
                        if ne > 1e-4 or pe > 1e-4 or de > 1e-5:
                            print("      g %d-%d (types %d/%d) dist %.6f / %.6f  pos err %.2e  normal err %.2e   n_gpu %s n_orc %s" % (
                                c["geom1"], c["geom2"], g.geom_type[c["geom1"]], g.geom_type[c["geom2"]], a[0], c["dist"], pe, ne, np.round(a[6:9], 4), np.round(c["frame"][0], 4)))
                else:
                    print("      gpu", [(int(a[1]), int(a[2]), round(float(a[0]), 6)) for a in gc])
                    print("      orc", [(c["geom1"], c["geom2"], round(c["dist"], 6)) for c in cons])
                f = so.get("efc_force"); fg = d[640:640 + len(f)]
                if len(f): print("      force err max %.3e (max force %.2f)" % (np.abs(f - fg).max(), np.abs(f).max()))
                if st < 3:   # next stage state from the ORACLE's evaluation
                    qs = integ(X0q, A[st] * so.get("qvel"), H)
                    vs = X0v + H * A[st] * qa
                    ws = qa.copy()
print("cases", cases)
