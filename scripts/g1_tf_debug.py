import numpy as np, torch, sys
sys.path.insert(0,"/root/repo")
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.g1 import G1HipEngine
from oracle import oracle_g1 as og
n, steps = 16, 60
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
clip = og.G1Clip(*mc.tables())
eng = G1HipEngine(n, auto_reset=False); eng.load_clip(mc); out = eng.alloc_outputs()
dbg = eng.enable_debug()
idx = torch.arange(n, dtype=torch.int32, device=eng.device) * 4
eng.reset(out["obs"], idx_init=idx)
sims = [og.G1Sim() for _ in range(n)]
for i, s in enumerate(sims):
    s.set_caps(48, 256); s.env_reset(clip, int(idx[i]))
rng = np.random.default_rng(1)
alive = np.ones(n, bool); recs=[]; PRE={}
for t in range(steps):
    q, v, w = [x.cpu().numpy().astype(np.float64) for x in eng.get_state()]
    act = rng.uniform(-1, 1, (n, 23)).astype(np.float32)
    eng.step(torch.tensor(act, device=eng.device), out); torch.cuda.synchronize()
    q2, v2, _ = [x.cpu().numpy() for x in eng.get_state()]
    d = dbg.cpu().numpy(); done = out["done"].cpu().numpy()
    for i, s in enumerate(sims):
        if not alive[i]: continue
        PRE[(t, i)] = (q[i].copy(), v[i].copy(), w[i].copy())
        s.set("qpos", q[i]); s.set("qvel", v[i]); s.set("qacc_warmstart", w[i])
        o, r, dn, terms, reason = s.env_step(clip, act[i].astype(np.float64))
        eq = np.abs(q2[i] - s.get("qpos")).max(); ev = np.abs(v2[i]-s.get("qvel")).max()
        recs.append((eq, ev, t, i, int(d[i][203]), s.geti("stage_ncon3"), int(d[i][204]), s.geti("stage_nefc3"), int(d[i][205]), s.geti("solver_iter"), [s.geti("stage_ncon%d"%k) for k in range(4)], np.abs(s.get("qvel")).max(), [int(x) for x in d[i][1000:1004]], [int(x) for x in d[i][1004:1008]], [s.geti("stage_nefc%d"%k)&0xFF for k in range(4)]))
        if dn: alive[i] = False
recs.sort(key=lambda r:-r[0])
for r in recs[:14]: print("qpos err %.2e qvel err %.2e t %d env %d | last-stage ncon gpu %d oracle %d | nefc %d %d | iter %d %d | oracle stage ncon %s | max|qvel| %.1f | gpu stage ncon %s nefc %s oracle nefc %s"%r)
e=np.array([r[0] for r in recs]); print("n", len(e), "median", np.median(e), "p90", np.percentile(e,90), "p99", np.percentile(e,99))

# ---- replay the worst pre-step states through a single forward evaluation on both sides
print("---- stage-0 comparison at the worst pre-step states")
eng2 = G1HipEngine(1, auto_reset=False); eng2.load_clip(mc); dbg2 = eng2.enable_debug()
g, _ = og.g1_model()
for r in recs[:6]:
    t, i = r[2], r[3]
    q, v, w = PRE[(t, i)]
    eng2.set_state(torch.tensor(q[None], dtype=torch.float32, device=eng2.device), torch.tensor(v[None], dtype=torch.float32, device=eng2.device),
                   torch.tensor(w[None], dtype=torch.float32, device=eng2.device))
    torch.cuda.synchronize()
    d = dbg2.cpu().numpy()[0]
    s = og.G1Sim(); s.set_caps(48, 256); s.set("qacc_warmstart", w); s.set_state(q, v)
    cons = s.contacts(); nc = int(d[203])
    print("t %d env %d: ncon %d/%d nefc %d/%d iter %d/%d  qacc err %.3e (max %.1f)  qas err %.3e" % (t, i, nc, len(cons), int(d[204]), s.geti("nefc"), int(d[205]), s.geti("solver_iter"),
          np.abs(d[160:203] - s.get("qacc")).max(), np.abs(s.get("qacc")).max(), np.abs(d[117:160] - s.get("qacc_smooth")).max()))
    gc = d[208:208 + 9 * nc].reshape(-1, 9)
    for a, c in zip(gc, cons):
        print("    g %d-%d (%s/%s) dist %.6f / %.6f  pos err %.2e  normal err %.2e" % (c["geom1"], c["geom2"], g.geom_type[c["geom1"]], g.geom_type[c["geom2"]], a[0], c["dist"],
              np.abs(a[3:6] - c["pos"]).max(), np.abs(a[6:9] - c["frame"][0]).max()))
    f = s.get("efc_force"); fg = d[640:640 + len(f)]
    print("    force err max %.3e (max force %.2f) at row %d" % (np.abs(f - fg).max(), np.abs(f).max(), int(np.argmax(np.abs(f - fg)))))
