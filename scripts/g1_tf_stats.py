"""Teacher-forced G1 stepping against the fp64 oracle, outlier statistics: python scripts/g1_tf_stats.py [lib file] [n] [steps] [scale]
Prints the distribution of the per-step qpos error and, for every step beyond 1e-4, what differed."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import deepmimic_mujoco_amd._lib as _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = _lib.LIB_PATH.replace("libdeepmimic_hip.so", sys.argv[1])
from deepmimic_mujoco_amd.config import MotionConfig
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.g1 import G1HipEngine
from oracle import oracle_g1 as og
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 80
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
mc = MocapDM(robot="unitree_g1"); mc.load_mocap(MotionConfig("walk", robot="unitree_g1").mocap_path)
clip = og.G1Clip(*mc.tables())
eng = G1HipEngine(n, auto_reset=False); eng.load_clip(mc); out = eng.alloc_outputs()
dbg = eng.enable_debug()
idx = (torch.arange(n, dtype=torch.int32, device=eng.device) * 2) % 70
eng.reset(out["obs"], idx_init=idx)
sims = [og.G1Sim() for _ in range(n)]
for i, s in enumerate(sims):
    s.set_caps(48, 256); s.env_reset(clip, int(idx[i]))
rng = np.random.default_rng(1)
alive = np.ones(n, bool); recs = []
for t in range(steps):
    q, v, w = [x.cpu().numpy().astype(np.float64) for x in eng.get_state()]
    act = (rng.uniform(-1, 1, (n, 23)) * scale).astype(np.float32)
    eng.step(torch.tensor(act, device=eng.device), out); torch.cuda.synchronize()
    q2, v2, _ = [x.cpu().numpy() for x in eng.get_state()]
    d = dbg.cpu().numpy(); done = out["done"].cpu().numpy()
    for i, s in enumerate(sims):
        if not alive[i]: continue
        s.set("qpos", q[i]); s.set("qvel", v[i]); s.set("qacc_warmstart", w[i])
        o, r, dn, terms, reason = s.env_step(clip, act[i].astype(np.float64))
        eq = np.abs(q2[i] - s.get("qpos")).max(); ev = np.abs(v2[i] - s.get("qvel")).max()
        g_nc = [int(x) for x in d[i][1000:1004]]; o_nc = [s.geti("stage_ncon%d" % k) for k in range(4)]
        g_ne = [int(x) for x in d[i][1004:1008]]; o_ne = [s.geti("stage_nefc%d" % k) & 0xFF for k in range(4)]
        recs.append((eq, ev, t, i, g_nc == o_nc and g_ne == o_ne, bool(done[i]) == bool(dn), int(d[i][1011])))
        if dn or bool(done[i]): alive[i] = False
e = np.array([r[0] for r in recs]); ev = np.array([r[1] for r in recs])
print("lib", _lib.LIB_PATH.split("/")[-1], "env-steps", len(e), "| qpos err median %.2e p90 %.2e p99 %.2e max %.2e" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
for thr in (1e-6, 1e-5, 1e-4, 1e-3):
    print("   > %.0e: %d (%.2f %%)" % (thr, int((e > thr).sum()), 100 * (e > thr).mean()))
print("   qvel err median %.2e p99 %.2e max %.2e" % (np.median(ev), np.percentile(ev, 99), ev.max()))
print("   steps with differing per-stage contact / row counts: %d; differing done: %d" % (sum(not r[4] for r in recs), sum(not r[5] for r in recs)))
for r in sorted(recs, key=lambda r: -r[0])[:12]:
    print("   qpos %.2e qvel %.2e t %d env %d counts-equal %s done-equal %s mpr pairs (last stage) %d" % r)
