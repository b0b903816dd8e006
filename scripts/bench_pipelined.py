"""Throughput of 4096 envs stepped as K independent sub-batches on K streams (no barrier between sub-batches):
the ramp-down of one sub-batch's launch overlaps the next launch of another.  Compare with bench.py (K = 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deepmimic_mujoco_amd._lib as L
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
NT = 4096
for K in (1, 2, 4):
    N = NT // K
    engs, outs, acts, streams = [], [], [], []
    for k in range(K):
        e = L.HipEngine(model, N, seed=1234 + k); e.load_clip(0, mc)
        o = e.alloc_outputs(); a = torch.zeros(N, 28, device=e.device)
        e.reset(o["obs"], idx_init=((torch.arange(N, device=e.device) + k * N) % 76).to(torch.int32))
        engs.append(e); outs.append(o); acts.append(a); streams.append(torch.cuda.Stream())
    def run(nsteps, base):
        for i in range(nsteps):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    engs[k].fill_random_actions(acts[k], base + i); engs[k].step(acts[k], outs[k])
    run(50, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200, 50); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("K=%d sub-batches of %d envs: %.0f env-steps/s, %.3f ms per %d-env step" % (K, N, 200 * NT / dt, dt / 200 * 1e3, NT))
    for e in engs: e.close()
