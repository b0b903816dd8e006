"""How well can the cost of an env-step be predicted for the longest-first launch order?  Correlation of this step's measured
wave ticks with (a) the kernel's work estimate of the previous step (what dm_schedule_kernel sorts by), (b) the previous step's
measured ticks, (c) this step's own work estimate; list-scheduling makespans for the orders each would give.
*prof* build (-DDM_PROFILE=2, libdeepmimic_hip_prof.so).  usage: sched_predict.py [slots]"""
import os, sys, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deepmimic_mujoco_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libdeepmimic_hip_prof.so")
from deepmimic_mujoco_amd.model import load_model
from deepmimic_mujoco_amd.mocap import MocapDM
from deepmimic_mujoco_amd.config import MotionConfig
model = load_model(); mc = MocapDM(model=model); mc.load_mocap(MotionConfig("walk").mocap_path)
N = 4096
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
eng = L.HipEngine(model, N); eng.load_clip(0, mc)
out = eng.alloc_outputs(); act = torch.zeros(N, 28, device=eng.device)
eng.reset(out["obs"], idx_init=(torch.arange(N, device=eng.device) % 76).to(torch.int32))
dbg = eng.enable_debug(); eng.enable_timing(True)


def makespan(order, t):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for e in order:
        s = heapq.heappop(h)
        f = s + t[e]
        end = max(end, f)
        heapq.heappush(h, f)
    return end


prev_work = prev_t = prev_done = w_prev2 = prev_last = None
for i in range(140):
    eng.fill_random_actions(act, i)
    torch.cuda.synchronize()
    w_before = eng.get_work().cpu().numpy().copy()
    eng.step(act, out)
    torch.cuda.synchronize()
    t = dbg[:, 352:368].sum(1).cpu().numpy()
    w_after = eng.get_work().cpu().numpy().copy()
    done = out["done"].cpu().numpy().astype(bool)
    if i >= 100 and i % 8 == 0 and prev_t is not None:
        c = lambda a, b: np.corrcoef(a, b)[0, 1]
        print("step %d: corr(ticks, prev estimate) %.2f  corr(ticks, prev ticks) %.2f  corr(ticks, own estimate) %.2f  (not-done envs only: %.2f / %.2f / %.2f)  done %.3f"
              % (i, c(t, w_before), c(t, prev_t), c(t, w_after), c(t[~done], w_before[~done]), c(t[~done], prev_t[~done]), c(t[~done], w_after[~done]), done.mean()))
        print("   makespan on %d slots: prev-estimate order %.0f | prev-ticks order %.0f | own-estimate order %.0f | perfect %.0f | sum/slots %.0f max %.0f | mean ticks done %.0f not done %.0f"
              % (slots, makespan(np.argsort(-w_before, kind="stable"), t), makespan(np.argsort(-prev_t, kind="stable"), t),
                 makespan(np.argsort(-w_after, kind="stable"), t), makespan(np.argsort(-t), t), t.sum() / slots, t.max(), t[done].mean() if done.any() else 0, t[~done].mean()))
    if i >= 100 and i % 8 == 0:
        st = dbg[:, 368:372].cpu().numpy()                 # ticks from entry to the end of RK stage 0..3 (stage 3 incl. nothing of the task layer)
        s0, rest = st[:, 0], t - st[:, 0]
        c = lambda a, b: np.corrcoef(a, b)[0, 1]
        # two launches: A = stage 0 in the previous-estimate order, B = the rest in the order of the measured stage-0 ticks
        mA = makespan(np.argsort(-w_before, kind="stable"), s0)
        mB = makespan(np.argsort(-s0, kind="stable"), rest)
        mBp = makespan(np.argsort(-rest), rest)
        print("   stage 0: mean %.0f max %.0f corr(stage 0, rest) %.2f | two launches: A %.0f + B(order by stage 0) %.0f = %.0f  (B perfect %.0f; sum rest / slots %.0f, max rest %.0f)"
              % (s0.mean(), s0.max(), c(s0, rest), mA, mB, mA + mB, mBp, rest.sum() / slots, rest.max()))
    if i >= 100 and i % 8 == 0 and prev_done is not None:
        # what if the evaluation at the reset state moved to the START of the env's next step (known before the launch)?
        extra = t[done].mean() - t[~done].mean()
        top = np.argsort(-t)[:100]
        t2 = t - extra * done + extra * prev_done
        est2 = w_before + (w_before.mean() * extra / t[~done].mean()) * prev_done
        print("   reset evaluation ~%.0f ticks; done among the 100 heaviest: %d; deferred-reset what-if: estimate order %.0f perfect %.0f max %.0f"
              % (extra, int(done[top].sum()), makespan(np.argsort(-est2, kind="stable"), t2), makespan(np.argsort(-t2), t2), t2.max()))
    if i >= 100 and i % 8 == 0 and prev_done is not None and prev_t is not None:
        # measured ticks of the previous step as the key, with the envs that were reset in it (their ticks include the evaluation
        # at the reset state and belong to an episode that is over) keyed by the batch median instead
        key = prev_t.copy()
        key[prev_done] = np.median(prev_t[~prev_done])
        print("   previous ticks with reset envs at the median: corr %.2f makespan %.0f" % (np.corrcoef(key, t)[0, 1], makespan(np.argsort(-key, kind="stable"), t)))
        # mixture: average of the ranks under the two predictors
        ra, rb = np.argsort(np.argsort(-key)), np.argsort(np.argsort(-w_before))
        print("   rank average of that key and the kernel's estimate: makespan %.0f" % makespan(np.argsort(ra + rb, kind="stable"), t))
    if i >= 100 and i % 8 == 0 and w_prev2 is not None:
        # the order computed one step earlier (from the estimate of step t - 2): the schedule kernel could then run beside step t - 1
        print("   estimate of two steps ago: corr %.2f makespan %.0f" % (np.corrcoef(w_prev2, t)[0, 1], makespan(np.argsort(-w_prev2, kind="stable"), t)))
    st_all = dbg[:, 368:372].cpu().numpy()
    if i >= 100 and i % 8 == 0 and prev_last is not None:
        # fresher keys: the LAST RK stage of the previous step alone (ticks), and its mix with the kernel's estimate
        print("   previous step's last-stage ticks: corr %.2f makespan %.0f" % (np.corrcoef(prev_last, t)[0, 1], makespan(np.argsort(-prev_last, kind="stable"), t)))
    prev_last = st_all[:, 3] - st_all[:, 2]
    prev_t = t
    prev_done = done
    w_prev2 = w_before
