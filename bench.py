#!/usr/bin/env python3
"""bench.py — env-steps/s of the HIP DPEnv.step() path (BASELINE.json configs[1]).

Workload ("cfg2_random_torque"): 4096 humanoid3d envs per GPU on the `walk` clip, each reset to
frame env%L, actions ~ U(-2,2)^28 from the counter-based generator shared with the oracle,
full step() = RK4 mj_step-equivalent + obs + imitation reward + termination + auto-reset.
One "step" = one dm_step over the whole batch; inputs/outputs stay resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the contract in the task statement) with `roofline`
(HBM-bound accounting of the step kernel, measured live with HIP events on the launch stream)
and `cpu_baseline` (the fp64 oracle timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 1260      # 315 fp32 words of state in/out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
EVENT_STRIDE = 4                    # every 4th launch of the timed region carries a HIP event pair


def _newest_pmc():
    """Counter means of dm_step_kernel from the newest committed PMC profile (same command, 4096 envs)."""
    import csv
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dm_step_kernel.csv"))
    path = max(files, key=lambda f: tuple(int(x) for x in re.findall(r"\d+", os.path.basename(f))))   # newest round / version
    vals = {r["counter"]: float(r["mean_per_dispatch_over_last_10_dispatches"]) for r in csv.DictReader(open(path))}
    return vals, os.path.basename(path)


def _newest_g1_pmc():
    """Counters of the split G1 pipeline from the newest committed PMC profile (scripts/pmc_g1_split.sh, 4 096 envs): one step
    is six launches of g1_env_kernel and five of g1_pair_kernel; the files hold means per launch.  Returns
    ({"g1_env_kernel": {...}, "g1_pair_kernel": {...}}, [file names]) or (None, [])."""
    import csv
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_g1*_pmc_g1_env_kernel.csv"))
    if not files:
        return None, []
    envf = max(files, key=lambda f: tuple(int(x) for x in re.findall(r"\d+", os.path.basename(f))))
    pairf = envf.replace("g1_env_kernel", "g1_pair_kernel")
    out = {}
    for k, f in (("g1_env_kernel", envf), ("g1_pair_kernel", pairf)):
        rows = list(csv.DictReader(open(f)))
        col = [c for c in rows[0] if c != "counter"][0]
        out[k] = {r["counter"]: float(r[col]) for r in rows}
    return out, [os.path.basename(envf), os.path.basename(pairf)]


G1_LAUNCHES = {"g1_env_kernel": 6, "g1_pair_kernel": 5}     # per step of the split pipeline (dm_g1.hip: four RK stages + reset evaluation)


def measured_traffic_bytes():
    """HBM bytes per launch of dm_step_kernel from the committed PMC profile."""
    try:
        vals, _ = _newest_pmc()
        return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def valu_view(kernel_ms, n_envs):
    """What actually bounds dm_step_kernel (the HBM roofline is nominal): vector-instruction issue.
    Counters from the newest committed PMC file (one wave = one env-step; SQ_* cycle counters are in quad-cycles);
    issue cost 2 cycles per wave64 VALU instruction on a SIMD-32 (MI355X_MICROARCH.md), 1024 SIMDs, two or three resident
    waves per SIMD (kernel variant); the launch duration is the one measured live in THIS run."""
    try:
        v, src = _newest_pmc()
        waves = v["SQ_WAVES"]
        insts = v["SQ_INSTS_VALU"] / waves
        wave_cycles = 4.0 * v["SQ_WAVE_CYCLES"] / waves
        clock_ghz = 2.4
        simds, resident = 1024, (3 if n_envs >= 3072 else 2)     # dm_step launches the three-wave build from 3 072 envs up
        rounds = n_envs / float(simds * resident)
        steady_cycles = rounds * wave_cycles                       # launch length if every SIMD always held two waves
        launch_cycles = kernel_ms * 1e-3 * clock_ghz * 1e9
        return {"insts_per_env_step": insts, "salu_per_env_step": v["SQ_INSTS_SALU"] / waves, "lds_per_env_step": v["SQ_INSTS_LDS"] / waves,
                "vmem_per_env_step": v["SQ_INSTS_VMEM"] / waves, "wave_cycles_per_env_step": wave_cycles,
                "issue_frac_steady": resident * insts * 2.0 / wave_cycles,
                "issue_frac_launch": n_envs * insts * 2.0 / (simds * launch_cycles),
                "wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], "wait_inst_frac": v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"],
                "rampdown_frac": max(0.0, 1.0 - steady_cycles / launch_cycles),
                "vector_tflops_frac": None, "clock_ghz_assumed": clock_ghz, "counters_from": "profiles/" + src}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def _mlp_flops_per_step(B, D, H1, H2, A):
    """Algorithmic FLOPs of one PPO minibatch gradient (policy trunk D-H1-H2-A, value trunk D-H1-H2-1): forward,
    input gradients (no gradient into the observations), weight gradients."""
    def trunk(a):
        l1, l2, l3 = D * H1, H1 * H2, H2 * a
        return 2.0 * B * ((l1 + l2 + l3) + (l2 + l3) + (l1 + l2 + l3))
    return trunk(A) + trunk(1)


def g1_record(local_rank, n=4096, steps=20, warmup=5, with_cpu=False):
    """Auxiliary record: the second robot of the reference, DPEnv(robot="unitree_g1") — the robot its published figure
    (~1 390 env-steps/s, src/plot_profiling.py:486) was measured on.  Full step() with auto-reset, random actions in [-1, 1]
    (x 20 torque scale inside), walk clip, dmg1_step through the C-ABI; kernel time from HIP events inside the library."""
    import torch
    from deepmimic_mujoco_amd.g1 import G1HipEngine, NACT
    from deepmimic_mujoco_amd.config import MotionConfig as MC
    from deepmimic_mujoco_amd.mocap import MocapDM as MD
    mc = MD(robot="unitree_g1")
    mc.load_mocap(MC("walk", robot="unitree_g1").mocap_path)
    eng = G1HipEngine(n, device=local_rank, auto_reset=True, seed=3)
    eng.load_clip(mc)
    out = eng.alloc_outputs()
    eng.reset(out["obs"])
    g = torch.Generator(device=eng.device).manual_seed(0)
    acts = [torch.rand(n, NACT, device=eng.device, generator=g) * 2 - 1 for _ in range(8)]
    for t in range(warmup):
        eng.step(acts[t % 8], out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        eng.step(acts[t % 8], out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms_list = []
    for t in range(10):           # kernel time: ten more steps, each read back (a host sync per step, outside the timed window)
        eng.step(acts[t % 8], out)
        torch.cuda.synchronize()
        kms_list.append(eng.last_kernel_ms())
    kms = float(np.mean(kms_list))
    G1_ALGO_BYTES = 4 * (44 + 43 + 43 + 23 + 4 + 44 + 43 + 43 + 85 + 1 + 5 + 2 + 2)     # state row in / out, action, outputs: 1 528 B
    rec = {"robot": "unitree_g1 (43 DoF, 32 convex meshes, friction loss)", "envs_per_gpu": n, "steps": steps,
           "window": "steps %d..%d after reset (throughput), %d..%d (kernel time)" % (warmup, warmup + steps - 1, warmup + steps, warmup + steps + 9),
           "env_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3,
           "kernel": "6 x g1_env_kernel + 5 x g1_pair_kernel (split pipeline)" if eng.split else "g1_step_kernel",
           "kernel_ms": kms, "done_fraction_last_step": float(out["done"].float().mean()),
           "reference_published_env_steps_per_s": 1390, "note": "auxiliary: SURVEY 8f-2 (next row), not the headline metric"}
    pmc, pmc_src = _newest_g1_pmc()
    traffic = None
    try:
        traffic = sum(G1_LAUNCHES[k] * (pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024.0 for k in G1_LAUNCHES)
    except Exception:
        pass
    ach = n * G1_ALGO_BYTES / (kms * 1e-3) / 1e9
    rec["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                       "traffic": traffic, "kernel": rec["kernel"], "kernel_ms": kms, "kernel_launches_timed": len(kms_list),
                       "algorithmic_bytes_per_env_step": G1_ALGO_BYTES,
                       "note": "nominal bound like the humanoid's: both kernels are latency / issue bound (valu below); kernel_ms = HIP events "
                               "around the eleven launches of one step; traffic = 6 x g1_env_kernel + 5 x g1_pair_kernel launches, from "
                               "profiles/%s" % " + ".join(pmc_src or ["-"])}
    try:   # what bounds the two kernels: instruction issue / latency, from the committed PMC files of scripts/bench_g1.py at 4 096 envs
        view = {}
        for k, v in pmc.items():
            w = v["SQ_WAVES"]
            insts = v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"] + v["SQ_INSTS_VMEM"]
            view[k] = {"launches_per_step": G1_LAUNCHES[k], "waves_per_launch": w, "valu_per_launch": v["SQ_INSTS_VALU"],
                       "salu_per_launch": v["SQ_INSTS_SALU"], "lds_per_launch": v["SQ_INSTS_LDS"], "vmem_per_launch": v["SQ_INSTS_VMEM"],
                       "wave_cycles_per_wave": 4.0 * v["SQ_WAVE_CYCLES"] / w, "cycles_per_instruction": 4.0 * v["SQ_WAVE_CYCLES"] / insts,
                       "wait_frac": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
                       "hbm_bytes_per_launch": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0}
        tot_valu = sum(G1_LAUNCHES[k] * pmc[k]["SQ_INSTS_VALU"] for k in G1_LAUNCHES)
        view["valu_per_env_step"] = tot_valu / n
        view["issue_frac_step"] = tot_valu * 2.0 / (1024 * kms * 1e-3 * 2.4e9)
        view["hbm_bytes_per_env_step"] = None if traffic is None else traffic / n
        view["algorithmic_bytes_per_env_step"] = G1_ALGO_BYTES
        view["counters_from"] = ["profiles/" + x for x in pmc_src]
        rec["valu"] = view
    except Exception as e:  # noqa: BLE001
        rec["valu"] = {"error": repr(e)[:200]}
    eng.close()
    # the same at 16 384 envs: at 4 096 the launch lasts as long as its heaviest env (longest-first order, two rounds); with more
    # envs per GPU the mean cost is what counts
    nb = 4 * n
    eng = G1HipEngine(nb, device=local_rank, auto_reset=True, seed=3)
    eng.load_clip(mc)
    outb = eng.alloc_outputs()
    eng.reset(outb["obs"])
    actb = [torch.rand(nb, NACT, device=eng.device, generator=g) * 2 - 1 for _ in range(4)]
    for t in range(warmup):
        eng.step(actb[t % 4], outb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps // 2):
        eng.step(actb[t % 4], outb)
    torch.cuda.synchronize()
    dtb = time.perf_counter() - t0
    rec["larger_batch"] = {"envs_per_gpu": nb, "env_steps_per_s": nb * (steps // 2) / dtb, "ms_per_step": dtb / (steps // 2) * 1e3,
                           "kernel_ms_last": eng.last_kernel_ms(), "window": "steps %d..%d after reset" % (warmup, warmup + steps // 2 - 1)}
    eng.close()
    del outb, actb
    # DPCombinedEnv() as src/sb3_ppo.py:277-278 trains it: walk / run / getup state machine on the G1, RSI auto-reset
    from deepmimic_mujoco_amd.g1 import HipG1CombinedVecEnv
    venv = HipG1CombinedVecEnv(n, device=local_rank, seed=3)
    venv.reset_tensor()
    for t in range(warmup):
        venv.step_tensor(acts[t % 8] * 0.25)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        o = venv.step_tensor(acts[t % 8] * 0.25)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec["dp_combined_env"] = {"window": "steps %d..%d after an RSI reset (the cost drifts upwards over hundreds of steps as robots fall)" % (warmup, warmup + steps - 1),
                              "env_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3,
                              "done_fraction_last_step": float(o["done"].float().mean()), "mean_reward": float(o["rew"].mean())}
    venv.close()
    # the same two envs with the batch split into two engines of n / 2 envs, each on its own HIP stream (VecEnv sub_batches=2;
    # one step still ends with both halves joined): one half's kernels fill the CUs the other half's heaviest envs leave idle
    from deepmimic_mujoco_amd.g1 import HipG1VecEnv
    rec["two_streams"] = {"sub_batches": 2, "window": "steps %d..%d after reset" % (warmup, warmup + steps - 1),
                          "note": "auxiliary: HipG1VecEnv(sub_batches=2).step_tensor, every step joins both sub-batches"}
    for key, make, sc in (("dp_env", lambda: HipG1VecEnv(n, motion="walk", device=local_rank, seed=3, sub_batches=2), 1.0),
                          ("dp_combined_env", lambda: HipG1CombinedVecEnv(n, device=local_rank, seed=3, sub_batches=2), 0.25)):
        venv = make()
        venv.reset_tensor()
        for t in range(warmup):
            venv.step_tensor(acts[t % 8] * sc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            venv.step_tensor(acts[t % 8] * sc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rec["two_streams"][key] = {"env_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3}
        venv.close()
    if with_cpu:   # the fp64 G1 oracle on 16 host threads, bounded sample of the same workload (checker code: CPU leg only)
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle_g1 as og
        L = og.lib()
        L.dmo_bench_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint64]
        L.dmo_bench_steps.restype = C.c_double
        _, cm = og.g1_model()
        clip = og.G1Clip(*mc.tables())
        limit, affinity, quota = cpu_share()
        t0 = time.perf_counter()
        L.dmo_bench_steps(C.byref(cm), C.byref(clip.c), 1, 20, 76)
        rate1 = 20 / (time.perf_counter() - t0)                  # single-thread rate, to size the sample

        def run(c, seconds):
            nenv, nst = 2, max(10, min(200, int(rate1 * seconds / 2)))
            with ThreadPoolExecutor(c) as ex:
                t0 = time.perf_counter()
                list(ex.map(lambda k: L.dmo_bench_steps(C.byref(cm), C.byref(clip.c), nenv, nst, 77 + k), range(c)))
                dtc = time.perf_counter() - t0
            return c * nenv * nst, dtc, "%d threads x %d envs x %d random-torque steps, walk clip, fp64 G1 oracle" % (c, nenv, nst)
        cores, rate, txt, calib = best_thread_count(run, limit, 9.0)
        rec["cpu_baseline"] = {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port", "affinity_cores": affinity,
                               "cgroup_cpu_quota": quota, "thread_calibration_env_steps_per_s": calib,
                               "single_thread_value": rate1, "sample": txt}
    return rec


def ppo_loop_record(args, dev, local_rank, rank, world, launched, barrier):
    """PPO loop of BASELINE configs 3 (one GPU: 4096 envs `walk`) / 4 (per-GPU share: 4096 envs `spinkick`): rollout incl.
    policy inference, GAE, 20 epochs x 32 minibatches of 4096 with ONE all-reduce of the flat gradient per optimizer step
    when several ranks run.  One untimed iteration (graph capture, warm-up), then `iters` timed ones."""
    import torch
    import torch.distributed as dist
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO, FusedMlpGrad
    motion = "spinkick" if world > 1 else "walk"
    rec = {"motion": motion, "envs_per_gpu": args.envs, "horizon": 32, "epochs": 20, "minibatch": 4096, "n_gpus": world,
           "timed_iterations": args.ppo_iters, "window": "PPO iterations 1..%d (iteration 0 untimed: graph capture), 32 env steps each, "
                                                         "fresh envs, untrained policy" % args.ppo_iters}
    for arch, mdt in (((256, 128), torch.float32), ((256, 128), torch.bfloat16), ((1024, 512), torch.float32), ((1024, 512), torch.bfloat16)):
        key = "%d,%d" % arch + ("" if mdt == torch.float32 else " bf16-gemm")     # bf16-gemm: mixed-precision learner (dm_ppo_wide_grad)
        env = ppo = err = None
        try:
            env = HipDeepMimicVecEnv(args.envs, motion=motion, device=local_rank, seed=1234 + 7919 * rank)
            ppo = PPO(env, net_arch=arch, n_steps=32, batch_size=4096, n_epochs=20, seed=0, mlp_dtype=mdt)
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:300]
        if launched and world > 1:   # every rank must enter this net's collectives, or none (a rank that failed to build would
            okt = torch.tensor([0 if err else 1], device=dev if args.dist_backend == "nccl" else "cpu")   # leave the others waiting)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if int(okt.item()) == 0 and err is None:
                err = "another rank failed to build this net"
        if err is not None:
            rec[key] = {"error": err}
            if env is not None:
                env.close()
            continue
        try:
            ppo.train(ppo.collect_rollouts())                    # untimed: captures the graphs
            torch.cuda.synchronize()
            barrier()
            t_roll = t_train = 0.0
            t0 = time.perf_counter()
            for _ in range(args.ppo_iters):
                a = time.perf_counter()
                buf = ppo.collect_rollouts()
                torch.cuda.synchronize()
                b = time.perf_counter()
                ppo.train(buf)
                torch.cuda.synchronize()
                t_roll += b - a
                t_train += time.perf_counter() - b
            barrier()
            dt = time.perf_counter() - t0
            coll_us = None
            if launched and world > 1:
                g = ppo.optimizer.flat_g if hasattr(ppo.optimizer, "flat_g") else ppo.grad_sync.flat
                for _ in range(5):
                    dist.all_reduce(g)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                for _ in range(50):
                    dist.all_reduce(g)
                torch.cuda.synchronize()
                coll_us = (time.perf_counter() - c0) / 50 * 1e6
                t = torch.tensor([dt, t_roll, t_train], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, t_roll, t_train = (float(x) for x in t)
            steps_per_iter = 32 * args.envs * world
            nopt = 20 * (32 * args.envs // 4096)
            opt_us = t_train / (args.ppo_iters * nopt) * 1e6
            flops = _mlp_flops_per_step(4096, 67, arch[0], arch[1], 28)
            rec[key] = {"loop_env_steps_per_s": args.ppo_iters * steps_per_iter / dt,
                        "rollout_env_steps_per_s": args.ppo_iters * steps_per_iter / t_roll,
                        "optimizer_step_us": opt_us, "optimizer_steps_per_iteration": nopt,
                        "mfma_frac": flops / (opt_us * 1e-6) / (157.3e12 if mdt == torch.float32 else 2.5e15),
                        "mfma_peak_tflops": 157.3 if mdt == torch.float32 else 2500.0, "flops_per_optimizer_step": flops,
                        "collective_us": coll_us, "collectives_per_iteration": nopt if world > 1 else 0,
                        "grad_floats": int(sum(p.numel() for p in ppo.policy.parameters())),
                        "mlp_dtype": "f32" if mdt == torch.float32 else "bf16 GEMMs / activations, f32 master weights, loss and Adam",
                        "learner_path": ("dist two-graph" if getattr(ppo, "_dg", None) is not None else
                                         "epoch graph" if getattr(ppo, "_eg", None) is not None else "eager"),
                        "gradient_kernels": ("dm_ppo_wide_grad (fused bf16 chain + split-K weight gradients)"
                                             if getattr(ppo, "_wide_ok", False) else
                                             "dm_ppo_mlp_grad (fused fp32 chain)" if (ppo.fused_mlp and FusedMlpGrad.supported(ppo.policy, ppo.batch_size)) else
                                             "library GEMMs"),
                        "mean_reward": ppo.stats.get("mean_reward")}
            env.close()
        except Exception as e:  # noqa: BLE001  (never lose the headline line to the auxiliary record)
            rec[key] = {"error": repr(e)[:300]}
    return rec


def cpu_share():
    """(thread limit, affinity size, cgroup quota) of this process: the affinity mask, capped by the cgroup CPU quota when one
    is set (the one-GPU box shows 256 cores in the mask and grants 16 of them: 256 threads there run at half the rate of 16)."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    return max(1, affinity if quota is None else min(affinity, int(quota + 0.999))), affinity, quota


def best_thread_count(run, limit, budget_s):
    """A container may grant fewer CPUs than its affinity mask shows without any quota file saying so: time the same bounded
    sample (`run(threads, seconds) -> (env-steps, wall seconds, sample text)`) at the mask size and at 64 / 16 threads and
    keep the best.  Returns (threads, rate, sample text, {threads: rate})."""
    cand = sorted({c for c in (16, 64, limit) if c <= limit} | {limit}, reverse=True)
    calib, best = {}, None
    for c in cand:
        n, dt, txt = run(c, budget_s / len(cand))
        calib[c] = n / dt
        if best is None or calib[c] > calib[best[0]]:
            best = (c, txt)
    return best[0], calib[best[0]], best[1], calib


def cpu_baseline(model, mocap, budget_s=12.0):
    """Oracle DPEnv.step() on the host cores: one thread per core of this process's CPU share (each thread owns
    its envs, as one SubprocVecEnv worker does), bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleClip, bench_steps
    clip = OracleClip(*mocap.tables())
    limit, affinity, quota = cpu_share()
    t0 = time.perf_counter()
    bench_steps(model, clip, 4, 250, 1234)                      # single-thread rate, to size the sample
    rate1 = 1000 / (time.perf_counter() - t0)
    def run(c, seconds):
        nenv = max(2, min(64, int(rate1 * seconds / 1000)))     # envs per thread, 1000 steps each
        with ThreadPoolExecutor(c) as ex:                       # ctypes releases the GIL; each call owns its DmoData
            t0 = time.perf_counter()
            list(ex.map(lambda k: bench_steps(model, clip, nenv, 1000, 1234 + k), range(c)))
            dt = time.perf_counter() - t0
        return c * nenv * 1000, dt, "%d threads x %d envs x 1000 random-torque steps, walk clip, fp64 oracle (oracle/dm_oracle.c)" % (c, nenv)
    cores, rate, txt, calib = best_thread_count(run, limit, budget_s)
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "affinity_cores": affinity, "cgroup_cpu_quota": quota, "thread_calibration_env_steps_per_s": calib,
            "single_thread_value": rate1, "sample": txt}


def _pci_bus_id(torch, idx):
    try:
        p = torch.cuda.get_device_properties(idx)
        return "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", -1), getattr(p, "pci_device_id", 0))
    except Exception:
        return "?"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (BASELINE: 4096)")
    ap.add_argument("--motion", default="walk")
    ap.add_argument("--actions", default="random", choices=["random", "zero"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the auxiliary two-sub-batch measurement (profiling runs)")
    ap.add_argument("--no-physics-only", action="store_true", help="skip the physics-only / full-step kernel comparison (profiling runs: "
                                                                   "its launches would be averaged into the step kernel's counters)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--integrator", default="model", choices=["model", "RK4", "Euler"],
                    help="model = the XML's (RK4, the headline); Euler = north_star's semi-implicit Euler option (auxiliary figure)")
    ap.add_argument("--no-ppo-loop", action="store_true", help="skip the auxiliary PPO-loop record (configs 3 / 4)")
    ap.add_argument("--no-g1", action="store_true", help="skip the auxiliary Unitree G1 record")
    ap.add_argument("--ppo-iters", type=int, default=2, help="timed PPO iterations per net in the ppo_loop record")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rendezvous only: launch / join the ranks, all-reduce the rank ids, print {n_gpus, ranks_seen}; no GPU needed")
    args = ap.parse_args()

    launched = "RANK" in os.environ
    if args.gpus > 1 and not launched:
        # `python bench.py --gpus N` (no launcher): become the launcher BEFORE anything touches the GPU — the N ranks are children of
        # torch.distributed.run, rank 0's JSON line goes to the inherited stdout, the exit code is the children's
        import socket
        import subprocess
        import torch
        if args.dist_backend == "nccl" and not args.dry_launch and torch.cuda.device_count() < args.gpus:   # device_count() does not initialise HIP
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; ranks are never folded onto one device"
                             % (args.gpus, torch.cuda.device_count()))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    import datetime
    tmo = datetime.timedelta(minutes=5)        # a rank that dies must not leave the others waiting forever
    if args.dry_launch:
        ranks = [0]
        if launched:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=tmo)
            t = torch.zeros(world, dtype=torch.int64)
            t[rank] = rank + 1
            dist.all_reduce(t)
            ranks = [int(x) - 1 for x in t]
            assert dist.get_world_size() == args.gpus
        if rank == 0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks_seen": ranks, "self_launched": launched}))
        if launched:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU.  Ranks are never folded onto one device under nccl; --dist-backend gloo + several ranks on one GPU is the
    # explicit rehearsal mode of the N > 1 path on a one-GPU box (the line then says devices_distinct: false)
    ndev = torch.cuda.device_count()
    if launched and args.dist_backend == "nccl" and ndev < world:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible; refusing to fold ranks onto one device" % (world, ndev))
    local_dev = local_rank % ndev
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), timeout=tmo)
        else:
            dist.init_process_group(args.dist_backend, timeout=tmo)
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    local_rank = local_dev
    # which device every rank really sits on: (rank, local device index, PCI bus id) gathered on all ranks
    ranks_seen = [[rank, local_dev, _pci_bus_id(torch, local_dev)]]
    if launched:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_seen[0])
        ranks_seen = gathered
    devices_distinct = len({(r[1], r[2]) for r in ranks_seen}) == world
    if launched and args.dist_backend == "nccl":
        assert devices_distinct, "two ranks share a GPU: %r" % (ranks_seen,)

    def barrier():
        if launched:
            dist.barrier()

    from deepmimic_mujoco_amd.model import load_model
    from deepmimic_mujoco_amd.mocap import MocapDM
    from deepmimic_mujoco_amd.config import MotionConfig
    from deepmimic_mujoco_amd._lib import HipEngine

    model = load_model()
    mocap = MocapDM(model=model)
    mocap.load_mocap(MotionConfig(args.motion).mocap_path)
    N = args.envs
    eng = HipEngine(model, N, device=local_rank, seed=1234 + rank, auto_reset=True, integrator=args.integrator)
    eng.load_clip(0, mocap)
    L = eng.clip_len[0]
    out = eng.alloc_outputs()
    actions = torch.zeros(N, 28, device=dev)
    eng.reset(out["obs"], idx_init=(torch.arange(N, device=dev) % L).to(torch.int32))

    def one_step(i):
        if args.actions == "random":
            eng.fill_random_actions(actions, i)
        eng.step(actions, out)

    for i in range(args.warmup):
        one_step(i)
    # inputs of the timed region resident in HBM before it starts (the bench contract): the K action batches of the same
    # counter-based generator (the oracle driver draws the same numbers), K x N x 28 floats; beyond 1 GiB they are generated
    # inside the loop as before (one 4.7 us launch per step)
    pre_actions = None
    if args.actions == "random" and args.steps * N * 28 * 4 <= (1 << 30):
        pre_actions = torch.empty(args.steps, N, 28, device=dev)
        for i in range(args.steps):
            eng.fill_random_actions(pre_actions[i], args.warmup + i)

    def timed_step(i):
        if pre_actions is not None:
            eng.step(pre_actions[i], out)
        else:
            one_step(args.warmup + i)
    # HIP event pair around every EVENT_STRIDE-th dm_step launch of the timed region, on the launch stream (a pair costs ~8 us
    # of stream time, 2.5 % of a step: sampling keeps the kernel-time measurement live without taxing the throughput)
    eng.enable_timing(True, stride=EVENT_STRIDE)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        timed_step(i)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kms, kcount = eng.mean_step_ms()           # read after the timed region: no host sync inside it
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    done_frac = float(out["done"].float().mean().item())

    # SURVEY 8(d) asks for physics-only throughput beside the full step(): for `nphys` consecutive steps the pre-step state is
    # saved, dm_physics_step (the mj_step-equivalent alone: same kernel, task layer skipped, same launch order) is timed on it,
    # the state is put back and the full dm_step is timed on the SAME state.  Kernel times from the library's HIP events.
    physics_only = None
    if args.actions == "random" and args.integrator != "Euler" and not args.no_physics_only:
        nphys, tp, tf = 24, 0.0, 0.0
        eng.enable_timing(True, stride=1)
        for i in range(nphys):
            q, v, w, c = eng.get_state()
            one_fill = args.warmup + args.steps + i
            eng.fill_random_actions(actions, one_fill)
            eng.physics_step(actions)
            torch.cuda.synchronize()
            tp += eng.last_step_ms()
            eng.set_state(q, v, warm=w, ctrl=c, run_forward=False)
            eng.step(actions, out)
            torch.cuda.synchronize()
            tf += eng.last_step_ms()
        physics_only = {"kernel_ms_physics_only": tp / nphys, "kernel_ms_full_step_same_states": tf / nphys,
                        "env_steps_per_s_physics_only": N / (tp / nphys * 1e-3), "env_steps_per_s_full_step": N / (tf / nphys * 1e-3),
                        "task_layer_frac": 1.0 - tp / tf, "window": "steps %d..%d of the run, one launch of each kind per step" %
                        (args.warmup + args.steps, args.warmup + args.steps + nphys - 1),
                        "note": "kernel time only (one launch per batch step, no host gap); the full step also resets finished envs "
                                "(a fifth forward evaluation for ~%.1f %% of them), which physics-only never does" % (100 * done_frac)}
        eng.enable_timing(False)

    # Auxiliary figure (never `value`): the same 4096 envs stepped as two independent 2048-env sub-batches on two
    # streams with no barrier between them, as a double-buffered rollout does (policy on one half while the other
    # half simulates): the ramp-down of one launch overlaps the next launch of the other half.
    pipelined = None
    if world == 1 and N % 2 == 0 and args.actions == "random" and not args.no_pipelined:
        from deepmimic_mujoco_amd.streams import concurrent_streams
        K, n2 = 2, N // 2
        subs, sts = [], concurrent_streams(dev, 2)
        for k in range(K):
            e2 = HipEngine(model, n2, device=local_rank, seed=1234 + 17 * (k + 1), auto_reset=True)
            e2.load_clip(0, mocap)
            o2 = e2.alloc_outputs()
            a2 = torch.zeros(n2, 28, device=dev)
            e2.reset(o2["obs"], idx_init=((torch.arange(n2, device=dev) + k * n2) % L).to(torch.int32))
            subs.append((e2, o2, a2, sts[k]))

        def run(nsteps, base):
            for i in range(nsteps):
                for e2, o2, a2, st in subs:
                    with torch.cuda.stream(st):
                        e2.fill_random_actions(a2, base + i)
                        e2.step(a2, o2)
        run(args.warmup, 0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(args.steps, args.warmup)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        pipelined = {"sub_batches": K, "envs_per_sub_batch": n2, "value": args.steps * N / dt2, "unit": "env-steps/s",
                     "ms_per_step": dt2 / args.steps * 1e3, "window": "steps %d..%d after reset" % (args.warmup, args.warmup + args.steps - 1),
                     "note": "auxiliary: no barrier between the sub-batches (double-buffered rollout); not the headline value"}
        for e2, _, _, _ in subs:
            e2.close()

    ppo_loop = None
    if not args.no_ppo_loop and args.actions == "random" and N == 4096 and args.integrator != "Euler":
        try:
            ppo_loop = ppo_loop_record(args, dev, local_rank, rank, world, launched, barrier)
        except Exception as e:  # noqa: BLE001
            ppo_loop = {"error": repr(e)[:300]}

    g1 = None
    if world == 1 and not args.no_g1 and args.actions == "random" and N == 4096 and args.integrator != "Euler":   # single-GPU runs only
        try:
            g1 = g1_record(local_rank, with_cpu=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:  # noqa: BLE001
            g1 = {"error": repr(e)[:300]}

    if rank == 0:
        total_steps = args.steps * N * world
        value = total_steps / dt
        achieved = N * ALGO_BYTES_PER_ENV_STEP / (kms * 1e-3) / 1e9
        traffic = measured_traffic_bytes() if (N == 4096 and args.actions == "random" and args.integrator != "Euler") else None
        line = {
            "metric": "env-steps/sec (whole node), 34-DoF humanoid, 4096 envs, at 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "ranks_seen": ranks_seen, "devices_distinct": devices_distinct,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2_random_torque: %d envs/GPU, humanoid3d, clip %s, %s h=0.0166 PGS<=50, "
                                   "full DPEnv.step (physics+obs+reward+done+auto-reset), actions %s"
                                   % (N, args.motion, "Euler (implicit damping)" if args.integrator == "Euler" else "RK4",
                                      "U(-2,2) device RNG%s" % (", drawn before the timed region" if pre_actions is not None else ", one launch per step") if args.actions == "random" else "zero"),
                       "envs_per_gpu": N, "parallelism": "env-sharded x%d, no data-path collective" % world,
                       "done_fraction_last_step": done_frac},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes/launch = (FETCH_SIZE + WRITE_SIZE) * 1024 from the separate rocprofv3 --pmc passes "
                                         "committed under profiles/ (4-byte-per-lane accesses: FETCH_SIZE uncalibrated on gfx950)",
                         "kernel": "dm_step_kernel_w3" if N >= 3072 else "dm_step_kernel", "kernel_ms": kms, "kernel_launches_timed": kcount, "kernel_event_stride": EVENT_STRIDE,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP,
                         "valu": valu_view(kms, N) if (N == 4096 and args.actions == "random" and args.integrator != "Euler") else None},
        }
        if ppo_loop is not None:
            line["ppo_loop"] = ppo_loop
        if g1 is not None:
            line["g1"] = g1
        if pipelined is not None:
            line["pipelined"] = pipelined
        if physics_only is not None:
            line["physics_only"] = physics_only
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, mocap)
        print(json.dumps(line))
    eng.close()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
