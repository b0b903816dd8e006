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


def measured_traffic_bytes():
    """HBM bytes per launch of dm_step_kernel from the committed PMC profile (same command, 4096 envs)."""
    import csv
    import glob
    try:
        import re
        files = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_dm_step_kernel.csv"))
        path = max(files, key=lambda f: tuple(int(x) for x in re.findall(r"\d+", os.path.basename(f))))   # newest round / version
        vals = {r["counter"]: float(r["mean_per_dispatch_over_last_10_dispatches"]) for r in csv.DictReader(open(path))}
        return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


def cpu_baseline(model, mocap, budget_s=12.0):
    """Oracle DPEnv.step() on the host cores: one thread per core of this process's CPU share (each thread owns
    its envs, as one SubprocVecEnv worker does), bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import OracleClip, bench_steps
    clip = OracleClip(*mocap.tables())
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                              # the one-GPU box gives a 16-core share whatever the affinity mask says
    t0 = time.perf_counter()
    bench_steps(model, clip, 4, 250, 1234)                      # single-thread rate, to size the sample
    rate1 = 1000 / (time.perf_counter() - t0)
    nenv = max(2, min(64, int(rate1 * budget_s / 1000)))        # envs per thread, 1000 steps each
    with ThreadPoolExecutor(cores) as ex:                       # ctypes releases the GIL; each call owns its DmoData
        t0 = time.perf_counter()
        list(ex.map(lambda k: bench_steps(model, clip, nenv, 1000, 1234 + k), range(cores)))
        dt = time.perf_counter() - t0
    return {"value": cores * nenv * 1000 / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "single_thread_value": rate1,
            "sample": "%d threads x %d envs x 1000 random-torque steps, walk clip, fp64 oracle (oracle/dm_oracle.c)"
                      % (cores, nenv)}


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
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one rank per GPU; --dist-backend gloo + several ranks on one GPU is only for rehearsing the N>1 path
    local_dev = local_rank % torch.cuda.device_count()
    launched = "RANK" in os.environ
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    local_rank = local_dev

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
    eng = HipEngine(model, N, device=local_rank, seed=1234 + rank, auto_reset=True)
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
    # HIP event pair around every EVENT_STRIDE-th dm_step launch of the timed region, on the launch stream (a pair costs ~8 us
    # of stream time, 2.5 % of a step: sampling keeps the kernel-time measurement live without taxing the throughput)
    eng.enable_timing(True, stride=EVENT_STRIDE)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    kms, kcount = eng.mean_step_ms()           # read after the timed region: no host sync inside it
    if launched:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    done_frac = float(out["done"].float().mean().item())

    # Auxiliary figure (never `value`): the same 4096 envs stepped as two independent 2048-env sub-batches on two
    # streams with no barrier between them, as a double-buffered rollout does (policy on one half while the other
    # half simulates): the ramp-down of one launch overlaps the next launch of the other half.
    pipelined = None
    if world == 1 and N % 2 == 0 and args.actions == "random" and not args.no_pipelined:
        K, n2 = 2, N // 2
        subs = []
        for k in range(K):
            e2 = HipEngine(model, n2, device=local_rank, seed=1234 + 17 * (k + 1), auto_reset=True)
            e2.load_clip(0, mocap)
            o2 = e2.alloc_outputs()
            a2 = torch.zeros(n2, 28, device=dev)
            e2.reset(o2["obs"], idx_init=((torch.arange(n2, device=dev) + k * n2) % L).to(torch.int32))
            subs.append((e2, o2, a2, torch.cuda.Stream(device=dev)))

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
                     "ms_per_step": dt2 / args.steps * 1e3,
                     "note": "auxiliary: no barrier between the sub-batches (double-buffered rollout); not the headline value"}
        for e2, _, _, _ in subs:
            e2.close()

    if rank == 0:
        total_steps = args.steps * N * world
        value = total_steps / dt
        achieved = N * ALGO_BYTES_PER_ENV_STEP / (kms * 1e-3) / 1e9
        traffic = measured_traffic_bytes() if (N == 4096 and args.actions == "random") else None
        line = {
            "metric": "env-steps/sec (whole node), 34-DoF humanoid, 4096 envs, at 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2_random_torque: %d envs/GPU, humanoid3d, clip %s, RK4 h=0.0166 PGS<=50, "
                                   "full DPEnv.step (physics+obs+reward+done+auto-reset), actions %s"
                                   % (N, args.motion, "U(-2,2) device RNG" if args.actions == "random" else "zero"),
                       "envs_per_gpu": N, "parallelism": "env-sharded x%d, no data-path collective" % world,
                       "done_fraction_last_step": done_frac},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes/launch = (FETCH_SIZE + WRITE_SIZE) * 1024 from the separate rocprofv3 --pmc passes "
                                         "committed under profiles/ (4-byte-per-lane accesses: FETCH_SIZE uncalibrated on gfx950)",
                         "kernel": "dm_step_kernel", "kernel_ms": kms, "kernel_launches_timed": kcount, "kernel_event_stride": EVENT_STRIDE,
                         "algorithmic_bytes_per_env_step": ALGO_BYTES_PER_ENV_STEP},
        }
        if pipelined is not None:
            line["pipelined"] = pipelined
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, mocap)
        print(json.dumps(line))
    eng.close()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
