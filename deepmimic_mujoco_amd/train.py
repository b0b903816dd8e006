"""Training driver — the build's counterpart of src/sb3_ppo.py:244-314 / src/ppo.py:16-40.

    python -m deepmimic_mujoco_amd.train --motion walk --envs 4096 --horizon 32 --total 2000000
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m deepmimic_mujoco_amd.train ...

One process per GPU; every rank owns `--envs` environments (sharded, no collective on the env path)
and a policy replica; gradients are all-reduced once per optimizer step (RCCL over xGMI).
"""
import argparse
import json
import os
import time

import torch
import torch.distributed as dist


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--motion", default="walk", help="clip name or comma list (multi-clip: env i -> clip i mod k)")
    ap.add_argument("--env", default="deep_mimic_mujoco", choices=["deep_mimic_mujoco", "dp_combined_env"],
                    help="env_name of src/sb3_ppo.py:247-248 (dp_combined_env: walk/run/getup state machine on humanoid3d)")
    ap.add_argument("--robot", default="humanoid3d", choices=["humanoid3d", "unitree_g1"],
                    help="src/sb3_ppo.py:251 trains unitree_g1 (its dp_combined_env is hard-wired to it)")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=32)          # 32 x 4096 = the reference's 4096 x 32 batch
    ap.add_argument("--epochs", type=int, default=20)           # src/sb3_ppo.py:259
    ap.add_argument("--minibatch", type=int, default=4096)      # :271
    ap.add_argument("--lr", type=float, default=4e-4)           # :260
    ap.add_argument("--arch", default="256,128")                # :265 ([1024,512] is BASELINE config 3)
    ap.add_argument("--total", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bf16-buffer", action="store_true", help="store rollout obs/actions in bf16 (config 5)")
    ap.add_argument("--bf16-learner", action="store_true",
                    help="PPO(mlp_dtype=bfloat16): bf16 matrix-pipe products in the learner (dm_ppo_wide_grad for [256,128] .. [1024,512]-class "
                         "nets; fp32 master weights, loss, gradients and Adam) — 106 us instead of 378 us per optimizer step on MLP(1024,512)")
    ap.add_argument("--save", default="")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="> 1: the env batch as that many engines, each on its own probed concurrent HIP stream (double-buffered rollout: "
                         "one sub-batch simulates while the policy runs on another; Unitree G1: 2 is ~10 %% faster than 1)")
    ap.add_argument("--rollout-graph", action="store_true", help="with --sub-batches > 1: capture the rollout as one hipGraph")
    ap.add_argument("--json", action="store_true", help="print a JSON throughput summary on rank 0")
    ap.add_argument("--eval-every", type=int, default=0, help="eval dashboard every N global steps (src/sb3_ppo.py:313 EVAL_N); 0 = off")
    ap.add_argument("--run-name", default="run")
    ap.add_argument("--eval-dir", default="~/deep_mimic")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + several ranks on one GPU only rehearses the multi-rank path")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
            local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    rank = dist.get_rank() if world > 1 else 0

    from .deepmimic_env import HipDeepMimicVecEnv
    from .ppo import PPO

    motions = args.motion.split(",")
    sb = args.sub_batches if args.envs % args.sub_batches == 0 else 1
    if args.env == "dp_combined_env":                                   # src/sb3_ppo.py:276-278
        from .combined_env import HipCombinedVecEnv
        env = HipCombinedVecEnv(args.envs, robot=args.robot, device=local_rank, seed=1234 + 7919 * rank,
                                **({"sub_batches": sb} if args.robot == "unitree_g1" else {}))      # (the humanoid3d variant has one engine)
    elif args.robot == "unitree_g1":                                    # src/sb3_ppo.py:274-275 with robot = "unitree_g1"
        env = HipDeepMimicVecEnv(args.envs, motion=motions[0], robot="unitree_g1", device=local_rank, seed=1234 + 7919 * rank, sub_batches=sb)
    else:
        env = HipDeepMimicVecEnv(args.envs, motion=motions if len(motions) > 1 else motions[0], device=local_rank,
                                 seed=1234 + 7919 * rank, sub_batches=sb)
    ppo = PPO(env, net_arch=tuple(int(x) for x in args.arch.split(",")), n_steps=args.horizon,
              batch_size=args.minibatch, n_epochs=args.epochs, learning_rate=args.lr, seed=args.seed,
              buffer_dtype=torch.bfloat16 if args.bf16_buffer else torch.float32, rollout_graph=args.rollout_graph,
              mlp_dtype=torch.bfloat16 if args.bf16_learner else torch.float32)
    hist = []
    dash = None
    if args.eval_every > 0 and rank == 0:                               # src/sb3_ppo.py:273-313: one eval env next to the batch
        from .eval_dashboard import EvalDashboardCallback
        if args.env == "dp_combined_env":
            from .combined_env import DPCombinedEnv
            eval_env = DPCombinedEnv(robot=args.robot, device=local_rank)
        else:
            from .deepmimic_env import DPEnv
            eval_env = DPEnv(motions[0], robot=args.robot, device=local_rank)
        dash = EvalDashboardCallback(eval_env, args.motion + "_" + args.run_name, every_n_global_steps=args.eval_every, out_root=args.eval_dir)

    def _cb(p):
        hist.append(dict(p.stats))
        if dash is not None:
            dash(p)
        return True
    t0 = time.perf_counter()
    ppo.learn(args.total, log_interval=0 if args.json else 1, callback=_cb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        if args.save:
            ppo.save(os.path.expanduser(args.save))
        if args.json:
            steady = hist[1:] if len(hist) > 1 else hist
            roll = sum(h["rollout_s"] for h in steady) / len(steady)
            trn = sum(h["train_s"] for h in steady) / len(steady)
            per_it = args.envs * args.horizon * world
            print(json.dumps({"workload": "ppo", "n_gpus": world, "envs_per_gpu": args.envs, "horizon": args.horizon,
                              "arch": args.arch, "epochs": args.epochs, "minibatch": args.minibatch,
                              "iterations": len(hist), "rollout_env_steps_per_s": per_it / roll,
                              "train_s_per_iter": trn, "overall_env_steps_per_s": per_it / (roll + trn),
                              "mean_reward_last": hist[-1]["mean_reward"], "wall_s": dt,
                              "grad_allreduce_calls": ppo.grad_sync.calls}))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
