"""Worker of tests/test_dist_two_ranks.py (not collected by pytest): one rank of a 2-rank PPO run over gloo with BOTH
ranks on the one GPU — rehearses the multi-GPU learner path (env shards + the one all-reduce of FlatAdam.flat_g per
optimizer step, src/sb3_ppo.py:307-313 scaled out as BASELINE configs 4/5 ask).  Launched by
`python -m torch.distributed.run --nproc-per-node 2 tests/dist_two_rank_worker.py --out DIR`."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--arch", default="256,128", help="hidden sizes: 256,128 (dm_ppo_mlp_grad) or 1024,512 (library-GEMM learner, BASELINE cfg3-5)")
    ap.add_argument("--bf16", action="store_true", help="PPO(mlp_dtype=torch.bfloat16): the dm_ppo_wide_grad learner")
    args = ap.parse_args()
    arch = tuple(int(x) for x in args.arch.split(","))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    from deepmimic_mujoco_amd.deepmimic_env import HipDeepMimicVecEnv
    from deepmimic_mujoco_amd.ppo import PPO
    res = {}
    for tag, dg in (("graph", True), ("eager", False)):
        env = HipDeepMimicVecEnv(64, motion="spinkick", device=0, seed=1234 + 7919 * rank)
        ppo = PPO(env, net_arch=arch, n_steps=6, batch_size=128, n_epochs=1, seed=3, dist_graph=dg,
                  mlp_dtype=torch.bfloat16 if args.bf16 else torch.float32)
        assert not args.bf16 or ppo._wide_ok
        buf = ppo.collect_rollouts()
        with torch.no_grad():
            mean = ppo.policy.action_net(ppo.policy.pi(buf["obs"].reshape(-1, 67)))
            noise = (buf["act"].reshape(-1, 28) - mean).cpu()
        gen = torch.Generator(device=ppo.device).manual_seed(17 + rank)
        ppo.train(buf, generator=gen)
        torch.cuda.synchronize()
        res[tag] = dict(params=ppo.optimizer.flat_p.detach().cpu().clone(), calls=ppo.optimizer.calls,
                        used_dist_graph=getattr(ppo, "_dg", None) is not None, noise=noise[:64].clone(),
                        loss=ppo.stats["loss"], obs0=buf["obs"][0, :4].cpu().clone())
        env.close()
    torch.save(res, os.path.join(args.out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
