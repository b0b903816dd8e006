"""CPU tests of the learner side: policy MLP, GAE, PPO update, and the world_size-2 gradient
all-reduce over gloo (the N>1 path; RCCL on GPUs uses the same code)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepmimic_mujoco_amd.ppo import PPO, ExtractedPolicy, FlatGradAllReduce, MlpPolicy, compute_gae

G = os.path.join(os.path.dirname(__file__), "golden")


def test_mlp_policy_parameter_counts_match_survey():
    n = lambda arch: sum(p.numel() for p in MlpPolicy(net_arch=arch).parameters())
    assert n((256, 128)) == 104377
    assert n((1024, 512)) == 1203769


def test_extracted_policy_known_answer():
    """src/extracted_policy.py:480-485 through the torch path."""
    pol = ExtractedPolicy(os.path.join(G, "policy_kat.npz"))
    z = np.load(os.path.join(G, "policy_kat.npz"))
    assert np.allclose(pol.act(z["kat_obs"]).numpy(), z["kat_expected"], rtol=1e-4, atol=1e-5)
    assert np.allclose(pol.act(z["extra_obs"]).numpy(), z["extra_act"], rtol=1e-4, atol=1e-5)


def test_gae_matches_naive_recursion():
    torch.manual_seed(0)
    T, N = 12, 5
    r, v = torch.rand(T, N), torch.rand(T, N)
    d = (torch.rand(T, N) < 0.2).float()
    lv = torch.rand(N)
    adv, ret = compute_gae(r, v, d, lv, 0.99, 0.95)
    for n in range(N):
        for t in range(T):
            a, disc = 0.0, 1.0
            for k in range(t, T):
                nv = lv[n] if k == T - 1 else v[k + 1, n]
                delta = r[k, n] + 0.99 * nv * (1 - d[k, n]) - v[k, n]
                a += disc * delta
                if d[k, n]:
                    break
                disc *= 0.99 * 0.95
            assert abs(adv[t, n] - a) < 1e-5
    assert torch.allclose(ret, adv + v)


def _fake_buffer(T, N, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(obs=torch.randn(T, N, 67, generator=g), act=torch.randn(T, N, 28, generator=g),
                rew=torch.rand(T, N, generator=g), done=torch.zeros(T, N), val=torch.randn(T, N, generator=g),
                logp=-40 + torch.randn(T, N, generator=g), adv=torch.randn(T, N, generator=g),
                ret=torch.randn(T, N, generator=g))


def test_ppo_update_reduces_loss_on_fixed_batch():
    ppo = PPO(None, net_arch=(64, 32), n_epochs=1, batch_size=64, device=torch.device("cpu"))
    buf = _fake_buffer(8, 8, 0)
    with torch.no_grad():
        _, lp, _ = ppo.policy.evaluate_actions(buf["obs"].reshape(-1, 67), buf["act"].reshape(-1, 28))
        buf["logp"] = lp.reshape(8, 8)
    first = ppo.train(buf)
    for _ in range(20):
        last = ppo.train(buf)
    assert np.isfinite(first) and last < first


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    pol = MlpPolicy(net_arch=(32, 16))
    sync = FlatGradAllReduce(pol.parameters())
    full = _fake_buffer(4, 8, 1)
    sl = slice(rank * 4, rank * 4 + 4)                     # envs sharded across ranks
    obs, act = full["obs"][:, sl].reshape(-1, 67), full["act"][:, sl].reshape(-1, 28)
    value, logp, _ = pol.evaluate_actions(obs, act)
    loss = (logp * full["adv"][:, sl].reshape(-1)).mean() + ((value - full["ret"][:, sl].reshape(-1)) ** 2).mean()
    loss.backward()
    sync()
    q.put((rank, sync.calls, torch.cat([p.grad.reshape(-1) for p in pol.parameters()]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_world_size_2_equals_full_batch_gradient():
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    pol = MlpPolicy(net_arch=(32, 16))
    full = _fake_buffer(4, 8, 1)
    value, logp, _ = pol.evaluate_actions(full["obs"].reshape(-1, 67), full["act"].reshape(-1, 28))
    loss = (logp * full["adv"].reshape(-1)).mean() + ((value - full["ret"].reshape(-1)) ** 2).mean()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in pol.parameters()]).numpy()
    for rank, calls, g in res:
        assert calls == 1                                   # exactly one collective per optimizer step
        assert np.allclose(g, ref, rtol=1e-4, atol=1e-6)
    assert np.array_equal(res[0][2], res[1][2])             # ranks stay bit-identical
