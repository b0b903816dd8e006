"""The multi-GPU learner path as it runs on GPUs, rehearsed with two ranks on the one GPU of the test box (gloo).

Covers what the CPU gloo test (tests/test_ppo_cpu.py) cannot: `FlatAdam.all_reduce` on the device-resident flat gradient
(deepmimic_mujoco_amd/ppo.py), the [gather + dm_ppo_mlp_grad] / [dm_flat_adam_update] graphs captured around it, and the
per-rank exploration noise of the fused samplers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["256,128", "1024,512", "1024,512,bf16"])
def test_two_ranks_one_gpu_flat_adam_allreduce(tmp_path, arch):
    """arch 256,128: graph A = gather + dm_ppo_mlp_grad; 1024,512 (the net BASELINE configs 3-5 name): graph A = gather + the
    library-GEMM forward / dm_ppo_loss / backward (dm_linear_tanh, dm_tanh_linear_wgrad, ...) — VERDICT r2 item 3; "1024,512,bf16":
    graph A = gather + dm_ppo_wide_grad (PPO(mlp_dtype=torch.bfloat16), r3)."""
    import torch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", {"256,128": "29533", "1024,512": "29534"}.get(arch, "29535"), os.path.join(ROOT, "tests", "dist_two_rank_worker.py"),
           "--out", str(tmp_path), "--arch", arch.replace(",bf16", "")] + (["--bf16"] if arch.endswith("bf16") else [])
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    a, b = (torch.load(os.path.join(tmp_path, "rank%d.pt" % k)) for k in (0, 1))
    for tag in ("graph", "eager"):
        # one collective per optimizer step (3 minibatches of 128 out of 6 x 64 samples, 1 epoch)
        assert a[tag]["calls"] == 3 and b[tag]["calls"] == 3
        # data-parallel replicas stay BIT-identical: same init (shared seed), same averaged gradient every step
        assert torch.equal(a[tag]["params"], b[tag]["params"])
        # each rank owns different envs (different reset frames) and explores with different noise
        assert not torch.equal(a[tag]["obs0"], b[tag]["obs0"])
        assert float((a[tag]["noise"] - b[tag]["noise"]).abs().max()) > 0.1
        assert torch.isfinite(a[tag]["params"]).all()
    assert a["graph"]["used_dist_graph"] and not a["eager"]["used_dist_graph"]
    # the captured two-graph step computes what the eager step computes
    # (bf16 wide learner at this small minibatch: split-K with fp32 atomics, summation order differs from launch to launch)
    tol = dict(rtol=1e-3, atol=1e-5) if arch.endswith("bf16") else dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(a["graph"]["params"], a["eager"]["params"], **tol)
    assert abs(a["graph"]["loss"] - a["eager"]["loss"]) < (1e-3 if arch.endswith("bf16") else 1e-4) * max(1.0, abs(a["eager"]["loss"]))
