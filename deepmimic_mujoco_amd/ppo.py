"""PPO rollout + update loop on device tensors — the driver side of the hot path.

Mirrors what the reference's training scripts ask Stable-Baselines to do:
  src/sb3_ppo.py:254-271,307-312  PPO(MlpPolicy, envs, net_arch=[256,128], n_steps=4096, lr=4e-4,
                                      n_epochs=20, batch_size=4096)
  src/ppo.py:21-39                PPO2(MlpPolicy, n_steps=128, nminibatches=4, noptepochs=12, lr=2.5e-4)
with SB3's defaults for everything the scripts leave unset [EXT]: gamma 0.99, gae_lambda 0.95,
clip_range 0.2, ent_coef 0, vf_coef 0.5, max_grad_norm 0.5, per-minibatch advantage
normalisation, tanh MLPs with separate policy / value trunks, orthogonal init, state-independent
log-std initialised to 0, actions clipped to the action space before env.step.

Differences by design (MI355X-first): observations, actions, rewards and dones never leave HBM
(``HipDeepMimicVecEnv.step_tensor``); the MLP GEMMs run on MFMA through PyTorch-ROCm; multi-GPU is
one process per GPU with ONE all-reduce of the flat gradient per optimizer step over RCCL/xGMI
(``FlatGradAllReduce``), envs sharded across ranks with no other collective.
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch
import torch.distributed as dist
from torch import nn
from .streams import concurrent_streams


class _HipLinearFn(torch.autograd.Function):
    """y = x W^T + b; the backward computes dW and db with `dm_linear_wgrad` (MFMA split-K over the batch, csrc/dm_ppo.hip)
    and dX with the library GEMM."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        ctx.save_for_backward(x, w)
        ctx.mod = mod
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        import ctypes as C
        from . import _lib
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gy = gy.contiguous()
        arena = getattr(ctx.mod, "_grad_arena", None)    # (gw, gb) views of one flat buffer zeroed once per step
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        L = _lib.load_library()
        if max(w.shape) > 256 and min(w.shape) > 128:
            # large square-ish layers: dW on the library GEMM, written straight into the arena (no autograd-owned gradient, so
            # no copy into the flat buffer afterwards), db by dm_colsum.  Skinny layers of a big net (1024 x 67, 28 x 512,
            # 1 x 512: K = 4096 reductions into small outputs, ~38 us each in the library) take dm_linear_wgrad below.
            if arena is not None:
                gw, gb = arena
                torch.mm(gy.t(), x, out=gw)
            else:
                gw = gy.t() @ x
                gb = torch.zeros(w.shape[0], device=w.device, dtype=w.dtype)
            rc = L.dm_colsum(p(gy), x.shape[0], w.shape[0], p(gb), stream)
            if rc != 0:
                raise RuntimeError("dm_colsum failed (%d)" % rc)
            return (gx, None, None, None) if arena is not None else (gx, gw, gb, None)
        if arena is not None:
            gw, gb = arena
        else:
            gw = torch.zeros_like(w)
            gb = torch.zeros(w.shape[0], device=w.device, dtype=w.dtype)
        rc = L.dm_linear_wgrad(p(gy), p(x), p(gw), p(gb), x.shape[0], w.shape[0], w.shape[1], stream)
        if rc != 0:
            raise RuntimeError("dm_linear_wgrad failed (%d)" % rc)
        if arena is not None:
            # the gradients already sit in the optimizer's flat buffer: handing views to autograd would make
            # AccumulateGrad clone them (a view cannot be stolen) and the optimizer copy them back — 2 copies per tensor
            return gx, None, None, None
        return gx, gw, gb, None


class _HipLinearBf16Fn(torch.autograd.Function):
    """y = x W^T + b with bf16 MFMA GEMMs (fp32 accumulation inside the GEMM, bf16 activations) against bf16 shadows of the
    fp32 master weights kept by ``FlatAdam`` — the mixed-precision option of the library-GEMM learner (``PPO(mlp_dtype=
    torch.bfloat16)``).  dW / db are written into the optimizer's fp32 gradient arena, as in the fp32 path."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        wb, bb = mod._bf16
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        ctx.save_for_backward(xb)
        ctx.mod = mod
        return torch.addmm(bb, xb, wb.t())

    @staticmethod
    def backward(ctx, gy):
        (xb,) = ctx.saved_tensors
        mod = ctx.mod
        wb, _ = mod._bf16
        gy = gy.contiguous()
        gx = (gy @ wb) if ctx.needs_input_grad[0] else None
        gw, gb = mod._grad_arena
        gw.copy_(gy.t() @ xb)                        # bf16 product (fp32 accumulation), widened into the fp32 arena
        gb.copy_(gy.sum(0, dtype=torch.float32))
        return gx, None, None, None


class _HipLinearTanhFn(torch.autograd.Function):
    """tanh(x W^T + b) of a wide trunk ([1024,512]) with the activation fused around the GEMM (csrc/dm_ppo.hip):
    first layer (x = observations, <= 128 wide, no input gradient): `dm_linear_tanh` forward, `dm_tanh_linear_wgrad` backward —
    no activation pass, no dZ; deeper layers: library GEMM + in-place tanh forward, `dm_tanh_bwd_colsum` (tanh' and the bias
    gradient in one pass) + library GEMMs backward.  Gradients go straight into the optimizer's flat arena."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        import ctypes as C
        from . import _lib
        first = x.shape[1] <= 128 and not x.requires_grad
        if first:
            y = torch.empty(x.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
            rc = _lib.load_library().dm_linear_tanh(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()),
                                                    C.c_void_p(y.data_ptr()), x.shape[0], w.shape[0], x.shape[1],
                                                    C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("dm_linear_tanh failed (%d)" % rc)
        else:
            y = torch.addmm(b, x, w.t()).tanh_()
        ctx.save_for_backward(x, w, y)
        ctx.mod, ctx.first = mod, first
        return y

    @staticmethod
    def backward(ctx, gy):
        import ctypes as C
        from . import _lib
        x, w, y = ctx.saved_tensors
        gw, gb = ctx.mod._grad_arena
        gy = gy.contiguous()
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        L = _lib.load_library()
        if ctx.first:
            rc = L.dm_tanh_linear_wgrad(p(gy), p(y), p(x), p(gw), p(gb), x.shape[0], w.shape[0], w.shape[1], stream)
            if rc != 0:
                raise RuntimeError("dm_tanh_linear_wgrad failed (%d)" % rc)
            return None, None, None, None
        gz = torch.empty_like(gy)
        rc = L.dm_tanh_bwd_colsum(p(gy), p(y), p(gz), p(gb), x.shape[0], w.shape[0], stream)
        if rc != 0:
            raise RuntimeError("dm_tanh_bwd_colsum failed (%d)" % rc)
        torch.mm(gz.t(), x, out=gw)
        gx = gz @ w if ctx.needs_input_grad[0] else None
        return gx, None, None, None


def _fused_tanh_trunk(seq, x):
    """Run an nn.Sequential of (HipLinear, Tanh) pairs through `_HipLinearTanhFn` where it applies (wide fp32 layers of a CUDA
    minibatch whose gradients live in the flat arena); anything else through the modules themselves."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        if (isinstance(m, HipLinear) and isinstance(nxt, nn.Tanh) and m._bf16 is None and x.is_cuda and x.dim() == 2
                and x.dtype == torch.float32 and x.is_contiguous() and x.shape[0] >= 1024 and x.shape[0] % 64 == 0
                and torch.is_grad_enabled() and getattr(m, "_grad_arena", None) is not None and m.weight.shape[0] > 256
                and ((x.shape[1] <= 128 and not x.requires_grad) or min(m.weight.shape) > 128)):
            x = _HipLinearTanhFn.apply(x, m.weight, m.bias, m)
            i += 2
        else:
            x = m(x)
            i += 1
    return x


class HipLinear(nn.Linear):
    """nn.Linear whose weight / bias gradients bypass autograd's generic kernels when the batch is a large CUDA minibatch:
    layers up to 256 x 256 and skinny layers (one side <= 128) on `dm_linear_wgrad` (the library's K = 4096 GEMMs into small
    outputs take 25-38 us each), large ones on the library GEMM + `dm_colsum`, all written straight into the optimizer's flat
    gradient arena."""

    _bf16 = None            # (weight, bias) bf16 shadow views, set by FlatAdam.enable_bf16_shadow

    def forward(self, x):
        if (self._bf16 is not None and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled()
                and getattr(self, "_grad_arena", None) is not None):
            return _HipLinearBf16Fn.apply(x, self.weight, self.bias, self)
        if (x.is_cuda and x.dim() == 2 and x.shape[0] >= 1024 and x.shape[0] % 64 == 0 and x.dtype == torch.float32
                and x.is_contiguous() and torch.is_grad_enabled()):
            return _HipLinearFn.apply(x, self.weight, self.bias, self)
        return super().forward(x)


class MlpPolicy(nn.Module):
    """SB3 ``ActorCriticPolicy`` with ``net_arch=[h1, h2]`` (shared sizes, separate trunks), tanh."""

    def __init__(self, obs_dim=67, act_dim=28, net_arch=(256, 128), log_std_init=0.0):
        super().__init__()

        def trunk():
            layers, d = [], obs_dim
            for hdim in net_arch:
                layers += [HipLinear(d, hdim), nn.Tanh()]
                d = hdim
            return nn.Sequential(*layers), d

        self.pi, dpi = trunk()
        self.vf, dvf = trunk()
        self.action_net = HipLinear(dpi, act_dim)
        self.value_net = HipLinear(dvf, 1)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std_init)))
        for seq in (self.pi, self.vf):
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=math.sqrt(2))
                    nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.action_net.weight, gain=0.01)
        nn.init.zeros_(self.action_net.bias)
        nn.init.orthogonal_(self.value_net.weight, gain=1.0)
        nn.init.zeros_(self.value_net.bias)

    def forward(self, obs, deterministic=False):
        mean = self.action_net(self.pi(obs))
        value = self.value_net(self.vf(obs)).squeeze(-1)
        std = self.log_std.exp()
        act = mean if deterministic else mean + std * torch.randn_like(mean)
        logp = self._logp(act, mean)
        return act, value, logp

    def _logp(self, act, mean):
        var = (2 * self.log_std).exp()
        return (-0.5 * ((act - mean) ** 2 / var) - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def evaluate_actions(self, obs, act):
        mean = self.action_net(self.pi(obs))
        value = self.value_net(self.vf(obs)).squeeze(-1)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum().expand(obs.shape[0])
        return value, self._logp(act, mean), entropy

    def predict_values(self, obs):
        return self.value_net(self.vf(obs)).squeeze(-1)


class FusedPolicyForward:
    """``dm_policy_forward`` (csrc/dm_policy.hip) for an ``MlpPolicy`` with two hidden layers: both trunks, the sampling
    head and the policy-side rollout-buffer writes as one launch.  ``pack()`` re-orders the weights into MFMA operand
    order and must be called again after the weights changed (once per ``collect_rollouts``)."""

    def __init__(self, policy, device):
        from . import _lib
        self.lib = _lib.load_library()
        self.policy, self.device = policy, device
        lin = lambda seq: [m for m in seq if isinstance(m, nn.Linear)]
        self.pi, self.vf = lin(policy.pi) + [policy.action_net], lin(policy.vf) + [policy.value_net]
        self.D, self.H1, self.H2, self.A = self.pi[0].in_features, self.pi[0].out_features, self.pi[1].out_features, self.pi[2].out_features
        n = int(self.lib.dm_policy_packed_floats(self.D, self.H1, self.H2, self.A))
        if n <= 0:
            raise ValueError("dm_policy_forward does not support this net")
        self.packed = [torch.zeros(n, device=device), torch.zeros(n, device=device)]

    @staticmethod
    def supported(policy, device):
        if device.type != "cuda" or not isinstance(policy, MlpPolicy):
            return False
        lin = [m for m in policy.pi if isinstance(m, nn.Linear)]
        if len(lin) != 2 or len([m for m in policy.vf if isinstance(m, nn.Linear)]) != 2:
            return False
        h1, h2, a, d = lin[0].out_features, lin[1].out_features, policy.action_net.out_features, lin[0].in_features
        lds = 32 * (h1 + 4 + max(((d + 7) // 8) * 8 + 4, 132)) * 4
        return h1 % 32 == 0 and h2 % 32 == 0 and a <= 32 and lds <= 160 * 1024 and all(
            m.weight.dtype == torch.float32 and m.weight.is_contiguous() for m in lin)

    def _stream(self):
        import ctypes as C
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def pack(self):
        import ctypes as C
        p = lambda t: C.c_void_p(t.data_ptr())
        for layers, out, a in ((self.pi, self.packed[0], self.A), (self.vf, self.packed[1], 1)):
            rc = self.lib.dm_policy_pack(p(layers[0].weight), p(layers[1].weight), p(layers[2].weight), self.D, self.H1, self.H2, a,
                                         p(out), self._stream())
            if rc != 0:
                raise RuntimeError("dm_policy_pack failed (%d)" % rc)

    def __call__(self, obs, seed, counter, draw_offset, lo, hi, act, act_env, logp, val, obs_copy=None, mean_out=None,
                 deterministic=False):
        import ctypes as C
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        pi, vf = self.pi, self.vf
        rc = self.lib.dm_policy_forward(
            p(obs), obs.shape[0], self.D, self.H1, self.H2, self.A, p(self.packed[0]), p(pi[0].bias), p(pi[1].bias), p(pi[2].bias),
            p(self.packed[1]), p(vf[0].bias), p(vf[1].bias), p(vf[2].bias), p(self.policy.log_std), C.c_uint64(seed), p(counter),
            C.c_uint32(draw_offset), int(bool(deterministic)), p(lo), p(hi), p(mean_out), p(act), p(act_env), p(logp), p(val),
            p(obs_copy), self._stream())
        if rc != 0:
            raise RuntimeError("dm_policy_forward failed (%d)" % rc)


class FusedMlpGrad:
    """``dm_ppo_mlp_grad`` (csrc/dm_ppo_mlp.hip): loss and every gradient of one PPO minibatch for an ``MlpPolicy`` with
    two hidden layers of at most 256 units, in three launches, written into the flat gradient arena of ``FlatAdam``
    (which the caller zeroes).  Returns the loss as a view of the device-side ``out8`` record."""

    def __init__(self, policy, opt, B, loss_acc=None, fold=True):
        import ctypes as C
        from . import _lib
        self.lib, self.C, self.St = _lib.load_library(), C, _lib.DmPpoMlpStep
        lin = lambda seq: [m for m in seq if isinstance(m, nn.Linear)]
        pi, vf = lin(policy.pi) + [policy.action_net], lin(policy.vf) + [policy.value_net]
        dev = policy.log_std.device
        D, H1, H2, A = pi[0].in_features, pi[0].out_features, pi[1].out_features, pi[2].out_features
        n = int(self.lib.dm_ppo_mlp_workspace_floats(B, D, H1, H2, A))
        if n <= 0:
            raise ValueError("dm_ppo_mlp_grad does not support this net / minibatch")
        self.ws = torch.zeros(n, device=dev)
        self.out8 = torch.zeros(8, device=dev)
        grad = {id(p): g for p, g in zip(opt.params, opt.slices)}
        st = self.St()
        st.B, st.D, st.H1, st.H2, st.A = B, D, H1, H2, A
        for t, layers in enumerate((pi, vf)):
            for l, m in enumerate(layers):
                st.W[t][l], st.b[t][l] = m.weight.data_ptr(), m.bias.data_ptr()
                st.gW[t][l], st.gb[t][l] = grad[id(m.weight)].data_ptr(), grad[id(m.bias)].data_ptr()
        st.log_std, st.g_log_std = policy.log_std.data_ptr(), grad[id(policy.log_std)].data_ptr()
        st.out8, st.workspace, st.workspace_floats = self.out8.data_ptr(), self.ws.data_ptr(), n
        st.reserved = int(__import__("os").environ.get("DM_WGRAD_SPLITK", "0"))      # 0: library default (experiments)
        self.folded = bool(fold)
        if fold:    # the launch also clears the gradient arena, does Adam's begin and keeps the running loss sum
            self.loss_acc = loss_acc if loss_acc is not None else torch.zeros(2, device=dev)
            st.zero_ptr, st.zero_floats = opt.flat_g.data_ptr(), opt.n
            st.adam_state2, st.loss_acc = opt.state2.data_ptr(), self.loss_acc.data_ptr()
        self.st, self.B, self.dev = st, B, dev

    @staticmethod
    def supported(policy, B):
        if not isinstance(policy, MlpPolicy) or policy.log_std.device.type != "cuda":
            return False
        lin = [m for m in policy.pi if isinstance(m, nn.Linear)]
        if len(lin) != 2 or len([m for m in policy.vf if isinstance(m, nn.Linear)]) != 2:
            return False
        h1, h2 = lin[0].out_features, lin[1].out_features
        return (B >= 64 and B % 64 == 0 and h1 % 32 == 0 and h2 % 32 == 0 and h1 <= 256 and h2 <= 256
                and policy.action_net.out_features <= 32)

    def __call__(self, obs, act, adv, ret, old_logp, clip_range, vf_coef, ent_coef, normalize):
        st = self.st
        for t in (obs, act, adv, ret, old_logp):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.shape[0] == self.B
        st.obs, st.act, st.adv, st.ret, st.old_logp = (t.data_ptr() for t in (obs, act, adv, ret, old_logp))
        st.clip_range, st.vf_coef, st.ent_coef, st.normalize_advantage = clip_range, vf_coef, ent_coef, int(bool(normalize))
        rc = self.lib.dm_ppo_mlp_grad(self.C.byref(st), self.C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_ppo_mlp_grad failed (%d)" % rc)
        return self.out8[0]


class WideMlpGrad:
    """One minibatch gradient of the WIDE net ([1024,512]-class, the net BASELINE configs 3-5 name) with bf16 matrix-pipe products
    (`dm_ppo_wide_grad`, csrc/dm_ppo_wide.hip): weights -> bf16, the fused forward / loss / input-gradient chain of both trunks, the
    six weight gradients — three launches.  fp32 master weights, fp32 loss, gradients and Adam.  Leaves every gradient in
    ``opt.flat_g`` (cleared by the first launch, which also performs Adam's begin) and adds the loss to ``loss_acc``."""

    def __init__(self, policy, opt, B, loss_acc):
        import ctypes as C
        from . import _lib
        self.lib, self.C = _lib.load_library(), C
        lin = lambda seq: [m for m in seq if isinstance(m, nn.Linear)]
        pi, vf = lin(policy.pi) + [policy.action_net], lin(policy.vf) + [policy.value_net]
        dev = policy.log_std.device
        D, H1, H2, A = pi[0].in_features, pi[0].out_features, pi[1].out_features, pi[2].out_features
        Dp = int(self.lib.dm_ppo_wide_dp(D))
        bf = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.bfloat16)
        npk = int(self.lib.dm_ppo_wide_packed_elems(D, H1, H2))
        self.buf = dict(wpk=[bf(npk), bf(npk)], xbT=bf((Dp + 31) // 32 * 32, B), h1T=[bf(H1, B), bf(H1, B)], dz1T=[bf(H1, B), bf(H1, B)], h2T=[bf(H2, B), bf(H2, B)],
                        dz2T=[bf(H2, B), bf(H2, B)], dz3T=[bf(32, B), bf(32, B)], part=torch.zeros(2 * (B // 32) * 40, device=dev),
                        stats8=torch.zeros(8, device=dev), out8=torch.zeros(8, device=dev))
        grad = {id(p): g for p, g in zip(opt.params, opt.slices)}
        st = _lib.DmPpoWideStep()
        st.B, st.D, st.H1, st.H2, st.A = B, D, H1, H2, A
        for t, layers in enumerate((pi, vf)):
            for l, m in enumerate(layers):
                st.W[t][l], st.b[t][l] = m.weight.data_ptr(), m.bias.data_ptr()
                st.gW[t][l], st.gb[t][l] = grad[id(m.weight)].data_ptr(), grad[id(m.bias)].data_ptr()
            for k in ("wpk", "h1T", "dz1T", "h2T", "dz2T", "dz3T"):
                getattr(st, k)[t] = self.buf[k][t].data_ptr()
        st.xbT, st.part, st.stats8, st.out8 = (self.buf[k].data_ptr() for k in ("xbT", "part", "stats8", "out8"))
        st.log_std, st.g_log_std = policy.log_std.data_ptr(), grad[id(policy.log_std)].data_ptr()
        st.zero_ptr, st.zero_floats = opt.flat_g.data_ptr(), opt.n
        st.adam_state2, st.loss_acc = opt.state2.data_ptr(), loss_acc.data_ptr()
        self.st, self.B, self.dev = st, B, dev

    @staticmethod
    def supported(policy, B):
        if not isinstance(policy, MlpPolicy) or policy.log_std.device.type != "cuda":
            return False
        lin = [m for m in policy.pi if isinstance(m, nn.Linear)]
        if len(lin) != 2 or len([m for m in policy.vf if isinstance(m, nn.Linear)]) != 2:
            return False
        from . import _lib
        return bool(_lib.load_library().dm_ppo_wide_supported(int(B), lin[0].in_features, lin[0].out_features, lin[1].out_features,
                                                              policy.action_net.out_features))

    def __call__(self, obs, act, adv, ret, old_logp, clip_range, vf_coef, ent_coef, normalize):
        st = self.st
        for t in (obs, act, adv, ret, old_logp):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.shape[0] == self.B
        st.obs, st.act, st.adv, st.ret, st.old_logp = (t.data_ptr() for t in (obs, act, adv, ret, old_logp))
        st.clip_range, st.vf_coef, st.ent_coef, st.normalize_advantage = clip_range, vf_coef, ent_coef, int(bool(normalize))
        rc = self.lib.dm_ppo_wide_grad(self.C.byref(st), self.C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_ppo_wide_grad failed (%d)" % rc)
        return self.buf["out8"][0]


class ExtractedPolicy:
    """The reference's exported walk policy: a = tanh(tanh(o W0 + B0) W2 + B2) WA + BA
    (src/extracted_policy.py:471-478; used with obs[:66] and clip +-0.5, src/play_extracted.py:36-38)."""

    def __init__(self, npz_path, device="cpu"):
        z = np.load(npz_path)
        self.p = {k: torch.tensor(z[k], dtype=torch.float32, device=device) for k in ("W0", "B0", "W2", "B2", "WA", "BA")}
        assert self.p["W0"].shape == (66, 256) and self.p["W2"].shape == (256, 128) and self.p["WA"].shape == (128, 28)
        self.obs_shape, self.act_shape = 66, 28

    def act(self, obs):
        o = torch.as_tensor(obs, dtype=torch.float32, device=self.p["W0"].device)
        f = torch.tanh(o @ self.p["W0"] + self.p["B0"])
        f = torch.tanh(f @ self.p["W2"] + self.p["B2"])
        return f @ self.p["WA"] + self.p["BA"]


class FusedPPOLoss(torch.autograd.Function):
    """loss = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss of SB3's PPO.train, forward and backward in
    one pass of `dm_ppo_loss` (csrc/dm_ppo.hip) on the current stream.  The gradient buffers are static per shape, so
    the call can sit inside a captured hipGraph.  `stats` (8 floats, see include/deepmimic_hip.h) stays on the device."""

    _bufs = {}

    @staticmethod
    def buffers(B, A, dev):
        key = (B, A, str(dev))
        if key not in FusedPPOLoss._bufs:
            z = lambda *s: torch.zeros(*s, device=dev)
            FusedPPOLoss._bufs[key] = dict(gm=z(B, A), gl=z(A), gv=z(B), out=z(8), scratch=z(8))
        return FusedPPOLoss._bufs[key]

    @staticmethod
    def forward(ctx, mean, log_std, value, act, old_logp, adv, ret, clip_range, vf_coef, ent_coef, normalize):
        from . import _lib
        import ctypes as C
        L = _lib.load_library()
        B, A = mean.shape
        b = FusedPPOLoss.buffers(B, A, mean.device)
        args = [t.detach().contiguous().float() for t in (mean, log_std, value, act, old_logp, adv, ret)]
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = L.dm_ppo_loss(*[p(t) for t in args], B, A, float(clip_range), float(vf_coef), float(ent_coef),
                           1 if normalize else 0, p(b["gm"]), p(b["gl"]), p(b["gv"]), p(b["out"]), p(b["scratch"]),
                           C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_ppo_loss failed (%d)" % rc)
        ctx.b = b
        ctx._keep = args
        return b["out"][0]

    @staticmethod
    def backward(ctx, g):
        b = ctx.b
        return (g * b["gm"], g * b["gl"], g * b["gv"]) + (None,) * 8

    @staticmethod
    def raw(mean, log_std, value, act, old_logp, adv, ret, clip_range, vf_coef, ent_coef, normalize, grad_log_std=None):
        """The same launch outside autograd: returns (loss, d loss / d mean, d loss / d value) as static buffers and
        writes d loss / d log_std into ``grad_log_std`` (e.g. the optimizer's arena slice).  The caller seeds the
        backward pass with ``torch.autograd.backward([mean, value], [gm, gv])``: no `g * grad` products, no gradient
        tensors handed back through a Function node (three multiplies, an add and a copy per step on the library path)."""
        from . import _lib
        import ctypes as C
        B, A = mean.shape
        b = FusedPPOLoss.buffers(B, A, mean.device)
        gl = grad_log_std if grad_log_std is not None else b["gl"]
        args = [t.detach() for t in (mean, log_std, value, act, old_logp, adv, ret)]
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in args)
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = _lib.load_library().dm_ppo_loss(*[p(t) for t in args], B, A, float(clip_range), float(vf_coef), float(ent_coef),
                                             1 if normalize else 0, p(b["gm"]), p(gl), p(b["gv"]), p(b["out"]), p(b["scratch"]),
                                             C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_ppo_loss failed (%d)" % rc)
        return b["out"][0], b["gm"], b["gv"]


def compute_gae(rewards, values, dones, last_values, gamma, lam):
    """SB3 ``RolloutBuffer.compute_returns_and_advantage`` [EXT]: tensors [T, N]; dones[t] is the done
    flag returned by step t (so the value after it is not bootstrapped)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        next_v = last_values if t == T - 1 else values[t + 1]
        nonterm = 1.0 - dones[t]
        delta = rewards[t] + gamma * next_v * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv, adv + values


class FlatGradAllReduce:
    """One collective per optimizer step: flatten every gradient into a single fp32 buffer,
    all-reduce (sum) it over the process group (RCCL over xGMI on GPUs, gloo in CPU tests), divide by
    the world size, scatter back.  104 377 floats for [256,128], 1 203 769 for [1024,512]."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.calls = 0

    def __call__(self):
        if self.world == 1:
            return
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.grad.reshape(-1) if p.grad is not None else torch.zeros(n, device=self.flat.device))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.calls += 1
        self.flat.div_(self.world)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


class FlatAdam:
    """clip_grad_norm_ + Adam on one flat buffer (`dm_flat_adam_step`, csrc/dm_ppo.hip): every parameter of the policy
    becomes a view of `flat_p`, every gradient is gathered in `flat_g` (the HipLinear layers write theirs there
    directly), and the whole update is two launches.  With several ranks `flat_g` is also what is all-reduced: the
    one collective of the data-parallel learner, without staging copies.  GPU only; `state` mimics torch.optim's layout
    far enough for the snapshot / restore around hipGraph capture."""

    def __init__(self, policy, lr, betas=(0.9, 0.999), eps=1e-5, max_grad_norm=0.5):
        self.params = [p for p in policy.parameters()]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n, self.lr, self.betas, self.eps, self.max_grad_norm = n, lr, betas, eps, max_grad_norm
        self.flat_p = torch.empty(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.state2 = torch.zeros(2 + 1024, device=dev)   # [scratch, step count, DM_ADAM_PARTIALS per-block sums of squares]
        self.slices = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                self.slices.append(self.flat_g[off:off + k].view_as(p))
                off += k
        # HipLinear layers accumulate their weight / bias gradients straight into flat_g
        byid = {id(p): g for p, g in zip(self.params, self.slices)}
        for mod in policy.modules():
            if isinstance(mod, HipLinear):
                mod._grad_arena = (byid[id(mod.weight)], byid[id(mod.bias)])
        self.state = {"flat": {"exp_avg": self.m, "exp_avg_sq": self.v, "state2": self.state2}}
        self.calls = 0          # collectives issued (multi-rank)
        self.flat_pb = None     # bf16 shadow of flat_p (mixed-precision learner), refreshed after every update
        self._policy = policy

    def enable_bf16_shadow(self):
        """bf16 copies of the fp32 master weights for the GEMMs of the library-path learner; every HipLinear gets views."""
        self.flat_pb = self.flat_p.to(torch.bfloat16)
        views, off = {}, 0
        for p in self.params:
            k = p.numel()
            views[id(p)] = self.flat_pb[off:off + k].view_as(p)
            off += k
        for mod in self._policy.modules():
            if isinstance(mod, HipLinear):
                mod._bf16 = (views[id(mod.weight)], views[id(mod.bias)])

    def zero_grad(self, set_to_none=True):
        self.flat_g.zero_()
        for p in self.params:
            p.grad = None

    def gather_grads(self):
        """Gradients that autograd produced outside flat_g (log_std, layers on the library path) are copied in."""
        for p, g in zip(self.params, self.slices):
            if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad)

    def all_reduce(self):
        """The ONE collective of the data-parallel learner: sum of the flat gradient over the ranks, in place.  The division by
        the world size is not a launch of its own: `step()` hands 1 / world to the update kernel as `grad_scale`."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
            self.calls += 1

    @property
    def grad_scale(self):
        return 1.0 / dist.get_world_size() if (dist.is_initialized() and dist.get_world_size() > 1) else 1.0

    def step(self, begin=True, gather_next=None):
        """begin=False: state2 was prepared by dm_ppo_mlp_grad (adam_state2 fold) — two launches instead of three.
        gather_next = (flat, idx, out): the gather of the NEXT minibatch (rows idx of flat["obs" | "act" | "adv" | "ret" | "logp"] ->
        out[...]) rides on the norm launch (dm_flat_adam_step_gather): one launch less per optimizer step."""
        import ctypes as C
        from . import _lib
        p = lambda t: C.c_void_p(t.data_ptr())
        L = _lib.load_library()
        common = (p(self.flat_p), p(self.flat_g), p(self.m), p(self.v), self.n, self.lr, self.betas[0], self.betas[1], self.eps,
                  self.max_grad_norm, self.grad_scale, p(self.state2), int(self.state2.numel()))
        stream = C.c_void_p(torch.cuda.current_stream(self.flat_p.device).cuda_stream)
        if gather_next is None:
            rc = (L.dm_flat_adam_step if begin else L.dm_flat_adam_update)(*common, stream)
        else:
            flat, idx, out = gather_next
            assert idx.dtype == torch.int64 and idx.is_contiguous()
            for k in ("obs", "act", "adv", "ret", "logp"):
                assert flat[k].dtype == torch.float32 and flat[k].is_contiguous() and out[k].dtype == torch.float32 and out[k].is_contiguous()
            gs = _lib.DmGatherSpec()
            gs.idx, gs.B, gs.D, gs.A = idx.data_ptr(), int(idx.numel()), int(flat["obs"].shape[1]), int(flat["act"].shape[1])
            gs.obs, gs.act, gs.adv, gs.ret, gs.logp = (flat[k].data_ptr() for k in ("obs", "act", "adv", "ret", "logp"))
            gs.o_obs, gs.o_act, gs.o_adv, gs.o_ret, gs.o_logp = (out[k].data_ptr() for k in ("obs", "act", "adv", "ret", "logp"))
            rc = L.dm_flat_adam_step_gather(*common, int(bool(begin)), C.byref(gs), stream)
        if rc != 0:
            raise RuntimeError("dm_flat_adam_step failed (%d)" % rc)
        if self.flat_pb is not None:
            self.flat_pb.copy_(self.flat_p)

    def state_dict(self):
        return {"flat_adam": True, "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone(), "state2": self.state2.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "max_grad_norm": self.max_grad_norm}

    def load_state_dict(self, sd):
        self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"])
        # state2 = [scratch, step count, per-block partial sums]: only the step count is state.  Checkpoints written before the
        # fixed-order norm (r1 / early r2, incl. the eval dashboard's "_best" models) carry two floats: take those, zero the rest.
        st2 = sd["state2"].to(self.state2.device).reshape(-1)
        k = min(int(st2.numel()), int(self.state2.numel()))
        self.state2.zero_()
        self.state2[:k].copy_(st2[:k])
        self.lr, self.betas, self.eps = sd["lr"], tuple(sd["betas"]), sd["eps"]


class PPO:
    def __init__(self, env, net_arch=(256, 128), n_steps=4096, batch_size=4096, n_epochs=20, learning_rate=4e-4,
                 gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.0, vf_coef=0.5, max_grad_norm=0.5,
                 normalize_advantage=True, seed=0, device=None, buffer_dtype=torch.float32, policy=None,
                 use_hip_graph=None, fused_loss=True, flat_adam=True, two_stream=True, rollout_graph=True, fused_rollout=True,
                 fused_policy=True, fused_mlp=True, epoch_graph=True, dist_graph=True, mlp_dtype=torch.float32, fused_wide=True):
        # rollout_graph only takes effect for an env built with sub_batches > 1
        self.env = env
        self.device = device if device is not None else getattr(env, "device", torch.device("cpu"))
        self.n_envs = env.num_envs if env is not None else 0
        self.n_steps, self.batch_size, self.n_epochs = n_steps, batch_size, n_epochs
        self.gamma, self.gae_lambda, self.clip_range = gamma, gae_lambda, clip_range
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.normalize_advantage = normalize_advantage
        self.fused_loss = fused_loss          # dm_ppo_loss (HIP) for the loss tail when the batch is on the GPU
        self.two_stream = two_stream          # value trunk on a second stream (parallel graph branches)
        self.rollout_graph = rollout_graph    # env with sub_batches > 1: the T-step rollout is one captured hipGraph
        self.fused_rollout = fused_rollout    # dm_policy_sample + dm_rollout_store instead of ~20 small kernels per step
        self.fused_policy = fused_policy      # dm_policy_forward: the whole policy side of a rollout step as one launch
        self.epoch_graph = epoch_graph        # one hipGraph replay per epoch (gathers + optimizer steps) instead of one per minibatch
        self.fused_mlp = fused_mlp            # dm_ppo_mlp_grad: loss + all gradients of a minibatch in three launches
        self.dist_graph = dist_graph          # several ranks: [gather + gradient] and [Adam] stay captured graphs around the all-reduce
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # the fused samplers key their draws on (seed, LOCAL env index, counter, action index): the rank must be part of
        # the seed or every rank would explore with bit-identical noise (weights still start identical: torch seed below)
        self._rollout_seed = (0x5EED0000 + seed + 0x9E3779B1 * self.rank) & 0x7FFFFFFFFFFFFFFF
        self.buffer_dtype = buffer_dtype
        torch.manual_seed(seed)  # identical initial weights on every rank; no parameter broadcast needed
        self.obs_dim = int(env.observation_space.shape[0]) if env is not None else 67   # 67 DPEnv, 72 DPCombinedEnv; G1: 85 / 98
        self.act_dim = int(env.action_space.shape[0]) if env is not None else 28        # 28 humanoid3d, 23 Unitree G1
        self.policy = (policy if policy is not None else MlpPolicy(obs_dim=self.obs_dim, act_dim=self.act_dim, net_arch=tuple(net_arch))).to(self.device)
        on_gpu = self.device.type == "cuda"
        # The optimizer step of one minibatch is ~60 small kernels: launch-bound.  On one GPU it is captured
        # once into a hipGraph and replayed (640 replays per PPO iteration with the reference's settings).
        self.use_hip_graph = (on_gpu and not dist.is_initialized()) if use_hip_graph is None else bool(use_hip_graph)
        self.flat_adam = on_gpu and flat_adam
        if self.flat_adam:
            self.optimizer = FlatAdam(self.policy, lr=learning_rate, eps=1e-5, max_grad_norm=max_grad_norm)
        else:
            self.optimizer = torch.optim.Adam(self.policy.parameters(), lr=learning_rate, eps=1e-5, fused=on_gpu,
                                              capturable=on_gpu and self.use_hip_graph)
        # mixed-precision learner (library-GEMM path only): bf16 MFMA GEMMs and activations against bf16 shadows of the fp32
        # master weights, fp32 loss / gradients-arena / Adam.  fp32 (the reference's dtype) is the default.
        self.mlp_dtype = mlp_dtype
        self._wide_ok = False
        if mlp_dtype == torch.bfloat16:
            if not self.flat_adam:
                raise ValueError("mlp_dtype=bfloat16 needs the flat Adam path (GPU)")
            self.fused_mlp = False                     # the fused [256,128]-class kernel is fp32
            self._wide_ok = bool(fused_wide) and self.fused_loss and WideMlpGrad.supported(self.policy, self.batch_size)
            if not self._wide_ok:
                self.optimizer.enable_bf16_shadow()    # library path: bf16 shadow weights for hipBLASLt
        self._graph = None
        self._mlp_grads = {}                                                        # FusedMlpGrad per minibatch size
        self._loss_acc = torch.zeros(2, device=self.device) if on_gpu else None     # device-side (sum of losses, count)
        self.grad_sync = self.optimizer if self.flat_adam else FlatGradAllReduce(self.policy.parameters())
        if dist.is_initialized():  # decorrelate action noise across ranks after the common init
            torch.manual_seed(seed + 1000 * (self.rank + 1))
        lo = torch.as_tensor(env.action_space.low, device=self.device) if env is not None else None
        hi = torch.as_tensor(env.action_space.high, device=self.device) if env is not None else None
        self.act_lo, self.act_hi = lo, hi
        self.num_timesteps = 0
        self._last_obs = None
        self.stats = {}

    # ------------------------------------------------------------------ rollout
    # ---- rollout-side fused kernels (csrc/dm_ppo.hip): policy head -> sampled / clamped action + logp, and the
    # per-step stores into the rollout buffer, two launches instead of ~20 small PyTorch kernels per env step
    def _fused_rollout_ok(self):
        return (self.fused_rollout and self.device.type == "cuda" and self.buffer_dtype == torch.float32)

    def _rollout_scratch(self, key):
        sc = getattr(self, "_rsc", {})
        if key not in sc:
            n = key[0]
            z = lambda *shape: torch.zeros(*shape, device=self.device)
            sc[key] = dict(act=z(n, self.act_dim), act_env=z(n, self.act_dim), logp=z(n))
            self._rsc = sc
        if getattr(self, "_rctrs", None) is None:
            # draw counters, advanced on the device: ONE PER SUB-BATCH — sub-batch chains run on their own streams (or as
            # independent branches of a captured graph), so a shared counter bumped by one chain would be read by the
            # others at unordered times (same noise at consecutive steps, non-reproducible rollouts)
            self._rctrs = torch.zeros(max(16, int(getattr(self.env, "sub_batches", 1))), dtype=torch.int32, device=self.device)
            self._rctr = self._rctrs[0:1]
        return sc[key]

    def _ctr(self, k):
        return self._rctrs[k:k + 1]

    def _policy_step_fused(self, obs, env_index=0):
        """mean/value by the MLP (library GEMMs), then one launch for sample + logp + clamp."""
        import ctypes as C
        from . import _lib
        n = obs.shape[0]
        sc = self._rollout_scratch((n, env_index))
        mean = self.policy.action_net(self.policy.pi(obs))
        val = self.policy.value_net(self.policy.vf(obs)).squeeze(-1).contiguous()
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = _lib.load_library().dm_policy_sample(p(mean.contiguous()), p(self.policy.log_std), n, self.act_dim,
                                                  C.c_uint64(self._rollout_seed + 7919 * env_index), p(self._ctr(env_index)), p(self.act_lo),
                                                  p(self.act_hi), p(sc["act"]), p(sc["act_env"]), p(sc["logp"]),
                                                  C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_policy_sample failed (%d)" % rc)
        return sc, val

    def _store_fused(self, rb, t, sl, last, sc, val, out, bump, env_index=0):
        import ctypes as C
        from . import _lib
        p = lambda x: C.c_void_p(x.data_ptr())
        n = val.shape[0]
        rc = _lib.load_library().dm_rollout_store(
            n, self.obs_dim, self.act_dim, p(last[sl]), p(sc["act"]), p(val), p(sc["logp"]), p(out["rew"]), p(out["done"]), p(out["obs"]),
            p(rb["obs"][t, sl]), p(rb["act"][t, sl]), p(rb["val"][t, sl]), p(rb["logp"][t, sl]), p(rb["rew"][t, sl]),
            p(rb["done"][t, sl]), p(last[sl]), p(self._ctr(env_index)) if bump else None,
            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_rollout_store failed (%d)" % rc)

    # ---- one-launch policy side (csrc/dm_policy.hip): per env step dm_policy_forward + dm_step, nothing else
    def _fused_policy_ok(self):
        return (self.fused_policy and self._fused_rollout_ok() and self.env is not None
                and (hasattr(self.env, "engines") or hasattr(self.env, "engine"))
                and FusedPolicyForward.supported(self.policy, self.device))

    def _rollout_fused_policy(self):
        """Rollout with two host calls per (sub-batch) step: ``dm_policy_forward`` reads the observations in place and
        writes action / value / log-prob / observation copy straight into row t of the rollout buffer, ``dm_step``
        writes reward and done flag into row t and the next observation over the one just consumed.  With
        ``env.sub_batches`` > 1 every sub-batch runs on its own stream (the policy kernel of one fills the ramp-down of
        the other's step kernel), host-driven or, with ``rollout_graph``, as one captured hipGraph of the T steps."""
        env, T, N, dev = self.env, self.n_steps, self.n_envs, self.device
        K = getattr(env, "sub_batches", 1)
        engines = getattr(env, "engines", None) or [env.engine]
        st = getattr(self, "_fp", None)
        if st is None:
            z = lambda *shape, dt=torch.float32: torch.zeros(*shape, device=dev, dtype=dt)
            rb = dict(obs=z(T, N, self.obs_dim), act=z(T, N, self.act_dim), rew=z(T, N), done_u8=z(T, N, dt=torch.uint8), val=z(T, N),
                      logp=z(T, N))
            last = (env.reset_tensor() if self._last_obs is None else self._last_obs).clone()
            self._rollout_scratch((N, 0))
            st = self._fp = dict(rb=rb, last=last, fwd=FusedPolicyForward(self.policy, dev), act_env=z(N, self.act_dim),
                                 streams=concurrent_streams(dev, K) if K > 1 else None, graph=None)
        rb, last, fwd = st["rb"], st["last"], st["fwd"]
        if self._last_obs is not None and self._last_obs.data_ptr() != last.data_ptr():
            last.copy_(self._last_obs)

        def sub_step(k, t):
            sl = env.sub_slices[k] if K > 1 else slice(0, N)
            fwd(last[sl], self._rollout_seed + 7919 * k, self._rctr, t, self.act_lo, self.act_hi, rb["act"][t, sl], st["act_env"][sl],
                rb["logp"][t, sl], rb["val"][t, sl], obs_copy=rb["obs"][t, sl])
            engines[k].step(st["act_env"][sl], dict(obs=last[sl], rew=rb["rew"][t, sl], done=rb["done_u8"][t, sl]))

        def whole():
            cur = torch.cuda.current_stream(dev)
            fwd.pack()
            if K == 1:
                for t in range(T):
                    sub_step(0, t)
            else:
                for s_ in st["streams"]:
                    s_.wait_stream(cur)
                for t in range(T):
                    for k in range(K):
                        with torch.cuda.stream(st["streams"][k]):
                            sub_step(k, t)
                for s_ in st["streams"]:
                    cur.wait_stream(s_)
            self._rctr += T
            rb["done"] = rb["done_u8"].float()
            last_val = self.policy.predict_values(last)
            rb["adv"], rb["ret"] = compute_gae(rb["rew"], rb["val"], rb["done"], last_val, self.gamma, self.gae_lambda)

        with torch.no_grad():
            if self.rollout_graph and K > 1:
                if st["graph"] is None:
                    side = torch.cuda.Stream(device=dev)        # warm-up off the default stream (real but uncounted env steps)
                    side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        fwd.pack()
                        for k in range(K):
                            sub_step(k, 0)
                        self.policy.predict_values(last)
                    torch.cuda.current_stream(dev).wait_stream(side)
                    torch.cuda.synchronize(dev)
                    st["graph"] = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(st["graph"], **(dict(capture_error_mode="thread_local") if dist.is_initialized() else {})):
                        whole()
                st["graph"].replay()
            else:
                whole()
        self._last_obs = last
        self.num_timesteps += T * N
        self.stats["mean_reward"] = float(rb["rew"].mean())
        self.stats["done_rate"] = float(rb["done"].mean())
        return {k: v for k, v in rb.items() if k != "done_u8"}

    def _rollout_graph_build(self):
        """Capture the whole T-step rollout as ONE hipGraph with one chain per env sub-batch (own stream each): the
        policy forward of one half overlaps the step kernel of the other, so the ramp-down of every launch is filled
        (INTEGRATION.md "double-buffered halves"), and the ~35 launches per step cost no host time at replay."""
        env, T, N, dev, bd = self.env, self.n_steps, self.n_envs, self.device, self.buffer_dtype
        K = env.sub_batches
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, device=dev, dtype=dt)
        rb = dict(obs=z(T, N, self.obs_dim, dt=bd), act=z(T, N, self.act_dim, dt=bd), rew=z(T, N), done=z(T, N), val=z(T, N), logp=z(T, N))
        last = env.reset_tensor().clone() if self._last_obs is None else self._last_obs.clone()
        streams = concurrent_streams(dev, K)

        def chain(k, steps):
            sl = env.sub_slices[k]
            for t in range(steps):
                obs = last[sl]
                if self._fused_rollout_ok():
                    sc, val = self._policy_step_fused(obs, env_index=k)
                    out = env.step_sub(k, sc["act_env"])
                    self._store_fused(rb, t, sl, last, sc, val, out, bump=True, env_index=k)
                    continue
                act, val, logp = self.policy(obs)
                rb["obs"][t, sl] = obs
                rb["act"][t, sl] = act
                rb["val"][t, sl] = val
                rb["logp"][t, sl] = logp
                out = env.step_sub(k, torch.clamp(act, self.act_lo, self.act_hi))
                rb["rew"][t, sl] = out["rew"]
                rb["done"][t, sl] = out["done"].float()
                last[sl].copy_(out["obs"])

        with torch.no_grad():
            # warm-up on the side streams (library workspaces, allocator pools); these env steps are real but uncounted
            cur = torch.cuda.current_stream(dev)
            for k in range(K):
                streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    chain(k, 1)
            for k in range(K):
                cur.wait_stream(streams[k])
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **(dict(capture_error_mode="thread_local") if dist.is_initialized() else {})):
                cap = torch.cuda.current_stream(dev)
                for k in range(K):
                    streams[k].wait_stream(cap)
                    with torch.cuda.stream(streams[k]):
                        chain(k, T)
                for k in range(K):
                    cap.wait_stream(streams[k])
                last_val = self.policy.predict_values(last)
                adv, ret = compute_gae(rb["rew"], rb["val"], rb["done"], last_val, self.gamma, self.gae_lambda)
        rb["adv"], rb["ret"] = adv, ret
        self._rollout = (g, rb, last)

    def _rollout_pipelined_eager(self):
        """sub_batches > 1 without graph capture: the host issues step t of every sub-batch on that sub-batch's stream,
        so the policy kernels of one overlap the step kernel of the other (14 launches per sub-batch step)."""
        env, T, N, dev = self.env, self.n_steps, self.n_envs, self.device
        K = env.sub_batches
        if getattr(self, "_pipe", None) is None:
            z = lambda *shape: torch.zeros(*shape, device=dev)
            rb = dict(obs=z(T, N, self.obs_dim), act=z(T, N, self.act_dim), rew=z(T, N), done=z(T, N), val=z(T, N), logp=z(T, N))
            last = env.reset_tensor().clone() if self._last_obs is None else self._last_obs.clone()
            self._pipe = (rb, last, concurrent_streams(dev, K))
        rb, last, streams = self._pipe
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for st in streams:
                st.wait_stream(cur)
            for t in range(T):
                for k in range(K):
                    with torch.cuda.stream(streams[k]):
                        sl = env.sub_slices[k]
                        sc, val = self._policy_step_fused(last[sl], env_index=k)
                        out = env.step_sub(k, sc["act_env"])
                        self._store_fused(rb, t, sl, last, sc, val, out, bump=True, env_index=k)
            for st in streams:
                cur.wait_stream(st)
            last_val = self.policy.predict_values(last)
            rb["adv"], rb["ret"] = compute_gae(rb["rew"], rb["val"], rb["done"], last_val, self.gamma, self.gae_lambda)
        self._last_obs = last
        self.num_timesteps += T * N
        self.stats["mean_reward"] = float(rb["rew"].mean())
        self.stats["done_rate"] = float(rb["done"].mean())
        return rb

    def collect_rollouts(self):
        if self._fused_policy_ok():
            return self._rollout_fused_policy()
        if (not self.rollout_graph and self._fused_rollout_ok() and getattr(self.env, "sub_batches", 1) > 1
                and hasattr(self.env, "step_sub")):
            return self._rollout_pipelined_eager()
        if (self.rollout_graph and self.device.type == "cuda" and getattr(self.env, "sub_batches", 1) > 1
                and hasattr(self.env, "step_sub")):
            if getattr(self, "_rollout", None) is None:
                self._rollout_graph_build()
            g, rb, last = self._rollout
            g.replay()
            self._last_obs = last
            self.num_timesteps += self.n_steps * self.n_envs
            self.stats["mean_reward"] = float(rb["rew"].mean())
            self.stats["done_rate"] = float(rb["done"].mean())
            return rb
        T, N, dev = self.n_steps, self.n_envs, self.device
        bd = self.buffer_dtype
        buf = dict(obs=torch.zeros(T, N, self.obs_dim, device=dev, dtype=bd), act=torch.zeros(T, N, self.act_dim, device=dev, dtype=bd),
                   rew=torch.zeros(T, N, device=dev), done=torch.zeros(T, N, device=dev),
                   val=torch.zeros(T, N, device=dev), logp=torch.zeros(T, N, device=dev))
        if self._last_obs is None:
            self._last_obs = self.env.reset_tensor().clone()
        ep_done = 0
        with torch.no_grad():
            fused = self._fused_rollout_ok() and self._last_obs.is_contiguous()
            full = slice(0, N)
            for t in range(T):
                obs = self._last_obs
                if fused:
                    sc, val = self._policy_step_fused(obs)
                    out = self.env.step_tensor(sc["act_env"])
                    self._store_fused(buf, t, full, self._last_obs, sc, val, out, bump=True)
                    continue
                act, val, logp = self.policy(obs)
                out = self.env.step_tensor(torch.clamp(act, self.act_lo, self.act_hi))
                buf["obs"][t] = obs
                buf["act"][t] = act
                buf["val"][t] = val
                buf["logp"][t] = logp
                buf["rew"][t] = out["rew"]
                buf["done"][t] = out["done"].float()
                self._last_obs = out["obs"].clone()
            last_val = self.policy.predict_values(self._last_obs)
            adv, ret = compute_gae(buf["rew"], buf["val"], buf["done"], last_val, self.gamma, self.gae_lambda)
        buf["adv"], buf["ret"] = adv, ret
        self.num_timesteps += T * N
        self.stats["mean_reward"] = float(buf["rew"].mean())
        self.stats["done_rate"] = float(buf["done"].mean())
        return buf

    # ------------------------------------------------------------------ update
    def train(self, buf, generator=None):
        """``n_epochs`` passes over the flattened rollout in minibatches of ``batch_size`` (SB3 PPO.train)."""
        flat = {k: v.reshape(-1, *v.shape[2:]) for k, v in buf.items()}
        n = flat["obs"].shape[0]
        loss_sum = torch.zeros((), device=self.device)
        nsteps = 0
        on_dev = (self.flat_adam and self.fused_loss and self.device.type == "cuda" and n % self.batch_size == 0
                  and ((self.fused_mlp and FusedMlpGrad.supported(self.policy, self.batch_size)) or self._wide_ok))
        self._on_dev = on_dev
        if on_dev:
            self._loss_acc.zero_()
        if self._epoch_graph_ok(flat, n):
            # one replay per epoch: the 32 x (gather + optimizer step) of an epoch are one captured hipGraph reading the
            # permutation from a static buffer, so the host issues two calls per epoch instead of two per minibatch
            eg = self._epoch_graph(flat, n)
            for k in eg["flat"]:
                eg["flat"][k].copy_(flat[k])
            eg["loss"].zero_()
            for _ in range(self.n_epochs):
                torch.randperm(n, device=self.device, generator=generator, out=eg["perm"])
                eg["graph"].replay()
            nsteps = self.n_epochs * (n // self.batch_size)
            acc = self._loss_acc if on_dev else torch.stack([eg["loss"], torch.full((), float(nsteps), device=self.device)])
            self.stats["loss"] = float(acc[0] / torch.clamp(acc[1], min=1.0))
            return self.stats["loss"]
        if self._dist_graph_ok(flat, n):
            # several ranks: a minibatch is [graph A: gather + minibatch gradient (dm_ppo_mlp_grad, or for nets beyond its class
            # the library-GEMM forward / loss / backward incl. dm_linear_tanh, dm_tanh_linear_wgrad, dm_ppo_loss)] -> all-reduce of
            # the flat gradient (the ONE collective, eager, on the same stream) -> [graph B: dm_flat_adam_step / _update with
            # grad_scale = 1 / world]: three host calls and no eager kernel launches, instead of the un-captured step
            dg = self._dist_graphs(flat, n)
            for k in dg["flat"]:
                dg["flat"][k].copy_(flat[k])
            dg["loss"].zero_()
            B = self.batch_size
            for _ in range(self.n_epochs):
                torch.randperm(n, device=self.device, generator=generator, out=dg["perm"])
                for s in range(0, n, B):
                    dg["idx"].copy_(dg["perm"][s:s + B])
                    dg["grad"].replay()
                    self.optimizer.all_reduce()
                    dg["adam"].replay()
                    nsteps += 1
            acc = self._loss_acc if on_dev else torch.stack([dg["loss"], torch.full((), float(nsteps), device=self.device)])
            self.stats["loss"] = float(acc[0] / torch.clamp(acc[1], min=1.0))
            return self.stats["loss"]
        for _ in range(self.n_epochs):
            perm = torch.randperm(n, device=self.device, generator=generator)
            for s in range(0, n, self.batch_size):
                idx = perm[s:s + self.batch_size]
                if self.use_hip_graph and len(idx) == self.batch_size:
                    loss = self._graph_step(flat, idx)
                else:
                    loss = self._minibatch_step(flat["obs"][idx].float(), flat["act"][idx].float(), flat["adv"][idx],
                                                flat["ret"][idx], flat["logp"][idx])
                if not on_dev:
                    loss_sum += loss
                nsteps += 1
        if on_dev:      # dm_ppo_mlp_grad kept the sum on the device (loss_acc fold)
            self.stats["loss"] = float(self._loss_acc[0] / torch.clamp(self._loss_acc[1], min=1.0))
        else:
            self.stats["loss"] = float(loss_sum / max(nsteps, 1))   # one host sync per train() call
        return self.stats["loss"]

    def _loss_torch(self, obs, act, adv, ret, old_logp):
        """The loss of SB3 PPO.train written with PyTorch ops (CPU tests, and the reference for the fused kernel)."""
        if self.normalize_advantage and obs.shape[0] > 1:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        value, logp, entropy = self.policy.evaluate_actions(obs, act)
        ratio = torch.exp(logp - old_logp)
        pg = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - self.clip_range, 1 + self.clip_range)).mean()
        vl = torch.nn.functional.mse_loss(ret, value)
        return pg + self.vf_coef * vl - self.ent_coef * entropy.mean()

    def _trunks(self, obs):
        # The policy and value trunks are independent until the loss: the value trunk runs on a second stream, forward
        # and (autograd replays each node on its forward stream) backward, so the two chains of small launch-bound
        # kernels overlap — also as parallel branches of the captured hipGraph.
        cur = torch.cuda.current_stream(obs.device)
        if self.two_stream:
            if getattr(self, "_vf_stream", None) is None:
                self._vf_stream = torch.cuda.Stream(device=obs.device)
            self._vf_stream.wait_stream(cur)
            with torch.cuda.stream(self._vf_stream):
                value = self.policy.value_net(_fused_tanh_trunk(self.policy.vf, obs)).squeeze(-1)
            mean = self.policy.action_net(_fused_tanh_trunk(self.policy.pi, obs))
            cur.wait_stream(self._vf_stream)
        else:
            mean = self.policy.action_net(_fused_tanh_trunk(self.policy.pi, obs))
            value = self.policy.value_net(_fused_tanh_trunk(self.policy.vf, obs)).squeeze(-1)
        if mean.dtype != torch.float32:               # bf16 learner: the loss kernel reads fp32 heads
            mean, value = mean.float(), value.float()
        return mean, value

    def _loss_fused(self, obs, act, adv, ret, old_logp):
        mean, value = self._trunks(obs)
        return FusedPPOLoss.apply(mean, self.policy.log_std, value, act, old_logp, adv, ret, self.clip_range, self.vf_coef,
                                  self.ent_coef, self.normalize_advantage and obs.shape[0] > 1)

    def _minibatch_grad(self, obs, act, adv, ret, old_logp):
        """Gradient half of an optimizer step on the flat-Adam paths: leaves the minibatch gradient in ``optimizer.flat_g``
        (the operand of the ONE collective of the data-parallel learner).  Returns (loss, begin): begin is False when Adam's
        begin launch was folded into the gradient launches (dm_ppo_mlp_grad)."""
        if (self.fused_mlp and self.fused_loss and obs.is_cuda and FusedMlpGrad.supported(self.policy, obs.shape[0])):
            # the whole minibatch gradient in three launches, written into the flat arena
            # (that launch sequence also clears the arena, performs Adam's begin and adds the loss to a device-side sum)
            mg = self._mlp_grads.get(obs.shape[0])
            if mg is None:
                mg = self._mlp_grads[obs.shape[0]] = FusedMlpGrad(self.policy, self.optimizer, obs.shape[0], loss_acc=self._loss_acc)
            loss = mg(obs.contiguous(), act.contiguous(), adv.contiguous(), ret.contiguous(), old_logp.contiguous(), self.clip_range,
                      self.vf_coef, self.ent_coef, self.normalize_advantage and obs.shape[0] > 1)
            return loss, False
        if self._wide_ok and obs.is_cuda and obs.shape[0] == self.batch_size:
            # wide nets with bf16 matrix-pipe products: the fused chain of dm_ppo_wide_grad + six library weight-gradient GEMMs
            wg = self._mlp_grads.get(("wide", obs.shape[0]))
            if wg is None:
                wg = self._mlp_grads[("wide", obs.shape[0])] = WideMlpGrad(self.policy, self.optimizer, obs.shape[0], self._loss_acc)
            loss = wg(obs.contiguous(), act.contiguous(), adv.contiguous(), ret.contiguous(), old_logp.contiguous(), self.clip_range,
                      self.vf_coef, self.ent_coef, self.normalize_advantage and obs.shape[0] > 1)
            return loss, False
        # library-GEMM learner (nets beyond the [256,128] class): every gradient lands in one flat buffer (one memset)
        self.optimizer.zero_grad()
        if self.fused_loss and obs.is_cuda:
            mean, value = self._trunks(obs)
            gls = self.optimizer.slices[[id(q) for q in self.optimizer.params].index(id(self.policy.log_std))]
            loss, gm, gv = FusedPPOLoss.raw(mean.contiguous(), self.policy.log_std, value.contiguous(), act, old_logp, adv, ret,
                                            self.clip_range, self.vf_coef, self.ent_coef,
                                            self.normalize_advantage and obs.shape[0] > 1, grad_log_std=gls)
            torch.autograd.backward([mean, value], [gm, gv])
        else:
            loss = (self._loss_fused if self.fused_loss else self._loss_torch)(obs, act, adv, ret, old_logp)
            loss.backward()
        if getattr(self, "_vf_stream", None) is not None:
            # the layer kernels of the value trunk hand no gradient tensor to autograd (they write flat_g), so the
            # engine has no leaf to synchronise on: join the second stream explicitly before the update
            torch.cuda.current_stream(obs.device).wait_stream(self._vf_stream)
        self.optimizer.gather_grads()
        return loss.detach(), True

    def _minibatch_step(self, obs, act, adv, ret, old_logp, gather_next=None):
        if self.flat_adam:
            # clipping + Adam are one fused update on the flat buffer; with several ranks that buffer is all-reduced in between
            loss, begin = self._minibatch_grad(obs, act, adv, ret, old_logp)
            self.optimizer.all_reduce()
            self.optimizer.step(begin=begin, gather_next=gather_next)
            return loss
        assert gather_next is None
        loss = (self._loss_fused if (self.fused_loss and obs.is_cuda) else self._loss_torch)(obs, act, adv, ret, old_logp)
        # grads are re-created by backward (no zero-fill, no accumulate-add per parameter); inside a captured
        # hipGraph they live in the graph's private pool, so their addresses are the same at every replay
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.grad_sync()                         # the ONE collective of the data-parallel learner
        nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return loss.detach()

    def _gather_minibatch(self, flat, idx, g):
        import ctypes as C
        from . import _lib
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = _lib.load_library().dm_ppo_gather(
            p(idx), int(idx.numel()), p(flat["obs"]), self.obs_dim, p(flat["act"]), flat["act"].shape[1], p(flat["adv"]),
            p(flat["ret"]), p(flat["logp"]), p(g["obs"]), p(g["act"]), p(g["adv"]), p(g["ret"]), p(g["logp"]),
            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("dm_ppo_gather failed (%d)" % rc)

    def _capture_with_restore(self, warm, body):
        """Run ``warm`` three times on a side stream, capture ``body`` into a hipGraph, then put parameters and optimizer
        state back: warm-up and capture must not change the model."""
        dev = self.device
        snap_p = [p.detach().clone() for p in self.policy.parameters()]
        snap_o = {k: {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                  for k, st in self.optimizer.state.items()}
        had_state = len(self.optimizer.state) > 0
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                warm()
        torch.cuda.current_stream(dev).wait_stream(side)
        # with a process group alive, its watchdog thread queries events while we capture: "global" capture mode would
        # fail the capture on that foreign call, "thread_local" only polices this thread
        mode = dict(capture_error_mode="thread_local") if dist.is_initialized() else {}
        if isinstance(body, (list, tuple)):      # several graphs captured back to back (e.g. around a collective)
            graph, out = [], []
            for b in body:
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_, **mode):
                    out.append(b())
                graph.append(g_)
        else:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, **mode):
                out = body()
        with torch.no_grad():
            for p, q in zip(self.policy.parameters(), snap_p):
                p.copy_(q)
            for k, st in self.optimizer.state.items():
                for kk, vv in st.items():
                    if torch.is_tensor(vv):
                        vv.copy_(snap_o[k][kk]) if had_state and k in snap_o else vv.zero_()
            if self._loss_acc is not None:
                self._loss_acc.zero_()          # the warm-up steps are not part of the statistics
            if getattr(self.optimizer, "flat_pb", None) is not None:
                self.optimizer.flat_pb.copy_(self.optimizer.flat_p)
        return graph, out

    def _static_minibatch(self):
        B, dev = self.batch_size, self.device
        return dict(obs=torch.zeros(B, self.obs_dim, device=dev), act=torch.zeros(B, self.act_dim, device=dev), adv=torch.zeros(B, device=dev),
                    ret=torch.zeros(B, device=dev), logp=torch.zeros(B, device=dev))

    def _epoch_graph_ok(self, flat, n):
        return (self.use_hip_graph and self.device.type == "cuda" and n % self.batch_size == 0 and self.epoch_graph
                and all(flat[k].dtype == torch.float32 for k in ("obs", "act", "adv", "ret", "logp")))

    def _dist_graph_ok(self, flat, n):
        return (self.dist_graph and self.flat_adam and self.device.type == "cuda" and n % self.batch_size == 0
                and dist.is_initialized() and dist.get_world_size() > 1
                and all(flat[k].dtype == torch.float32 for k in ("obs", "act", "adv", "ret", "logp")))

    def _dist_graphs(self, flat, n):
        """Two captured graphs per minibatch for the multi-rank learner (see train())."""
        dg = getattr(self, "_dg", None)
        if dg is not None and dg["n"] == n:
            return dg
        dev, B = self.device, self.batch_size
        dg = dict(n=n, perm=torch.arange(n, device=dev), idx=torch.arange(B, device=dev), loss=torch.zeros((), device=dev), begin=True,
                  flat={k: torch.zeros_like(flat[k]) for k in ("obs", "act", "adv", "ret", "logp")})
        gin = self._static_minibatch()
        kw = {("old_logp" if k == "logp" else k): v for k, v in gin.items()}

        def grad():
            self._gather_minibatch(dg["flat"], dg["idx"], gin)
            loss, dg["begin"] = self._minibatch_grad(**kw)
            if not self._on_dev:
                dg["loss"].add_(loss)

        def adam():
            self.optimizer.step(begin=dg["begin"])

        def warm():              # local only: no collective during warm-up / capture, every rank does the same
            grad()
            adam()

        (dg["grad"], dg["adam"]), _ = self._capture_with_restore(warm, [grad, adam])
        self._dg = dg
        return dg

    def _epoch_graph(self, flat, n):
        eg = getattr(self, "_eg", None)
        if eg is not None and eg["n"] == n:
            return eg
        dev, B = self.device, self.batch_size
        eg = dict(n=n, perm=torch.arange(n, device=dev), loss=torch.zeros((), device=dev),
                  flat={k: torch.zeros_like(flat[k]) for k in ("obs", "act", "adv", "ret", "logp")})
        gin = self._static_minibatch()
        kw = {("old_logp" if k == "logp" else k): v for k, v in gin.items()}

        def warm():
            self._gather_minibatch(eg["flat"], eg["perm"][:B], gin)
            self._minibatch_step(**kw)

        def body():
            # one gather launch for the first minibatch; every later one rides on the previous optimizer step's norm launch (it
            # depends on nothing that step computes, and the static minibatch was last read by that step's gradient launches)
            ride = self.flat_adam
            self._gather_minibatch(eg["flat"], eg["perm"][:B], gin)
            for s in range(0, n, B):
                nxt = (eg["flat"], eg["perm"][s + B:s + 2 * B], gin) if (ride and s + B < n) else None
                loss = self._minibatch_step(gather_next=nxt, **kw)
                if not ride and s + B < n:
                    self._gather_minibatch(eg["flat"], eg["perm"][s + B:s + 2 * B], gin)
                if not self._on_dev:
                    eg["loss"].add_(loss)

        eg["graph"], _ = self._capture_with_restore(warm, body)
        self._eg = eg
        return eg

    def _graph_step(self, flat, idx):
        """Replay the captured optimizer step on static input buffers (capture on first use)."""
        if self._graph is None:
            self._gin = self._static_minibatch()
            kw = {("old_logp" if k == "logp" else k): v for k, v in self._gin.items()}
            self._graph, self._gloss = self._capture_with_restore(lambda: self._minibatch_step(**kw), lambda: self._minibatch_step(**kw))
        g = self._gin
        if flat["obs"].dtype == torch.float32 and flat["act"].dtype == torch.float32 and idx.dtype == torch.int64:
            self._gather_minibatch(flat, idx, g)
        else:  # bf16 rollout buffers (config 5): gather + widen with PyTorch ops
            g["obs"].copy_(flat["obs"][idx]); g["act"].copy_(flat["act"][idx])
            torch.index_select(flat["adv"], 0, idx, out=g["adv"])
            torch.index_select(flat["ret"], 0, idx, out=g["ret"])
            torch.index_select(flat["logp"], 0, idx, out=g["logp"])
        self._graph.replay()
        return self._gloss

    def learn(self, total_timesteps, log_interval=1, callback=None):
        it = 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        while self.num_timesteps * world < total_timesteps:
            t0 = time.perf_counter()
            buf = self.collect_rollouts()
            t1 = time.perf_counter()
            self.train(buf)
            t2 = time.perf_counter()
            it += 1
            self.stats.update(rollout_s=t1 - t0, train_s=t2 - t1, iteration=it)
            if callback is not None:
                callback(self)
            if self.rank == 0 and log_interval and it % log_interval == 0:
                print("iter %d  steps %d  rew/step %.4f  done %.4f  loss %.4f  rollout %.2fs  train %.2fs" % (
                    it, self.num_timesteps * world, self.stats["mean_reward"], self.stats["done_rate"],
                    self.stats["loss"], t1 - t0, t2 - t1), flush=True)
        return self

    def predict(self, obs, deterministic=True):
        with torch.no_grad():
            o = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
            act, _, _ = self.policy(o, deterministic=deterministic)
            return torch.clamp(act, self.act_lo, self.act_hi)

    def save(self, path):
        torch.save({"policy": self.policy.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "num_timesteps": self.num_timesteps}, path)

    def load(self, path):
        ck = torch.load(path, map_location=self.device)
        self.policy.load_state_dict(ck["policy"])
        self.optimizer.load_state_dict(ck["optimizer"])
        if getattr(self.optimizer, "flat_pb", None) is not None:
            self.optimizer.flat_pb.copy_(self.optimizer.flat_p)
        self.num_timesteps = ck["num_timesteps"]
        return self
