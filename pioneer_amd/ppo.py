"""Thin PyTorch-ROCm PPO host driver.

Replaces what RLlib's PPOTrainer + rollout workers do for pioneer/launch/pioneer_knm_train.py:
the sampling loop calls ``env.vector_step`` with device-resident tensors (no Ray object store,
no host copies), the learner is data-parallel over GPUs with ONE flat gradient all-reduce per
minibatch (RCCL over xGMI).  Hyper-parameter names and defaults follow the reference's config
dict (pioneer_knm_train.py:45-67) and the RLlib-0.8.x PPO defaults it did not override
(SURVEY.md Appendix D).  Works on CPU tensors too (gloo tests use synthetic rollouts).
"""
import math
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import dist as pdist

# Output rows of the two head GEMMs (12 and 1 wide) are padded to a multiple of this (zero rows, sliced off).
# PNR_PPO_HEAD_PAD=1 runs them at their natural widths: only tools/graph_nan_repro.py sets it.
import os as _os
_HEAD_PAD = max(1, int(_os.environ.get("PNR_PPO_HEAD_PAD", "16")))


@dataclass
class PPOConfig:
    # model (pioneer_knm_train.py:59-61; RLlib defaults: tanh, vf_share_layers False)
    fcnet_hiddens: Sequence[int] = (256, 256)
    obs_dim: int = 137
    act_dim: int = 6
    # optimisation (pioneer_knm_train.py:62-65)
    lr: float = 2e-5
    num_sgd_iter: int = 20
    sgd_minibatch_size: int = 128
    train_batch_size: int = 8000          # informational: T * num_envs * world replaces it here
    rollout_fragment_length: int = 32     # T steps per env per iteration
    # PPO defaults of the era (Appendix D)
    gamma: float = 0.99
    lambda_: float = 1.0
    clip_param: float = 0.3
    kl_coeff: float = 0.2
    kl_target: float = 0.01
    vf_loss_coeff: float = 1.0
    vf_clip_param: float = 10.0
    clip_actions: bool = True
    grad_clip: Optional[float] = None
    # entropy_coeff_schedule [(0, x), (decay_steps, 0)] (pioneer_knm_train.py:32-41, :63)
    entropy_coeff_start: float = 1e-2
    entropy_decay_steps: int = 1_000_000
    # 'observation_filter': 'ConcurrentMeanStdFilter' (pioneer_knm_train.py:66)
    observation_filter: str = "MeanStdFilter"
    filter_clip: float = 10.0
    seed: int = 0
    # engine options (no reference counterpart)
    amp_bf16: bool = False                # run the MLP GEMMs in bf16 (MFMA) under autocast; losses stay fp32


def sample_entropy_start(rng: np.random.RandomState, min_start: float = 1e-3, max_start: float = 1e-1,
                         base: float = 10.0) -> float:
    """entropy_coeff_schedule's log-uniform start value (pioneer_knm_train.py:32-41)."""
    logmin = np.log(min_start) / np.log(base)
    logmax = np.log(max_start) / np.log(base)
    return float(base ** rng.uniform(logmin, logmax))


class MeanStdFilter:
    """Running mean/std observation normaliser with RLlib MeanStdFilter semantics
    ((x - mean) / (std + 1e-8), clipped), kept on the device; per-iteration deltas are
    all-reduced so every rank holds the same statistics (ConcurrentMeanStdFilter's role).

    Moments are accumulated as SHIFTED sums about a pivot row (the first sample of the pending delta):
    a constant feature (36 of the 137 obs entries are the joint limits and their cos / sin) then has
    exactly zero deviation sums, so its mean is the constant, its variance exactly 0 and its filtered value
    exactly 0 — what RLlib's float64 RunningStat gives.  Plain float32 column sums of x and x^2 left such
    a column with mean off by ~1e-6 and m2 <= 0, i.e. a filtered value of +-clip that depended on the
    summation order."""

    def __init__(self, dim: int, device, clip: float = 10.0):
        self.n = torch.zeros((), dtype=torch.float64, device=device)
        self.mean = torch.zeros(dim, dtype=torch.float64, device=device)
        self.m2 = torch.zeros(dim, dtype=torch.float64, device=device)
        self.clip = clip
        self._dn = torch.zeros((), dtype=torch.float64, device=device)
        self._dsum = torch.zeros(dim, dtype=torch.float64, device=device)     # sum of (x - pivot)
        self._dsq = torch.zeros(dim, dtype=torch.float64, device=device)      # sum of (x - pivot)^2
        self._pivot = torch.zeros(dim, dtype=torch.float32, device=device)
        self._pending = 0       # observe() calls since the last sync(): the first one sets the pivot

    def observe(self, x: torch.Tensor) -> None:
        """Accumulate a batch [..., dim] into the pending delta.  Partial sums of the deviations from the
        pivot run in float32 over chunks of <= 16 384 rows, the accumulation across chunks and calls in
        float64.  (Which call is the first after a sync() is host-side state: a captured hipGraph replays
        the pattern it was captured with, i.e. collect -> sync -> collect.)"""
        x = x.reshape(-1, x.shape[-1])
        if self._pending == 0:
            self._pivot.copy_(x[0])
        self._pending += 1
        m = x.shape[0]
        if x.is_cuda and x.shape[-1] == 137 and x.dtype == torch.float32 and x.is_contiguous():
            # one read of the buffer (pnr_filter_moments): float32 partial sums over 512 rows, float64 across them
            import ctypes
            from . import _lib
            lib = _lib.load_library()
            need = int(lib.pnr_filter_moments_scratch(m))
            if getattr(self, "_fm_scratch", None) is None or self._fm_scratch.numel() < need:
                self._fm_scratch = torch.empty(need, dtype=torch.float32, device=x.device)
            _lib.check(lib.pnr_filter_moments(m, _dp(x), _dp(self._pivot), _dp(self._fm_scratch), self._fm_scratch.numel(), _dp(self._dsum),
                                              _dp(self._dsq), _dp(self._dn), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
            return
        chunk = m
        for c in (16384, 8192, 4096, 2048, 1024):
            if m % c == 0:
                chunk = c
                break
        d = (x - self._pivot).view(m // chunk, chunk, -1)
        self._dn += m
        self._dsum += d.sum(1).double().sum(0)
        self._dsq += (d * d).sum(1).double().sum(0)

    def sync(self) -> None:
        """Merge the pending deltas of all ranks into the running statistics (Chan et al.).
        In place and without host synchronisation, so a captured hipGraph keeps seeing them."""
        d = self.mean.numel()
        if self.mean.is_cuda and d == 137 and not pdist.is_dist():
            # one rank: the same float64 operations in one launch (pnr_filter_merge), bit-identical to the formulation below
            import ctypes
            from . import _lib
            _lib.check(_lib.load_library().pnr_filter_merge(_dp(self._dn), _dp(self._dsum), _dp(self._dsq), _dp(self._pivot), _dp(self.n),
                                                            _dp(self.mean), _dp(self.m2),
                                                            ctypes.c_void_p(torch.cuda.current_stream(self.mean.device).cuda_stream)))
            self._pending = 0
            return
        dn_r = self._dn
        dn_r_safe = torch.clamp(dn_r, min=1.0)
        mean_r = self._pivot.double() + self._dsum / dn_r_safe
        m2_r = torch.clamp(self._dsq - self._dsum * self._dsum / dn_r_safe, min=0.0)
        # the ranks' (n, mean, m2) combined exactly: n = sum n_r, mean = sum n_r mean_r / n,
        # m2 = sum (m2_r + n_r (mean_r - mean)^2); two small all-reduces
        pack_a = torch.cat([dn_r.reshape(1), dn_r * mean_r])
        pdist.allreduce_sum_(pack_a)
        dn = pack_a[0]
        dn_safe = torch.clamp(dn, min=1.0)
        bmean = pack_a[1:1 + d] / dn_safe
        dev = mean_r - bmean
        bm2 = m2_r + dn_r * dev * dev
        pdist.allreduce_sum_(bm2)
        tot = self.n + dn
        tot_safe = torch.clamp(tot, min=1.0)
        delta = bmean - self.mean
        self.m2.add_(bm2 + delta * delta * (self.n * dn / tot_safe))
        self.mean.add_(delta * (dn / tot_safe))
        self.n.copy_(tot)
        self._dn.zero_(); self._dsum.zero_(); self._dsq.zero_()
        self._pending = 0

    @property
    def std(self) -> torch.Tensor:
        var = self.m2 / torch.clamp(self.n - 1, min=1.0)
        return torch.sqrt(torch.clamp(var, min=0.0))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        y = (x - self.mean.to(x.dtype)) / (self.std.to(x.dtype) + 1e-8)
        if self.clip:
            y = torch.clamp(y, -self.clip, self.clip)
        return torch.where(self.n < 2, x, y)      # identity until two samples exist (no host sync)

    # The statistics only change at sync(): the rollout prepares shift / scale / clip vectors once
    # (in place, float32) and applies them with three kernels per step instead of a dozen.
    def prepare(self) -> None:
        if not hasattr(self, "_loc"):
            d, dev = self.mean.numel(), self.mean.device
            self._loc = torch.zeros(d, device=dev); self._inv = torch.ones(d, device=dev)
            self._lo = torch.empty(d, device=dev); self._hi = torch.empty(d, device=dev)
        ident = self.n < 2                                   # identity until two samples exist (no host sync)
        clip = self.clip if self.clip else float("inf")
        self._loc.copy_(torch.where(ident, torch.zeros_like(self.mean), self.mean))
        self._inv.copy_(torch.where(ident, torch.ones_like(self.mean), 1.0 / (self.std + 1e-8)))
        self._hi.copy_(torch.where(ident, torch.full_like(self.mean, float("inf")), torch.full_like(self.mean, clip)))
        self._lo.copy_(-self._hi)

    def apply_(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out = clip((x - mean) / (std + 1e-8)) with the vectors of the last prepare()."""
        torch.sub(x, self._loc, out=out)
        out.mul_(self._inv)
        return torch.clamp(out, min=self._lo, max=self._hi, out=out)

    def state_dict(self):
        return {"n": self.n, "mean": self.mean, "m2": self.m2}

    def load_state_dict(self, sd):
        self.n.copy_(sd["n"]); self.mean.copy_(sd["mean"]); self.m2.copy_(sd["m2"])


class NoFilter:
    def observe(self, x): pass
    def sync(self): pass
    def prepare(self): pass
    def apply_(self, x, out): return out.copy_(x)
    def __call__(self, x): return x
    def state_dict(self): return {}
    def load_state_dict(self, sd): pass


def _mlp(sizes: Sequence[int], out_dim: int, out_gain: float) -> nn.Sequential:
    layers: List[nn.Module] = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        lin = nn.Linear(a, b)
        nn.init.orthogonal_(lin.weight, gain=math.sqrt(2)); nn.init.zeros_(lin.bias)
        layers += [lin, nn.Tanh()]
    head = nn.Linear(sizes[-1], out_dim)
    nn.init.orthogonal_(head.weight, gain=out_gain); nn.init.zeros_(head.bias)
    layers.append(head)
    return nn.Sequential(*layers)


_ONES_CACHE: Dict[Tuple, torch.Tensor] = {}


def _ones_rows(S: int, k: int, dtype, device) -> torch.Tensor:
    key = (S, k, dtype, str(device))
    if key not in _ONES_CACHE:
        _ONES_CACHE[key] = torch.ones((S, 8, k), dtype=dtype, device=device)
    return _ONES_CACHE[key]


class _LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with a split-K weight gradient.

    For the PPO minibatches dW = dy^T x has a tiny output (<= 256 x 256) and K = the minibatch size
    (131 072): the BLAS library runs it as <= 16 workgroups on a 256-CU chip (368 us per call, a third
    of the learn phase in rocprof).  Slicing the batch axis into S chunks turns it into one batched
    GEMM with S x more workgroups plus a small sum."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.bfloat16)   # casts only when autocast is on
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return nn.functional.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy.to(weight.dtype) @ weight if ctx.needs_input_grad[0] else None
        B = x.shape[0]
        S = 1
        # measured on MI355X at B = 131 072 (tools/gemm_probe.py): bf16 42 us at S = 64 (plain 360 us),
        # fp32 95 us at S = 32 (plain 320 us)
        for cand in ((64, 32, 16, 8, 4, 2) if dy.dtype == torch.bfloat16 else (32, 16, 8, 4, 2)):
            if B % cand == 0 and B // cand >= 512:
                S = cand
                break
        x2 = x.reshape(B, -1).to(dy.dtype)
        if S > 1:
            out = dy.shape[1]
            # a bf16 batched GEMM with a single output row (the value head) takes ~11 ms of HOST time per
            # call in the BLAS library (tools/gemm_probe2.py): pad tiny heads to 8 rows and slice
            dyp = nn.functional.pad(dy, (0, 8 - out)) if out < 8 else dy
            dw = torch.bmm(dyp.view(S, B // S, -1).transpose(1, 2), x2.view(S, B // S, -1)).sum(0, dtype=torch.float32)[:out]
        else:
            dw = (dy.t() @ x2).float()
        if S > 1 and dy.is_cuda:
            # the bias gradient as a GEMM too (eight rows of ones against the same slices): a torch column-sum over
            # B rows takes the multi-block reduction path, which returned wrong sums from the second hipGraph replay
            # on (tools/graph_reduce_probe.py); GEMMs replay exactly (profiles/r02_nan_repro_C_blas_backward.json)
            ones = _ones_rows(S, B // S, dy.dtype, dy.device)
            db = torch.bmm(ones, dy.view(S, B // S, -1)).sum(0, dtype=torch.float32)[0]
        else:
            db = dy.sum(0, dtype=torch.float32)      # one reduction with a float32 accumulator, no cast pass
        return dx, dw.to(weight.dtype), db.to(weight.dtype)


class ActorCritic(nn.Module):
    """Separate policy and value MLPs [137 -> 256 -> 256 -> 12 | 1], tanh (RLlib FullyConnectedNetwork
    with vf_share_layers False).  The policy head emits the Gaussian's 6 means and 6 log-stds:
    104 204 + 101 377 = 205 581 parameters."""

    def __init__(self, cfg: PPOConfig):
        super().__init__()
        sizes = [cfg.obs_dim, *cfg.fcnet_hiddens]
        self.policy = _mlp(sizes, 2 * cfg.act_dim, 0.01)
        self.value = _mlp(sizes, 1, 1.0)
        self.act_dim = cfg.act_dim

    def dist_params(self, obs):
        out = self.policy(obs)
        mean, log_std = out[..., :self.act_dim], out[..., self.act_dim:]
        return mean, torch.clamp(log_std, -20.0, 2.0)

    @staticmethod
    def _run(net: nn.Sequential, x_pad, pad: int, full: bool = False):
        """First layer on a K padded to a multiple of 16 (137 -> 144): the BLAS library's kernels for an
        unaligned K = 137 run at ~3 TFLOP/s (391 us for a 16384 x 137 x 256 GEMM); the zero columns are
        appended to input and weight on the fly, so the parameters are the reference's."""
        first = net[0]
        h = _LinearSplitK.apply(x_pad, nn.functional.pad(first.weight, (0, pad)), first.bias)
        layers = list(net)[1:]
        for layer in layers[:-1]:
            h = _LinearSplitK.apply(h, layer.weight, layer.bias) if isinstance(layer, nn.Linear) else layer(h)
        # the heads (12 and 1 output rows) run as 16-row GEMMs (zero rows): aligned kernels, and the fused loss
        # kernel consumes rows of 16.  (r01 blamed these widths for NaNs under graph replay; the cause was a torch
        # reduction inside the captured loop, see PPOTrainer._collect_tail.)
        head = layers[-1]
        rp = (-head.out_features) % _HEAD_PAD
        out = _LinearSplitK.apply(h, nn.functional.pad(head.weight, (0, 0, 0, rp)), nn.functional.pad(head.bias, (0, rp)))
        return out if full else out[:, :head.out_features]

    def forward_heads(self, obs, amp_bf16: bool = False):
        """The two nets' raw head outputs as float32 rows of 16 (policy: means 0..5, unclamped log-stds
        6..11; value: column 0) — the layout pnr_ppo_loss consumes."""
        pad = (-obs.shape[-1]) % 16
        x = nn.functional.pad(obs, (0, pad)) if pad else obs
        if amp_bf16 and obs.is_cuda:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                hp, hv = self._run(self.policy, x, pad, True), self._run(self.value, x, pad, True)
        else:
            hp, hv = self._run(self.policy, x, pad, True), self._run(self.value, x, pad, True)
        return hp.float(), hv.float()

    # -- no-grad inference on cached weights (the sampling loop) ----------------------------------
    # The rollout runs the nets T times on unchanged parameters: the K-padded, compute-dtype copies of the
    # weights are made once per rollout (in place, so a captured hipGraph keeps using them) instead of
    # ~14 pad / cast kernels per net and step.  Same GEMMs on the same values as forward().
    def refresh_inference_cache(self, amp_bf16: bool) -> None:
        dt = torch.bfloat16 if (amp_bf16 and next(self.parameters()).is_cuda) else torch.float32
        cache = getattr(self, "_icache", None)
        if cache is None or cache["dtype"] != dt:
            cache = {"dtype": dt, "nets": []}
            for net in (self.policy, self.value):
                lins = [l for l in net if isinstance(l, nn.Linear)]
                pad = (-lins[0].in_features) % 16
                dev = lins[0].weight.device
                # every cached matrix is [rows padded to 16, K padded to 16]: the zero rows / columns cost nothing
                # and keep the BLAS library on its aligned kernels (the 1-row value head is the odd one out)
                p16 = lambda n: n + (-n) % (_HEAD_PAD if n < 16 else 16)      # noqa: E731
                ws = [torch.zeros((p16(l.out_features), l.in_features + (pad if i == 0 else 0)), dtype=dt, device=dev)
                      for i, l in enumerate(lins)]
                bs = [torch.zeros(p16(l.out_features), dtype=dt, device=dev) for l in lins]
                cache["nets"].append((lins, ws, bs))
            self._icache = cache
        with torch.no_grad():
            for lins, ws, bs in cache["nets"]:
                for l, w, b in zip(lins, ws, bs):
                    w[:l.out_features, :l.in_features].copy_(l.weight)
                    b[:l.out_features].copy_(l.bias)

    @torch.no_grad()
    def forward_cached(self, x_pad):
        """x_pad: [N, 144] in the cache's dtype (zero pad columns).  Returns (head [N, 12], v [N, 1]) in that dtype."""
        outs = []
        for lins, ws, bs in self._icache["nets"]:
            h = x_pad
            for i, (w, b) in enumerate(zip(ws, bs)):
                h = nn.functional.linear(h, w, b)
                if i + 1 < len(ws):
                    h = torch.tanh(h)
            outs.append(h[:, :lins[-1].out_features])
        return outs[0], outs[1]

    def forward(self, obs, amp_bf16: bool = False):
        pad = (-obs.shape[-1]) % 16
        x = nn.functional.pad(obs, (0, pad)) if pad else obs
        if amp_bf16 and obs.is_cuda:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out, v = self._run(self.policy, x, pad), self._run(self.value, x, pad)
            out, v = out.float(), v.float()
        else:
            out, v = self._run(self.policy, x, pad), self._run(self.value, x, pad)
        mean, log_std = out[..., :self.act_dim], torch.clamp(out[..., self.act_dim:], -20.0, 2.0)
        return mean, log_std, v.squeeze(-1)


def gaussian_logp(x, mean, log_std):
    z = (x - mean) * torch.exp(-log_std)
    return (-0.5 * z * z - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)


def gaussian_entropy(log_std):
    return (log_std + 0.5 * math.log(2 * math.pi * math.e)).sum(-1)


def gaussian_kl(mean0, log_std0, mean1, log_std1):
    """KL(N0 || N1) for diagonal Gaussians."""
    var0, var1 = torch.exp(2 * log_std0), torch.exp(2 * log_std1)
    return (log_std1 - log_std0 + (var0 + (mean0 - mean1) ** 2) / (2 * var1) - 0.5).sum(-1)


class FusedPPOLoss(torch.autograd.Function):
    """PPOLearner.loss()'s element-wise part as ONE HIP kernel (pnr_ppo_loss, csrc/pnr_ppo.h): forward
    values and d loss / d head in the same launch instead of ~120 small kernels per minibatch."""

    @staticmethod
    def forward(ctx, head_p, head_v, mb, kl_c, ent_c, clip, vf_clip, vf_coeff):
        from . import _lib
        import ctypes as C
        lib = _lib.load_library()
        B = head_p.shape[0]
        head_p, head_v = head_p.contiguous(), head_v.contiguous()
        assert head_p.shape == (B, 16) and head_v.shape == (B, 16) and head_p.dtype == head_v.dtype == torch.float32
        t = {k: mb[k].contiguous() for k in ("actions", "logp", "mean", "log_std", "adv", "vtarg", "values")}
        assert all(v.dtype == torch.float32 and v.is_cuda and v.shape[0] == B for v in t.values())
        assert t["actions"].shape == t["mean"].shape == t["log_std"].shape == (B, 6)
        g_p, g_v = torch.empty_like(head_p), torch.empty_like(head_v)
        rows = (B + 255) // 256
        partials = torch.empty((rows, 8), dtype=torch.float32, device=head_p.device)
        means = torch.empty(8, dtype=torch.float32, device=head_p.device)   # policy_loss, vf_loss, kl, entropy, total
        P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
        _lib.check(lib.pnr_ppo_loss(B, None, P(head_p), P(head_v), P(t["actions"]), P(t["logp"]), P(t["mean"]), P(t["log_std"]),
                                    P(t["adv"]), P(t["vtarg"]), P(t["values"]), P(kl_c), P(ent_c),
                                    C.c_float(clip), C.c_float(vf_clip), C.c_float(vf_coeff), P(g_p), P(g_v), P(partials),
                                    rows, P(means), C.c_void_p(torch.cuda.current_stream(head_p.device).cuda_stream)))
        ctx.save_for_backward(g_p, g_v)
        ctx.mark_non_differentiable(means)
        return means[4].clone(), means

    @staticmethod
    def backward(ctx, g_total, _g_means):
        g_p, g_v = ctx.saved_tensors
        return g_p * g_total, g_v * g_total, None, None, None, None, None, None


def compute_gae(rewards, values, last_value, terminals, gamma, lam):
    """rewards/values/terminals [T, N]; terminals = done | truncated (RLlib of that era treats the
    TimeLimit cut as terminal).  Returns advantages, value targets [T, N]."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    nxt_v, nxt_a = last_value, torch.zeros_like(last_value)
    for t in range(T - 1, -1, -1):
        live = 1.0 - terminals[t]
        delta = rewards[t] + gamma * nxt_v * live - values[t]
        nxt_a = delta + gamma * lam * live * nxt_a
        adv[t] = nxt_a
        nxt_v = values[t]
    return adv, adv + values


def _dp(t):
    import ctypes
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def hip_gae_logp(reward, values, last_value, done, trunc, actions, mean, log_std, gamma, lam, *, logp, adv, vtarg, terminals=None,
                 stats=None, adv_stats=None):
    """compute_gae + gaussian_logp of a [T, N] rollout in ONE launch (pnr_ppo_gae) on the current stream; outputs are
    written in place.  Advantages bit-identical to compute_gae (same float32 operations in the same order).  With
    ``stats`` (an EpisodeStats) and ``adv_stats`` (float64 [3]) the same launch also does the rollout's bookkeeping: the
    episode statistics (EpisodeStats.step()'s arithmetic) and the advantages' sum / sum of squares / count."""
    import ctypes
    from . import _lib
    T, N = reward.shape
    f32 = [reward, values, last_value, actions, mean, log_std, logp, adv, vtarg] + ([terminals] if terminals is not None else [])
    assert all(x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda for x in f32)
    assert done.dtype == torch.uint8 and done.is_contiguous() and (trunc is None or (trunc.dtype == torch.uint8 and trunc.is_contiguous()))
    assert values.shape == (T, N) and last_value.shape == (N,) and actions.shape == (T, N, 6) and adv.shape == (T, N)
    lib = _lib.load_library()
    st = None
    if stats is not None:
        assert adv_stats is not None and adv_stats.dtype == torch.float64 and adv_stats.numel() == 3 and adv_stats.is_cuda
        need = int(lib.pnr_ppo_gae_scratch(N))
        if getattr(stats, "_scratch", None) is None or stats._scratch.numel() < need:
            stats._scratch = torch.empty(need, dtype=torch.float64, device=reward.device)
        for x, dt, shape in ((stats.ret, torch.float32, (N,)), (stats.len, torch.float32, (N,)), (stats.w_sum, torch.float64, ()),
                             (stats.w_len, torch.float64, ()), (stats.w_cnt, torch.float64, ()), (stats.w_max, torch.float32, ()),
                             (stats.w_min, torch.float32, ())):
            assert x.dtype == dt and tuple(x.shape) == shape and x.is_cuda and x.is_contiguous()
        st = _lib.PnrRolloutStats(stats.ret.data_ptr(), stats.len.data_ptr(), stats._scratch.data_ptr(), stats._scratch.numel(),
                                  stats.w_sum.data_ptr(), stats.w_len.data_ptr(), stats.w_cnt.data_ptr(), stats.w_max.data_ptr(),
                                  stats.w_min.data_ptr(), adv_stats.data_ptr())
    _lib.check(lib.pnr_ppo_gae(T, N, _dp(reward), _dp(values), _dp(last_value), _dp(done), _dp(trunc), _dp(actions), _dp(mean), _dp(log_std),
                               float(gamma), float(lam), _dp(logp), _dp(adv), _dp(vtarg), _dp(terminals),
                               ctypes.byref(st) if st is not None else None,
                               ctypes.c_void_p(torch.cuda.current_stream(reward.device).cuda_stream)))


def hip_permutation(n: int, seed: int, stream_id: int, out: torch.Tensor) -> torch.Tensor:
    """out[:n] = a pseudo-random permutation of 0..n-1 keyed by (seed, stream_id): pnr_permutation, one launch, no sort."""
    import ctypes
    from . import _lib
    assert out.dtype == torch.int64 and out.is_contiguous() and out.numel() >= n and out.is_cuda
    _lib.check(_lib.load_library().pnr_permutation(n, seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1), _dp(out),
                                                   ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
    return out[:n]


class EpisodeStats:
    """episode_reward_{max,min,mean}, episode_len_mean, episodes_total (cli.py:32-38), on device.

    Every tensor is updated strictly IN PLACE: the sampling loop may be replayed from a captured
    hipGraph, which keeps reading and writing the buffers that existed at capture time."""

    def __init__(self, n, device):
        self.ret = torch.zeros(n, device=device)
        self.len = torch.zeros(n, device=device)
        self.total = 0
        self.w_sum = torch.zeros((), dtype=torch.float64, device=device)
        self.w_len = torch.zeros((), dtype=torch.float64, device=device)
        self.w_cnt = torch.zeros((), dtype=torch.float64, device=device)
        self.w_max = torch.full((), -float("inf"), device=device)
        self.w_min = torch.full((), float("inf"), device=device)

    def _reset_window(self):
        self.w_sum.zero_(); self.w_len.zero_(); self.w_cnt.zero_()
        self.w_max.fill_(-float("inf")); self.w_min.fill_(float("inf"))

    def step(self, reward, terminal):
        self.ret.add_(reward)
        self.len.add_(1.0)
        m = terminal > 0
        self.w_cnt.add_(m.sum())
        self.w_sum.add_(torch.where(m, self.ret, torch.zeros_like(self.ret)).sum().double())
        self.w_len.add_(torch.where(m, self.len, torch.zeros_like(self.len)).sum().double())
        self.w_max.copy_(torch.maximum(self.w_max, torch.where(m, self.ret, torch.full_like(self.ret, -float("inf"))).max()))
        self.w_min.copy_(torch.minimum(self.w_min, torch.where(m, self.ret, torch.full_like(self.ret, float("inf"))).min()))
        keep = (~m).to(self.ret.dtype)
        self.ret.mul_(keep)
        self.len.mul_(keep)

    def rollout(self, rewards: torch.Tensor, terminals: torch.Tensor) -> None:
        """The same bookkeeping as T calls of step(), from the rollout's [T, N] reward / terminal (0/1)
        buffers in ~25 kernels instead of ~30 per step.  With k(t) the 1-based index of the most recent
        terminal strictly before step t (0 = none in this rollout), the episode ending at a terminal t
        has return csum[t] - csum[k(t)] (+ the carried-in return if k(t) = 0) and length t - k(t)
        (+ the carried-in length)."""
        T, N = rewards.shape
        term = terminals.to(rewards.dtype)
        idx = torch.arange(1, T + 1, device=rewards.device, dtype=rewards.dtype).unsqueeze(1)
        last = torch.cummax(term * idx, dim=0).values                       # most recent terminal at or before t
        zero = torch.zeros((1, N), device=rewards.device, dtype=rewards.dtype)
        prev = torch.cat([zero, last[:-1]], dim=0)                          # ... strictly before t
        csum0 = torch.cat([zero, torch.cumsum(rewards, dim=0)], dim=0)      # csum0[k] = sum of the first k rewards
        fresh = prev == 0
        ep_ret = csum0[1:] - torch.gather(csum0, 0, prev.long()) + torch.where(fresh, self.ret.unsqueeze(0), zero)
        ep_len = idx - prev + torch.where(fresh, self.len.unsqueeze(0), zero)
        m = term > 0
        self.w_cnt.add_(m.sum())
        self.w_sum.add_(torch.where(m, ep_ret, zero).sum().double())
        self.w_len.add_(torch.where(m, ep_len, zero).sum().double())
        self.w_max.copy_(torch.maximum(self.w_max, torch.where(m, ep_ret, torch.full_like(zero, -float("inf"))).max()))
        self.w_min.copy_(torch.minimum(self.w_min, torch.where(m, ep_ret, torch.full_like(zero, float("inf"))).min()))
        end = last[-1]                                                       # [N]
        open_ = end == 0
        tail = csum0[-1] - torch.gather(csum0, 0, end.long().unsqueeze(0)).squeeze(0)
        self.ret.copy_(tail + torch.where(open_, self.ret, torch.zeros_like(self.ret)))
        self.len.copy_(float(T) - end + torch.where(open_, self.len, torch.zeros_like(self.len)))

    def window_tensor(self) -> torch.Tensor:
        """The window's (return sum, length sum, episode count, max, min) over all ranks as ONE float64 device tensor: read
        it back with whatever else the iteration reports (one host synchronisation), then call finish_window()."""
        packed = torch.stack([self.w_sum, self.w_len, self.w_cnt])
        pdist.allreduce_sum_(packed)
        mx = pdist.allreduce_max_(self.w_max.clone()); mn = pdist.allreduce_min_(self.w_min.clone())
        return torch.cat([packed, mx.double().reshape(1), mn.double().reshape(1)])

    def summarize(self) -> Dict[str, float]:
        return self.finish_window(self.window_tensor().tolist())

    def finish_window(self, vals) -> Dict[str, float]:
        s, l, c, mx, mn = vals
        self.total += int(c)
        out = {"episode_reward_mean": s / c if c else float("nan"),
               "episode_len_mean": l / c if c else float("nan"),
               "episode_reward_max": float(mx) if c else float("nan"),
               "episode_reward_min": float(mn) if c else float("nan"),
               "episodes_this_iter": int(c), "episodes_total": self.total}
        self._reset_window()
        return out


# other threads (the RCCL watchdog of torch.distributed) may touch the HIP runtime while this thread captures
_CAPTURE_MODE = "thread_local"


class PPOLearner:
    """The learn phase alone (usable on CPU with any rollout tensors): minibatch SGD on the clipped
    surrogate + adaptive KL + clipped value loss - entropy bonus, gradients averaged over ranks."""

    def __init__(self, cfg: PPOConfig, device, use_graph: bool = False):
        self.cfg = cfg
        self.device = torch.device(device)
        torch.manual_seed(cfg.seed)
        self.model = ActorCritic(cfg).to(self.device)
        pdist.broadcast_module_(self.model)
        # hipGraph capture of one minibatch update, GPU, no grad clip.  One rank: loss -> backward -> Adam
        # in ONE graph.  Several ranks: graph A (loss -> backward into a flat gradient bucket), the
        # bucket's all-reduce issued eagerly (collectives stay out of the graphs), graph B (Adam).
        self.use_graph = bool(use_graph) and self.device.type == "cuda" and not cfg.grad_clip
        self._split = self.use_graph and pdist.is_dist()
        self._flat_grad = None
        self._graph_b = None
        self.opt = torch.optim.Adam(self.model.parameters(), lr=cfg.lr, capturable=self.use_graph)
        self.kl_coeff = cfg.kl_coeff
        self.timesteps_total = 0
        # loss coefficients as device scalars so a captured graph sees their current values
        self._kl_c = torch.tensor(float(cfg.kl_coeff), device=self.device)
        self._ent_c = torch.tensor(float(cfg.entropy_coeff_start), device=self.device)
        self._graph = None
        self._static = None
        self._static_info = None
        self._eager_updates = 0
        self.fused_loss = self.device.type == "cuda" and cfg.act_dim == 6   # pnr_ppo_loss; torch ops otherwise (CPU)
        # the hand-written MLP kernels (csrc/pnr_mlp.h) carry the whole differentiable part of an update when the nets are
        # the reference's (137-256-256, pioneer_knm_train.py:59-61) and bf16 GEMMs were asked for; float32 runs stay on torch
        self.hip = (self.fused_loss and bool(cfg.amp_bf16) and cfg.obs_dim == 137 and tuple(cfg.fcnet_hiddens) == (256, 256))
        self._mlp = None            # HipMLP with a workspace for one minibatch
        self._means = None          # [minibatches, 8] per-update loss means (HIP path)
        self._perm = None           # the epoch's minibatch shuffle (HIP path)
        self._epochs = 0            # SGD epochs so far: the shuffle's stream id (saved with the optimiser state)
        self._hip_dirty = True      # the packed bf16 weights are stale (construction, restore)

    def drop_graphs(self) -> None:
        """Forget the captured minibatch update (it is re-captured after the eager warm-up updates)."""
        if self._graph is not None or self._graph_b is not None:
            torch.cuda.synchronize(self.device)
        self._graph = self._graph_b = self._static = self._static_info = self._flat_grad = None
        self._eager_updates = 0
        for p in self.model.parameters():
            p.grad = None                      # split mode made every .grad a view of the flat bucket

    def entropy_coeff(self) -> float:
        frac = min(1.0, self.timesteps_total / max(1, self.cfg.entropy_decay_steps))
        return self.cfg.entropy_coeff_start * (1.0 - frac)

    def loss(self, mb: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        cfg = self.cfg
        if self.hip and mb["obs"].is_cuda:
            idx = mb.get("idx")
            B = int(idx.numel()) if idx is not None else int(mb["obs"].shape[0])
            m = self.hip_mlp(B).policy_loss(mb["obs"], idx, mb.get("filt"), mb, self._kl_c, self._ent_c, cfg.clip_param,
                                      cfg.vf_clip_param, cfg.vf_loss_coeff)
            # the reported means are DETACHED views: kept across minibatches (agg, _static_info) they must not keep the
            # autograd graph alive — its AccumulateGrad nodes would remember the stream of an earlier eager update, and
            # replaying them under capture on the capture stream tied the two streams together (segfault at capture_end)
            d = m.detach()
            return m[4], {"policy_loss": d[0], "vf_loss": d[1], "kl": d[2], "entropy": d[3], "total_loss": d[4]}
        if self.fused_loss and mb["obs"].is_cuda:
            hp, hv = self.model.forward_heads(mb["obs"], cfg.amp_bf16)
            total, m = FusedPPOLoss.apply(hp, hv, mb, self._kl_c, self._ent_c, float(cfg.clip_param),
                                          float(cfg.vf_clip_param), float(cfg.vf_loss_coeff))
            return total, {"policy_loss": m[0], "vf_loss": m[1], "kl": m[2], "entropy": m[3], "total_loss": m[4]}
        mean, log_std, v = self.model(mb["obs"], cfg.amp_bf16)
        logp = gaussian_logp(mb["actions"], mean, log_std)
        ratio = torch.exp(logp - mb["logp"])
        adv = mb["adv"]
        surr = torch.minimum(adv * ratio, adv * torch.clamp(ratio, 1 - cfg.clip_param, 1 + cfg.clip_param))
        kl = gaussian_kl(mb["mean"], mb["log_std"], mean, log_std)
        ent = gaussian_entropy(log_std)
        vf1 = (v - mb["vtarg"]) ** 2
        v_clipped = mb["values"] + torch.clamp(v - mb["values"], -cfg.vf_clip_param, cfg.vf_clip_param)
        vf = torch.maximum(vf1, (v_clipped - mb["vtarg"]) ** 2)
        total = (-surr + self._kl_c * kl + cfg.vf_loss_coeff * vf - self._ent_c * ent).mean()
        return total, {"policy_loss": -surr.mean().detach(), "vf_loss": vf.mean().detach(),
                       "kl": kl.mean().detach(), "entropy": ent.mean().detach(), "total_loss": total.detach()}

    def update(self, batch: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None, readback: bool = True):
        """batch tensors are flat [B, ...] (this rank's share).  Every rank must run the same number
        of minibatches (equal shard sizes)."""
        cfg = self.cfg
        B = batch["obs"].shape[0]
        mbs = min(cfg.sgd_minibatch_size, B)
        adv = batch["adv"]
        # standardise advantages over the GLOBAL batch
        if batch.get("adv_stats") is not None:          # this rank's moments, made by the rollout's pnr_ppo_gae launch
            stats = batch["adv_stats"].clone()
        else:
            stats = torch.stack([adv.sum().double(), (adv.double() ** 2).sum(), torch.tensor(float(B), dtype=torch.float64, device=adv.device)])
        pdist.allreduce_sum_(stats)
        mu = stats[0] / stats[2]
        sd = torch.sqrt(torch.clamp(stats[1] / stats[2] - mu * mu, min=1e-12))
        hip_path = self.hip and adv.is_cuda
        # (the HIP path standardises while it packs the record, pnr_ppo_pack_record: the same two float32 operations)
        adv_n = adv if hip_path else (adv - mu.float()) / (sd.float() + 1e-8)
        self._adv_scalars = (mu.float().reshape(1), (sd.float() + 1e-8).reshape(1)) if hip_path else None
        self._kl_c.fill_(self.kl_coeff)
        self._ent_c.fill_(self.entropy_coeff())
        agg: Dict[str, torch.Tensor] = {}
        nmb = 0
        filt = batch.get("filt")
        tens = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor) and k != "adv_stats"}
        tens["adv"] = adv_n
        if self.hip and adv.is_cuda:
            m = self._update_hip(tens, filt, B, mbs, generator)
            if not readback:
                return m                                            # PPOTrainer reads it back together with the episode statistics
            return self.finish_update_values(m.tolist())

        def eager_step(mb):
            loss, info = self.loss(mb)
            self.opt.zero_grad(set_to_none=True)
            loss.backward()
            pdist.allreduce_mean_grads(self.model.parameters())   # one flat 0.82 MB bucket
            if cfg.grad_clip:
                nn.utils.clip_grad_norm_(self.model.parameters(), cfg.grad_clip)
            self.opt.step()
            return info

        graph_ok = self.use_graph and (self._static is None or self._static["obs"].shape[0] == mbs)
        for _ in range(cfg.num_sgd_iter):
            perm = torch.randperm(B, device=adv.device, generator=generator)
            for s in range(0, B - mbs + 1, mbs):
                idx = perm[s:s + mbs]
                if graph_ok and self._graph is None and self._eager_updates >= 3:
                    self._capture(tens, idx)                        # after a few eager updates (warm-up)
                if graph_ok and self._graph is not None:
                    for k, v in tens.items():
                        torch.index_select(v, 0, idx, out=self._static[k])
                    self._graph.replay()
                    if self._split:
                        pdist.allreduce_mean_(self._flat_grad)      # the one 0.82 MB bucket, eager
                        self._graph_b.replay()
                    info = self._static_info
                else:
                    info = eager_step({k: v[idx] for k, v in tens.items()})
                    self._eager_updates += 1
                for k, v in info.items():
                    agg[k] = agg.get(k, 0) + v
                nmb += 1
        out = {k: float(v) / max(1, nmb) for k, v in agg.items()}
        return self._finish_update(out, adv.device)

    def finish_update_values(self, m) -> Dict[str, float]:
        """The HIP path's loss means (already averaged over the ranks) as the result columns + the adaptive-KL step."""
        out = {"policy_loss": m[0], "vf_loss": m[1], "kl": m[2], "entropy": m[3], "total_loss": m[4]}
        cfg = self.cfg
        if out["kl"] > 2.0 * cfg.kl_target:
            self.kl_coeff *= 1.5
        elif out["kl"] < 0.5 * cfg.kl_target:
            self.kl_coeff *= 0.5
        out["cur_kl_coeff"] = self.kl_coeff
        out["entropy_coeff"] = self.entropy_coeff()
        return out

    def _finish_update(self, out: Dict[str, float], device) -> Dict[str, float]:
        cfg = self.cfg
        # adaptive KL (RLlib PPO: update_kl)
        kl_t = torch.tensor([out.get("kl", 0.0)], dtype=torch.float64, device=device)
        pdist.allreduce_sum_(kl_t)
        kl = float(kl_t) / (pdist.dist.get_world_size() if pdist.is_dist() else 1)
        if kl > 2.0 * cfg.kl_target:
            self.kl_coeff *= 1.5
        elif kl < 0.5 * cfg.kl_target:
            self.kl_coeff *= 0.5
        out["kl"] = kl
        out["cur_kl_coeff"] = self.kl_coeff
        out["entropy_coeff"] = self.entropy_coeff()
        return out

    def optimizer_state(self):
        """torch.optim.Adam's state_dict on the torch path; on the HIP path Adam's moments as the fused kernel keeps them
        (padded gradient layout) and its update count."""
        if self.hip and self._mlp is not None:
            m, v, step = self._mlp.adam_state()
            return {"hip_adam": {"m": m, "v": v, "step": step, "epochs": self._epochs}}
        return self.opt.state_dict()

    def load_optimizer_state(self, sd) -> None:
        if "hip_adam" in sd:
            if not self.hip:
                raise AssertionError("the checkpoint holds the HIP path's optimiser state; this learner runs on torch")
            mlp = self.hip_mlp(max(1, min(self.cfg.sgd_minibatch_size, 1 << 20)))
            for dst, k in zip(mlp.adam_state(), ("m", "v", "step")):
                dst.copy_(sd["hip_adam"][k])
            self._epochs = int(sd["hip_adam"].get("epochs", 0))
        elif not (self.hip and not sd.get("state")):
            self.opt.load_state_dict(sd)
        self._hip_dirty = True           # the master weights changed behind the packed bf16 copies

    # -- the HIP path: every minibatch update is pnr_mlp_train_step, three launches, no autograd, no hipGraph needed ----
    def hip_mlp(self, batch: int):
        if self._mlp is None or self._mlp.max_batch < batch:
            from .mlp import HipMLP
            old = self._mlp
            self._mlp = HipMLP(self.model, batch, self.device)
            if old is not None:                                   # keep the optimiser state across a workspace resize
                for a, b in zip(self._mlp.adam_state(), old.adam_state()):
                    a.copy_(b)
            self._hip_dirty = True
        return self._mlp

    def _update_hip(self, tens, filt, B, mbs, generator) -> Dict[str, float]:
        """The minibatch loop on the hand-written kernels.  The kernels gather minibatch rows themselves (a slice of the
        epoch's permutation is the row index), the float32 master parameters are updated in place by the fused
        reduction + Adam kernel, which also refreshes the packed bf16 weights for the next forward.  Several ranks:
        the reduced gradient goes to one flat bucket, is all-reduced (RCCL), and pnr_mlp_adam applies the mean."""
        cfg, dev = self.cfg, self.device
        mlp = self.hip_mlp(mbs)
        if self._hip_dirty:
            mlp.pack()
            self._hip_dirty = False
        rec = {k: v.contiguous() for k, v in tens.items()}
        nmb_epoch = len(range(0, B - mbs + 1, mbs))
        total = cfg.num_sgd_iter * nmb_epoch
        if self._means is None or self._means.shape[0] != total:
            self._means = torch.zeros((total, 8), dtype=torch.float32, device=dev)
        multi = pdist.is_dist()
        world = pdist.dist.get_world_size() if multi else 1
        if multi and self._flat_grad is None:
            self._flat_grad = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), dtype=torch.float32, device=dev)
        if self._perm is None or self._perm.numel() != B:
            self._perm = torch.empty(B, dtype=torch.int64, device=dev)
        # the record as one 96-byte row per sample, advantages standardised on the way: once per iteration
        rows = mlp.pack_record(rec, *(self._adv_scalars or (None, None)))
        k = 0
        for _ in range(cfg.num_sgd_iter):
            # each epoch's shuffle: pnr_permutation keyed by (seed, rank, epoch counter) — one launch, no sort
            perm = hip_permutation(B, cfg.seed * 1000003 + (pdist.dist.get_rank() if multi else 0), self._epochs, self._perm)
            self._epochs += 1
            # ... applied once per epoch (pnr_mlp_gather): the 16 updates then read contiguous rows
            g = mlp.gather_epoch(rec["obs"], perm, filt, None, rec_rows=rows, xs_rows=rec.get("xs"))
            for s in range(0, B - mbs + 1, mbs):
                mlp.train_step(None, None, None, {k: g[k][s:s + mbs] for k in mlp.REC_KEYS}, self._kl_c, self._ent_c, cfg.clip_param,
                               cfg.vf_clip_param, cfg.vf_loss_coeff, self._means[k], cfg.lr,
                               flat_grad=self._flat_grad if multi else None, xs_in=g["xs"][s:s + mbs])
                if multi:
                    pdist.allreduce_sum_(self._flat_grad)          # the one 0.86 MB bucket
                    mlp.adam(self._flat_grad, 1.0 / world, cfg.lr)
                k += 1
        m = self._means.mean(0).double()
        if multi:                                                   # the reported losses are averages over the ranks
            pdist.allreduce_sum_(m)
            m = m / world
        return m                                                    # device tensor [8]: read back by the caller


def _learner_capture(self, batch, idx):
    """Capture one minibatch update on static buffers: one hipGraph (single rank) or two with the
    gradient all-reduce between them (several ranks).  The capture pass only records work."""
    try:
        self._static = {k: v[idx].clone() for k, v in batch.items()}
        torch.cuda.synchronize(self.device)
        if not self._split:
            self.opt.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                loss, info = self.loss(self._static)
                loss.backward()
                self.opt.step()
            self._graph, self._static_info = g, info
            return
        # every parameter's .grad becomes a view of one flat bucket; backward accumulates in place
        params = [p for p in self.model.parameters() if p.requires_grad]
        flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=self.device)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode=_CAPTURE_MODE):
            flat.zero_()
            loss, info = self.loss(self._static)
            loss.backward()
        with torch.cuda.graph(gb, capture_error_mode=_CAPTURE_MODE):
            self.opt.step()
        self._graph, self._graph_b, self._flat_grad, self._static_info = ga, gb, flat, info
    except Exception:            # capture not possible here: stay eager
        self._graph, self._graph_b, self._static, self.use_graph, self._split = None, None, None, False, False
        torch.cuda.synchronize(self.device)


PPOLearner._capture = _learner_capture


class PPOTrainer:
    """Rollout + learn loop over a PioneerVectorEnv shard (one process per GPU).

    ``use_graph=True`` captures the whole T-step sampling loop (policy forward, action sampling, ``pnr_step``,
    log-probs, GAE) into ONE hipGraph after an eager warm-up iteration.  With the reference's nets and bf16 GEMMs
    (``PPOLearner.hip``) the nets run on the hand-written MFMA kernels: one launch per step computes both heads from
    the RAW observation the env kernel left in the rollout buffer (the MeanStdFilter is applied on load), and the
    learner reads the same raw buffer through a row index — filtered observations are never materialised."""

    def __init__(self, env, cfg: Optional[PPOConfig] = None, use_graph: bool = False):
        self.env = env
        self.cfg = cfg or PPOConfig()
        self.device = env.device
        self.rank, _, self.world = pdist.world_info()
        self.learner = PPOLearner(self.cfg, self.device, use_graph=use_graph)
        self.hip = self.learner.hip
        self.filter = (MeanStdFilter(self.cfg.obs_dim, self.device, self.cfg.filter_clip)
                       if self.cfg.observation_filter in ("MeanStdFilter", "ConcurrentMeanStdFilter") else NoFilter())
        self.stats = EpisodeStats(env.num_envs, self.device)
        self._ev = None             # timing events of train()
        self.gen = torch.Generator(device=self.device).manual_seed(self.cfg.seed * 1000003 + self.rank)
        self.a_max = torch.from_numpy(env.a_max).to(self.device)
        self._a_lo = -self.a_max
        self.iteration = 0
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        self._graph = None
        self._xin = None
        T, N, D, A = self.cfg.rollout_fragment_length, env.num_envs, self.cfg.obs_dim, self.cfg.act_dim
        f32 = dict(dtype=torch.float32, device=self.device)
        self._xlast = torch.empty((N, D), **f32)
        self._env_act = torch.empty((N, A), **f32)
        # raw observations: slot 0 = what the rollout starts from, slot t + 1 = written in place by pnr_step at step t;
        # the nets' inputs of a rollout are slots 0 .. T-1, slot T carries over to the next rollout's slot 0
        self.raw_in = torch.empty((T + 1, N, D), **f32)
        self.raw_in[T].copy_(env.reset())
        self.buf = {
            "raw_obs": self.raw_in[1:],
            "actions": torch.empty((T, N, A), **f32),
            "mean": torch.empty((T, N, A), **f32), "log_std": torch.empty((T, N, A), **f32),   # the policy head, clamped log-stds
            "logp": torch.empty((T, N), **f32), "values": torch.empty((T, N), **f32),
            "reward": torch.empty((T, N), **f32),
            "adv": torch.empty((T, N), **f32), "vtarg": torch.empty((T, N), **f32),
            "done": torch.empty((T, N), dtype=torch.uint8, device=self.device),
            "trunc": torch.empty((T, N), dtype=torch.uint8, device=self.device),
            "term_u8": torch.empty((T, N), dtype=torch.uint8, device=self.device),
            "terminals": torch.empty((T, N), **f32),                   # done | truncated as 0 / 1
        }
        if self.hip:
            from .mlp import HipMLP
            self.sample_mlp = HipMLP(self.learner.model, N, self.device)   # packed weights for the rollout's T forwards
            self._last_heads = torch.empty((2, N, 16), **f32)
            self._last_v = torch.empty((N,), **f32)
            # the nets' inputs as the sampler saw them (filtered, bf16): what the learner trains on, 288 bytes per sample
            self.buf["xs"] = torch.empty((T, N, 144), dtype=torch.bfloat16, device=self.device)
            self._adv_stats = torch.zeros(3, dtype=torch.float64, device=self.device)     # sum, sum of squares, count (pnr_ppo_gae)
        else:
            self.buf["obs"] = torch.empty((T, N, D), **f32)            # filtered, what the nets saw

    @property
    def raw_obs(self) -> torch.Tensor:
        """The observation the next rollout starts from."""
        return self.raw_in[self.cfg.rollout_fragment_length]

    def _filt(self):
        f = self.filter
        return (f._loc, f._inv, f._lo, f._hi) if isinstance(f, MeanStdFilter) else None

    def _noise(self):
        shape = tuple(self.buf["actions"].shape)
        # in-graph noise comes from the default (graph-safe) generator
        return torch.randn(shape, device=self.device) if self._capturing else torch.randn(shape, generator=self.gen, device=self.device)

    def _finish_rollout(self, last_v: torch.Tensor) -> None:
        cfg, buf = self.cfg, self.buf
        if self.hip:
            hip_gae_logp(buf["reward"], buf["values"], last_v.contiguous(), buf["done"], buf["trunc"], buf["actions"], buf["mean"],
                         buf["log_std"], cfg.gamma, cfg.lambda_, logp=buf["logp"], adv=buf["adv"], vtarg=buf["vtarg"],
                         terminals=buf["terminals"], stats=self.stats, adv_stats=self._adv_stats)
            return
        buf["logp"].copy_(gaussian_logp(buf["actions"], buf["mean"], buf["log_std"]))
        torch.bitwise_or(buf["done"], buf["trunc"], out=buf["term_u8"])
        buf["terminals"].copy_(buf["term_u8"])
        adv, vtarg = compute_gae(buf["reward"], buf["values"], last_v, buf["terminals"], cfg.gamma, cfg.lambda_)
        buf["adv"].copy_(adv); buf["vtarg"].copy_(vtarg)

    def _env_step(self, t: int, act: torch.Tensor) -> None:
        buf = self.buf
        env_act = torch.clamp(act, self._a_lo, self.a_max, out=self._env_act) if self.cfg.clip_actions else act
        self.env.vector_step(env_act, out={"obs": self.raw_in[t + 1], "reward": buf["reward"][t], "done": buf["done"][t],
                                           "truncated": buf["trunc"][t]})

    @torch.no_grad()
    def _collect_impl(self) -> None:
        """T steps into the static buffers; pure device work (capturable).  Log-probs and GAE are computed once, after
        the loop, from the [T, N] buffers; episode statistics and filter moments in _collect_tail(), outside any
        capture."""
        cfg, buf, model = self.cfg, self.buf, self.learner.model
        T, A = cfg.rollout_fragment_length, cfg.act_dim
        self.raw_in[0].copy_(self.raw_in[T])
        self.filter.prepare()
        noise = self._noise()
        if self.hip:
            # per step TWO launches: pnr_mlp_act (both nets on the raw observation, filter applied on load, action draw and
            # clip in the policy net's epilogue) and pnr_step
            mlp, filt = self.sample_mlp, self._filt()
            mlp.pack()
            clip = self.cfg.clip_actions
            for t in range(T):
                mlp.act(self.raw_in[t], filt, noise[t], self.a_max if clip else None, mean=buf["mean"][t], log_std=buf["log_std"][t],
                        values=buf["values"][t], actions=buf["actions"][t], env_actions=self._env_act if clip else None,
                        xs_out=buf["xs"][t])
                self.env.vector_step(self._env_act if clip else buf["actions"][t],
                                     out={"obs": self.raw_in[t + 1], "reward": buf["reward"][t], "done": buf["done"][t],
                                          "truncated": buf["trunc"][t]})
            last = mlp.forward_nograd(self.raw_in[T], None, filt, out=self._last_heads)
            self._last_v.copy_(last[1, :, 0])           # bootstrap value of the state after the last step
            self._finish_rollout(self._last_v)
            return
        model.refresh_inference_cache(cfg.amp_bf16)
        cdt = model._icache["dtype"]
        if self._xin is None or self._xin.dtype != cdt:
            self._xin = torch.zeros((self.raw_in.shape[1], cfg.obs_dim + (-cfg.obs_dim) % 16), dtype=cdt, device=self.device)
        xin = self._xin
        for t in range(T):
            x = buf["obs"][t]
            self.filter.apply_(self.raw_in[t], out=x)
            xin[:, :cfg.obs_dim].copy_(x)
            head, v = model.forward_cached(xin)
            buf["mean"][t].copy_(head[:, :A])
            log_std = torch.clamp(head[:, A:], -20.0, 2.0, out=buf["log_std"][t])
            buf["values"][t].copy_(v.squeeze(-1))
            act = torch.addcmul(buf["mean"][t], torch.exp(log_std), noise[t], out=buf["actions"][t])
            self._env_step(t, act)
        xin[:, :cfg.obs_dim].copy_(self.filter.apply_(self.raw_in[T], out=self._xlast))
        self._finish_rollout(model.forward_cached(xin)[1].squeeze(-1).float())

    def _collect_tail(self) -> None:
        """The rollout's bookkeeping reductions — episode statistics and the filter's moments — run EAGERLY after
        the (possibly replayed) loop, never inside a captured graph.  Root cause of the r01 "NaNs after graph replay"
        finding (tools/graph_reduce_probe.py, profiles/r02_graph_reduce_probe.json): a torch reduction over the
        middle axis of a large tensor ([31, 16384, 137].sum(1): the multi-block path with per-output semaphores that
        the launcher zeroes by hipMemsetAsync) returns wrong sums from the SECOND replay of a hipGraph on (first
        replay exact, eager always exact) — depending on the pool layout, which is why changing head widths or
        minibatch sizes made it come and go.  It fed garbage into the filter's moments.  Nothing of the BLAS
        library was involved.  The filter only changes at sync(), so observing all inputs here, in one pass, is
        identical to observing them one by one inside the loop."""
        buf, T = self.buf, self.cfg.rollout_fragment_length
        if not self.hip:                                  # the HIP path's pnr_ppo_gae launch has done the episode statistics
            self.stats.rollout(buf["reward"], buf["terminals"])
        self.filter.observe(self.raw_in[:T])

    _capturing = False

    def collect(self) -> Dict[str, torch.Tensor]:
        if self.use_graph and self._graph is None and self.iteration >= 1:
            # capture after one eager iteration (allocator and library warm-up done)
            torch.cuda.synchronize(self.device)
            self._capturing = True              # in-graph noise comes from the default (graph-safe) generator
            torch.cuda.manual_seed(self.cfg.seed * 7919 + self.rank + 1)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                self._collect_impl()
            self._graph = g
            # the capture pass itself only recorded work: run it for real below
        if self._graph is not None:
            self._graph.replay()
        else:
            self._collect_impl()
        self._collect_tail()
        buf, T = self.buf, self.cfg.rollout_fragment_length
        flat = lambda x: x.reshape(-1, *x.shape[2:])  # noqa: E731
        batch = {"actions": flat(buf["actions"]), "mean": flat(buf["mean"]), "log_std": flat(buf["log_std"]),
                 "logp": flat(buf["logp"]), "values": flat(buf["values"]), "adv": flat(buf["adv"]), "vtarg": flat(buf["vtarg"])}
        if self.hip:
            batch.update(obs=flat(self.raw_in[:T]), filt=self._filt())      # raw: the kernels filter on load
            batch["adv_stats"] = self._adv_stats
            batch["xs"] = flat(buf["xs"])
        else:
            batch["obs"] = flat(buf["obs"])
        return batch

    def train(self) -> Dict[str, float]:
        """One iteration: collect, merge the filter, update.  On the GPU the phase split (sample_time_s / learn_time_s, RLlib's
        sample_time_ms / learn_time_ms) comes from an event recorded between the phases, not from a host synchronisation there:
        the learner's first kernels are queued while the sampler's last ones still run."""
        cuda = self.device.type == "cuda"
        t0 = time.perf_counter()
        if cuda:
            if self._ev is None:
                self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            self._ev[0].record()
        batch = self.collect()
        if cuda:
            self._ev[1].record()
        t1 = time.perf_counter()
        self.filter.sync()
        steps = batch["obs"].shape[0] * self.world
        self.learner.timesteps_total += steps
        info = self.learner.update(batch, self.gen, readback=not self.hip)
        win = self.stats.window_tensor()
        if self.hip:                       # ONE device -> host read per iteration: loss means + episode statistics
            vals = torch.cat([info, win]).tolist()
            info, win = self.learner.finish_update_values(vals[:8]), vals[8:]
        else:
            win = win.tolist()
        if cuda:
            self._ev[2].record()
            torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
        if cuda:
            gpu_sample = self._ev[0].elapsed_time(self._ev[1]) * 1e-3
            t1 = t0 + min(max(gpu_sample, t1 - t0), t2 - t0)       # the sampler's share of the wall time of this iteration
        self.iteration += 1
        res = self.stats.finish_window(win)
        res.update(info)
        res.update({"training_iteration": self.iteration, "timesteps_total": self.learner.timesteps_total,
                    "timesteps_this_iter": steps, "sample_time_s": t1 - t0, "learn_time_s": t2 - t1,
                    "time_this_iter_s": t2 - t0, "env_steps_per_s": steps / (t2 - t0)})
        return res

    # -- checkpoint / resume (Tune's checkpoint_freq / checkpoint_at_end, pioneer_knm_train.py:72-73) --
    def _env_state(self) -> Dict[str, torch.Tensor]:
        st = {"env_state": self.env.get_state().cpu(), "env_id_offset": int(self.env.env_id_offset),
              "num_envs": int(self.env.num_envs)}
        if self.env.engine_config.mode == "dynamic":
            st["dyn_state"] = self.env.get_dyn_state().cpu()        # q, qd and the per-env randomised parameters
        return st

    @staticmethod
    def _env_path(path: str, rank: int) -> str:
        return f"{path}.env_rank{rank}"

    def save(self, path: str) -> str:
        """Rank 0 writes the learner (weights, optimiser, filter, counters) and its own env shard; every other
        rank writes its env shard next to it (`<path>.env_rank<r>`), so restore_env gives each rank ITS envs back."""
        est = self._env_state()
        if self.rank == 0:
            torch.save({"model": self.learner.model.state_dict(), "opt": self.learner.optimizer_state(),
                        "filter": self.filter.state_dict(), "kl_coeff": self.learner.kl_coeff,
                        "timesteps_total": self.learner.timesteps_total, "iteration": self.iteration,
                        "episodes_total": self.stats.total, "world": self.world, **est,
                        "cfg": self.cfg.__dict__}, path)
        else:
            torch.save(est, self._env_path(path, self.rank))
        pdist.barrier()
        return path

    def restore(self, path: str, restore_env: bool = False) -> None:
        ck = torch.load(path, map_location=self.device, weights_only=True)    # tensors and plain values only
        self.learner.model.load_state_dict(ck["model"]); self.learner.load_optimizer_state(ck["opt"])
        # load_state_dict REPLACES the optimiser's state tensors: a learner graph captured before this call would
        # keep replaying on the old exp_avg / exp_avg_sq / step.  Drop the captures; they are rebuilt after the
        # usual eager warm-up updates.  (Model weights and filter moments are copied in place: the sampling
        # graph keeps seeing them.)
        self.learner.drop_graphs()
        self.filter.load_state_dict(ck["filter"]); self.learner.kl_coeff = ck["kl_coeff"]
        self.learner.timesteps_total = ck["timesteps_total"]; self.iteration = ck["iteration"]
        self.stats.total = ck["episodes_total"]
        if restore_env:
            est = ck if self.rank == 0 else torch.load(self._env_path(path, self.rank), map_location=self.device,
                                                      weights_only=True)
            if int(ck.get("world", 1)) != self.world or int(est.get("num_envs", est["env_state"].shape[1])) != self.env.num_envs \
                    or int(est.get("env_id_offset", 0)) != int(self.env.env_id_offset):
                raise AssertionError("restore_env: the checkpoint's env shards (world size, envs per rank, env id offsets) "
                                     "do not match this run")
            dynamic = self.env.engine_config.mode == "dynamic"
            if dynamic != ("dyn_state" in est):
                raise AssertionError("restore_env: the checkpoint's env mode (kinematic / dynamic) does not match this run")
            self.env.set_state(est["env_state"].to(self.device))
            if dynamic:
                self.env.set_dyn_state(est["dyn_state"].to(self.device))
            self.raw_obs.copy_(self.env.observe())
