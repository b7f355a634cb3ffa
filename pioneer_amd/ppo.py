"""Thin PyTorch-ROCm PPO host driver.

Replaces what RLlib's PPOTrainer + rollout workers do for pioneer/launch/pioneer_knm_train.py:
the sampling loop steps the env with device-resident tensors (no Ray object store, no host copies),
the learner is data-parallel over GPUs with the gradients all-reduced per minibatch (RCCL over xGMI).
Hyper-parameter names and defaults follow the reference's config dict (pioneer_knm_train.py:45-67) and
the RLlib-0.8.x PPO defaults it did not override.

On a HIP device the nets of the reference (137 -> 256 -> 256 tanh, separate value net) run on the
hand-written kernels of csrc/pnr_mlp.h / pnr_ppo.h (bf16 MFMA operands, float32 accumulation, float32
master weights and Adam moments): that is what ``PPOConfig()`` selects (``hip_kernels=True``).  The
torch formulation below (float32 autograd, ``torch.optim.Adam``) is what CPU tensors use (the gloo
tests, synthetic rollouts) and the numerical reference the kernels are tested against; on a GPU it
only runs when asked for (``hip_kernels=False``) or for nets of another shape, eagerly.
"""
import math
import os
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import dist as pdist


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
    grad_clip: Optional[float] = None     # torch formulation only
    # entropy_coeff_schedule [(0, x), (decay_steps, 0)] (pioneer_knm_train.py:32-41, :63)
    entropy_coeff_start: float = 1e-2
    entropy_decay_steps: int = 1_000_000
    # 'observation_filter': 'ConcurrentMeanStdFilter' (pioneer_knm_train.py:66)
    observation_filter: str = "MeanStdFilter"
    filter_clip: float = 10.0
    seed: int = 0
    # engine option (no reference counterpart): on a HIP device, run the reference-shaped nets, the loss, Adam, GAE and the
    # shuffle on the hand-written kernels.  "f32": every operand as two scaled fp16 planes = float32-accurate products (22 significant
    # bits, measured at torch float32's own distance from float64) — the reference's learner is float32 torch,
    # pioneer_knm_train.py:47; True / "bf16": bf16 MFMA operands (8 significant bits: the reduced-precision fast variant);
    # "bf16x3": three bf16 planes (24 bits, twice the MFMAs of "f32"); False = the float32 torch formulation
    hip_kernels: object = True

    @classmethod
    def from_dict(cls, d: Dict) -> "PPOConfig":
        """From a checkpoint's / params.json's dict: lists back to tuples, r01 / r02's `amp_bf16` read as `hip_kernels`, r04's "bf16x2" as "f32"."""
        d = dict(d)
        if "amp_bf16" in d:
            d.setdefault("hip_kernels", bool(d.pop("amp_bf16")))
        if d.get("hip_kernels") == "bf16x2":          # r04's two-bf16-plane form no longer exists: its successor is the two-fp16-plane "f32"
            d["hip_kernels"] = "f32"
        return cls(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items() if k in cls.__dataclass_fields__})

    def mlp_planes(self) -> int:
        """bf16 planes per MFMA operand of the hand-written kernels (pioneer_amd.mlp.PLANES)."""
        from .mlp import PLANES
        if self.hip_kernels not in PLANES:
            raise AssertionError(f"hip_kernels must be one of True, 'bf16', 'f32', 'bf16x3' or False, got {self.hip_kernels!r}")
        return PLANES[self.hip_kernels]

    def mlp_dtype(self) -> str:
        return {1: "bf16", 2: "f32 (two scaled fp16 planes per operand: float32-accurate products)", 3: "bf16x3 (three bf16 planes per operand)"}[self.mlp_planes()]

    def wants_hip(self, device) -> bool:
        if self.hip_kernels:
            self.mlp_planes()          # a misspelt precision fails here, not as a silent torch run
        return (bool(self.hip_kernels) and torch.device(device).type == "cuda" and self.obs_dim == 137 and self.act_dim == 6
                and tuple(self.fcnet_hiddens) == (256, 256) and not self.grad_clip)


def sample_entropy_start(rng: np.random.RandomState, min_start: float = 1e-3, max_start: float = 1e-1,
                         base: float = 10.0) -> float:
    """entropy_coeff_schedule's log-uniform start value (pioneer_knm_train.py:32-41)."""
    logmin = np.log(min_start) / np.log(base)
    logmax = np.log(max_start) / np.log(base)
    return float(base ** rng.uniform(logmin, logmax))


from .filters import MeanStdFilter, NoFilter   # noqa: E402,F401  (re-exported: the driver's observation filters)


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

class ActorCritic(nn.Module):
    """Separate policy and value MLPs [137 -> 256 -> 256 -> 12 | 1], tanh (RLlib FullyConnectedNetwork
    with vf_share_layers False).  The policy head emits the Gaussian's 6 means and 6 log-stds:
    104 204 + 101 377 = 205 581 parameters.  These float32 tensors are the master weights; on a
    HIP device the kernels read packed bf16 copies of them (pioneer_amd.mlp.HipMLP)."""

    def __init__(self, cfg: PPOConfig):
        super().__init__()
        sizes = [cfg.obs_dim, *cfg.fcnet_hiddens]
        self.policy = _mlp(sizes, 2 * cfg.act_dim, 0.01)
        self.value = _mlp(sizes, 1, 1.0)
        self.act_dim = cfg.act_dim

    def forward(self, obs):
        out, v = self.policy(obs), self.value(obs)
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


from .filters import _dp   # noqa: E402


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


class PPOLearner:
    """The learn phase alone (usable on CPU with any rollout tensors): minibatch SGD on the clipped
    surrogate + adaptive KL + clipped value loss - entropy bonus, gradients averaged over ranks."""

    def __init__(self, cfg: PPOConfig, device):
        self.cfg = cfg
        self.device = torch.device(device)
        torch.manual_seed(cfg.seed)
        self.model = ActorCritic(cfg).to(self.device)
        pdist.broadcast_module_(self.model)
        # the hand-written kernels carry the whole update when the nets are the reference's (pioneer_knm_train.py:59-61) on a
        # HIP device; otherwise float32 autograd + torch.optim.Adam
        self.hip = cfg.wants_hip(self.device)
        self.opt = None if self.hip else torch.optim.Adam(self.model.parameters(), lr=cfg.lr)
        self.kl_coeff = cfg.kl_coeff
        self.timesteps_total = 0
        # loss coefficients as device scalars: the kernels (and a captured sampling graph) read their current values
        self._kl_c = torch.tensor(float(cfg.kl_coeff), device=self.device)
        self._ent_c = torch.tensor(float(cfg.entropy_coeff_start), device=self.device)
        self._mlp = None            # HipMLP with a workspace for one minibatch
        self._means = None          # [minibatches, 8] per-update loss means (HIP path)
        self._perm = None           # the epoch's minibatch shuffle (HIP path)
        self._flat_grad = None      # several ranks: the gradient bucket that is all-reduced (HIP path)
        self._epochs = 0            # SGD epochs so far: the shuffle's stream id (saved with the optimiser state)
        self._hip_dirty = True      # the packed bf16 weights are stale (construction, restore)
        # several ranks: the two nets as two SGD chains on two streams, each all-reducing its half of the bucket (False: one bucket,
        # serial).  Bit-identical to the one-bucket form on gloo (tests/test_gpu_multirank.py); its overlap gain and its behaviour
        # under real RCCL stream semantics are UNMEASURED (no multi-GPU node so far): PNR_NET_CHAINS=0 is the fallback switch.
        self.net_chains = os.environ.get("PNR_NET_CHAINS", "1") != "0"
        self._net_streams = None
        self._side = None           # the stream the next epoch's gather runs on (HIP path)

    def entropy_coeff(self) -> float:
        frac = min(1.0, self.timesteps_total / max(1, self.cfg.entropy_decay_steps))
        return self.cfg.entropy_coeff_start * (1.0 - frac)

    def loss(self, mb: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """The torch formulation (float32, autograd): what pnr_mlp_train_step's fused kernel is tested against."""
        cfg = self.cfg
        mean, log_std, v = self.model(mb["obs"])
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
        self._kl_c.fill_(self.kl_coeff)
        self._ent_c.fill_(self.entropy_coeff())
        tens = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor) and k != "adv_stats"}
        if self.hip:
            if not adv.is_cuda:
                raise AssertionError("this learner runs on the HIP kernels: its batches must live on the device")
            # (the kernels standardise while they pack the record, pnr_ppo_pack_record: the same two float32 operations)
            m = self._update_hip(tens, batch.get("filt"), B, mbs, (mu.float().reshape(1), (sd.float() + 1e-8).reshape(1)))
            # PPOTrainer reads the means back together with the episode statistics (one host synchronisation per iteration)
            return self.finish_update_values(m.tolist()) if readback else m
        tens["adv"] = (adv - mu.float()) / (sd.float() + 1e-8)
        agg: Dict[str, torch.Tensor] = {}
        nmb = 0
        for _ in range(cfg.num_sgd_iter):
            perm = torch.randperm(B, device=adv.device, generator=generator)
            for s in range(0, B - mbs + 1, mbs):
                idx = perm[s:s + mbs]
                loss, info = self.loss({k: v[idx] for k, v in tens.items()})
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                pdist.allreduce_mean_grads(self.model.parameters())   # one flat 0.82 MB bucket
                if cfg.grad_clip:
                    nn.utils.clip_grad_norm_(self.model.parameters(), cfg.grad_clip)
                self.opt.step()
                for k, v in info.items():
                    agg[k] = agg.get(k, 0) + v
                nmb += 1
        m = torch.stack([agg[k] for k in ("policy_loss", "vf_loss", "kl", "entropy", "total_loss")]).double() / max(1, nmb)
        if pdist.is_dist():                                         # the reported losses are averages over the ranks
            m = pdist.allreduce_sum_(m) / pdist.dist.get_world_size()
        return self.finish_update_values(m.tolist())

    def finish_update_values(self, m) -> Dict[str, float]:
        """The loss means (already averaged over the ranks) as the result columns + the adaptive-KL step (RLlib PPO: update_kl)."""
        out = {"policy_loss": m[0], "vf_loss": m[1], "kl": m[2], "entropy": m[3], "total_loss": m[4]}
        cfg = self.cfg
        if out["kl"] > 2.0 * cfg.kl_target:
            self.kl_coeff *= 1.5
        elif out["kl"] < 0.5 * cfg.kl_target:
            self.kl_coeff *= 0.5
        out["cur_kl_coeff"] = self.kl_coeff
        out["entropy_coeff"] = self.entropy_coeff()
        return out

    def optimizer_state(self):
        """torch.optim.Adam's state_dict on the torch path; on the HIP path Adam's moments as the fused kernel keeps them
        (padded gradient layout) and its update count."""
        if self.hip:
            m, v, step = self.hip_mlp(1).adam_state()
            return {"hip_adam": {"m": m, "v": v, "step": step, "epochs": self._epochs}}
        return self.opt.state_dict()

    def load_optimizer_state(self, sd) -> None:
        """A checkpoint restores into the formulation that wrote it: silently dropping Adam's moments (the other layout)
        would restart their bias correction, so a mismatch raises."""
        if ("hip_adam" in sd) != self.hip:
            raise AssertionError(f"the checkpoint holds the {'HIP kernels' if 'hip_adam' in sd else 'torch'} learner's optimiser state, "
                                 f"this learner runs on {'the HIP kernels' if self.hip else 'torch'}: restore it with "
                                 f"PPOConfig(hip_kernels={'hip_adam' in sd})")
        if self.hip:
            for dst, k in zip(self.hip_mlp(1).adam_state(), ("m", "v", "step")):
                dst.copy_(sd["hip_adam"][k])
            self._epochs = int(sd["hip_adam"].get("epochs", 0))
            self._hip_dirty = True           # the master weights changed behind the packed bf16 copies
        else:
            self.opt.load_state_dict(sd)

    # -- the HIP path: every minibatch update is pnr_mlp_train_step, three launches, no autograd, no hipGraph needed ----
    def hip_mlp(self, batch: int):
        if self._mlp is None or self._mlp.max_batch < batch:
            from .mlp import HipMLP
            old = self._mlp
            self._mlp = HipMLP(self.model, batch, self.device, planes=self.cfg.mlp_planes())
            if old is not None:                                   # keep the optimiser state across a workspace resize
                for a, b in zip(self._mlp.adam_state(), old.adam_state()):
                    a.copy_(b)
            self._hip_dirty = True
        return self._mlp

    def _update_hip(self, tens, filt, B, mbs, adv_scalars) -> torch.Tensor:
        """The minibatch loop on the hand-written kernels.  Per iteration the record is packed into 96-byte rows, per epoch
        the shuffle is drawn and applied (the updates then read contiguous rows); the float32 master parameters are updated
        in place by the fused reduction + Adam kernel, which also refreshes the packed bf16 weights for the next forward.
        Several ranks: the reduced gradient goes to a flat bucket, is all-reduced (RCCL), and pnr_mlp_adam applies the mean —
        per net, as two chains on two streams (net_chains), or as one bucket after both nets' kernels."""
        cfg, dev = self.cfg, self.device
        mlp = self.hip_mlp(mbs)
        if self._hip_dirty:
            mlp.pack()
            self._hip_dirty = False
        rec = {k: v.contiguous() for k, v in tens.items()}
        total = cfg.num_sgd_iter * len(range(0, B - mbs + 1, mbs))
        if self._means is None or self._means.shape[0] != total:
            self._means = torch.zeros((total, 8), dtype=torch.float32, device=dev)
        multi = pdist.is_dist()
        world = pdist.dist.get_world_size() if multi else 1
        if multi and self._flat_grad is None:
            self._flat_grad = torch.zeros(int(mlp.lib.pnr_mlp_grad_floats()), dtype=torch.float32, device=dev)
        if self._perm is None or self._perm.shape[1] != B:
            self._perm = torch.empty((2, B), dtype=torch.int64, device=dev)
        rows = mlp.pack_record(rec, *adv_scalars)
        chains = multi and self.net_chains
        cur = torch.cuda.current_stream(dev)
        if chains:
            if self._net_streams is None:
                self._net_streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
                self._means2 = None
            if self._means2 is None or self._means2.shape[1] != total:
                self._means2 = torch.zeros((2, total, 8), dtype=torch.float32, device=dev)
            mlp.sync_step_counters(True)
        ge = int(mlp.lib.pnr_mlp_grad_floats()) // 2
        k = 0
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side

        def gather(e):
            # an epoch's shuffle: pnr_permutation keyed by (seed, rank, epoch counter) — one launch, no sort — and its application
            perm = hip_permutation(B, cfg.seed * 1000003 + (pdist.dist.get_rank() if multi else 0), self._epochs, self._perm[e % 2])
            self._epochs += 1
            return mlp.gather_epoch(rec["obs"], perm, filt, None, rec_rows=rows, xs_rows=rec.get("xs"), slot=e % 2)

        xs_rows = (lambda g_, s_: g_["xs"][s_:s_ + mbs]) if mlp.planes == 1 else (lambda g_, s_: g_["xs"][:, s_:s_ + mbs])
        nxt, nxt_ready = gather(0), None
        for e in range(cfg.num_sgd_iter):
            g = nxt
            if nxt_ready is not None:
                cur.wait_event(nxt_ready)
            if e + 1 < cfg.num_sgd_iter:
                # The next epoch's gather depends on the rollout and the shuffle, not on the weights: it runs on a side stream
                # under this epoch's updates, into the other buffer set (whose last readers, epoch e - 1's updates, are already
                # enqueued on the main stream in front of the event the side stream waits for).  102 us per epoch off the chain.
                ev = torch.cuda.Event()
                ev.record(cur)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    nxt = gather(e + 1)
                    nxt_ready = torch.cuda.Event()
                    nxt_ready.record(side)
            if chains:
                # The two nets share nothing but their input, so each is its own SGD chain on its own stream: fused kernel ->
                # weight gradients -> all-reduce of ITS half of the bucket -> Adam.  One net's all-reduce (latency-bound: 0.43 MB
                # over xGMI) then runs under the other net's kernels instead of idling the GPU 2 x per update.  Same arithmetic
                # per element as the one-bucket form: the weights come out bit-identical (tests/test_gpu_multirank.py).
                ev = torch.cuda.Event()
                ev.record(cur)
                for st in self._net_streams:
                    st.wait_event(ev)
                # (enqueued alternately: the process group runs its collectives in call order, so net 0's all-reduce k must
                # not queue behind all of net 1's epoch)
                for s in range(0, B - mbs + 1, mbs):
                    for net, st in enumerate(self._net_streams):
                        with torch.cuda.stream(st):
                            mlp.train_step(None, None, None, {k_: g[k_][s:s + mbs] for k_ in mlp.REC_KEYS}, self._kl_c, self._ent_c,
                                           cfg.clip_param, cfg.vf_clip_param, cfg.vf_loss_coeff, self._means2[net, k], cfg.lr,
                                           flat_grad=self._flat_grad, xs_in=xs_rows(g, s), nets=(net, 1))
                            pdist.allreduce_sum_(self._flat_grad[net * ge:(net + 1) * ge])
                            mlp.adam(self._flat_grad, 1.0 / world, cfg.lr, nets=(net, 1))
                    k += 1
                for st in self._net_streams:                        # the next epoch's gather overwrites what the chains read
                    e2 = torch.cuda.Event()
                    e2.record(st)
                    cur.wait_event(e2)
                continue
            for s in range(0, B - mbs + 1, mbs):
                mlp.train_step(None, None, None, {k_: g[k_][s:s + mbs] for k_ in mlp.REC_KEYS}, self._kl_c, self._ent_c, cfg.clip_param,
                               cfg.vf_clip_param, cfg.vf_loss_coeff, self._means[k], cfg.lr,
                               flat_grad=self._flat_grad if multi else None, xs_in=xs_rows(g, s))
                if multi:
                    pdist.allreduce_sum_(self._flat_grad)          # the one 0.86 MB bucket
                    mlp.adam(self._flat_grad, 1.0 / world, cfg.lr)
                k += 1
        m = (self._means2.sum(0) if chains else self._means).mean(0).double()
        if multi:                                                   # the reported losses are averages over the ranks
            m = pdist.allreduce_sum_(m) / world
        return m                                                    # device tensor [8]


# other threads (the RCCL watchdog of torch.distributed) may touch the HIP runtime while this thread captures
_CAPTURE_MODE = "thread_local"


class PPOTrainer:
    """Rollout + learn loop over a PioneerVectorEnv shard (one process per GPU).

    With the reference's nets (``PPOLearner.hip``) everything runs on the hand-written kernels.  Kinematic mode: the
    sampler's T steps are ONE resident launch (pnr_ppo_rollout: a workgroup owns 64 envs — both nets on the RAW
    observation, the MeanStdFilter applied on load, the action draw and clip in the policy net's epilogue, the env step
    — for the whole rollout); dynamics mode: per step two launches, pnr_mlp_act and pnr_step, with the same arithmetic.
    Then one pnr_ppo_gae launch for log-probs, GAE and the episode statistics; ``use_graph=True`` replays the whole
    sampling phase from ONE hipGraph after an eager warm-up iteration.  The torch formulation (``hip_kernels=False`` or other net
    shapes) samples eagerly with float32 torch ops."""

    def __init__(self, env, cfg: Optional[PPOConfig] = None, use_graph: bool = False):
        self.env = env
        self.cfg = cfg or PPOConfig()
        self.device = env.device
        self.rank, _, self.world = pdist.world_info()
        self.learner = PPOLearner(self.cfg, self.device)
        self.hip = self.learner.hip
        self.filter = (MeanStdFilter(self.cfg.obs_dim, self.device, self.cfg.filter_clip)
                       if self.cfg.observation_filter in ("MeanStdFilter", "ConcurrentMeanStdFilter") else NoFilter())
        self.stats = EpisodeStats(env.num_envs, self.device)
        self._ev = None             # timing events of train()
        self.gen = torch.Generator(device=self.device).manual_seed(self.cfg.seed * 1000003 + self.rank)
        self.a_max = torch.from_numpy(env.a_max).to(self.device)
        self.iteration = 0
        self.use_graph = bool(use_graph) and self.hip
        # kinematic mode with env-major layouts: the sampler's T steps are one resident launch (dynamics mode keeps the per-step
        # pnr_mlp_act + pnr_step pair); False forces the per-step form (the A/B and the equality test)
        ec = env.engine_config
        self.resident_rollout = self.hip and ec.mode == "kinematic" and ec.obs_layout == "env_major" and ec.action_layout == "env_major" \
            and self.cfg.mlp_planes() == 1          # (the resident kernel keeps bf16 weights in registers; split operands sample per step)
        self._graph = None
        self._eager_collects = 0        # eager (uncaptured) collects done by this trainer object: gates the graph capture
        T, N, D, A = self.cfg.rollout_fragment_length, env.num_envs, self.cfg.obs_dim, self.cfg.act_dim
        f32 = dict(dtype=torch.float32, device=self.device)
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
            "terminals": torch.empty((T, N), **f32),                   # done | truncated as 0 / 1
        }
        if self.hip:
            from .mlp import HipMLP
            self.sample_mlp = HipMLP(self.learner.model, N, self.device, planes=self.cfg.mlp_planes())   # packed weights for the rollout's T forwards
            self._last_heads = torch.empty((2, N, 16), **f32)
            self._last_v = torch.empty((N,), **f32)
            # the nets' inputs as the sampler saw them (filtered, bf16): what the learner trains on, 288 bytes per sample
            if self.cfg.mlp_planes() == 1:
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

    def _step_env(self, t: int, act: torch.Tensor) -> None:
        buf = self.buf
        self.env.vector_step(act, out={"obs": self.raw_in[t + 1], "reward": buf["reward"][t], "done": buf["done"][t],
                                       "truncated": buf["trunc"][t]})

    @torch.no_grad()
    def _collect_impl(self) -> None:
        """T steps into the static buffers; pure device work (capturable on the HIP path).  Log-probs, GAE and the episode
        statistics come from ONE launch after the loop; the filter's moments in collect(), outside any capture."""
        cfg, buf = self.cfg, self.buf
        T, clip = cfg.rollout_fragment_length, cfg.clip_actions
        self.raw_in[0].copy_(self.raw_in[T])
        self.filter.prepare()
        shape = tuple(buf["actions"].shape)
        # in-graph noise comes from the default (graph-safe) generator
        noise = torch.randn(shape, device=self.device) if self._capturing else torch.randn(shape, generator=self.gen, device=self.device)
        if self.hip:
            mlp, filt = self.sample_mlp, self._filt()
            mlp.pack()
            if self.resident_rollout:
                # the whole closed loop in ONE launch: a workgroup owns 64 envs for all T steps (pnr_ppo_rollout)
                mlp.rollout(self.env, filt, noise, self.a_max if clip else None, obs=self.raw_in, mean=buf["mean"], log_std=buf["log_std"],
                            values=buf["values"], actions=buf["actions"], reward=buf["reward"], done=buf["done"], truncated=buf["trunc"],
                            xs_out=buf["xs"])
            xs = buf.get("xs")                              # (split operands: the learner re-makes its input planes from the observations)
            for t in range(0 if not self.resident_rollout else T, T):
                mlp.act(self.raw_in[t], filt, noise[t], self.a_max if clip else None, mean=buf["mean"][t], log_std=buf["log_std"][t],
                        values=buf["values"][t], actions=buf["actions"][t], env_actions=self._env_act if clip else None,
                        xs_out=xs[t] if xs is not None else None)
                self._step_env(t, self._env_act if clip else buf["actions"][t])
            last = mlp.forward_nograd(self.raw_in[T], None, filt, out=self._last_heads)
            self._last_v.copy_(last[1, :, 0])           # bootstrap value of the state after the last step
            hip_gae_logp(buf["reward"], buf["values"], self._last_v, buf["done"], buf["trunc"], buf["actions"], buf["mean"],
                         buf["log_std"], cfg.gamma, cfg.lambda_, logp=buf["logp"], adv=buf["adv"], vtarg=buf["vtarg"],
                         terminals=buf["terminals"], stats=self.stats, adv_stats=self._adv_stats)
            return
        model = self.learner.model
        for t in range(T):
            mean, log_std, v = model(self.filter.apply_(self.raw_in[t], out=buf["obs"][t]))
            buf["mean"][t].copy_(mean); buf["log_std"][t].copy_(log_std); buf["values"][t].copy_(v)
            act = torch.addcmul(mean, torch.exp(log_std), noise[t], out=buf["actions"][t])
            self._step_env(t, torch.clamp(act, -self.a_max, self.a_max, out=self._env_act) if clip else act)
            term = (buf["done"][t] | buf["trunc"][t]).to(torch.float32)
            buf["terminals"][t].copy_(term)
            self.stats.step(buf["reward"][t], term)
        last_v = model(self.filter.apply_(self.raw_in[T], out=torch.empty_like(self.raw_in[T])))[2]
        buf["logp"].copy_(gaussian_logp(buf["actions"], buf["mean"], buf["log_std"]))
        adv, vtarg = compute_gae(buf["reward"], buf["values"], last_v, buf["terminals"], cfg.gamma, cfg.lambda_)
        buf["adv"].copy_(adv); buf["vtarg"].copy_(vtarg)

    _capturing = False

    def collect(self) -> Dict[str, torch.Tensor]:
        if self.use_graph and self._graph is None and self._eager_collects >= 1:
            # capture after one eager collect BY THIS OBJECT (allocator and library warm-up done: first launches of pnr_ppo_rollout /
            # pnr_ppo_gae / pnr_filter_prepare, the scratch allocations) — not `self.iteration`, which restore() sets from the
            # checkpoint: a fresh trainer that restored would otherwise capture on its very first collect
            torch.cuda.synchronize(self.device)
            self._capturing = True
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
            self._eager_collects += 1
        # The filter's moment pass runs EAGERLY after the (possibly replayed) loop, never inside a captured graph: a torch
        # reduction over the middle axis of a large tensor returned wrong sums from the second replay of a hipGraph on (r01's
        # "NaNs after graph replay"; layout-dependent, cause not pinned below torch: tools/graph_reduce_probe.py,
        # profiles/r02_graph_reduce_probe.json), and nothing captured here reduces with torch any more.  The filter only changes
        # at sync(), so observing all inputs in one pass equals observing them one by one inside the loop.
        buf, T = self.buf, self.cfg.rollout_fragment_length
        self.filter.observe(self.raw_in[:T])
        flat = lambda x: x.reshape(-1, *x.shape[2:])  # noqa: E731
        batch = {"actions": flat(buf["actions"]), "mean": flat(buf["mean"]), "log_std": flat(buf["log_std"]),
                 "logp": flat(buf["logp"]), "values": flat(buf["values"]), "adv": flat(buf["adv"]), "vtarg": flat(buf["vtarg"])}
        if self.hip:
            batch.update(obs=flat(self.raw_in[:T]), filt=self._filt())      # raw: the kernels filter on load
            batch["adv_stats"] = self._adv_stats
            if "xs" in buf:
                batch["xs"] = flat(buf["xs"])
        else:
            batch["obs"] = flat(buf["obs"])
        return batch

    def train(self) -> Dict[str, float]:
        """One iteration: collect, merge the filter, update.  The phase split (sample_time_s / learn_time_s, RLlib's
        sample_time_ms / learn_time_ms) comes from an event recorded between the phases, not from a host synchronisation there:
        the learner's first kernels are queued while the sampler's last ones still run."""
        t0 = time.perf_counter()
        if self._ev is None:
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self._ev[0].record()
        batch = self.collect()
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
        torch.cuda.synchronize(self.device)
        t2 = time.perf_counter()
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
              "num_envs": int(self.env.num_envs),
              # this rank's running episode accumulators and its noise generator: the rollout continues where it stopped
              "stats_ret": self.stats.ret.cpu(), "stats_len": self.stats.len.cpu(), "gen_state": self.gen.get_state()}
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
        """Weights, optimiser state, filter and counters are copied IN PLACE (a captured sampling graph keeps seeing them).
        With restore_env the env shards, the running episode accumulators and the noise generator come back too: the run
        then continues exactly as the uninterrupted one would have (eager sampling; a replayed graph draws its noise from
        the default generator)."""
        ck = torch.load(path, map_location=self.device, weights_only=True)    # tensors and plain values only
        self.learner.model.load_state_dict(ck["model"]); self.learner.load_optimizer_state(ck["opt"])
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
            if "stats_ret" in est:
                self.stats.ret.copy_(est["stats_ret"]); self.stats.len.copy_(est["stats_len"])
                self.gen.set_state(est["gen_state"].cpu())
