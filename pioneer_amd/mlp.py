"""The PPO driver's two MLPs on the hand-written bf16 MFMA kernels of csrc/pnr_mlp.h (C ABI: pnr_mlp_*).

``planes`` selects the operands' precision (include/pioneer_amd.h, pnr_mlp_pack): 1 = bf16 (reduced precision), 2 = every float32
operand as two power-of-two-scaled fp16 planes (22 significant bits: the accuracy of a float32 GEMM — what the reference's torch
learner computes in — at three MFMAs per product), 3 = three bf16 planes (24 bits, six MFMAs per product).

``HipMLP`` owns the packed bf16 weights and the activation / gradient workspaces for one ``ActorCritic`` and exposes

* ``forward_nograd``  the sampling path: heads of a batch of RAW observations, the MeanStdFilter applied on load;
* ``apply``           the learner path: an autograd function (forward kernel saves the activations; backward =
                      backward-data + weight-gradient + reduction kernels) whose outputs are the two nets' raw head rows
                      ``[B, 16]`` — what ``pnr_ppo_loss`` consumes — and whose gradients land on the float32 master
                      parameters of the module, so torch's Adam (or any optimiser) keeps working unchanged.

The master parameters stay float32 ``nn.Linear`` tensors (checkpoints, CPU tests, the reference's [256, 256] tanh nets of
pioneer/launch/pioneer_knm_train.py:59-61); there is no fallback: without the HIP library this module raises.
"""
import ctypes as C
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib

HEAD = 16
IN_PAD = 144
HID = 256


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# PPOConfig.hip_kernels -> 16-bit planes per MFMA operand: 1 = bf16 (reduced precision); 2 = two scaled fp16 planes (float32-class:
# 22 significant bits, three MFMAs per product — "f32"); 3 = three bf16 planes (24 bits, six MFMAs per product — "bf16x3")
PLANES = {True: 1, "bf16": 1, "f32": 2, "bf16x3": 3}


class HipMLP:
    def __init__(self, model: nn.Module, max_batch: int, device, planes: int = 1):
        self.lib = _lib.load_library()
        self.planes = int(planes)
        if self.planes not in (1, 2, 3):
            raise AssertionError(f"planes must be 1, 2 or 3, got {planes}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise AssertionError("HipMLP needs a HIP device")
        self.model = model
        self.max_batch = int(max_batch)
        pol = [l for l in model.policy if isinstance(l, nn.Linear)]
        val = [l for l in model.value if isinstance(l, nn.Linear)]
        for lins in (pol, val):
            shapes = [tuple(l.weight.shape) for l in lins]
            if len(lins) != 3 or shapes[0] != (HID, 137) or shapes[1] != (HID, HID) or shapes[2][1] != HID or shapes[2][0] > HEAD:
                raise AssertionError(f"HipMLP is built for 137-256-256-n (n <= 16) nets, got {shapes}")
        self.n3 = (pol[2].out_features, val[2].out_features)
        self.params = [t for lins in (pol, val) for l in lins for t in (l.weight, l.bias)]     # 12, net-major
        assert all(p.dtype == torch.float32 and p.is_contiguous() and p.device == self.device for p in self.params)
        bf, f32 = dict(dtype=torch.bfloat16, device=self.device), dict(dtype=torch.float32, device=self.device)
        self.wpack = torch.empty(self.planes * int(self.lib.pnr_mlp_pack_elems()), **bf)
        self.bias = torch.empty(int(self.lib.pnr_mlp_bias_elems()), **f32)
        B = self.max_batch
        self._train_ws = None

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptrs(tensors: Sequence[torch.Tensor]):
        """A ctypes array of device pointers (read by the library during the call only)."""
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _workspace(self):
        if self._train_ws is None:
            B = self.max_batch
            bf, f32 = dict(dtype=torch.bfloat16, device=self.device), dict(dtype=torch.float32, device=self.device)
            P = self.planes
            self._train_ws = {
                "xs": torch.empty((B, IN_PAD), **bf),
                # (planes > 1: H2 is never stored — the fused kernel makes layer 3's products itself — so one plane of it is enough)
                "h1": torch.empty((P, 2, B, HID), **bf), "h2": torch.empty((2, B, HID), **bf),
                "dz1": torch.empty((P, 2, B, HID), **bf), "dz2": torch.empty((P, 2, B, HID), **bf),
                "slabs": torch.empty(int(self.lib.pnr_mlp_slab_floats(B)), **f32),
                "head": torch.empty(2 * B * HEAD, **f32), "g": torch.empty(2 * B * HEAD, **f32),
                "partials": torch.empty((2 * ((B + 63) // 64), 8), **f32),      # one row per (64-sample tile, net)
                "w3part": torch.empty(int(self.lib.pnr_mlp_w3_partial_floats(B)), **f32),   # layer 3's weight-gradient products per tile
            }
        return self._train_ws

    def pack(self) -> None:
        """bf16 copies of the current master weights (padded, plus the transposes the backward pass reads)."""
        _lib.check(self.lib.pnr_mlp_pack(self._ptrs(self.params), self.n3[0], self.n3[1], _p(self.wpack), _p(self.bias),
                                         self.planes, self._stream()))

    def _launch_forward(self, B, obs, idx, filt, head, save):
        if save and self.planes != 1:
            raise AssertionError("the autograd binding (saved activations + pnr_mlp_backward) is bf16-only: use train_step for planes > 1")
        ws = self._workspace() if save else None
        f = filt if filt is not None else (None, None, None, None)
        self._batch_of_ws(B)       # the kernels index the saved activations as [2][B][256] with the CURRENT batch size
        _lib.check(self.lib.pnr_mlp_forward(B, _p(obs), _p(idx), _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(self.wpack), _p(self.bias),
                                            _p(head), _p(ws["xs"]) if ws else None, _p(ws["h1"]) if ws else None,
                                            _p(ws["h2"]) if ws else None, 0, 2, self.planes, self._stream()))

    def _batch_of_ws(self, B):
        if B > self.max_batch:
            raise AssertionError(f"batch {B} exceeds the workspace ({self.max_batch})")
        return B

    @staticmethod
    def _check_inputs(obs, idx, filt, device):
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[-1] == 137 and obs.device == device
        if idx is not None:
            assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.device == device
        if filt is not None:
            assert len(filt) == 4 and all(v.dtype == torch.float32 and v.numel() == 137 and v.is_contiguous() for v in filt)

    # -- the sampling path ------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_nograd(self, obs: torch.Tensor, idx: Optional[torch.Tensor] = None, filt=None,
                       out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Heads [2, B, 16] (policy rows: means 0..5, raw log-stds 6..11; value rows: v at column 0) of ``obs[idx]``
        (or ``obs``) on the weights of the last ``pack()``."""
        self._check_inputs(obs, idx, filt, self.device)
        B = int(idx.numel()) if idx is not None else int(obs.numel() // 137)
        head = out if out is not None else torch.empty((2, B, HEAD), dtype=torch.float32, device=self.device)
        assert head.is_contiguous() and tuple(head.shape) == (2, B, HEAD)
        self._launch_forward(B, obs, idx, filt, head, save=False)
        return head

    @torch.no_grad()
    def act(self, obs: torch.Tensor, filt, noise: torch.Tensor, a_max: Optional[torch.Tensor], *, mean: torch.Tensor,
            log_std: torch.Tensor, values: torch.Tensor, actions: torch.Tensor, env_actions: Optional[torch.Tensor] = None,
            head: Optional[torch.Tensor] = None, xs_out: Optional[torch.Tensor] = None) -> None:
        """One sampler step in ONE launch (pnr_mlp_act): both nets on ``obs`` [B, 137] and the DiagGaussian draw
        ``actions = mean + exp(clamp(log_std, -20, 2)) * noise`` in the policy net's epilogue; ``env_actions`` =
        ``actions`` clipped to +-``a_max`` (what the env is stepped with) when ``a_max`` is given."""
        self._check_inputs(obs, None, filt, self.device)
        B = int(obs.numel() // 137)
        f = filt if filt is not None else (None, None, None, None)
        for name, x, shape in (("noise", noise, (B, 6)), ("mean", mean, (B, 6)), ("log_std", log_std, (B, 6)), ("values", values, (B,)),
                               ("actions", actions, (B, 6)), ("env_actions", env_actions, (B, 6)), ("head", head, (2, B, HEAD)),
                               ("a_max", a_max, (6,))):
            if x is not None:
                assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == shape and x.device == self.device, name
        assert (a_max is None) or (env_actions is not None and env_actions.data_ptr() != actions.data_ptr())
        assert xs_out is None or (self.planes == 1 and xs_out.dtype == torch.bfloat16 and xs_out.is_contiguous() and tuple(xs_out.shape) == (B, 144))
        _lib.check(self.lib.pnr_mlp_act(B, _p(obs), _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(self.wpack), _p(self.bias), _p(noise),
                                        _p(a_max), _p(head), _p(mean), _p(log_std), _p(values), _p(actions), _p(env_actions),
                                        _p(xs_out), self.planes, self._stream()))

    @torch.no_grad()
    def rollout(self, env, filt, noise: torch.Tensor, a_max: Optional[torch.Tensor], *, obs: torch.Tensor, mean: torch.Tensor,
                log_std: torch.Tensor, values: torch.Tensor, actions: torch.Tensor, reward: torch.Tensor, done: torch.Tensor,
                truncated: Optional[torch.Tensor], xs_out: Optional[torch.Tensor] = None) -> None:
        """The sampler's closed loop for T steps in ONE resident launch (pnr_ppo_rollout): per step both nets on slot t of ``obs``
        [T + 1, N, 137], the draw and clip of ``act``, and the step of ``env`` (a kinematic-mode PioneerVectorEnv with env-major
        layouts) with that action — reward / done / truncated [T, N], the next observation into slot t + 1.  Equals T x (act,
        env.vector_step) bit for bit; the env's state advances as if they had been called."""
        assert self.planes == 1, "the resident rollout kernel keeps bf16 weights in registers: planes > 1 samples with act() + vector_step()"
        T, N = int(noise.shape[0]), int(env.num_envs)
        f = filt if filt is not None else (None, None, None, None)
        for name, x, shape in (("noise", noise, (T, N, 6)), ("mean", mean, (T, N, 6)), ("log_std", log_std, (T, N, 6)), ("values", values, (T, N)),
                               ("actions", actions, (T, N, 6)), ("obs", obs, (T + 1, N, 137)), ("reward", reward, (T, N)), ("a_max", a_max, (6,))):
            if x is not None:
                assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == shape and x.device == self.device, name
        for name, x in (("done", done), ("truncated", truncated)):
            assert x is None or (x.dtype == torch.uint8 and x.is_contiguous() and tuple(x.shape) == (T, N) and x.device == self.device), name
        for v in f:
            assert v is None or (v.dtype == torch.float32 and v.is_contiguous() and v.numel() == 137 and v.device == self.device)
        assert xs_out is None or (xs_out.dtype == torch.bfloat16 and xs_out.is_contiguous() and tuple(xs_out.shape) == (T, N, 144))
        _lib.check(self.lib.pnr_ppo_rollout(env._h, T, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(self.wpack), _p(self.bias), _p(noise),
                                            _p(a_max), _p(obs), _p(mean), _p(log_std), _p(values), _p(actions), _p(xs_out), _p(reward),
                                            _p(done), _p(truncated), self._stream()))

    # -- the learner path -------------------------------------------------------------------------------------
    def apply(self, obs: torch.Tensor, idx: Optional[torch.Tensor] = None, filt=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(head_policy [B, 16], head_value [B, 16]) with autograd through the HIP backward kernels.  Packs the
        current weights first.  One application at a time: the saved activations live in this object's workspace."""
        self._check_inputs(obs, idx, filt, self.device)
        out = _FusedMLP.apply(self, obs, idx, filt, *self.params)
        return out[0], out[1]


    # -- the learner path without autograd: one PPO minibatch update = pnr_mlp_train_step (three launches) ------
    def adam_state(self):
        """(m, v, step): Adam's moments in the padded gradient layout [pnr_mlp_grad_floats()] and the update count."""
        if not hasattr(self, "_adam"):
            n = int(self.lib.pnr_mlp_grad_floats())
            f32 = dict(dtype=torch.float32, device=self.device)
            self._adam = (torch.zeros(n, **f32), torch.zeros(n, **f32), torch.zeros((), **f32))
            self._step_value_net = torch.zeros((), **f32)      # the value net's own count when the nets run as two chains
        return self._adam

    def _step_args(self, lr, betas, eps, nets=None):
        """nets = (first_net, n_nets) or None for both.  Every train_step call counts one update in its counter, so the value
        net's chain (nets == (1, 1)) counts in a counter of its own; it follows the common one whenever both nets step together."""
        s = _lib.PnrMlpStep()
        s.struct_size = C.sizeof(_lib.PnrMlpStep)
        for k, prm in enumerate(self.params):
            s.params[k] = prm.data_ptr()
        s.n3_policy, s.n3_value = self.n3
        m, v, step = self.adam_state()
        s.wpack, s.bias = self.wpack.data_ptr(), self.bias.data_ptr()
        s.adam_m, s.adam_v = m.data_ptr(), v.data_ptr()
        s.adam_step = (self._step_value_net if nets == (1, 1) else step).data_ptr()
        s.lr, s.beta1, s.beta2, s.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        if nets is not None:
            s.first_net, s.n_nets = int(nets[0]), int(nets[1])
        return s

    def sync_step_counters(self, to_chains: bool) -> None:
        """Two-chain mode keeps one update count per net: copied from the common count before the chains start, and back (they
        agree) after they have joined."""
        _, _, step = self.adam_state()
        if to_chains:
            self._step_value_net.copy_(step)

    REC_KEYS = ("actions", "logp", "mean", "log_std", "adv", "vtarg", "values")
    w3_partials = True          # train_step: layer 3's weight gradients per tile from the fused kernel (pnr_mlp_step.w3_partials)

    def pack_record(self, rec, adv_mu: Optional[torch.Tensor] = None, adv_den: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The rollout record as one 24-float row per sample (pnr_ppo_pack_record), advantages standardised as
        (adv - adv_mu) / adv_den when the two float32 device scalars are given.  Feed it to ``gather_epoch(rec_rows=...)``."""
        R = int(rec["actions"].shape[0])
        for k in self.REC_KEYS:
            v = rec[k]
            assert v.dtype == torch.float32 and v.is_contiguous() and v.device == self.device and v.shape[0] == R, k
        for x in (adv_mu, adv_den):
            assert x is None or (x.dtype == torch.float32 and x.numel() == 1 and x.device == self.device)
        if getattr(self, "_rec_rows", None) is None or self._rec_rows.shape[0] < R:
            self._rec_rows = torch.empty((R, 24), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pnr_ppo_pack_record(R, *[_p(rec[k]) for k in self.REC_KEYS], _p(adv_mu), _p(adv_den), _p(self._rec_rows),
                                                self._stream()))
        return self._rec_rows[:R]

    def gather_epoch(self, obs, idx, filt, rec, rec_rows: Optional[torch.Tensor] = None, xs_rows: Optional[torch.Tensor] = None,
                     slot: int = 0) -> dict:
        """An SGD epoch's shuffle applied ONCE (pnr_mlp_gather): returns {"xs": bf16 [n, 144] — the filtered, rounded net
        inputs of rows ``idx`` —, and the record fields gathered the same way}.  Slices of it feed ``train_step(xs_in=...)``.
        The buffers live in this object (one set per ``slot``: the learner gathers epoch e + 1 into the other set while epoch e's
        updates read this one) and are overwritten by the next call with the same slot."""
        if xs_rows is not None:       # the sampler's saved net inputs (act(xs_out=...)): copied, nothing is filtered
            assert self.planes == 1, "planes > 1: the input planes are made from the float32 observations"
            assert xs_rows.dtype == torch.bfloat16 and xs_rows.is_contiguous() and xs_rows.shape[1] == 144 and xs_rows.device == self.device
            assert idx.dtype == torch.int64 and idx.is_contiguous()
            obs_p, R, filt = None, int(xs_rows.shape[0]), None
        else:
            self._check_inputs(obs, idx, filt, self.device)
            obs_p, R = _p(obs), obs.shape[0]
        n = int(idx.numel())
        sets = getattr(self, "_gathered", None)
        if sets is None:
            sets = self._gathered = {}
        g = sets.get(slot)
        if g is None or g["xs"].shape[-2] != n:
            f32 = dict(dtype=torch.float32, device=self.device)
            # planes > 1: [planes, n, 144] — a minibatch is the slice [:, s:s + mbs] (train_step(xs_in=...) takes the strided view's rows
            # through a per-minibatch plane-major copy, see train_step)
            g = {"xs": torch.empty((n, 144) if self.planes == 1 else (self.planes, n, 144), dtype=torch.bfloat16, device=self.device)}
            for k in self.REC_KEYS:
                g[k] = torch.empty((n, 6) if k in ("actions", "mean", "log_std") else (n,), **f32)
            sets[slot] = g
        f = filt if filt is not None else (None, None, None, None)
        if rec_rows is not None:
            assert rec_rows.dtype == torch.float32 and rec_rows.is_contiguous() and tuple(rec_rows.shape) == (R, 24) and rec_rows.device == self.device
            src = [None] * len(self.REC_KEYS)
        else:
            for k in self.REC_KEYS:
                v = rec[k]
                assert v.dtype == torch.float32 and v.is_contiguous() and v.device == self.device and v.shape[0] == R, k
            src = [_p(rec[k]) for k in self.REC_KEYS]
        _lib.check(self.lib.pnr_mlp_gather(n, _p(idx), obs_p, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), *src, _p(g["xs"]),
                                           *[_p(g[k]) for k in self.REC_KEYS], _p(rec_rows), _p(xs_rows), self.planes, self._stream()))
        return {k: (v[:n] if (k != "xs" or self.planes == 1) else v) for k, v in g.items()}

    def train_step(self, obs, idx, filt, rec, kl_c, ent_c, clip: float, vf_clip: float, vf_coeff: float, means_out: torch.Tensor,
                   lr: float, betas=(0.9, 0.999), eps: float = 1e-8, flat_grad: Optional[torch.Tensor] = None,
                   xs_in: Optional[torch.Tensor] = None, nets=None) -> None:
        """Forward, loss, backward and — unless ``flat_grad`` is given — the Adam update of the float32 master
        parameters plus the refresh of the packed bf16 weights, all on the device (pnr_mlp_train_step).  With
        ``flat_grad`` ([pnr_mlp_grad_floats()]) the reduced gradient lands there instead and nothing is updated: all-reduce
        it and call ``adam(flat_grad, 1 / world)``.  ``pack()`` must have run once after the parameters last changed
        behind this object's back (construction, restore)."""
        if xs_in is not None:
            # the epoch's pre-gathered rows (gather_epoch): inputs and record are read row by row, nothing is filtered
            assert xs_in.dtype == torch.bfloat16 and xs_in.shape[-1] == 144 and xs_in.device == self.device
            assert obs is None and idx is None and filt is None
            if self.planes == 1:
                assert xs_in.dim() == 2 and xs_in.is_contiguous()
            else:
                # [planes, B, 144], possibly a slice [:, s:s + B] of an epoch's gathered planes: rows contiguous, planes stride(0) apart
                assert xs_in.dim() == 3 and xs_in.shape[0] == self.planes and xs_in.stride(2) == 1 and xs_in.stride(1) == 144
            B = R = int(xs_in.shape[-2])
        else:
            self._check_inputs(obs, idx, filt, self.device)
            B = int(idx.numel()) if idx is not None else int(obs.shape[0])
            R = obs.shape[0]
        self._batch_of_ws(B)
        ws = self._workspace()
        for k in ("actions", "logp", "mean", "log_std", "adv", "vtarg", "values"):
            v = rec[k]
            assert v.dtype == torch.float32 and v.is_contiguous() and v.device == self.device and v.shape[0] == R, k
        assert means_out.dtype == torch.float32 and means_out.numel() >= 8 and means_out.is_contiguous()
        s = self._step_args(lr, betas, eps, nets)
        s.batch = B
        if xs_in is not None:
            s.xs_in = xs_in.data_ptr()
        else:
            s.obs, s.idx = obs.data_ptr(), (idx.data_ptr() if idx is not None else None)
        if filt is not None:
            s.f_loc, s.f_inv, s.f_lo, s.f_hi = (t.data_ptr() for t in filt)
        s.actions, s.logp_old, s.mean_old, s.log_std_old = (rec[k].data_ptr() for k in ("actions", "logp", "mean", "log_std"))
        s.adv, s.value_target, s.value_old = (rec[k].data_ptr() for k in ("adv", "vtarg", "values"))
        s.kl_coeff, s.entropy_coeff = kl_c.data_ptr(), ent_c.data_ptr()
        s.clip_param, s.vf_clip_param, s.vf_loss_coeff = float(clip), float(vf_clip), float(vf_coeff)
        s.head, s.g_head = ws["head"].data_ptr(), ws["g"].data_ptr()
        s.xs, s.h1, s.h2, s.dz1, s.dz2 = (ws[k].data_ptr() for k in ("xs", "h1", "h2", "dz1", "dz2"))
        # one row of loss sums per (tile, net); a one-net call (two-chain mode: the other net's launch may run at the same time on
        # another stream) gets that net's half of the rows to itself
        part = ws["partials"]
        if nets is not None and tuple(nets) == (1, 1):
            part = part[part.shape[0] // 2:]
        s.partials, s.partial_rows = part.data_ptr(), part.shape[0]
        s.slabs, s.slab_floats = ws["slabs"].data_ptr(), ws["slabs"].numel()
        # the per-tile layer-3 partials: each net's one-net call gets its own half, like the loss-sum rows
        w3 = ws["w3part"]
        if nets is not None and tuple(nets) == (1, 1):
            w3 = w3[w3.numel() // 2:]
        if self.w3_partials or self.planes > 1:  # (False: H2 is stored and read back by the weight-gradient kernel — the A/B; same bits)
            s.w3_partials, s.w3_partial_floats = w3.data_ptr(), w3.numel()
        s.planes = self.planes
        if self.planes > 1:
            assert xs_in is not None, "planes > 1: train_step takes the pre-gathered input planes (gather_epoch)"
            s.xs_in_plane = int(xs_in.stride(0))
        s.means = means_out.data_ptr()
        if flat_grad is not None:
            assert flat_grad.dtype == torch.float32 and flat_grad.is_contiguous() and flat_grad.numel() == int(self.lib.pnr_mlp_grad_floats())
            s.flat_grad = flat_grad.data_ptr()
        _lib.check(self.lib.pnr_mlp_train_step(C.byref(s), self._stream()))

    def adam(self, flat_grad: torch.Tensor, grad_scale: float, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, nets=None) -> None:
        """Adam on the all-reduced bucket ``flat_grad`` ([pnr_mlp_grad_floats()], always the WHOLE bucket: ``nets`` selects the half
        that is applied)."""
        s = self._step_args(lr, betas, eps, nets)
        s.planes = self.planes
        _lib.check(self.lib.pnr_mlp_adam(C.byref(s), _p(flat_grad), C.c_float(grad_scale), self._stream()))

    def policy_loss(self, obs, idx, filt, rec, kl_c, ent_c, clip: float, vf_clip: float, vf_coeff: float) -> torch.Tensor:
        """The whole differentiable part of one PPO minibatch update in six launches: weight packing, the fused
        forward of both nets, the loss kernel (pnr_ppo_loss: values + d loss / d head) and its finishing sum; the
        backward pass (backward-data, weight gradients, reduction) runs when autograd reaches the returned tensor.
        Returns the batch means [8] = (policy_loss, vf_loss, kl, entropy, total, ...); differentiate means[4].
        ``rec``: the rollout record (actions [R, 6], logp, mean [R, 6], log_std [R, 6], adv, vtarg, values [R]); with
        ``idx`` (int64 [B]) sample i is row idx[i] of ``obs`` and of the record."""
        self._check_inputs(obs, idx, filt, self.device)
        R = obs.shape[0]
        for k in ("actions", "logp", "mean", "log_std", "adv", "vtarg", "values"):
            v = rec[k]
            assert v.dtype == torch.float32 and v.is_contiguous() and v.device == self.device and v.shape[0] == R, k
        assert tuple(rec["actions"].shape) == tuple(rec["mean"].shape) == tuple(rec["log_std"].shape) == (R, 6)
        return _HipPolicyLoss.apply(self, obs, idx, filt, rec, kl_c, ent_c, float(clip), float(vf_clip), float(vf_coeff),
                                    *self.params)


class _HipPolicyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp: HipMLP, obs, idx, filt, rec, kl_c, ent_c, clip, vf_clip, vf_coeff, *params):
        B = int(idx.numel()) if idx is not None else int(obs.shape[0])
        mlp._batch_of_ws(B)
        ws = mlp._workspace()
        head = ws["head"][:2 * B * HEAD].view(2, B, HEAD)
        g = ws["g"][:2 * B * HEAD].view(2, B, HEAD)
        mlp.pack()
        mlp._launch_forward(B, obs, idx, filt, head, save=True)
        rows = (B + 255) // 256
        means = torch.empty(8, dtype=torch.float32, device=mlp.device)
        _lib.check(mlp.lib.pnr_ppo_loss(B, _p(idx), _p(head[0]), _p(head[1]), _p(rec["actions"]), _p(rec["logp"]), _p(rec["mean"]),
                                        _p(rec["log_std"]), _p(rec["adv"]), _p(rec["vtarg"]), _p(rec["values"]), _p(kl_c),
                                        _p(ent_c), C.c_float(clip), C.c_float(vf_clip), C.c_float(vf_coeff), _p(g[0]), _p(g[1]),
                                        _p(ws["partials"]), rows, _p(means), mlp._stream()))
        ctx.mlp, ctx.B = mlp, B
        return means

    @staticmethod
    def backward(ctx, g_means):
        mlp, B = ctx.mlp, ctx.B
        ws = mlp._workspace()
        g_means = g_means.contiguous()
        assert g_means.dtype == torch.float32 and g_means.numel() == 8
        grads = [torch.empty_like(p) for p in mlp.params]
        scale = C.c_void_p(g_means.data_ptr() + 4 * 4)            # d / d means[4]: the total loss
        _lib.check(mlp.lib.pnr_mlp_backward(B, _p(ws["g"]), _p(mlp.wpack), _p(ws["xs"]), _p(ws["h1"]), _p(ws["h2"]), _p(ws["dz1"]),
                                            _p(ws["dz2"]), _p(ws["slabs"]), ws["slabs"].numel(), mlp._ptrs(grads),
                                            mlp.n3[0], mlp.n3[1], 0, scale, mlp._stream()))
        return (None,) * 10 + tuple(grads)


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp: HipMLP, obs, idx, filt, *params):
        B = int(idx.numel()) if idx is not None else int(obs.numel() // 137)
        mlp._batch_of_ws(B)
        mlp.pack()
        head = torch.empty((2, B, HEAD), dtype=torch.float32, device=mlp.device)
        mlp._launch_forward(B, obs, idx, filt, head, save=True)
        ctx.mlp, ctx.B = mlp, B
        return head

    @staticmethod
    def backward(ctx, g_head):
        mlp, B = ctx.mlp, ctx.B
        ws = mlp._workspace()
        g = g_head.contiguous()
        assert g.dtype == torch.float32 and tuple(g.shape) == (2, B, HEAD)
        grads = [torch.empty_like(p) for p in mlp.params]
        _lib.check(mlp.lib.pnr_mlp_backward(B, _p(g), _p(mlp.wpack), _p(ws["xs"]), _p(ws["h1"]), _p(ws["h2"]), _p(ws["dz1"]),
                                            _p(ws["dz2"]), _p(ws["slabs"]), ws["slabs"].numel(), mlp._ptrs(grads),
                                            mlp.n3[0], mlp.n3[1], 0, None, mlp._stream()))
        return (None, None, None, None, *grads)
