"""The PPO driver's observation filters: RLlib's MeanStdFilter semantics ('observation_filter': 'ConcurrentMeanStdFilter',
pioneer_knm_train.py:66) kept on the device, and the identity."""
import torch

from . import dist as pdist


def _dp(t):
    import ctypes
    return None if t is None else ctypes.c_void_p(t.data_ptr())


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
        clip = self.clip if self.clip else float("inf")
        if self.mean.is_cuda and self.mean.numel() == 137:
            # the same float64 operations in one launch (pnr_filter_prepare) instead of 17 element-wise ones: bit-identical
            import ctypes
            from . import _lib
            _lib.check(_lib.load_library().pnr_filter_prepare(_dp(self.n), _dp(self.mean), _dp(self.m2), ctypes.c_double(clip), _dp(self._loc),
                                                              _dp(self._inv), _dp(self._lo), _dp(self._hi),
                                                              ctypes.c_void_p(torch.cuda.current_stream(self.mean.device).cuda_stream)))
            return
        ident = self.n < 2                                   # identity until two samples exist (no host sync)
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
