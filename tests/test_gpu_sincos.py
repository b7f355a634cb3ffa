"""The engine's float32 sin/cos against float64 over dense sweeps (GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# np.cos/np.sin of a float32 array are float32 results within ~1.5 ulp of the true value;
# the engine's must stay within 1.3e-7 absolute (~1 ulp at 1.0) of float64.
ABS_TOL = 1.3e-7


def run(x, bounded):
    from pioneer_amd import _lib
    lib = _lib.load_library()
    xd = torch.from_numpy(x).cuda()
    s = torch.empty_like(xd); c = torch.empty_like(xd)
    _lib.check(lib.pnr_diag_sincos(C.c_void_p(xd.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(c.data_ptr()),
                                   xd.numel(), int(bounded), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return s.cpu().numpy().astype(np.float64), c.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("rng_hi", [3.2, 6.3, 13.0, 130.0, 700.0, 65536.0])
def test_bounded_path(rng_hi):
    rng = np.random.RandomState(int(rng_hi))
    x = np.concatenate([rng.uniform(-rng_hi, rng_hi, 4_000_000), np.linspace(-rng_hi, rng_hi, 1_000_001),
                        [0.0, -0.0, np.pi / 2, np.pi, -np.pi, 3.1416, 1.5708]]).astype(np.float32)
    s, c = run(x, True)
    xd = x.astype(np.float64)
    assert np.abs(s - np.sin(xd)).max() <= ABS_TOL
    assert np.abs(c - np.cos(xd)).max() <= ABS_TOL


def test_action_path_large_and_special():
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-1e6, 1e6, 1_000_000), rng.uniform(-1e30, 1e30, 100_000),
                        10.0 ** rng.uniform(-30, 38, 100_000)]).astype(np.float32)
    s, c = run(x, False)
    xd = x.astype(np.float64)
    assert np.abs(s - np.sin(xd)).max() <= ABS_TOL
    assert np.abs(c - np.cos(xd)).max() <= ABS_TOL
    s, c = run(np.array([np.inf, -np.inf, np.nan], dtype=np.float32), False)
    assert np.isnan(s).all() and np.isnan(c).all()   # np.sin(inf) = nan as well
