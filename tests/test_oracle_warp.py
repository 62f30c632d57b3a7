"""Pins the warp restatements against torch's own grid_sample, the op the
reference calls at ema_vfi.py:169."""
import numpy as np
import pytest
import torch

from oracle import emavfi_oracle as oracle


def _case(seed, B, C, H, W, sigma):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g), torch.randn(B, 2, H, W, generator=g) * sigma


@pytest.mark.parametrize("H,W,sigma", [(17, 23, 1.0), (64, 48, 5.0), (33, 127, 50.0), (1, 9, 2.0), (9, 1, 2.0), (1, 1, 0.3)])
def test_c_warp_matches_grid_sample(oracle_c, H, W, sigma):
    f2, flow = _case(7, 2, 3, H, W, sigma)
    ref = oracle.warp(f2, flow).numpy()
    got = oracle_c.warp(f2.numpy(), flow.numpy())
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_zero_flow_is_identity():
    f2, _ = _case(1, 1, 3, 20, 30, 0.0)
    out = oracle.warp(f2, torch.zeros(1, 2, 20, 30))
    assert torch.allclose(out, f2, atol=1e-5)


def test_integer_flow_shifts_and_zero_pads():
    f2, _ = _case(2, 1, 3, 16, 16, 0.0)
    flow = torch.zeros(1, 2, 16, 16)
    flow[:, 0] = 3.0   # channel 0 = dx (ema_vfi.py:157,162: grid = cat(xx, yy))
    flow[:, 1] = -2.0  # channel 1 = dy
    out = oracle.warp(f2, flow)
    assert torch.allclose(out[:, :, 2:, :13], f2[:, :, :14, 3:], atol=1e-4)
    # the normalise/un-normalise round trip (fact 6) leaves ~1e-7 of a neighbour, not an exact 0
    assert out[:, :, :2, :].abs().max() <= 1e-5 and out[:, :, :, 13:].abs().max() <= 1e-5


def test_far_flow_gives_zero_and_non_finite_is_unspecified(oracle_c):
    """Finite flow far outside the image -> exactly 0 in both restatements.
    Non-finite flow is unspecified by the reference: ATen's CPU grid_sample
    propagates NaN weights (inf - floor(inf)), its GPU kernel returns 0; the C
    restatement (and the HIP kernel) return 0, so only the finite case is pinned."""
    f2, flow = _case(3, 1, 3, 8, 8, 1.0)
    flow[0, 0, 4, 4] = 1e9
    flow[0, 1, 5, 5] = -300.0
    out = oracle_c.warp(f2.numpy(), flow.numpy())
    ref = oracle.warp(f2, flow).numpy()
    for y, x in ((4, 4), (5, 5)):
        assert np.all(out[0, :, y, x] == 0) and np.all(ref[0, :, y, x] == 0)
    flow[0, 0, 2, 2] = float("nan")
    flow[0, 1, 3, 3] = float("inf")
    out = oracle_c.warp(f2.numpy(), flow.numpy())
    assert np.all(out[0, :, 2, 2] == 0) and np.all(out[0, :, 3, 3] == 0)
