"""Known-answer tests that pin the restated DCNv2.

torchvision is not installed and not vendored by the reference, so these
tests - not an execution of torchvision - are what defines the operator for
this repository.  Each clause of the definition is isolated: dy/dx channel
meaning, row-major tap order, the static/dynamic channel routing of
ModulatedDeformConvPack (ema_vfi.py:57-58), half-pixel weights, and the
"<= -1 or >= size -> 0" border rule.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import emavfi_oracle as oracle


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _shifted(x, dy, dx):
    """Image s with s[y, x] = x[y + dy, x + dx], zero outside."""
    B, C, H, W = x.shape
    out = torch.zeros_like(x)
    ys, xs = max(0, -dy), max(0, -dx)
    ye, xe = min(H, H - dy), min(W, W - dx)
    if ye > ys and xe > xs:
        out[:, :, ys:ye, xs:xe] = x[:, :, ys + dy:ye + dy, xs + dx:xe + dx]
    return out


def test_zero_offset_unit_mask_is_conv2d():
    x, w, b = _rand(2, 5, 9, 11), _rand(7, 5, 3, 3, seed=1), _rand(7, seed=2)
    off, msk = torch.zeros(2, 18, 9, 11), torch.ones(2, 9, 9, 11)
    got = oracle.deform_conv2d(x, off, msk, w, b)
    assert torch.allclose(got, F.conv2d(x, w, b, padding=1), atol=1e-5)


def test_fresh_pack_is_half_conv_plus_bias():
    # zero offset_conv (ema_vfi.py:42-43) -> offsets 0, mask sigmoid(0) = 0.5
    x, w, b = _rand(1, 4, 8, 8), _rand(4, 4, 3, 3, seed=3), _rand(4, seed=4)
    p = {"attention_blocks.0.offset_conv.weight": torch.zeros(27, 4, 3, 3),
         "attention_blocks.0.offset_conv.bias": torch.zeros(27),
         "attention_blocks.0.dcn_v2.weight": w, "attention_blocks.0.dcn_v2.bias": b}
    got = oracle.attention_block(p, 0, x)
    assert torch.allclose(got, 0.5 * F.conv2d(x, w, None, padding=1) + b.view(1, -1, 1, 1), atol=1e-5)


@pytest.mark.parametrize("dy,dx", [(2, 0), (0, -3), (-1, 0), (0, 1), (1, 2)])
def test_integer_offsets_equal_shifted_conv(dy, dx):
    # even offset channel = dy, odd = dx: a constant (dy, dx) on all taps samples the shifted image
    x, w = _rand(1, 3, 10, 12), _rand(2, 3, 3, 3, seed=5)
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2] = dy
    off[:, 1::2] = dx
    got = oracle.deform_conv2d(x, off, torch.ones(1, 9, 10, 12), w, None)
    # zero padding must apply to the ORIGINAL image extent: pad, shift, conv-valid
    xp = F.pad(x, (8, 8, 8, 8))
    ref = F.conv2d(_shifted(xp, dy, dx), w, None, padding=1)[:, :, 8:-8, 8:-8]
    assert torch.allclose(got, ref, atol=1e-5)


@pytest.mark.parametrize("k", range(9))
def test_single_tap_offset_pins_row_major_tap_order(k):
    x, w = _rand(1, 2, 9, 9), _rand(3, 2, 3, 3, seed=6)
    off = torch.zeros(1, 18, 9, 9)
    off[:, 2 * k] = 1.0      # dy of tap k only
    off[:, 2 * k + 1] = -2.0  # dx of tap k only
    got = oracle.deform_conv2d(x, off, torch.ones(1, 9, 9, 9), w, None)
    i, j = divmod(k, 3)
    wk = torch.zeros_like(w)
    wk[:, :, i, j] = w[:, :, i, j]
    xp = F.pad(x, (6, 6, 6, 6))
    ref = F.conv2d(x, w - wk, None, padding=1) + \
        F.conv2d(_shifted(xp, 1, -2), wk, None, padding=1)[:, :, 6:-6, 6:-6]
    assert torch.allclose(got, ref, atol=1e-5)


def test_pack_channel_routing():
    # raw[0:9] -> offset[0:9], raw[18:27] -> offset[9:18], raw[9:18] -> mask (ema_vfi.py:57-59)
    x = _rand(1, 4, 6, 6)
    bias = torch.arange(27, dtype=torch.float32) * 0.1
    p = {"attention_blocks.0.offset_conv.weight": torch.zeros(27, 4, 3, 3),
         "attention_blocks.0.offset_conv.bias": bias}
    off, msk = oracle.offset_and_mask(p, 0, x)
    assert torch.allclose(off[0, :, 2, 2], torch.cat([bias[0:9], bias[18:27]]))
    assert torch.allclose(msk[0, :, 2, 2], torch.sigmoid(bias[9:18]))


def test_half_pixel_is_two_neighbour_average():
    x = _rand(1, 1, 8, 8)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0
    off = torch.zeros(1, 18, 8, 8)
    off[:, 2 * 4 + 1] = 0.5  # centre tap, dx = +0.5
    got = oracle.deform_conv2d(x, off, torch.ones(1, 9, 8, 8), w, None)
    ref = 0.5 * (x + _shifted(x, 0, 1))
    assert torch.allclose(got, ref, atol=1e-6)


def test_border_rule():
    H = W = 6
    x = torch.ones(1, 1, H, W)
    w = torch.zeros(1, 1, 3, 3)
    w[0, 0, 1, 1] = 1.0

    def centre_sample(dy, dx, y, xq):
        off = torch.zeros(1, 18, H, W)
        off[:, 8], off[:, 9] = dy, dx
        return oracle.deform_conv2d(x, off, torch.ones(1, 9, H, W), w, None)[0, 0, y, xq].item()

    assert centre_sample(-0.25, 0.0, 0, 3) == pytest.approx(0.75)   # row -0.25: partial corner weights
    assert centre_sample(-1.0, 0.0, 0, 3) == 0.0                     # row -1 exactly: outside
    assert centre_sample(-0.999, 0.0, 0, 3) == pytest.approx(0.001, abs=1e-5)
    assert centre_sample(0.5, 0.0, H - 1, 3) == pytest.approx(0.5)   # row H-0.5
    assert centre_sample(1.0, 0.0, H - 1, 3) == 0.0                  # row H exactly: outside
    assert centre_sample(0.0, 0.75, 2, W - 1) == pytest.approx(0.25)
    assert centre_sample(0.0, -7.0, 2, 3) == 0.0
    assert centre_sample(float("nan"), 0.0, 2, 3) == 0.0 or np.isnan(centre_sample(float("nan"), 0.0, 2, 3))


@pytest.mark.parametrize("k", [0, 4, 8])
def test_one_hot_mask_isolates_tap(k):
    x, w = _rand(1, 3, 7, 7), _rand(2, 3, 3, 3, seed=7)
    msk = torch.zeros(1, 9, 7, 7)
    msk[:, k] = 1.0
    got = oracle.deform_conv2d(x, torch.zeros(1, 18, 7, 7), msk, w, None)
    i, j = divmod(k, 3)
    wk = torch.zeros_like(w)
    wk[:, :, i, j] = w[:, :, i, j]
    assert torch.allclose(got, F.conv2d(x, wk, None, padding=1), atol=1e-5)


@pytest.mark.parametrize("seed", range(6))
def test_vectorised_matches_scalar_c(oracle_c, seed):
    g = torch.Generator().manual_seed(100 + seed)
    B, C, O = 1 + seed % 2, 3 + seed, 2 + seed
    H, W = 5 + 2 * seed, 13 - seed
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 18, H, W, generator=g) * (1.0 + seed)  # up to far outside the image
    msk = torch.rand(B, 9, H, W, generator=g)
    w = torch.randn(O, C, 3, 3, generator=g)
    b = torch.randn(O, generator=g)
    got = oracle.deform_conv2d(x, off, msk, w, b).numpy()
    ref = oracle_c.deform(x.numpy(), off.numpy(), msk.numpy(), w.numpy(), b.numpy())
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
