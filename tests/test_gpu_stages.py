"""Stage-level C-ABI entries against the oracle's stage functions (SURVEY 8b proposed emavfi_context / emavfi_reconstruct next to
emavfi_warp / emavfi_conv3x3 / emavfi_mdcn): the launches the forward runs for the stage - in the 16-bit modes at mid_channels 64 the
stride-2 ring kernel, the streamed-weight tile kernels, the deterministic pool + linear, the TAIL ring kernel and reconstruction.1 + .2
as ONE launch - on storage-rounded inputs and weights, so what is compared is the kernels' arithmetic."""
import math

import pytest
import torch

from emavfi import lib
from oracle import emavfi_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(t, dtype):
    return t.clone() if dtype == "fp32" else (t.bfloat16().float() if dtype == "bf16" else t.half().float())


def conv_w(g, cout, cin, dtype, scale=1.0):
    return rnd(torch.randn(cout, cin, 3, 3, generator=g) * (scale * math.sqrt(2.0 / (9 * cin))), dtype), torch.randn(cout, generator=g) * 0.1


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-6), ("bf16", 2e-3), ("fp16", 4e-4)])   # measured: 4e-7 / 6.6e-4 / 1.1e-4 (the mean over the pixels averages the roundings)
@pytest.mark.parametrize("mid,shape", [(64, (2, 37, 53)), (64, (1, 1, 7)), (64, (1, 64, 96)), (64, (1, 180, 320)), (8, (2, 23, 37)), (16, (1, 9, 33))])
def test_context_stage_matches_the_oracle(mid, shape, dtype, tol):
    """context_encoding (ema_vfi.py:79-86): ceil(H / 2) / ceil(H / 4) resolutions of odd sizes, one-row images, the 256 partial sums
    of the deterministic pool at 180 x 320."""
    B, H, W = shape
    g = torch.Generator().manual_seed(mid * 1000 + H)
    feat = rnd(torch.randn(B, mid, H, W, generator=g).relu(), dtype)
    w0, b0 = conv_w(g, 2 * mid, mid, dtype)
    w1, b1 = conv_w(g, 4 * mid, 2 * mid, dtype)
    w2, b2 = conv_w(g, 4 * mid, 4 * mid, dtype)
    lw, lb = torch.randn(mid, 4 * mid, generator=g) / math.sqrt(4 * mid), torch.randn(mid, generator=g) * 0.1
    params = [w0, b0, w1, b1, w2, b2, lw, lb]
    got = lib.context(feat.to(DEV), [p.to(DEV) for p in params], dtype=dtype).cpu()
    p = {"context_encoding.0.0.weight": w0, "context_encoding.0.0.bias": b0, "context_encoding.1.0.weight": w1, "context_encoding.1.0.bias": b1,
         "context_encoding.2.0.weight": w2, "context_encoding.2.0.bias": b2, "context_encoding.5.weight": lw, "context_encoding.5.bias": lb}
    ref = oracle.context_encoding(p, feat)
    err = (got - ref).abs().max().item()
    print(f"context {dtype} mid {mid} {shape}: max err {err:.3e} (|ctx| <= {ref.abs().max().item():.3g})")
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert err <= tol * max(1.0, ref.abs().max().item())
    again = lib.context(feat.to(DEV), [p.to(DEV) for p in params], dtype=dtype).cpu()
    assert torch.equal(got, again)          # fixed partition, fixed summation order: deterministic


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 1.5e-2), ("fp16", 2e-3)])   # measured: 2.4e-6 / 1.07e-2 / 1.2e-3 (two storage roundings in front of tanh / 2)
@pytest.mark.parametrize("mid,shape", [(64, (2, 75, 131)), (64, (1, 1, 7)), (64, (1, 2, 62)), (64, (1, 17, 124)), (64, (3, 40, 125)), (8, (2, 23, 37)), (32, (1, 19, 33))])
def test_reconstruction_stage_matches_the_oracle(mid, shape, dtype, tol):
    """reconstruction (ema_vfi.py:102-107) + tanh + (t + 1) / 2 (:146): widths around the fused tail kernel's 62-column strip pitch,
    one- and two-row images, several samples; the frame lies in [0, 1]."""
    B, H, W = shape
    g = torch.Generator().manual_seed(mid * 100 + W)
    fused = rnd(torch.randn(B, mid + 3, H, W, generator=g), dtype)
    w0, b0 = conv_w(g, mid, mid + 3, dtype)
    w1, b1 = conv_w(g, mid // 2, mid, dtype)
    w2, b2 = conv_w(g, 3, mid // 2, dtype, scale=1.5)
    params = [w0, b0, w1, b1, w2, b2]
    got = lib.reconstruct(fused.to(DEV), [p.to(DEV) for p in params], dtype=dtype).cpu()
    p = {"reconstruction.0.0.weight": w0, "reconstruction.0.0.bias": b0, "reconstruction.1.0.weight": w1, "reconstruction.1.0.bias": b1,
         "reconstruction.2.weight": w2, "reconstruction.2.bias": b2}
    ref = oracle.reconstruction(p, fused)
    err = (got - ref).abs().max().item()
    print(f"reconstruct {dtype} mid {mid} {shape}: max err {err:.3e}; frame in [{got.min().item():.3f}, {got.max().item():.3f}]")
    assert got.shape == ref.shape and torch.isfinite(got).all() and got.min() >= 0 and got.max() <= 1
    assert ref.max() - ref.min() > 0.5      # a non-degenerate frame
    assert err <= tol


def test_stage_entries_validate_their_arguments():
    L = lib.load()
    assert L.emavfi_context_workspace_bytes(1, 64, 32, 32, lib.BF16) > 0 and L.emavfi_reconstruct_workspace_bytes(1, 64, 32, 32, lib.F32) > 0
    assert L.emavfi_context_workspace_bytes(1, 7, 32, 32, lib.BF16) == 0 and "multiple of 8" in lib.last_error()
    assert L.emavfi_reconstruct_workspace_bytes(0, 64, 32, 32, lib.BF16) == 0
    x = torch.zeros(1, 64, 8, 8, device=DEV)
    with pytest.raises(ValueError, match="8 tensors"):
        lib.context(x, [x] * 8)
    with pytest.raises(ValueError, match="6 tensors"):
        lib.reconstruct(torch.zeros(1, 67, 8, 8, device=DEV), [x] * 6)
