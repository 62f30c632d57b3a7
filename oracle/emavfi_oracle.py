"""ORACLE - test infrastructure only, never the product path.

CPU fp32 restatement of the reference's EMA-VFI forward
(``/root/reference/src/models/ema_vfi.py:110-171``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the shipped model (``emavfi.model.EMA_VFI``) never does and
raises if the HIP library is missing.

Parity status
-------------
* Everything except the deformable convolution calls the same ATen ops the
  reference calls (``F.conv2d``, ``F.grid_sample``, ``F.adaptive_avg_pool2d``,
  ``F.linear``, ``tanh``, ``sigmoid``) and is pinned by running the reference's
  own ``EMA_VFI.forward`` in the authoring container
  (``tests/golden/make_golden.py``) - see ``tests/test_oracle_golden.py``.
* ``deform_conv2d`` restates ``torchvision.ops.deform_conv2d`` (DCNv2), which
  the reference imports at ``ema_vfi.py:18`` but which is neither vendored nor
  installed here (``requirements.txt:2`` pins no version).  **Parity for that
  one op is unpinned**: it is restated from the operator's published
  definition and pinned only by the known-answer tests of
  ``tests/test_oracle_deform.py`` and by agreement
  with the independent scalar C restatement in ``oracle/deform_warp_ref.c``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def conv3x3(x, w, b, stride: int = 1):
    """``conv`` / ``conv_block`` body: Conv2d(k=3, p=1) (ema_vfi.py:7-14)."""
    return F.conv2d(x, w, b, stride=stride, padding=1)


def feature_extraction(p: Params, frame1, frame2, num_blocks: int = 3):
    """ema_vfi.py:112-116 with modules from :73-76."""
    x = torch.cat([frame1, frame2], dim=1)
    x = F.relu(conv3x3(x, p["feat_ext_conv1.0.weight"], p["feat_ext_conv1.0.bias"]))
    for i in range(num_blocks):
        k = f"feat_ext_blocks.conv_block_{i}.0"
        x = F.relu(conv3x3(x, p[k + ".weight"], p[k + ".bias"]))
    return x


def context_encoding(p: Params, feat):
    """ema_vfi.py:120 with modules from :79-86 (two stride-2 convs, one conv,
    global average pool, flatten, linear)."""
    x = F.relu(conv3x3(feat, p["context_encoding.0.0.weight"], p["context_encoding.0.0.bias"], 2))
    x = F.relu(conv3x3(x, p["context_encoding.1.0.weight"], p["context_encoding.1.0.bias"], 2))
    x = F.relu(conv3x3(x, p["context_encoding.2.0.weight"], p["context_encoding.2.0.bias"]))
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, p["context_encoding.5.weight"], p["context_encoding.5.bias"])


def motion_estimation(p: Params, feat, ctx):
    """ema_vfi.py:124-126 with modules from :89-93.  Channel order of the
    concat is [feat, ctx broadcast]."""
    B, _, H, W = feat.shape
    x = torch.cat([feat, ctx[:, :, None, None].expand(B, ctx.shape[1], H, W)], dim=1)
    x = F.relu(conv3x3(x, p["motion_estimation.0.0.weight"], p["motion_estimation.0.0.bias"]))
    x = F.relu(conv3x3(x, p["motion_estimation.1.0.weight"], p["motion_estimation.1.0.bias"]))
    return conv3x3(x, p["motion_estimation.2.weight"], p["motion_estimation.2.bias"])


def warp(frame2, flow):
    """``EMA_VFI.warp`` (ema_vfi.py:149-171): pixel grid + flow, normalised
    with a true division, sampled by grid_sample (bilinear, zeros,
    align_corners=True)."""
    B, C, H, W = frame2.shape
    xx = torch.arange(0, W).view(1, 1, 1, W).expand(B, 1, H, W)
    yy = torch.arange(0, H).view(1, 1, H, 1).expand(B, 1, H, W)
    vgrid = torch.cat((xx, yy), 1).float() + flow
    gx = 2.0 * vgrid[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * vgrid[:, 1] / max(H - 1, 1) - 1.0
    grid = torch.stack((gx, gy), dim=-1)
    return F.grid_sample(frame2, grid, align_corners=True)


def offset_and_mask(p: Params, i: int, x):
    """``ModulatedDeformConvPack.forward`` up to the dcn call (ema_vfi.py:55-59):
    27 raw channels -> chunk(3) -> offset = cat(first, third), mask = sigmoid(second)."""
    k = f"attention_blocks.{i}.offset_conv"
    raw = conv3x3(x, p[k + ".weight"], p[k + ".bias"])
    o1, m, o2 = torch.chunk(raw, 3, dim=1)
    return torch.cat((o1, o2), dim=1), torch.sigmoid(m)


def _dcn_bilinear(x, py, px):
    """DCNv2 sampling rule for all channels at once.

    x [B,C,H,W]; py, px [B,H,W] absolute sample rows / columns.  Returns
    [B,C,H,W].  Rule (torchvision deform_conv2d CPU kernel, restated): the
    sample is 0 when ``py <= -1 or py >= H or px <= -1 or px >= W``; otherwise
    the four neighbours ``(floor, floor+1)`` contribute with the usual bilinear
    weights, and a neighbour outside the image contributes 0.
    """
    B, C, H, W = x.shape
    inside = (py > -1) & (py < H) & (px > -1) & (px < W)
    hl = torch.floor(py)
    wl = torch.floor(px)
    lh = py - hl
    lw = px - wl
    hh_, hw_ = 1 - lh, 1 - lw
    hl = hl.long()
    wl = wl.long()
    hh = hl + 1
    wh = wl + 1
    flat = x.reshape(B, C, H * W)

    def corner(r, c, wgt):
        ok = inside & (r >= 0) & (r <= H - 1) & (c >= 0) & (c <= W - 1)
        idx = (r.clamp(0, H - 1) * W + c.clamp(0, W - 1)).reshape(B, 1, H * W).expand(B, C, H * W)
        v = torch.gather(flat, 2, idx).reshape(B, C, H, W)
        return v * (wgt * ok.to(x.dtype)).unsqueeze(1)

    return (corner(hl, wl, hh_ * hw_) + corner(hl, wh, hh_ * lw)
            + corner(hh, wl, lh * hw_) + corner(hh, wh, lh * lw))


def deform_conv2d(x, offset, mask, weight, bias: Optional[torch.Tensor]):
    """DCNv2, 3x3, stride 1, pad 1, dilation 1, one offset group, one weight
    group - the configuration built at ema_vfi.py:45-51 and invoked at :60.

    out[b,o,y,x] = bias[o] + sum_c sum_{k=3i+j} W[o,c,i,j] * mask[b,k,y,x]
                   * bilin(x[b,c], y-1+i+offset[b,2k,y,x], x-1+j+offset[b,2k+1,y,x])
    (even offset channel = dy, odd = dx, taps row-major).
    """
    B, C, H, W = x.shape
    O = weight.shape[0]
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    out = torch.zeros(B, O, H, W, dtype=x.dtype)
    for k in range(9):
        i, j = divmod(k, 3)
        py = ys - 1 + i + offset[:, 2 * k]
        px = xs - 1 + j + offset[:, 2 * k + 1]
        col = _dcn_bilinear(x, py, px) * mask[:, k].unsqueeze(1)
        out += torch.einsum("oc,bchw->bohw", weight[:, :, i, j], col)
    if bias is not None:
        out += bias.view(1, O, 1, 1)
    return out


def attention_block(p: Params, i: int, x):
    """One ``ModulatedDeformConvPack`` (ema_vfi.py:53-60)."""
    off, msk = offset_and_mask(p, i, x)
    k = f"attention_blocks.{i}.dcn_v2"
    return deform_conv2d(x, off, msk, p[k + ".weight"], p[k + ".bias"])


def reconstruction(p: Params, x):
    """ema_vfi.py:144-146 with modules from :102-107."""
    x = F.relu(conv3x3(x, p["reconstruction.0.0.weight"], p["reconstruction.0.0.bias"]))
    x = F.relu(conv3x3(x, p["reconstruction.1.0.weight"], p["reconstruction.1.0.bias"]))
    x = torch.tanh(conv3x3(x, p["reconstruction.2.weight"], p["reconstruction.2.bias"]))
    return (x + 1) / 2


@torch.no_grad()
def forward(p: Params, frame1, frame2, num_blocks: int = 3, taps: Optional[dict] = None):
    """``EMA_VFI.forward`` (ema_vfi.py:110-147).  ``taps`` collects every
    intermediate the golden fixtures store."""
    feat = feature_extraction(p, frame1, frame2, num_blocks)
    ctx = context_encoding(p, feat)
    flow = motion_estimation(p, feat, ctx)
    warped = warp(frame2, flow)
    fused = torch.cat([feat, warped], dim=1)
    if taps is not None:
        taps.update(feat=feat, ctx=ctx, flow=flow, warped=warped)
    for i in range(num_blocks):
        if taps is not None:
            off, msk = offset_and_mask(p, i, fused)
            taps[f"offset_{i}"], taps[f"mask_{i}"] = off, msk
        fused = attention_block(p, i, fused)
        if taps is not None:
            taps[f"fused_{i}"] = fused
    out = reconstruction(p, fused)
    if taps is not None:
        taps["out"] = out
    return out


# --------------------------------------------------------------------------------------------------------------
# The reference's forward under ``torch.cuda.amp.autocast()`` (float16), which is what ``inference.py:159`` runs on a
# GPU.  Restated from PyTorch's published autocast op policy; since round 4 PINNED BY EXECUTION of the reference's own forward under
# torch.autocast("cpu", dtype=torch.float16) (tests/golden/make_golden.py amp; tests/test_oracle_golden.py: bit for bit at mid_channels 8) -
# the reference has never been run on a GPU here, and for every op on this path the CPU and CUDA autocast lists agree: ``conv2d`` / ``linear`` are on the float16 list (inputs, weight AND bias are cast to fp16, the
# products accumulate in fp32, the result is an fp16 tensor); ``grid_sampler``, ``torch.cat`` and the ``grid + flow``
# add promote to the widest input - fp32 for all three here, because frame2, the pixel grid and ``warped`` are fp32
# (ema_vfi.py:157-169, :134); element-wise ops without a list entry
# (relu, sigmoid, tanh, ``+ 1``, ``/ 2``, adaptive_avg_pool2d) run in their input's dtype with fp32 op-math and one
# rounding; torchvision registers ``deform_conv2d`` with an Autocast kernel that casts input, weight, offset, mask and
# bias to fp32 and casts the result back to the INPUT's dtype.  Consequences along ema_vfi.py:110-147:
#   feat, ctx, flow are fp16 tensors; warp runs in fp32 on the fp16-valued flow and returns fp32; cat(feat, warped) is
#   fp32; every attention block computes its offset_conv in fp16 (input rounded), sigmoid in fp16, the DCN in fp32 on
#   the UNROUNDED fp32 input with the fp32 master weights, and returns fp32; reconstruction is fp16 to the end, so
#   the frame is an fp16 tensor.  tests/test_gpu_runtime.py::test_autocast_op_policy_of_this_torch checks the
#   per-op dtypes this restatement assumes against the installed torch on the GPU box.
def _h(t):
    """Round to float16 (what storing a value in an autocast fp16 tensor does); values stay in fp32 containers."""
    return t.half().float()


def _conv16(x, w, b, stride: int = 1):
    return _h(F.conv2d(_h(x), _h(w), _h(b), stride=stride, padding=1))


@torch.no_grad()
def forward_autocast16(p: Params, frame1, frame2, num_blocks: int = 3, taps: Optional[dict] = None):
    x = torch.cat([frame1, frame2], dim=1)
    x = F.relu(_conv16(x, p["feat_ext_conv1.0.weight"], p["feat_ext_conv1.0.bias"]))
    for i in range(num_blocks):
        k = f"feat_ext_blocks.conv_block_{i}.0"
        x = F.relu(_conv16(x, p[k + ".weight"], p[k + ".bias"]))
    feat = x
    c = F.relu(_conv16(feat, p["context_encoding.0.0.weight"], p["context_encoding.0.0.bias"], 2))
    c = F.relu(_conv16(c, p["context_encoding.1.0.weight"], p["context_encoding.1.0.bias"], 2))
    c = F.relu(_conv16(c, p["context_encoding.2.0.weight"], p["context_encoding.2.0.bias"]))
    c = _h(F.adaptive_avg_pool2d(c, 1).flatten(1))
    ctx = _h(F.linear(c, _h(p["context_encoding.5.weight"]), _h(p["context_encoding.5.bias"])))
    B, _, H, W = feat.shape
    m = torch.cat([feat, ctx[:, :, None, None].expand(B, ctx.shape[1], H, W)], dim=1)
    m = F.relu(_conv16(m, p["motion_estimation.0.0.weight"], p["motion_estimation.0.0.bias"]))
    m = F.relu(_conv16(m, p["motion_estimation.1.0.weight"], p["motion_estimation.1.0.bias"]))
    flow = _conv16(m, p["motion_estimation.2.weight"], p["motion_estimation.2.bias"])
    warped = warp(frame2, flow)                       # fp32 op on the fp16-valued flow
    fused = torch.cat([feat, warped], dim=1)          # promotes to fp32: warped keeps its fp32 bits
    if taps is not None:
        taps.update(feat=feat, ctx=ctx, flow=flow, warped=warped)
    for i in range(num_blocks):
        k = f"attention_blocks.{i}"
        raw = _conv16(fused, p[k + ".offset_conv.weight"], p[k + ".offset_conv.bias"])
        o1, mk, o2 = torch.chunk(raw, 3, dim=1)
        fused = deform_conv2d(fused, torch.cat((o1, o2), dim=1), _h(torch.sigmoid(mk)),
                              p[k + ".dcn_v2.weight"], p[k + ".dcn_v2.bias"])      # fp32, master weights
        if taps is not None:
            taps[f"fused_{i}"] = fused
    r = F.relu(_conv16(fused, p["reconstruction.0.0.weight"], p["reconstruction.0.0.bias"]))
    r = F.relu(_conv16(r, p["reconstruction.1.0.weight"], p["reconstruction.1.0.bias"]))
    r = _h(torch.tanh(_conv16(r, p["reconstruction.2.weight"], p["reconstruction.2.bias"])))
    out = _h(r + 1.0) / 2.0
    if taps is not None:
        taps["out"] = out
    return out
