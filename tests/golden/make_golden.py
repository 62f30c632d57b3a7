#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the authoring
container only; /root/reference does not exist on the GPU box).

What runs: the REFERENCE's own code - ``EMA_VFI.forward``, ``EMA_VFI.warp`` and
``ModulatedDeformConvPack.forward`` imported from
``/root/reference/src/models/ema_vfi.py`` - on CPU in fp32.  The only thing
injected is ``src.models.ema_vfi.DeformConv2d``: torchvision is not installed
here, the reference soft-fails that import (ema_vfi.py:17-21), and the class
below supplies the operator from ``oracle.emavfi_oracle.deform_conv2d``
(parity for that one op is therefore by definition, not by execution; see the
oracle's header).

Outputs (data only - inputs and expected outputs):
  tiny_mid8_24x40.npz     EMA_VFI(mid_channels=8), B=2, every intermediate
  tiny_mid8_23x37.npz     same, odd sizes (ceil(H/2) stride handling, borders), stress input
  cfg1_rubberwhale_256.npz  BASELINE config 1: uint8 256x256 crops of the reference's
                          data/processed/train/RubberWhale/frame10..12.png + reference output
  large_checks.npz        mid=64: sampled pixels + per-stage statistics at 256x256 (B=2)
                          and 1280x720 (B=1); weights/inputs are regenerated from
                          emavfi.synth by whoever replays it
  large_1080.npz          the same at 1920x1080 (B=1; BASELINE configs[4]'s size) - `large1080`
  large_odd.npz           the same at 203x331 (B=2, stress input: odd in both dimensions, partial tiles at every level) - `odd`
  amp_mid8_23x37.npz, amp_mid64_40x56.npz
                          the reference's forward under torch.autocast("cpu", dtype=torch.float16) - `amp`: the only way
                          the reference's autocast path (inference.py:159, a CUDA autocast) can EXECUTE in this container.
                          The DCN stand-in is wrapped the way torchvision wraps deform_conv2d for autocast (every argument cast
                          to float, autocast disabled inside, result cast back to the input's dtype).  Every module's output
                          dtype is recorded next to its values.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
sys.path.insert(0, ROOT)

import importlib.util  # noqa: E402

from emavfi import synth  # noqa: E402
from oracle import emavfi_oracle as oracle  # noqa: E402


def load_reference_module():
    """The reference's src/models/ema_vfi.py, loaded BY FILE PATH.  `import src.models.ema_vfi` must not be used:
    this repository ships its own `src/` package (the drop-in import path of inference.py:4), which would win over
    the reference's namespace package and hand back the HIP-backed class instead of the reference."""
    path = os.path.join(REF, "src", "models", "ema_vfi.py")
    spec = importlib.util.spec_from_file_location("ref_ema_vfi", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF) + os.sep), mod.__file__
    assert mod.EMA_VFI.__module__ == "ref_ema_vfi"
    return mod


ref = load_reference_module()


class DeformConv2dStandIn(torch.nn.Module):
    """Parameter holder with torchvision.ops.DeformConv2d's constructor
    signature as used at ema_vfi.py:45-51; forward = the oracle's restatement."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
        super().__init__()
        assert (kernel_size, stride, padding, dilation) == (3, 1, 1, 1)
        self.weight = torch.nn.Parameter(torch.zeros(out_channels, in_channels, 3, 3))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, offset, mask):
        return oracle.deform_conv2d(x, offset, mask, self.weight, self.bias)


ref.DeformConv2d = DeformConv2dStandIn
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE  # --out DIR (or make_golden.OUT = ...) writes elsewhere: tests/test_oracle_golden.py regenerates into tmp_path


def run_reference(sd, f1, f2, mid, num_blocks=3):
    model = ref.EMA_VFI(in_channels=3, mid_channels=mid, num_blocks=num_blocks)
    model.load_state_dict(sd, strict=True)
    model.eval()
    taps = {}

    def hook(name):
        def fn(_m, _i, o):
            taps[name] = o.detach().clone()
        return fn

    model.feat_ext_blocks.register_forward_hook(hook("feat"))
    model.context_encoding.register_forward_hook(hook("ctx"))
    model.motion_estimation.register_forward_hook(hook("flow"))
    for i, blk in enumerate(model.attention_blocks):
        blk.register_forward_hook(hook(f"fused_{i}"))
        blk.offset_conv.register_forward_hook(hook(f"raw_{i}"))
    orig_warp = model.warp

    def warp_spy(frame2, feature, flow):
        out = orig_warp(frame2, feature, flow)
        taps["warped"] = out.detach().clone()
        return out

    model.warp = warp_spy
    with torch.no_grad():
        taps["out"] = model(f1, f2).detach().clone()
    return taps


def stage_stats(t):
    t = t.double()
    return np.array([t.mean().item(), t.pow(2).sum().sqrt().item(), t.abs().max().item()], dtype=np.float64)


def sample_positions(seed, name, numel, n):
    return (synth.hash_uniform(seed, name, n) * numel).astype(np.int64)


def tiny(name, H, W, kind, seed):
    mid = 8
    sd = synth.synthetic_state_dict(seed=seed, mid_channels=mid)
    f1, f2 = synth.synthetic_frames(seed, 2, H, W, kind)
    taps = run_reference(sd, f1, f2, mid)
    mine = {}
    oracle.forward(sd, f1, f2, taps=mine)
    for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
        d = (taps[k] - mine[k]).abs().max().item()
        print(f"  {name}: restatement vs reference {k:8s} max-abs {d:.3e}")
        assert d <= 2e-5, (k, d)
    arrays = {"frame1": f1.numpy(), "frame2": f2.numpy()}
    arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
    arrays.update({"tap." + k: v.numpy() for k, v in taps.items()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"  {name}: flow range [{taps['flow'].min():.2f}, {taps['flow'].max():.2f}], "
          f"out range [{taps['out'].min():.3f}, {taps['out'].max():.3f}]")


def cfg1():
    from PIL import Image
    crops = []
    for n in (10, 11, 12):
        im = Image.open(os.path.join(REF, "data/processed/train/RubberWhale", f"frame{n}.png")).convert("RGB")
        a = np.asarray(im)
        y0, x0 = (a.shape[0] - 256) // 2, (a.shape[1] - 256) // 2
        crops.append(a[y0:y0 + 256, x0:x0 + 256].copy())
    u8 = np.stack(crops)  # frame0, ground-truth middle, frame1 (data_utils.py:33-37 triplet order)
    f1 = synth._to_model_range(u8[0:1])
    f2 = synth._to_model_range(u8[2:3])
    sd = synth.synthetic_state_dict(seed=0)
    taps = run_reference(sd, f1, f2, 64)
    mine = {}
    oracle.forward(sd, f1, f2, taps=mine)
    d = (taps["out"] - mine["out"]).abs().max().item()
    print(f"  cfg1: restatement vs reference out max-abs {d:.3e}; flow range "
          f"[{taps['flow'].min():.2f}, {taps['flow'].max():.2f}]")
    assert d <= 2e-5
    arrays = {"triplet_u8": u8, "out": taps["out"].numpy(), "flow": taps["flow"].numpy().astype(np.float32)}
    for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
        arrays["stats." + k] = stage_stats(taps[k])
    np.savez_compressed(os.path.join(OUT, "cfg1_rubberwhale_256.npz"), **arrays)


def large():
    arrays = {}
    sd = synth.synthetic_state_dict(seed=0)
    for tag, B, H, W, kind, seed in (("256", 2, 256, 256, "natural", 1), ("256s", 1, 256, 256, "stress", 2),
                                     ("720", 1, 720, 1280, "natural", 3)):
        f1, f2 = synth.synthetic_frames(seed, B, H, W, kind)
        t0 = time.time()
        taps = run_reference(sd, f1, f2, 64)
        print(f"  large {tag}: reference forward {time.time() - t0:.1f}s, flow range "
              f"[{taps['flow'].min():.2f}, {taps['flow'].max():.2f}] out [{taps['out'].min():.3f},{taps['out'].max():.3f}]")
        arrays[f"{tag}.meta"] = np.array([B, H, W, seed, 0 if kind == "natural" else 1], dtype=np.int64)
        for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
            v = taps[k].contiguous().view(-1)
            pos = sample_positions(seed, f"sample.{tag}.{k}", v.numel(), min(4096, v.numel()))
            arrays[f"{tag}.pos.{k}"] = pos
            arrays[f"{tag}.val.{k}"] = v[torch.from_numpy(pos)].numpy()
            arrays[f"{tag}.stats.{k}"] = stage_stats(taps[k])
    np.savez_compressed(os.path.join(OUT, "large_checks.npz"), **arrays)


def large1080():
    """1920x1080 (BASELINE configs[4]'s frame size): the reference forward on CPU, ~1-2 minutes and ~10 GB here."""
    arrays = {}
    sd = synth.synthetic_state_dict(seed=0)
    tag, B, H, W, kind, seed = "1080", 1, 1080, 1920, "natural", 4
    f1, f2 = synth.synthetic_frames(seed, B, H, W, kind)
    t0 = time.time()
    taps = run_reference(sd, f1, f2, 64)
    print(f"  large {tag}: reference forward {time.time() - t0:.1f}s, flow range "
          f"[{taps['flow'].min():.2f}, {taps['flow'].max():.2f}] out [{taps['out'].min():.3f},{taps['out'].max():.3f}]")
    arrays[f"{tag}.meta"] = np.array([B, H, W, seed, 0], dtype=np.int64)
    for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
        v = taps[k].contiguous().view(-1)
        pos = sample_positions(seed, f"sample.{tag}.{k}", v.numel(), min(4096, v.numel()))
        arrays[f"{tag}.pos.{k}"] = pos
        arrays[f"{tag}.val.{k}"] = v[torch.from_numpy(pos)].numpy()
        arrays[f"{tag}.stats.{k}"] = stage_stats(taps[k])
    np.savez_compressed(os.path.join(OUT, "large_1080.npz"), **arrays)


def large_odd():
    """mid_channels 64 at an ODD size in both dimensions (B = 2, 203 x 331, stress input: flows far beyond one tile): every level of
    the context pyramid ends in a partial tile (102 x 166, 51 x 83), the warp takes its W % 4 != 0 path, every ring strip its ragged
    last columns.  Samples of every stage of the reference's forward."""
    arrays = {}
    sd = synth.synthetic_state_dict(seed=0)
    tag, B, H, W, kind, seed = "odd", 2, 203, 331, "stress", 5
    f1, f2 = synth.synthetic_frames(seed, B, H, W, kind)
    t0 = time.time()
    taps = run_reference(sd, f1, f2, 64)
    print(f"  large {tag}: reference forward {time.time() - t0:.1f}s, flow range "
          f"[{taps['flow'].min():.2f}, {taps['flow'].max():.2f}] out [{taps['out'].min():.3f},{taps['out'].max():.3f}]")
    arrays[f"{tag}.meta"] = np.array([B, H, W, seed, 1], dtype=np.int64)
    for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
        v = taps[k].contiguous().view(-1)
        pos = sample_positions(seed, f"sample.{tag}.{k}", v.numel(), min(4096, v.numel()))
        arrays[f"{tag}.pos.{k}"] = pos
        arrays[f"{tag}.val.{k}"] = v[torch.from_numpy(pos)].numpy()
        arrays[f"{tag}.stats.{k}"] = stage_stats(taps[k])
    np.savez_compressed(os.path.join(OUT, "large_odd.npz"), **arrays)


def large_offsets(recipe=(3.0, 3.0), fname="large_offsets.npz", tag="off", seed=6):
    """mid_channels 64 with LARGE deformable offsets (round 5): synthetic_state_dict(offset_std = 3, offset_bias = 3) makes the three
    packs' offsets span about +-8 px (the benchmark's recipe: +-2), so that most (wave, tap) groups of the HIP pack kernels leave their
    staged window (R = 2) and take the fix-up pass.  B = 1, 120 x 200 (7.5 x 12.5 tiles of 16 x 16), natural
    input; samples of every stage of the reference's forward, and the offsets' quantiles for the record.
    Round 6 ("offsets16": recipe (6, 6), large_offsets16.npz): offsets spanning about +-16 px - nearly every (wave, tap) group is outside,
    several arena rounds per wave in the rebuilt fix-up (deform_pack3.inl)."""
    arrays = {}
    sd = synth.synthetic_state_dict(seed=0, offset_std=recipe[0], offset_bias=recipe[1])
    B, H, W, kind = 1, 120, 200, "natural"
    f1, f2 = synth.synthetic_frames(seed, B, H, W, kind)
    t0 = time.time()
    taps = run_reference(sd, f1, f2, 64)
    raw = torch.cat([taps[f"raw_{i}"] for i in range(3)], 0)
    off = torch.cat([raw[:, 0:9], raw[:, 18:27]], 1).abs().flatten()
    print(f"  large {tag}: reference forward {time.time() - t0:.1f}s, |offset| p50 {off.quantile(0.5):.2f} p99 {off.quantile(0.99):.2f} max {off.max():.2f}, "
          f"out [{taps['out'].min():.3f},{taps['out'].max():.3f}]")
    arrays[f"{tag}.meta"] = np.array([B, H, W, seed, 0], dtype=np.int64)
    arrays[f"{tag}.recipe"] = np.array(recipe, dtype=np.float64)
    arrays[f"{tag}.abs_offset_quantiles"] = np.array([off.quantile(0.5).item(), off.quantile(0.99).item(), off.max().item()], dtype=np.float64)
    for k in ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out"):
        v = taps[k].contiguous().view(-1)
        pos = sample_positions(seed, f"sample.{tag}.{k}", v.numel(), min(4096, v.numel()))
        arrays[f"{tag}.pos.{k}"] = pos
        arrays[f"{tag}.val.{k}"] = v[torch.from_numpy(pos)].numpy()
        arrays[f"{tag}.stats.{k}"] = stage_stats(taps[k])
    np.savez_compressed(os.path.join(OUT, fname), **arrays)


class DeformConv2dAutocastStandIn(DeformConv2dStandIn):
    """torchvision registers an Autocast kernel for deform_conv2d (torchvision/csrc/ops/autocast/deform_conv2d_kernel.cpp, restated):
    autocast is switched off inside, input / weight / offset / mask / bias are cast to float, the op runs in fp32 and the result is
    cast `.to(input.scalar_type())`.  The same wrapper around the stand-in operator."""

    def forward(self, x, offset, mask):
        with torch.autocast("cpu", enabled=False):
            out = oracle.deform_conv2d(x.float(), offset.float(), mask.float(), self.weight.float(),
                                       None if self.bias is None else self.bias.float())
        return out.to(x.dtype)


def amp(name, mid, B, H, W, kind, seed):
    """The reference's own forward under CPU autocast (float16): values and dtypes of every stage."""
    sd = synth.synthetic_state_dict(seed=seed, mid_channels=mid)
    f1, f2 = synth.synthetic_frames(seed, B, H, W, kind)
    saved = ref.DeformConv2d
    ref.DeformConv2d = DeformConv2dAutocastStandIn
    try:
        model = ref.EMA_VFI(in_channels=3, mid_channels=mid, num_blocks=3)
    finally:
        ref.DeformConv2d = saved
    model.load_state_dict(sd, strict=True)
    model.eval()
    taps, dtypes = {}, {}

    def hook(name_):
        def fn(_m, _i, o):
            taps[name_] = o.detach().clone()
        return fn

    def dtype_hook(name_):
        def fn(_m, _i, o):
            dtypes[name_] = str(o.dtype).replace("torch.", "")
        return fn

    model.feat_ext_blocks.register_forward_hook(hook("feat"))
    model.context_encoding.register_forward_hook(hook("ctx"))
    model.context_encoding[3].register_forward_hook(hook("pooled"))
    model.motion_estimation.register_forward_hook(hook("flow"))
    for i, blk in enumerate(model.attention_blocks):
        blk.register_forward_hook(hook(f"fused_{i}"))
        blk.offset_conv.register_forward_hook(hook(f"raw_{i}"))
    for n, m in model.named_modules():
        if n:
            m.register_forward_hook(dtype_hook(n))
    orig_warp = model.warp

    def warp_spy(frame2, feature, flow):
        out = orig_warp(frame2, feature, flow)
        taps["warped"] = out.detach().clone()
        dtypes["warp"] = str(out.dtype).replace("torch.", "")
        return out

    model.warp = warp_spy
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
        taps["out"] = model(f1, f2).detach().clone()
    dtypes["out"] = str(taps["out"].dtype).replace("torch.", "")
    arrays = {"meta": np.array([mid, B, H, W, seed, 0 if kind == "natural" else 1], dtype=np.int64),
              "dtypes": np.array(sorted(f"{k}={v}" for k, v in dtypes.items()))}
    for k, v in taps.items():
        arrays["tap." + k] = v.float().numpy()      # fp16 values are exact in fp32
        arrays["dtype." + k] = np.array(str(v.dtype).replace("torch.", ""))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print(f"  {name}: out {taps['out'].dtype} range [{float(taps['out'].min()):.3f}, {float(taps['out'].max()):.3f}], "
          f"flow [{float(taps['flow'].min()):.2f}, {float(taps['flow'].max()):.2f}]; fp32 stages: "
          f"{sorted(k for k, v in taps.items() if v.dtype == torch.float32)}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    argv = sys.argv[1:]
    if "--out" in argv:
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        os.makedirs(OUT, exist_ok=True)
        del argv[i:i + 2]
    which = argv or ["tiny", "cfg1", "large"]
    if "tiny" in which:
        tiny("tiny_mid8_24x40", 24, 40, "natural", 11)
        tiny("tiny_mid8_23x37", 23, 37, "stress", 12)
    if "cfg1" in which:
        cfg1()
    if "large" in which:
        large()
    if "large1080" in which:
        large1080()
    if "odd" in which:
        large_odd()
    if "offsets" in which:
        large_offsets()
    if "offsets16" in which:
        large_offsets((6.0, 6.0), "large_offsets16.npz", "off16", 7)
    if "amp" in which:
        amp("amp_mid8_23x37", 8, 2, 23, 37, "stress", 12)
        amp("amp_mid64_40x56", 64, 1, 40, 56, "natural", 13)
    print("golden vectors written to", OUT)
