"""Parity tests proper: the HIP path, called through the C-ABI (ctypes, emavfi/lib.py), against
the CPU oracle and the golden vectors captured from the reference's forward().

Tolerances.  fp32 mode (exact-fp32 MFMA): BASELINE.json asks <= 1e-3 max-abs on the output;
intermediates are held to 2e-4 * max|ref| (summation order differs from MKLDNN's).
bf16 mode (BASELINE configs[2]): PSNR of the output vs the fp32 oracle, threshold in the test."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from emavfi import EMA_VFI, lib, synth
from oracle import emavfi_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
STAGES = ("feat", "ctx", "flow", "warped", "fused_0", "fused_1", "fused_2", "out")


def rel_err(got, ref):
    return (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())


def psnr(got, ref):
    mse = (got.double() - ref.double()).pow(2).mean().item()
    return 99.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)


def make_model(sd, mid=64, dtype="fp32"):
    m = EMA_VFI(mid_channels=mid, compute_dtype=dtype).to(DEV).eval()
    m.load_state_dict(sd, strict=True)
    return m


# ------------------------------------------------------------------ warp (row W)
@pytest.mark.parametrize("B,C,H,W,sigma", [(2, 3, 17, 23, 1.0), (1, 3, 64, 48, 5.0), (2, 3, 33, 128, 40.0),
                                           (1, 3, 1, 9, 2.0), (1, 3, 9, 1, 2.0), (1, 3, 1, 1, 0.3), (1, 1, 8, 12, 3.0),
                                           (1, 3, 720, 1280, 6.0)])
def test_warp_matches_oracle(B, C, H, W, sigma):
    g = torch.Generator().manual_seed(H * 1000 + W)
    f2 = torch.randn(B, C, H, W, generator=g)
    flow = torch.randn(B, 2, H, W, generator=g) * sigma
    got = lib.warp(f2.to(DEV), flow.to(DEV)).cpu()
    ref = oracle.warp(f2, flow)
    assert (got - ref).abs().max().item() <= 5e-6 * max(1.0, ref.abs().max().item())


def test_warp_far_and_nonfinite_flow_matches_the_cpu_reference():
    """ema_vfi.py:169 on the CPU forward (the oracle): a finite flow far outside samples zeros; a NaN / infinite flow - or one
    so large that the reference's own `2.0 * vgrid` overflows - gives NaN in every channel (ATen's CPU grid_sample: the corner
    weights are inf - inf there, and 0 * NaN = NaN).  The HIP kernels reproduce both, pixel for pixel, in the NCHW entry and
    in the tiled kernel the forward uses."""
    for H, W in ((8, 8), (40, 64)):          # 8x8: per-pixel kernel; 40x64 (W % 4 == 0): the tiled kernel
        f2 = torch.randn(1, 3, H, W)
        flow = torch.zeros(1, 2, H, W)
        flow[0, 0, 4, 4], flow[0, 1, 5, 5] = 1e9, -300.0
        flow[0, 0, 2, 2], flow[0, 1, 3, 3] = float("nan"), float("inf")
        flow[0, 0, 6, 6], flow[0, 1, 1, 1] = 3e38, -float("inf")
        got = lib.warp(f2.to(DEV), flow.to(DEV)).cpu()
        ref = oracle.warp(f2, flow)
        for y, x in ((4, 4), (5, 5)):
            assert torch.all(ref[0, :, y, x] == 0) and torch.all(got[0, :, y, x] == 0)
        for y, x in ((2, 2), (3, 3), (6, 6), (1, 1)):
            assert torch.isnan(ref[0, :, y, x]).all() and torch.isnan(got[0, :, y, x]).all(), (H, W, y, x, got[0, :, y, x])
        assert torch.equal(torch.isnan(got), torch.isnan(ref))
        fin = ~torch.isnan(ref)
        assert (got[fin] - ref[fin]).abs().max().item() <= 5e-6 * max(1.0, ref[fin].abs().max().item())


def test_model_warp_method_signature():
    m = EMA_VFI(mid_channels=8).to(DEV)
    f2, flow = torch.randn(1, 3, 12, 20), torch.randn(1, 2, 12, 20) * 2
    got = m.warp(f2.to(DEV), torch.zeros(1, device=DEV), flow.to(DEV)).cpu()
    assert torch.allclose(got, oracle.warp(f2, flow), atol=1e-5)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 40, 64), (1, 33, 132), (1, 75, 260), (3, 31, 68), (1, 360, 640)])
def test_warp_inside_the_forward_equals_the_standalone_warp(dtype, shape):
    """The warp the 16-bit forward runs at mid_channels 64 (warp_tiled_kernel, compact 16-byte pixels: a wave takes whole rows of the
    tile since round 4) repeats emavfi_warp's arithmetic pixel for pixel: its `warped` tap must be EXACTLY the stand-alone warp of
    frame2 by the forward's own `flow` tap, rounded to what the tail buffer stores (IEEE f16 in both 16-bit models at this width:
    Plan::feat16).  Tile-edge widths (W % 64 = 0, 4, 68 - 64), heights around the 32-row tile, flows that leave the 8-pixel window
    (synthetic "stress" frames), several samples."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=6)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(36, B, H, W, "stress"))
    m = make_model(sd, dtype=dtype)
    with torch.no_grad():
        _, taps = m(f1, f2, return_taps=True)
    flow = taps["flow"].float().contiguous()
    assert torch.isfinite(flow).all() and flow.abs().max().item() > 8.0      # some taps leave the LDS window: the global-gather path runs too
    want = lib.warp(f2, flow).half().float()
    got = taps["warped"].float()
    assert torch.equal(got, want), f"{int((got != want).sum())} of {got.numel()} differ, max {(got - want).abs().max().item():.3e}"


# ------------------------------------------------------------------ conv3x3 (row C)
CONV_CASES = [  # (Cin, Cout, H, W, stride, act): one per reference layer shape + ragged sizes
    (6, 64, 40, 72, 1, 1), (64, 64, 24, 40, 1, 1), (64, 128, 33, 47, 2, 1), (128, 256, 18, 34, 2, 1),
    (256, 256, 9, 21, 1, 1), (64, 2, 19, 33, 1, 0), (67, 27, 16, 35, 1, 0), (67, 64, 17, 64, 1, 1),
    (64, 32, 8, 32, 1, 1), (32, 3, 21, 45, 1, 2), (8, 8, 23, 37, 1, 1), (8, 16, 23, 37, 2, 1),
    (11, 27, 5, 7, 1, 0), (35, 32, 12, 33, 1, 1), (3, 5, 1, 1, 1, 0), (16, 16, 1, 70, 1, 1), (16, 32, 70, 1, 2, 1),
    # conv_wreg.inl (256 output channels, weights streamed into registers): several column tiles, ragged edges, 3 / 5 chunks, one-row images
    (128, 256, 37, 131, 2, 1), (256, 256, 19, 67, 1, 1), (192, 250, 11, 40, 1, 0), (160, 256, 21, 30, 2, 0), (256, 256, 1, 33, 1, 1), (128, 256, 1, 5, 2, 1),
]


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-2), ("fp16", 3e-3)])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_matches_aten(case, dtype, tol):
    Cin, Cout, H, W, stride, act = case
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(2, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    got = lib.conv3x3(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, act=act, dtype=dtype).cpu()
    ref = F.conv2d(x, w, b, stride=stride, padding=1)
    ref = F.relu(ref) if act == 1 else ((torch.tanh(ref) + 1) / 2 if act == 2 else ref)
    assert got.shape == ref.shape
    assert rel_err(got, ref) <= tol


@pytest.mark.parametrize("dtype,tol", [("bf16", 2e-2), ("fp16", 3e-3)])
@pytest.mark.parametrize("case", [(128, 256, 37, 131, 2, 1), (256, 256, 19, 67, 1, 1)])
def test_conv3x3_round3_plans_of_the_256_channel_layers_still_match_aten(case, dtype, tol, monkeypatch):
    """EMAVFI_CONV_WREG=0 (a layout switch: the stage entry reads it per call) puts context_encoding.1 / .2 back on the round-3 tile kernel
    plans - conv3x3<32,8,2> and conv3x3<64,4,1> in two passes - which the A/B measurements of csrc/conv_wreg.inl compare against."""
    monkeypatch.setenv("EMAVFI_CONV_WREG", "0")
    Cin, Cout, H, W, stride, act = case
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    x = torch.randn(2, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g) * 0.1
    got = lib.conv3x3(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, act=act, dtype=dtype).cpu()
    ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=1))
    assert rel_err(got, ref) <= tol


def test_conv_wreg_random_shapes_match_aten():
    """csrc/conv_wreg.inl through emavfi_conv3x3: 60 random (stride, Cin in 128..256, Cout in 232..256, B <= 3, H <= 41, W <= 150, dtype,
    activation) against ATen - partial tiles in both directions, 2 to 8 chunks, outputs narrower than the eight fragments."""
    import random
    rnd = random.Random(7)
    for it in range(60):
        stride = rnd.choice((1, 2))
        Cin = rnd.choice((128, 192, 256)) if stride == 1 else rnd.choice((128, 160, 192, 256))
        Cout = rnd.choice((256, 250, 232, 248))
        H, W, B = rnd.randint(1, 41), rnd.randint(1, 150), rnd.randint(1, 3)
        dtype, act = rnd.choice(("bf16", "fp16")), rnd.choice((0, 1))
        g = torch.Generator().manual_seed(it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
        b = torch.randn(Cout, generator=g) * 0.1
        got = lib.conv3x3(x.to(DEV), w.to(DEV), b.to(DEV), stride=stride, act=act, dtype=dtype).cpu()
        ref = F.conv2d(x, w, b, stride=stride, padding=1)
        ref = F.relu(ref) if act else ref
        assert got.shape == ref.shape and torch.isfinite(got).all(), (it, stride, Cin, Cout, H, W, B, dtype)
        assert rel_err(got, ref) <= (2e-2 if dtype == "bf16" else 3e-3), (it, stride, Cin, Cout, H, W, B, dtype, rel_err(got, ref))


# ------------------------------------------------------------------ deformable conv (rows O, D)
@pytest.mark.parametrize("dtype,tol", [("fp32", 3e-5), ("bf16", 3e-2), ("fp16", 4e-3)])
@pytest.mark.parametrize("C,O,H,W,spread", [(67, 67, 19, 41, 2.0), (67, 67, 8, 32, 12.0), (11, 11, 23, 37, 1.5),
                                            (5, 7, 1, 1, 1.0), (19, 19, 9, 33, 3.0), (35, 35, 16, 16, 2.0)])
def test_deform_conv2d_matches_oracle(C, O, H, W, spread, dtype, tol):
    g = torch.Generator().manual_seed(C * 7 + H)
    x = torch.randn(2, C, H, W, generator=g)
    off = torch.randn(2, 18, H, W, generator=g) * spread
    msk = torch.rand(2, 9, H, W, generator=g)
    w = torch.randn(O, C, 3, 3, generator=g) / math.sqrt(C * 9)
    b = torch.randn(O, generator=g) * 0.1
    got = lib.deform_conv2d(x.to(DEV), off.to(DEV), msk.to(DEV), w.to(DEV), b.to(DEV), dtype=dtype).cpu()
    ref = oracle.deform_conv2d(x, off, msk, w, b)
    assert rel_err(got, ref) <= tol


def test_deform_known_answers_on_gpu():
    """The same clauses that pin the oracle (tests/test_oracle_deform.py), replayed on the HIP kernel."""
    x = torch.randn(1, 3, 10, 12)
    w = torch.randn(2, 3, 3, 3)
    zero_off, ones = torch.zeros(1, 18, 10, 12), torch.ones(1, 9, 10, 12)
    got = lib.deform_conv2d(x.to(DEV), zero_off.to(DEV), ones.to(DEV), w.to(DEV), None).cpu()
    assert torch.allclose(got, F.conv2d(x, w, None, padding=1), atol=2e-5)
    xo = torch.ones(1, 1, 6, 6)
    wc = torch.zeros(1, 1, 3, 3)
    wc[0, 0, 1, 1] = 1.0
    for dy, y, want in ((-0.25, 0, 0.75), (-1.0, 0, 0.0), (0.5, 5, 0.5), (1.0, 5, 0.0)):
        off = torch.zeros(1, 18, 6, 6)
        off[:, 8] = dy
        v = lib.deform_conv2d(xo.to(DEV), off.to(DEV), torch.ones(1, 9, 6, 6).to(DEV), wc.to(DEV), None).cpu()[0, 0, y, 3]
        assert abs(v.item() - want) <= 1e-6, (dy, y, v.item())


def test_deform_bf16_is_deterministic_at_two_workgroups_per_cu():
    """Regression: at 1280x720 the bf16 kernel runs two workgroups per CU; an earlier version lost
    corner weights intermittently there (see the header of csrc/deform.inl)."""
    g = torch.Generator().manual_seed(0)
    H, W = 720, 1280
    x = torch.randn(1, 67, H, W, generator=g).to(DEV)
    off = (torch.randn(1, 18, H, W, generator=g) * 2).to(DEV)
    msk = torch.rand(1, 9, H, W, generator=g).to(DEV)
    w = (torch.randn(67, 67, 3, 3, generator=g) / math.sqrt(67 * 9)).to(DEV)
    b = torch.randn(67, generator=g).to(DEV)
    ref = lib.deform_conv2d(x, off, msk, w, b, dtype="fp32")
    runs = [lib.deform_conv2d(x, off, msk, w, b, dtype="bf16").clone() for _ in range(4)]
    for r in runs:
        assert torch.equal(r, runs[0])
        assert (r - ref).abs().max().item() <= 0.05


def test_bf16_pack_saturates_at_the_f16_range():
    """include/emavfi.h: in bf16 mode the LDS-window deformable kernel (reference width) works on the f16 image of its input -
    |x| > 65504 is CLAMPED to +-65504 (v_cvt_pkrtz saturates; nothing becomes inf), everything else is the bf16 value exactly.
    Held against the oracle on the clamped, bf16-rounded input; the unclamped oracle differs by far more than the gate."""
    g = torch.Generator().manual_seed(65504)
    H, W = 12, 20
    x = torch.randn(1, 67, H, W, generator=g) * 4e4                      # ~10 % of the values beyond the f16 range
    off = torch.randn(1, 18, H, W, generator=g) * 1.5
    msk = torch.rand(1, 9, H, W, generator=g)
    w = torch.randn(67, 67, 3, 3, generator=g) / math.sqrt(67 * 9) * 1e-3
    b = torch.randn(67, generator=g) * 0.1
    assert (x.abs() > 65504).float().mean().item() > 0.05
    got = lib.deform_conv2d(x.to(DEV), off.to(DEV), msk.to(DEV), w.to(DEV), b.to(DEV), dtype="bf16").cpu()
    xb, wb = x.bfloat16().float(), w.bfloat16().float()
    ref_clamped = oracle.deform_conv2d(xb.clamp(-65504.0, 65504.0), off, msk, wb, b)
    ref_plain = oracle.deform_conv2d(xb, off, msk, wb, b)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref_clamped) <= 2e-2
    assert rel_err(ref_plain, ref_clamped) >= 5e-2                      # the clamp is what the kernel does, not a rounding detail


def test_pack_module_matches_oracle_block():
    sd = synth.synthetic_state_dict(seed=4, mid_channels=8)
    m = make_model(sd, mid=8)
    x = torch.randn(1, 11, 14, 33)
    got = m.attention_blocks[1](x.to(DEV)).cpu()
    assert rel_err(got, oracle.attention_block(sd, 1, x)) <= 5e-5


# ------------------------------------------------------------------ full forward (row T)
@pytest.mark.parametrize("name", ["tiny_mid8_24x40.npz", "tiny_mid8_23x37.npz"])
def test_forward_tiny_golden_every_stage(name):
    g = load_golden(name)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    m = make_model(sd, mid=8)
    with torch.no_grad():
        out, taps = m(torch.from_numpy(g["frame1"]).to(DEV), torch.from_numpy(g["frame2"]).to(DEV), return_taps=True)
    for k in STAGES:
        ref = torch.from_numpy(g["tap." + k])
        assert rel_err(taps[k].cpu(), ref) <= 2e-4, k
    assert (out.cpu() - torch.from_numpy(g["tap.out"])).abs().max().item() <= 1e-3


def test_forward_config1_rubberwhale():
    """BASELINE.json configs[0] input, replayed on the GPU against the reference-run output."""
    g = load_golden("cfg1_rubberwhale_256.npz")
    u8 = g["triplet_u8"]
    f1, f2 = synth._to_model_range(u8[0:1]), synth._to_model_range(u8[2:3])
    m = make_model(synth.synthetic_state_dict(seed=0))
    with torch.no_grad():
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    err = (out.cpu() - torch.from_numpy(g["out"])).abs().max().item()
    assert err <= 1e-3, err
    assert (taps["flow"].cpu() - torch.from_numpy(g["flow"])).abs().max().item() <= 2e-3
    assert out.min().item() >= 0.0 and out.max().item() <= 1.0


def _large_file(tag):
    return {"1080": "large_1080.npz", "odd": "large_odd.npz"}.get(tag, "large_checks.npz")


@pytest.mark.parametrize("tag", ["256", "256s", "720", "1080", "odd"])
def test_forward_large_samples_vs_reference_run(tag):
    """mid=64 at 256x256 (natural, stress), 1280x720, 1920x1080 (BASELINE configs[4]'s frame size) and 203x331 (B = 2, stress input: odd
    in both dimensions - partial tiles at every pyramid level, the warp's W % 4 != 0 path, flows up to 16 px): sampled pixels of
    every stage recorded from the reference's forward (tests/golden/large_checks.npz, large_1080.npz, large_odd.npz)."""
    g = load_golden(_large_file(tag))
    B, H, W, seed, kind = (int(v) for v in g[f"{tag}.meta"])
    f1, f2 = synth.synthetic_frames(seed, B, H, W, "stress" if kind else "natural")
    m = make_model(synth.synthetic_state_dict(seed=0))
    with torch.no_grad():
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    for k in STAGES:
        got = taps[k].contiguous().view(-1).cpu()[torch.from_numpy(g[f"{tag}.pos.{k}"])]
        ref = torch.from_numpy(g[f"{tag}.val.{k}"])
        lim = 1e-3 if k == "out" else 5e-4 * max(1.0, ref.abs().max().item())
        assert (got - ref).abs().max().item() <= lim, (k, (got - ref).abs().max().item())
        mean, l2, amax = g[f"{tag}.stats.{k}"]
        t = taps[k].double()
        assert abs(t.mean().item() - mean) <= 1e-4 * max(1.0, abs(mean)) + 1e-5, k
        assert abs(t.pow(2).sum().sqrt().item() - l2) <= 1e-4 * l2, k


@pytest.mark.parametrize("fixture", ["large_offsets.npz:off", "large_offsets16.npz:off16"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_forward_large_offsets_vs_reference_run(dtype, fixture):
    """Round 6: the fix-up is an arena in the (dead) LDS window now, filled by LDS-DMA in rounds of 31 samples; large_offsets16.npz
    (offsets spanning about +-16 px: p50 3.6, p99 11.3, max 23.8 - nearly every (wave, tap) group outside, several rounds per wave) joins.
    Round 5: the reference's forward with deformable offsets that span about +-8 px (tests/golden/large_offsets.npz: p50 1.9, p99 5.5,
    max 9.4 px; the benchmark's recipe stays within +-2).  At these offsets a large share of the pack kernels' (wave, tap) groups leave the
    staged window (R = 2) and run the fix-up loop - rewritten this round in deform_pack3.inl (branch-free prefetched gathers), unchanged
    in deform_f32w.inl - inside the FORWARD, with the offsets the kernels computed themselves, three packs in a row, against the sampled
    pixels of the reference's own run.  fp32: the BASELINE gate (1e-3 on the frame, 5e-4 relative on every stage; measured 2.2e-6 on the
    frame); 16-bit: gates with a factor ~3 of head room over what this fixture measures (bf16 57.3 dB / 7.0e-3, fp16 74.0 dB / 1.1e-3)."""
    fname, tag = fixture.split(":")
    g = load_golden(fname)
    B, H, W, seed, kind = (int(v) for v in g[f"{tag}.meta"])
    std, bias = (float(v) for v in g[f"{tag}.recipe"])
    sd = synth.synthetic_state_dict(seed=0, offset_std=std, offset_bias=bias)
    f1, f2 = synth.synthetic_frames(seed, B, H, W, "natural")
    m = make_model(sd, dtype=dtype)
    with torch.no_grad():
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    report = []
    for k in STAGES:
        got = taps[k].contiguous().view(-1).cpu()[torch.from_numpy(g[f"{tag}.pos.{k}"])]
        ref = torch.from_numpy(g[f"{tag}.val.{k}"])
        err = (got - ref).abs().max().item()
        report.append(f"{k} {err:.2e}")
        if dtype == "fp32":
            lim = 1e-3 if k == "out" else 5e-4 * max(1.0, ref.abs().max().item())
            assert err <= lim, (k, err)
    got = out.contiguous().view(-1).cpu()[torch.from_numpy(g[f"{tag}.pos.out"])]
    ref = torch.from_numpy(g[f"{tag}.val.out"])
    p, err = psnr(got, ref), (got - ref).abs().max().item()
    print(f"large offsets ({tag}), {dtype}: PSNR {p:.1f} dB, max-abs {err:.3e}; stages: " + ", ".join(report))
    if dtype != "fp32":
        min_psnr, max_abs = {"bf16": (52.0, 2.5e-2), "fp16": (68.0, 4e-3)}[dtype]
        assert p >= min_psnr and err <= max_abs
        # the kernels' own census (round 6): these offsets do drive the packs through the fix-up, three packs in a row
        with torch.no_grad():
            m(f1.to(DEV), f2.to(DEV))
        rows = m.pack_census()
        print("  census:", [None if r is None else (round(r["fixup_share"], 3), round(r["samples_outside_share"], 4), round(r["abs_offset_px_max"], 1)) for r in rows])
        assert all(r is not None and r["fixup_share"] > (0.3 if tag == "off16" else 0.05) for r in rows)


def test_forward_config2_batch16_256():
    """BASELINE.json configs[1]: B=16 256x256 fp32.  Oracle (CPU) on 4 of the 16 samples, <= 1e-3 each;
    every sample of the batch equals the same sample run alone, bit for bit (no cross-sample op)."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.synthetic_frames(21, 16, 256, 256, "natural")
    m = make_model(sd)
    with torch.no_grad():
        out = m(f1.to(DEV), f2.to(DEV)).cpu()
        for i in (0, 5, 10, 15):
            ref = oracle.forward(sd, f1[i:i + 1], f2[i:i + 1])
            err = (out[i:i + 1] - ref).abs().max().item()
            assert err <= 1e-3, (i, err)
            assert psnr(out[i:i + 1], ref) >= 60.0
        for i in (3, 12):
            alone = m(f1[i:i + 1].to(DEV), f2[i:i + 1].to(DEV)).cpu()
            assert torch.equal(alone, out[i:i + 1])


@pytest.mark.parametrize("H,W", [(5, 7), (1, 40), (33, 1), (64, 96)])
def test_forward_ragged_sizes(H, W):
    sd = synth.synthetic_state_dict(seed=6, mid_channels=8)
    f1, f2 = synth.synthetic_frames(8, 2, H, W, "stress")
    m = make_model(sd, mid=8)
    with torch.no_grad():
        out = m(f1.to(DEV), f2.to(DEV)).cpu()
    assert (out - oracle.forward(sd, f1, f2)).abs().max().item() <= 1e-3


@pytest.mark.parametrize("H,W", [(5, 7), (1, 36), (40, 1), (17, 33), (31, 100)])
def test_forward_ragged_sizes_mid64_both_dtypes(H, W):
    """The mid_channels = 64 kernels (persistent conv, LDS-window deform, tiled warp when W % 4 == 0)
    on sizes that are not multiples of any tile: fp32 <= 1e-3 vs the oracle, bf16 close to it."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.synthetic_frames(13, 2, H, W, "natural")
    ref = oracle.forward(sd, f1, f2)
    with torch.no_grad():
        a = make_model(sd, dtype="fp32")(f1.to(DEV), f2.to(DEV)).cpu()
        b = make_model(sd, dtype="bf16")(f1.to(DEV), f2.to(DEV)).cpu()
        c = make_model(sd, dtype="fp16")(f1.to(DEV), f2.to(DEV)).cpu()
    assert (a - ref).abs().max().item() <= 1e-3
    # 16-bit gates derived from what these shapes measure (worst bf16 4.3e-3 / 60.5 dB, fp16 1.2e-3 / 74.7 dB over the five sizes,
    # printed below): a factor ~4 of head room, not the 0.15 / 0.03 of round 2 that a 30x regression would have passed; PSNR
    # on every shape >= 16 px
    eb, ec = (b - ref).abs().max().item(), (c - ref).abs().max().item()
    print(f"ragged {H}x{W}: bf16 max-abs {eb:.3e} PSNR {psnr(b, ref):.1f} dB; fp16 max-abs {ec:.3e} PSNR {psnr(c, ref):.1f} dB")
    assert torch.isfinite(b).all() and eb <= 2e-2
    assert torch.isfinite(c).all() and ec <= 5e-3
    if H * W >= 16:
        assert psnr(b, ref) >= 52.0 and psnr(c, ref) >= 66.0


def test_forward_bf16_psnr():
    """BASELINE configs[2] arithmetic (bf16 convs, fp32 warp) on a natural pair: PSNR vs the fp32 oracle."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.synthetic_frames(1, 2, 256, 256, "natural")
    m = make_model(sd, dtype="bf16")
    with torch.no_grad():
        out = m(f1.to(DEV), f2.to(DEV)).cpu()
    ref = oracle.forward(sd, f1, f2)
    p = psnr(out, ref)
    err = (out - ref).abs().max().item()
    print(f"bf16 vs fp32 oracle: PSNR {p:.1f} dB, max-abs {err:.3e}")
    # measured 58.7 dB / 9e-3 with the round-1 kernels; bf16 storage (8 significant bits) through 19 layers
    assert p >= 52.0 and err <= 2.5e-2


@pytest.mark.parametrize("tag", ["720", "1080", "odd"])
@pytest.mark.parametrize("dtype,min_psnr,max_abs", [("bf16", 50.0, 3e-2), ("fp16", 65.0, 6e-3)])
def test_forward_720p_16bit_vs_reference_run(dtype, min_psnr, max_abs, tag):
    """The headline arithmetic at the headline size - and at 1920x1080, BASELINE configs[4]'s frame size -: the pixels sampled from
    the REFERENCE's own fp32 run (tests/golden/large_checks.npz, large_1080.npz) replayed in bf16 and fp16.  PSNR over the 4096
    sampled output pixels and their max-abs error; bounds follow from the storage precision (bf16: 8 significant bits, fp16: 11),
    not from the oracle."""
    g = load_golden(_large_file(tag))
    B, H, W, seed, kind = (int(v) for v in g[f"{tag}.meta"])
    f1, f2 = synth.synthetic_frames(seed, B, H, W, "stress" if kind else "natural")
    m = make_model(synth.synthetic_state_dict(seed=0), dtype=dtype)
    with torch.no_grad():
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    got = out.contiguous().view(-1).cpu()[torch.from_numpy(g[f"{tag}.pos.out"])]
    ref = torch.from_numpy(g[f"{tag}.val.out"])
    p, err = psnr(got, ref), (got - ref).abs().max().item()
    print(f"{dtype} {W}x{H} vs reference-run samples: PSNR {p:.1f} dB, max-abs {err:.3e}")
    if kind:   # stress input (large_odd.npz: saturated high-contrast frames, flows up to 16 px): every storage rounding of the flow moves
        min_psnr, max_abs = min_psnr - 8.0, max_abs * 2.0   # a sharp edge by a fraction of a pixel (measured: bf16 44.8 dB / 3.2e-2, fp16 60.8 dB / 4.0e-3)
    assert p >= min_psnr and err <= max_abs
    # intermediate stages stay within the storage type's relative precision of the reference run
    tol = {"bf16": 4e-2, "fp16": 5e-3}[dtype]
    for k in ("feat", "flow", "fused_2"):
        gk = taps[k].contiguous().view(-1).cpu()[torch.from_numpy(g[f"{tag}.pos.{k}"])]
        rk = torch.from_numpy(g[f"{tag}.val.{k}"])
        assert (gk - rk).abs().max().item() <= tol * max(1.0, rk.abs().max().item()), k


def test_batch8_720p_sample_matches_its_own_fp32_run():
    """BASELINE configs[2] as benchmarked (B=8 x 1280x720, bf16 / fp16): one sample of the batch against the SAME sample
    run alone in the exact-fp32 mode (which the oracle and reference-run tests pin)."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.fast_frames(5, 8, 720, 1280, device=DEV)
    with torch.no_grad():
        ref = make_model(sd, dtype="fp32")(f1[6:7].contiguous(), f2[6:7].contiguous())
        for dtype, min_psnr, max_abs in (("bf16", 50.0, 4e-2), ("fp16", 65.0, 8e-3)):
            out = make_model(sd, dtype=dtype)(f1, f2)[6:7]
            p, err = psnr(out, ref), (out - ref).abs().max().item()
            print(f"{dtype} B=8 720p sample 6 vs its fp32 run: PSNR {p:.1f} dB, max-abs {err:.3e}")
            assert p >= min_psnr and err <= max_abs


def test_forward_fp16_fast_mode_and_dtype_resolution():
    """compute_dtype="fp16" is the FAST half mode (every contraction, the deformable one included, in fp16): held to the fp32
    oracle at what 10 mantissa bits must meet and bf16 does not.  compute_dtype=None resolves by the active autocast:
    float16 -> "amp16" (the policy-exact mode, next test), bfloat16 -> "bf16", none -> exact fp32."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.synthetic_frames(1, 2, 256, 256, "natural")
    ref = oracle.forward(sd, f1, f2)
    with torch.no_grad():
        h = make_model(sd, dtype="fp16")(f1.to(DEV), f2.to(DEV)).cpu()
        b = make_model(sd, dtype="bf16")(f1.to(DEV), f2.to(DEV)).cpu()
        a = make_model(sd, dtype="amp16")(f1.to(DEV), f2.to(DEV))
        m = make_model(sd, dtype=None)
        with torch.autocast("cuda", dtype=torch.float16):
            auto_h = m(f1.to(DEV), f2.to(DEV))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            auto_b = m(f1.to(DEV), f2.to(DEV)).float().cpu()
        plain = m(f1.to(DEV), f2.to(DEV)).cpu()
    ph, pb = psnr(h, ref), psnr(b, ref)
    print(f"fp16 vs fp32 oracle: PSNR {ph:.1f} dB, max-abs {(h - ref).abs().max().item():.3e} (bf16: {pb:.1f} dB)")
    assert ph >= 60.0 and ph >= pb + 8.0
    assert auto_h.dtype == torch.float16 and a.dtype == torch.float16 and torch.equal(auto_h, a)   # autocast(fp16) = amp16
    assert torch.equal(auto_b, b)                                  # autocast(bf16) selects the bf16 arithmetic
    assert (plain - ref).abs().max().item() <= 1e-3               # no autocast: exact-fp32 parity mode
    assert not torch.equal(a.float().cpu(), h)                     # the policy-exact mode is not the fast mode


def test_autocast_op_policy_of_this_torch():
    """The per-op dtypes oracle.forward_autocast16 assumes, checked against the installed torch ON THE GPU (torch's own
    autocast dispatch, not the reference): conv2d / linear -> fp16, grid_sample / cat / fp32 + fp16 promote (fp32 here),
    sigmoid / tanh / pooling / scalar arithmetic keep fp16.  (torchvision is not installed: its deform_conv2d Autocast
    kernel - cast every argument to float, result back to the input's dtype - is restated, not checked.)"""
    x32 = torch.randn(1, 4, 8, 8, device=DEV)
    w = torch.randn(4, 4, 3, 3, device=DEV)
    bias = torch.randn(4, device=DEV)
    with torch.autocast("cuda", dtype=torch.float16):
        y = F.conv2d(x32, w, bias, padding=1)
        assert y.dtype == torch.float16
        assert F.linear(x32.flatten(1), torch.randn(5, 256, device=DEV), torch.randn(5, device=DEV)).dtype == torch.float16
        assert F.relu(y).dtype == torch.float16 and torch.sigmoid(y).dtype == torch.float16 and torch.tanh(y).dtype == torch.float16
        assert ((torch.tanh(y) + 1) / 2).dtype == torch.float16
        assert F.adaptive_avg_pool2d(y, 1).dtype == torch.float16
        grid = torch.zeros(1, 8, 8, 2, device=DEV)
        assert F.grid_sample(x32, grid + y[:, :2].permute(0, 2, 3, 1), align_corners=True).dtype == torch.float32
        assert (grid + y[:, :2].permute(0, 2, 3, 1)).dtype == torch.float32
        # grid_sampler is on autocast's PROMOTE list (widest input type): fp32 in the reference, whose frame2 and grid
        # are both fp32 (ema_vfi.py:162-169); all-fp16 arguments would stay fp16
        assert F.grid_sample(y, grid.half(), align_corners=True).dtype == torch.float16
        assert torch.cat([y, x32], dim=1).dtype == torch.float32
    # the cast of conv2d covers the bias: the result equals the fp16-rounded-operand convolution
    ref = F.conv2d(x32.half().float(), w.half().float(), bias.half().float(), padding=1).half()
    assert (y.float() - ref.float()).abs().max().item() <= 2e-2 * ref.float().abs().max().item()


@pytest.mark.parametrize("mid,H,W", [(8, 23, 37), (64, 96, 128)])
def test_forward_amp16_matches_the_autocast_restatement(mid, H, W):
    """compute_dtype=None under torch.autocast("cuda", float16) - what inference.py:159 runs - is the
    policy-exact mode: fp16 Conv2d / Linear, fp32 grid_sample and fp32 deform_conv2d on the unrounded fp32 fusion tensor
    with fp32 master weights.  UNPINNED (no GPU run of the reference exists); the check is oracle.forward_autocast16, a
    CPU restatement of that op policy: both sides round to fp16 at the same places, so they differ only where fp32
    summation order moves a value across an fp16 rounding boundary - bounds are a few fp16 steps, not a PSNR."""
    sd = synth.synthetic_state_dict(seed=0, mid_channels=mid)
    f1, f2 = synth.synthetic_frames(17, 2, H, W, "natural")
    rt = {}
    ref = oracle.forward_autocast16(sd, f1, f2, taps=rt)
    m = make_model(sd, mid=mid, dtype=None)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    assert out.dtype == torch.float16
    got = out.float().cpu()
    flow = taps["flow"].cpu()
    assert torch.equal(flow.half().float(), flow)                 # flow is an fp16 tensor under autocast
    assert torch.equal(taps["feat"].cpu().half().float(), taps["feat"].cpu())
    assert not torch.equal(taps["fused_2"].cpu().half().float(), taps["fused_2"].cpu())   # the DCN output stays fp32
    err = (got - ref).abs()
    same = (err == 0).float().mean().item()
    print(f"amp16 mid={mid}: max-abs {err.max().item():.3e}, mean-abs {err.mean().item():.3e}, identical {100 * same:.1f} %; "
          f"flow max-abs {(flow - rt['flow']).abs().max().item():.3e}")
    # one fp16 step of the frame is 2^-11 = 4.9e-4 in [0.5, 1): a few steps at worst, almost all pixels within one
    assert err.max().item() <= 4e-3 and err.mean().item() <= 2e-4
    assert (flow - rt["flow"]).abs().max().item() <= 3e-2        # flow spans +-8 px: fp16 step 7.8e-3 there
    for k in ("feat", "warped", "fused_0", "fused_2"):
        r = rt[k]
        assert (taps[k].cpu() - r).abs().max().item() <= 1e-2 * max(1.0, r.abs().max().item()), k
    # and it is a half-precision forward of the same network: close to the fp32 oracle
    assert psnr(got, oracle.forward(sd, f1, f2)) >= 55.0


@pytest.mark.parametrize("name", ["amp_mid8_23x37.npz", "amp_mid64_40x56.npz"])
def test_forward_amp16_replays_the_reference_run_under_cpu_autocast(name):
    """the fixtures tests/golden/amp_*.npz hold the REFERENCE's own forward executed under
    torch.autocast("cpu", float16) (tests/golden/make_golden.py amp; the op policy on this path is the same as CUDA autocast's,
    tests/test_oracle_golden.py).  Until round 5 only the CPU restatement was compared with them; here the HIP amp16 path is, stage by
    stage, in steps of the stage's own fp16 grid (2^-10 x max(1, max|stage|)) as that CPU test does.  What legitimately differs from the
    executed run is the fp32 summation order inside a convolution (the restatement itself differs from oneDNN's by up to 6 steps at
    mid_channels 64), which moves a value across an fp16 rounding boundary now and then - and, downstream of the flow, a one-step
    flow difference times the frame's gradient (the mid 8 fixture warps i.i.d. bytes).  Gates: every stage's MEAN error below 0.35 step,
    99.9 % of its elements within 8 steps; stages in front of the warp within 8 steps everywhere."""
    g = load_golden(name)
    mid, B, H, W, seed, kind = (int(v) for v in g["meta"])
    sd = synth.synthetic_state_dict(seed=seed, mid_channels=mid)
    f1, f2 = synth.synthetic_frames(seed, B, H, W, "stress" if kind else "natural")
    m = make_model(sd, mid=mid, dtype="amp16")
    with torch.no_grad():
        out, taps = m(f1.to(DEV), f2.to(DEV), return_taps=True)
    assert out.dtype == torch.float16
    taps = {k: v.float().cpu() for k, v in taps.items()}
    report = []
    for k in STAGES:
        want, got = torch.from_numpy(g["tap." + k]).float(), taps[k]
        assert got.shape == want.shape, k
        if str(g["dtype." + k]) == "float16":
            assert torch.equal(got, got.half().float()), k          # an fp16 tensor's worth of values where the reference has an fp16 tensor
        step = 2.0 ** -10 * max(1.0, float(want.abs().max()))
        d = (got - want).abs() / step
        q999 = d.flatten().kthvalue(max(1, int(0.999 * d.numel()))).values.item()
        report.append(f"{k}: max {d.max().item():.2f} mean {d.mean().item():.3f} q99.9 {q999:.2f} steps, identical {100 * (d == 0).float().mean().item():.1f} %")
        assert d.mean().item() <= 0.35 and q999 <= 8.0, (k, report[-1])
        if k in ("feat", "ctx", "flow"):
            assert d.max().item() <= 8.0, (k, report[-1])
    print(f"amp16 replay of {name}: " + "; ".join(report))


def test_full_size_properties_config3():
    """BASELINE configs[2] size (B=8, 1280x720): size-independent properties instead of a CPU replay -
    determinism, batch-permutation equivariance (bit exact), output range."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = synth.fast_frames(3, 8, 720, 1280, device=DEV)
    for dtype in ("fp32", "bf16", "fp16"):
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            a = m(f1, f2)
            b = m(f1, f2)
            perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=DEV)
            c = m(f1[perm].contiguous(), f2[perm].contiguous())
        assert torch.equal(a, b)
        assert torch.equal(a[perm], c)
        assert a.min().item() >= 0.0 and a.max().item() <= 1.0 and torch.isfinite(a).all()
        del a, b, c


# ------------------------------------------------------------------ error behaviour at the C-ABI
def test_cabi_errors_are_codes_not_aborts():
    L = lib.load()
    x = torch.zeros(1, 3, 16, 16, device=DEV)
    o = torch.empty_like(x)
    blob = torch.empty(L.emavfi_packed_bytes(3, 64, 3, lib.F32), dtype=torch.uint8, device=DEV)
    small = torch.empty(1024, dtype=torch.uint8, device=DEV)
    rc = L.emavfi_forward(3, 64, 3, blob.data_ptr(), blob.numel(), x.data_ptr(), x.data_ptr(), o.data_ptr(), small.data_ptr(), small.numel(),
                          1, 16, 16, lib.F32, None, None)
    assert rc == -3 and "workspace" in lib.last_error()
    rc = L.emavfi_forward(3, 64, 3, blob.data_ptr(), blob.numel(), x.data_ptr() + 4, x.data_ptr(), o.data_ptr(), small.data_ptr(),
                          small.numel(), 1, 16, 16, lib.F32, None, None)
    assert rc == -1 and "aligned" in lib.last_error()
    with pytest.raises(RuntimeError, match="inference-only"):
        EMA_VFI(mid_channels=8).to(DEV)(x, x)
    with pytest.raises(RuntimeError, match="no kernel instantiation"):
        lib.conv3x3(torch.zeros(1, 100, 8, 8, device=DEV), torch.zeros(8, 100, 3, 3, device=DEV), None)


def test_reload_state_dict_repacks():
    sd_a, sd_b = synth.synthetic_state_dict(seed=1, mid_channels=8), synth.synthetic_state_dict(seed=2, mid_channels=8)
    f1, f2 = synth.synthetic_frames(9, 1, 24, 32, "natural")
    m = make_model(sd_a, mid=8)
    with torch.no_grad():
        a = m(f1.to(DEV), f2.to(DEV)).cpu()
        m.load_state_dict(sd_b)
        b = m(f1.to(DEV), f2.to(DEV)).cpu()
    assert (a - oracle.forward(sd_a, f1, f2)).abs().max().item() <= 1e-3
    assert (b - oracle.forward(sd_b, f1, f2)).abs().max().item() <= 1e-3


_FUSION_AB = r"""
import sys, numpy as np, torch
sys.path[:0] = [r"%(pkg)s"]
from emavfi import EMA_VFI, synth
f1, f2 = synth.synthetic_frames(22, 2, 75, 131, "natural")
for dt in ("bf16", "fp16", "fp32"):
    m = EMA_VFI(compute_dtype=dt).to("cuda:0").eval()
    m.load_state_dict(synth.synthetic_state_dict(seed=21, mid_channels=64))
    with torch.no_grad():
        out = m(f1.cuda(), f2.cuda()).cpu()
    np.save(r"%(out)s" + "_" + dt + ".npy", out.numpy())
"""


@pytest.fixture
def switch():
    """Flip A/B bits of the library's launch-sequence switch word (emavfi_debug_switches: the environment is read once per process,
    never per call) and restore the word afterwards."""
    old = lib.debug_switches()

    def set_(bit, on):
        lib.debug_switches(~bit, bit if on else 0)
    yield set_
    lib.debug_switches(0, old)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("cout,act", [(64, "relu"), (64, "none"), (48, "none"), (40, "relu"), (32, "relu"), (24, "none"), (2, "none"), (3, "tanh01")])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 16, 32), (1, 17, 33), (1, 360, 640)])
def test_mfma16_conv_agrees_with_the_32x32_kernels(dtype, cout, act, shape, monkeypatch):
    """conv3x3_persist16_kernel (v_mfma_f32_16x16x32, the product path for 16-bit 64 -> 33..64 layers) against the 32x32x16
    kernels it replaces (EMAVFI_CONV_MFMA16=0): the fp32 accumulation groups 32 instead of 16 channels per MFMA, so the two
    agree to fp32 rounding - after rounding to the storage type at most one unit in the last place, and almost everywhere
    exactly.  (Accuracy against ATen is gated by test_conv3x3_matches_aten for both.)  EMAVFI_CONV_RING=0: the 33..64-channel
    outputs are the ring kernel's in the product (next test)."""
    monkeypatch.setenv("EMAVFI_CONV_RING", "0")
    B, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 64, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, 64, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    kw = dict(dtype=dtype, act={"relu": lib.ACT_RELU, "none": lib.ACT_NONE, "tanh01": lib.ACT_TANH01}[act])
    monkeypatch.setenv("EMAVFI_CONV_MFMA16", "0")
    ref = lib.conv3x3(x, w, b, **kw).clone()
    monkeypatch.setenv("EMAVFI_CONV_MFMA16", "1")
    got = lib.conv3x3(x, w, b, **kw).clone()
    assert torch.isfinite(got).all() and got.shape == ref.shape
    ulp = 2.0 ** (-7 if dtype == "bf16" else -10)
    err = (got - ref).abs()
    if act == "tanh01":   # the planar head returns fp32: agreement to accumulation-order rounding
        ulp = 1e-5
    assert (err <= ulp * ref.abs().clamp_min(2.0 ** -6)).all(), f"max {err.max().item():.3e}"
    if act == "tanh01":
        return
    assert (err > 0).float().mean().item() < 0.02, "the two kernels should differ in rare last-place roundings only"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,act", [(64, 64, "relu"), (64, 48, "none"), (64, 33, "relu"), (67, 64, "relu"), (67, 40, "none"), (65, 64, "none")])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 5), (1, 2, 64), (1, 17, 65), (3, 40, 128), (1, 360, 640)])
def test_ring_conv_agrees_with_the_tile_kernels(dtype, cin, cout, act, shape, monkeypatch):
    """conv3x3_ring_kernel (weights in registers, input rows through an LDS ring, the im2col tail for channels 64..66: the product
    path of the 16-bit 64..67 -> 33..64 layers) against round 2's plans (EMAVFI_CONV_RING=0, EMAVFI_CONV_MFMA16=0: the 32x32x16
    tile / persistent kernels at CK = 64 or 80).  Same rounded operands and products; the ring kernel accumulates even and odd
    k-groups in two chains (and the tail first), so the two agree to fp32 rounding - after rounding to the storage type at most
    one unit in the last place, and almost everywhere exactly.  Shapes: one row, one strip, a strip of one column, several
    segments per strip (360 rows), several samples."""
    B, H, W = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    kw = dict(dtype=dtype, act=lib.ACT_RELU if act == "relu" else lib.ACT_NONE)
    monkeypatch.setenv("EMAVFI_CONV_RING", "0")
    monkeypatch.setenv("EMAVFI_CONV_MFMA16", "0")
    ref = lib.conv3x3(x, w, b, **kw).clone()
    monkeypatch.delenv("EMAVFI_CONV_RING")
    monkeypatch.delenv("EMAVFI_CONV_MFMA16")
    got = lib.conv3x3(x, w, b, **kw).clone()
    assert torch.isfinite(got).all() and got.shape == ref.shape
    ulp = 2.0 ** (-7 if dtype == "bf16" else -10)
    err = (got - ref).abs()
    assert (err <= ulp * ref.abs().clamp_min(2.0 ** -6)).all(), f"max {err.max().item():.3e}"
    assert (err > 0).float().mean().item() < 0.02, "the two kernels should differ in rare last-place roundings only"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 2, 62), (1, 3, 63), (1, 17, 124), (1, 40, 125), (1, 360, 640)])
def test_fused_flow_head_equals_the_two_launch_path(dtype, shape, switch):
    """16-bit modes at mid_channels 64 run motion_estimation.1 + .2 as ONE launch: .1's rows stay in an LDS ring (zeroed outside
    the image - they are .2's padding) and the flow head is computed from them two rows behind (csrc/conv_ring.inl, HEAD).
    EMAVFI_CONV_HEAD=0 writes .1's tensor and runs the planar-head kernel (conv_light.inl).  Same rounded .1 rows, same head
    weights and MFMA shape; the head accumulates in two chains instead of one, so the fp32 flow agrees to accumulation-order
    rounding.  Widths around the 62-column strip pitch, one- and two-row images, several segments per strip."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=5)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(31, B, H, W, "natural"))
    flows, outs = [], []
    for flag in ("1", "0"):
        switch(lib.SW_NO_HEAD, flag == "0")
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            out, taps = m(f1, f2, return_taps=True)
        flows.append(taps["flow"].clone()); outs.append(out.clone())
    names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
    assert sum("+head" in n for n in names) == 0   # (the switch is still set: the enumeration follows it)
    switch(lib.SW_NO_HEAD, False)
    names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
    assert sum("+head" in n for n in names) == 1
    err = (flows[0] - flows[1]).abs().max().item()
    scale = flows[1].abs().max().item()
    print(f"{dtype} {shape}: flow max-abs diff {err:.3e} (|flow| <= {scale:.3f}); frame diff {(outs[0] - outs[1]).abs().max().item():.3e}")
    assert torch.isfinite(flows[0]).all() and err <= 2e-5 * max(1.0, scale)
    assert (outs[0] - outs[1]).abs().max().item() <= (1.5e-2 if dtype == "bf16" else 3e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 2, 62), (3, 9, 200), (1, 17, 124), (5, 40, 125), (1, 360, 640), (2, 720, 1280)])
def test_ring_kernels_row_ranges_equal_their_segments(dtype, shape, switch):
    """Round 5: a workgroup of a persistent LDS-ring kernel walks ONE contiguous range of the launch's rows (csrc/conv_ring.inl,
    RingWork: 512 equal ranges over all strips, a range may end one strip and start the next) instead of 45-row segments dealt
    round-robin (EMAVFI_RING_CHUNK=0).  The decomposition must not be visible: every output row is computed from the same operands in
    the same order, so the whole forward - all five ring kernels are on its path - is BIT-identical, flow and frame.  Shapes: fewer
    rows than workgroups (ranges of < 8 rows are merged by the host), ranges that cross strips and samples, one-row images, 720p."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=6)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(37, B, H, W, "natural"))
    res = []
    for segments in (False, True):
        switch(lib.SW_NO_RING_CHUNK, segments)
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            out, taps = m(f1, f2, return_taps=True)
        res.append((out.clone(), taps["flow"].clone()))
    switch(lib.SW_NO_RING_CHUNK, False)
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 2, 62), (1, 3, 63), (1, 17, 124), (3, 40, 125), (1, 360, 640)])
def test_fused_first_two_layers_equal_the_two_launch_path(dtype, shape, switch):
    """16-bit modes at mid_channels 64 run cat + feat_ext_conv1 + ReLU + conv_block_0 + ReLU as ONE launch: feat_ext_conv1's rows
    exist only in an LDS ring (csrc/conv_ring_first.inl).  EMAVFI_CONV_FIRSTRING=0 runs conv_first + the ring kernel.  Both
    stages repeat the unfused kernels' arithmetic operation for operation, so `feat` (three layers later) and the frame must
    be bit-identical.  Widths around the 62-column strip pitch, one- and two-row images, several segments per strip."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=6)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(32, B, H, W, "natural"))
    feats, outs = [], []
    for flag in ("1", "0"):
        switch(lib.SW_NO_FIRSTRING, flag == "0")
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            out, taps = m(f1, f2, return_taps=True)
        feats.append(taps["feat"].clone()); outs.append(out.clone())
        names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
        assert sum(n.startswith("conv_first+conv3x3") for n in names) == int(flag)
    assert torch.isfinite(feats[0]).all()
    assert torch.equal(feats[0], feats[1]), f"feat: {int((feats[0] != feats[1]).sum())} of {feats[0].numel()} differ, max {(feats[0] - feats[1]).abs().max().item():.3e}"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 2, 62), (1, 3, 63), (1, 4, 61), (1, 17, 124), (3, 40, 125), (1, 360, 640), (8, 90, 187)])
def test_fused_block_pair_equals_the_two_launch_path(dtype, shape, switch):
    """16-bit modes at mid_channels 64 run feat_ext_blocks.conv_block_1 + conv_block_2 (64 -> 64 -> 64, both + ReLU) as ONE launch
    (csrc/conv_ring2.inl): an eight-wave workgroup whose waves 0-3 compute conv_block_1's rows into an LDS ring (rounded to the storage
    type, zero outside the image) and whose waves 4-7 compute conv_block_2 from that ring two rows behind.  EMAVFI_CONV_RING2=0 runs
    the ring kernel twice.  Both stages repeat the unfused kernel's arithmetic operation for operation, so `feat` and the frame must
    be BIT-IDENTICAL.  Widths around the 62-column strip pitch, one- to four-row images (fewer rows than the two-row lag), several
    segments per strip, more strips than workgroups."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=8)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(34, B, H, W, "natural"))
    feats, outs = [], []
    for flag in ("1", "0"):
        switch(lib.SW_NO_RING2, flag == "0")
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            out, taps = m(f1, f2, return_taps=True)
        feats.append(taps["feat"].clone()); outs.append(out.clone())
        names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
        assert sum(n.startswith("conv3x3+conv3x3") for n in names) == int(flag)
    assert torch.isfinite(feats[0]).all()
    assert torch.equal(feats[0], feats[1]), f"feat: {int((feats[0] != feats[1]).sum())} of {feats[0].numel()} differ, max {(feats[0] - feats[1]).abs().max().item():.3e}"
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "amp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 4, 4), (1, 13, 129), (3, 40, 125), (1, 360, 640), (1, 509, 517)])
def test_pool_fused_context_conv_equals_the_stored_tensor_path(dtype, shape, switch):
    """16-bit modes at mid_channels 64: context_encoding.2 (csrc/conv_wreg.inl) does not store its output - AdaptiveAvgPool2d is its only
    reader (ema_vfi.py:82-83) - but writes the per-channel sums of every 4 x 32 pixel tile, which avg_pool_partial and
    context_linear_fold add in a fixed order.  EMAVFI_CONV_POOLFUSE=0 stores the tensor and pools it as before.  The same rounded
    values are summed in another order: ctx agrees to fp32 summation error (amp16: to one fp16 unit of the mean / Linear output), the
    frame to what that moves it.  Image sizes with partial tiles in both directions, fewer pixels than one tile, many tiles."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=5)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(35, B, H, W, "natural"))
    ctxs, outs = [], []
    for flag in ("1", "0"):
        switch(lib.SW_NO_POOLFUSE, flag == "0")
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            out, taps = m(f1, f2, return_taps=True)
        ctxs.append(taps["ctx"].clone()); outs.append(out.clone())
        names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
        assert sum("pool (tile sums)" in n for n in names) == int(flag)
    assert torch.isfinite(ctxs[0]).all()
    scale = ctxs[1].abs().max().item()
    err = (ctxs[0] - ctxs[1]).abs().max().item()
    assert err <= (2e-3 if dtype == "amp16" else 2e-6) * max(scale, 1e-3), (err, scale)
    # (a ctx that moves by 1e-7 flips a few storage-type roundings in motion_estimation.0: single pixels of the frame move by some units of bf16 / fp16)
    assert (outs[0] - outs[1]).abs().max().item() <= {"bf16": 1.5e-2, "fp16": 2e-3, "amp16": 4e-3}[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 1, 7), (1, 2, 62), (1, 3, 63), (1, 17, 124), (3, 40, 125), (1, 360, 640)])
def test_fused_reconstruction_tail_equals_the_two_launch_path(dtype, shape, switch):
    """16-bit modes at mid_channels 64 run reconstruction.1 + .2 (64 -> 32 + ReLU, 32 -> 3 + tanh, (t + 1) / 2) as ONE launch
    (csrc/conv_ring_tail.inl): .1's rows stay in an LDS ring, its 18 k-steps are split between two waves (another fp32 summation
    order than the unfused kernel's: .1's stored values may differ in the last place of the storage type in a few elements).
    EMAVFI_CONV_TAILFUSE=0 runs conv3x3_persist16_kernel + conv_light_kernel.  The frames must agree to a few such units; widths
    around the 62-column strip pitch, one- and two-row images, several segments per strip, several samples."""
    B, H, W = shape
    sd = synth.synthetic_state_dict(seed=7)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(33, B, H, W, "natural"))
    outs = []
    for flag in ("1", "0"):
        switch(lib.SW_NO_TAILFUSE, flag == "0")
        m = make_model(sd, dtype=dtype)
        with torch.no_grad():
            outs.append(m(f1, f2).clone())
        names = [n for n, _, _ in lib.forward_launches(3, 64, 3, B, H, W, dtype)]
        assert sum(n.startswith("conv3x3+tail") for n in names) == int(flag)
    err = (outs[0] - outs[1]).abs()
    print(f"{dtype} {shape}: frame max-abs diff {err.max().item():.3e}, {100 * (err > 0).float().mean().item():.2f} % of the elements differ")
    assert torch.isfinite(outs[0]).all() and outs[0].min() >= 0 and outs[0].max() <= 1
    assert err.max().item() <= (8e-3 if dtype == "bf16" else 1e-3)


def test_fused_first_layer_equals_pack_input_plus_conv(switch):
    """16-bit modes at mid_channels 64 compute cat(frame1, frame2) + feat_ext_conv1 + ReLU in ONE launch straight from the NCHW
    fp32 frames (csrc/conv_first.inl); EMAVFI_CONV_FIRST=0 runs pack_input + conv3x3.  Same rounded inputs, weights and products;
    the fp32 accumulation groups two taps per MFMA instead of one, so conv1's stored output may differ by one unit in the last place
    of the storage type in a few elements; three more layers later (the `feat` tap) and at the frame the two paths must still
    agree to a few such units.  Ragged size on purpose (tile remainders)."""
    sd = synth.synthetic_state_dict(seed=0)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(41, 2, 45, 77, "natural"))
    for dt, step, tol in (("bf16", 2.0 ** -7, 1.5e-2), ("fp16", 2.0 ** -10, 3e-3)):
        feats, outs = [], []
        for flag in ("1", "0"):
            switch(lib.SW_NO_CONV_FIRST, flag == "0")
            m = make_model(sd, dtype=dt)
            with torch.no_grad():
                out, taps = m(f1, f2, return_taps=True)
            feats.append(taps["feat"].clone()); outs.append(out.clone())
        rel = ((feats[0] - feats[1]).abs() / feats[1].abs().clamp_min(1.0)).max().item()
        frac = (feats[0] != feats[1]).float().mean().item()
        print(f"{dt}: feat max rel diff {rel:.3e} ({100 * frac:.2f} % of the elements differ), frame max-abs {(outs[0] - outs[1]).abs().max().item():.3e}")
        assert rel <= 8 * step and (outs[0] - outs[1]).abs().max().item() <= tol


def test_fused_pack_equals_the_two_launch_path(tmp_path):
    """bf16 / fp16 at the reference width run offset_conv inside the deform kernel (one launch per
    ModulatedDeformConvPack, csrc/deform_pack.inl); EMAVFI_NO_FUSED_OFFSET=1 runs conv3x3(EPI_OM) + the deform kernel
    reading the offsets from memory.  Ragged size on purpose.
      fp16: the same f16 products, but the one-launch pack sums the three warped-frame channels of all nine taps as one im2col
            step behind the 64 feature channels (csrc/deform_pack3.inl) where the stand-alone layers sum them tap by tap: the
            fp32 accumulation order differs, so offsets and samples differ in their last fp32 bits and a few output values by
            one or two fp16 steps - the frames must agree to four fp16 steps of a [0, 1] frame (2e-3) and be equally close to the
            fp32 frame (PSNR within 0.3 dB).
      bf16: the one-launch pack runs its offset_conv on the f16 image of the window (f16 MFMA on exactly converted bf16
            values; values below 2^-14 lose trailing bits) while the stand-alone conv is a bf16 MFMA on the bf16 tensor,
            so the offsets differ in the last bits: the two frames must agree to a fraction of a bf16 output step and
            be equally close to the exact-fp32 frame (which the oracle tests pin)."""
    import os
    import subprocess
    import sys
    from conftest import PKG
    outs = []
    for tag, extra in (("fused", {}), ("split", {"EMAVFI_NO_FUSED_OFFSET": "1"})):
        env = dict(os.environ, **extra)
        if not extra:
            env.pop("EMAVFI_NO_FUSED_OFFSET", None)
        code = _FUSION_AB % {"pkg": PKG, "out": str(tmp_path / tag)}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({dt: torch.from_numpy(np.load(str(tmp_path / tag) + f"_{dt}.npy")) for dt in ("bf16", "fp16", "fp32")})
    fused, split = outs
    assert torch.equal(fused["fp32"], split["fp32"])          # fp32 never fuses: identical runs
    dh = (fused["fp16"] - split["fp16"]).abs().max().item()
    ph, phs = psnr(fused["fp16"], fused["fp32"]), psnr(split["fp16"], fused["fp32"])
    print(f"fp16 fused vs two launches: max-abs {dh:.3e}; PSNR vs fp32 frame: fused {ph:.2f} dB, two launches {phs:.2f} dB")
    assert dh <= 2e-3 and ph >= 65.0 and ph >= phs - 0.3
    d = (fused["bf16"] - split["bf16"]).abs().max().item()
    pf, ps = psnr(fused["bf16"], fused["fp32"]), psnr(split["bf16"], fused["fp32"])
    print(f"bf16 fused vs two launches: max-abs {d:.3e}; PSNR vs fp32 frame: fused {pf:.2f} dB, two launches {ps:.2f} dB")
    assert d <= 8e-3 and pf >= 52.0 and pf >= ps - 0.5


def test_bf16_feat_as_f16_is_at_least_as_close_to_fp32(tmp_path):
    """bf16 model with the one-launch packs: `feat` and the warped tail are stored as IEEE f16 (what the first pack wants on chip: no
    conversion pass in it), and their other readers - context_encoding.0, motion_estimation.0 - run the f16 ring kernels on the
    bf16-rounded weights (Plan::feat16).  EMAVFI_PACK_F16_CHAIN=0 (read once per process: subprocesses) keeps bf16 everywhere.
    f16 keeps three more mantissa bits of `feat`, so the frame must be as close to the exact-fp32 frame as before, and the two
    bf16-model frames must agree to a fraction of a bf16 output step.  fp16 and fp32 models are not touched by the switch."""
    import os
    import subprocess
    import sys
    from conftest import PKG
    outs = []
    for tag, extra in (("f16feat", {}), ("bf16feat", {"EMAVFI_PACK_F16_CHAIN": "0"})):
        env = dict(os.environ, **extra)
        if not extra:
            env.pop("EMAVFI_PACK_F16_CHAIN", None)
        env["EMAVFI_CACHE"] = "0"
        code = _FUSION_AB % {"pkg": PKG, "out": str(tmp_path / tag)}
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append({dt: torch.from_numpy(np.load(str(tmp_path / tag) + f"_{dt}.npy")) for dt in ("bf16", "fp16", "fp32")})
    new, old = outs
    assert torch.equal(new["fp32"], old["fp32"]) and torch.equal(new["fp16"], old["fp16"])
    d = (new["bf16"] - old["bf16"]).abs().max().item()
    pn, po = psnr(new["bf16"], new["fp32"]), psnr(old["bf16"], new["fp32"])
    print(f"bf16 model, feat as f16 vs bf16: max-abs {d:.3e}; PSNR vs fp32 frame: {pn:.2f} dB vs {po:.2f} dB")
    assert d <= 1.5e-2 and pn >= 52.0 and pn >= po - 0.3


def test_bf16_feat_stored_as_f16_saturates_instead_of_overflowing():
    """include/emavfi.h: in the bf16 model `feat` is stored as IEEE f16 for the one-launch packs (Plan::feat16) and SATURATES at +-65504
    like the packs' own conversion: activations that bf16 could hold but f16 cannot must not become inf (inf - inf = NaN two layers
    later).  conv_block_2's weights and bias are scaled until `feat` exceeds the f16 range by far."""
    sd = synth.synthetic_state_dict(seed=3)
    sd = {k: v.clone() for k, v in sd.items()}
    for k in ("feat_ext_blocks.conv_block_2.0.weight", "feat_ext_blocks.conv_block_2.0.bias"):
        sd[k] = sd[k] * 3.0e5
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(34, 1, 40, 70, "natural"))
    m = make_model(sd, dtype="bf16")
    with torch.no_grad():
        out, taps = m(f1, f2, return_taps=True)
    feat = taps["feat"]
    assert torch.isfinite(feat).all() and feat.max().item() == 65504.0, feat.max().item()
    assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1


@pytest.mark.parametrize("nb", [1, 2, 4])
def test_other_block_counts_take_the_fused_paths_correctly(nb):
    """num_blocks != 3 at mid_channels 64: the plan's fusions depend on it (the fused first two layers write `feat` themselves when
    there is one block; `feat` is stored as f16 only from two blocks on; the pack chain has nb links).  The 16-bit paths must stay
    as close to the exact-fp32 path as at the reference's three blocks."""
    sd = synth.synthetic_state_dict(seed=2, mid_channels=64, num_blocks=nb)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(5, 2, 70, 131, "natural"))
    outs = {}
    for dt in ("fp32", "bf16", "fp16"):
        m = EMA_VFI(num_blocks=nb, compute_dtype=dt).to(DEV).eval()
        m.load_state_dict(sd)
        with torch.no_grad():
            outs[dt] = m(f1, f2)
    for dt, tol in (("bf16", 4e-2), ("fp16", 8e-3)):
        err = (outs[dt] - outs["fp32"]).abs().max().item()
        assert torch.isfinite(outs[dt]).all() and err <= tol, (nb, dt, err)


def test_forward_is_capturable_in_a_hip_graph():
    """The forward only enqueues kernels on the stream it is given and allocates nothing once its workspace exists, so
    it can be captured with torch.cuda.graph (hipGraph) and replayed on new frame contents - what a launch-bound small
    configuration (one 256x256 pair = 20 launches) wants."""
    sd = synth.synthetic_state_dict(seed=0)
    m = make_model(sd, dtype="bf16")
    a1, a2 = synth.synthetic_frames(31, 1, 256, 256, "natural")
    b1, b2 = synth.synthetic_frames(32, 1, 256, 256, "natural")
    s1, s2 = a1.to(DEV).clone(), a2.to(DEV).clone()
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # warm-up on the capture stream: packs weights, sizes the workspace
            for _ in range(2):
                eager_a = m(s1, s2).clone()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = m(s1, s2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager_a)
        s1.copy_(b1.to(DEV)); s2.copy_(b2.to(DEV))
        graph.replay()
        torch.cuda.synchronize()
        replay_b = out.clone()
        eager_b = m(b1.to(DEV), b2.to(DEV))
    assert torch.equal(replay_b, eager_b)


def test_forward_fuzz_sizes_16bit_against_fp32_path():
    """30 seeded random shapes (1..150 per side, batch 1..3): the bf16 and fp16 paths (one-launch packs, compact warp
    tail, persistent convs, every tile remainder) stay close to the exact-fp32 path, which the oracle tests pin."""
    import random
    rnd = random.Random(20260101)
    sd = synth.synthetic_state_dict(seed=0)
    m32, mb, mh = make_model(sd, dtype="fp32"), make_model(sd, dtype="bf16"), make_model(sd, dtype="fp16")
    worst, low = {"bf16": 0.0, "fp16": 0.0}, {"bf16": (1e9,), "fp16": (1e9,)}
    for k in range(30):
        B, H, W = rnd.randint(1, 3), rnd.randint(1, 150), rnd.randint(1, 150)
        f1, f2 = synth.synthetic_frames(1000 + k, B, H, W, "natural" if k % 3 else "stress")
        with torch.no_grad():
            ref = m32(f1.to(DEV), f2.to(DEV))
            # the worst single element of a bf16 frame is a maximum over ~1e5 rounding decisions: it moved 3.5e-2 -> 4.0e-2 (other
            # shape, same seeds) when the 64 -> 64 kernels changed their fp32 accumulation order; the PSNR floor below did not move
            for name, m, tol in (("bf16", mb, 5e-2), ("fp16", mh, 8e-3)):
                out = m(f1.to(DEV), f2.to(DEV))
                err = (out - ref).abs().max().item()
                assert torch.isfinite(out).all() and err <= tol, (name, B, H, W, err)
                if H * W >= 16:
                    ps = psnr(out, ref)
                    if ps < low[name][0]:
                        low[name] = (ps, B, H, W, "natural" if k % 3 else "stress")
                worst[name] = max(worst[name], err)
    print("worst max-abs vs fp32 path:", worst, "lowest PSNR:", low)
    # measured (deterministic): worst max-abs 4.0e-2 / 5.9e-3 and lowest PSNR 45.0 / 61.3 dB, all on the i.i.d.-byte "stress" frames
    # (natural frames stay below 1e-2 / 2e-3); round 2 allowed 0.2 / 0.04 and had no PSNR floor
    assert low["bf16"][0] >= 43.0 and low["fp16"][0] >= 59.0, low
