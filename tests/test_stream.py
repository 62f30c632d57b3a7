"""GPU pre/post-processing and the streaming harness, against a plain
restatement of the reference's host loop (inference.py:38-58, 146-205)."""
import numpy as np
import pytest
import torch

from emavfi import EMA_VFI, FrameInterpolator, lib, synth
from oracle import emavfi_oracle as oracle

MEAN = np.array([0.485, 0.456, 0.406])   # inference.py:40 (float64 in numpy, as there)
STD = np.array([0.229, 0.224, 0.225])


def ref_preprocess(u8_hwc):
    """transforms.ToTensor() + Normalize (inference.py:38-41): /255 then (x - mean) / std, fp32."""
    t = torch.from_numpy(u8_hwc).permute(2, 0, 1).float().div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub(mean).div(std)


def ref_denormalize(t_chw):
    """denormalize_frame (inference.py:51-58) verbatim in numpy."""
    frame = t_chw.float().numpy()
    frame = np.transpose(frame, (1, 2, 0))
    frame = (frame * STD) + MEAN
    frame = np.clip(frame, 0, 1)
    return (frame * 255).astype(np.uint8)


def ref_loop(frames, forward, factor, interval):
    """The reference's while-loop (inference.py:158-201) on in-memory frames."""
    out, it = [], iter(frames)
    frame1 = next(it, None)
    if frame1 is None:
        return out
    t1, frame_num = ref_preprocess(frame1), 0
    while True:
        frame_num += 1
        frame2 = next(it, None)
        if frame_num % interval == 0:
            if frame2 is None:
                out.append(frame1)
                break
            t2 = ref_preprocess(frame2)
            for _ in range(factor):
                out.append(ref_denormalize(forward(t1[None], t2[None])[0]))
            out.append(ref_denormalize(t1))
            frame1, t1 = frame2, t2
        else:
            if frame2 is None:
                out.append(ref_denormalize(t1))
                break
            frame1, t1 = frame2, ref_preprocess(frame2)
    return out


def test_schedule_matches_reference_loop_counts():
    for n in range(0, 9):
        for interval in (1, 2, 3):
            frames = [np.full((2, 2, 3), i, np.uint8) for i in range(n)]
            ref = ref_loop(frames, lambda a, b: torch.zeros(1, 3, 2, 2), 2, interval)
            pairs, last, roundtrip = FrameInterpolator.schedule(n, interval)
            assert (0 if last is None else len(pairs) * 3 + 1) == len(ref), (n, interval)
            # the last frame is written raw exactly when the loop ends in its pair branch (always for interval 1): with
            # 2x2 constant frames the round trip is the identity, so compare against the loop's own branch instead
            if n:
                frame_num, left = 0, n - 1
                while True:
                    frame_num += 1
                    if left == 0:
                        break
                    left -= 1
                assert roundtrip == (frame_num % interval != 0), (n, interval)
    assert FrameInterpolator.schedule(7, 1)[2] is False and FrameInterpolator.schedule(7, 2)[2] is True
    assert FrameInterpolator.schedule(6, 2)[2] is False and FrameInterpolator.schedule(4, 3)[2] is True and FrameInterpolator.schedule(3, 3)[2] is False


def test_segments_concatenate_to_the_single_process_stream():
    """Segment sharding (one process per GPU, BASELINE configs[4]): for n = 0..9 frames, world = 1..4, interval 1..3 and
    factor 0..3 the ranks' emission plans, concatenated in rank order, ARE the single-process plan; a rank reads a contiguous
    frame range that covers its pairs, and with interval 1 its last frame is the next non-empty rank's first."""
    for n in range(0, 10):
        for interval in (1, 2, 3):
            for factor in (0, 1, 3):
                whole = FrameInterpolator.emission_plan(n, factor, interval)
                pairs, last, _ = FrameInterpolator.schedule(n, interval)
                assert len(whole) == (0 if last is None else len(pairs) * (factor + 1) + 1)
                for world in (1, 2, 3, 4):
                    cat, prev_hi, covered = [], None, []
                    for rank in range(world):
                        mine, lo, hi, tail = FrameInterpolator.segment(n, interval, rank, world)
                        cat += FrameInterpolator.emission_plan(n, factor, interval, rank, world)
                        covered += mine
                        assert tail == (last is not None and rank == world - 1)
                        assert all(lo <= f < hi for p in mine for f in p) and 0 <= lo <= hi <= max(n, 0)
                        if mine:
                            assert (lo, hi) == (mine[0][0], max(mine[-1][1], last if tail else 0) + 1)
                            if interval == 1 and prev_hi is not None:
                                assert lo == prev_hi - 1          # segment boundaries share exactly one frame
                            prev_hi = mine[-1][1] + 1
                    assert cat == whole, (n, interval, factor, world)
                    assert covered == pairs


@pytest.mark.gpu
def test_sharded_runs_concatenate_bit_exactly():
    """run(frames, rank, world) for every rank of a 3-way split, one after the other on one device: the concatenation is the
    single-process output frame for frame (a pair's forward depends on its two frames only; batches are cut differently, and
    the forward is batch-invariant bit for bit).  A lazy sequence is indexed inside the rank's segment only."""
    sd = synth.synthetic_state_dict(seed=24, mid_channels=8)
    base, _ = synth.synthetic_frames_u8(9, 1, 24, 32, "natural")
    frames = [np.roll(base[0], 3 * i, axis=1) for i in range(8)]
    model = EMA_VFI(mid_channels=8, compute_dtype="bf16").cuda().eval()
    model.load_state_dict(sd)

    class Lazy:
        def __init__(self):
            self.touched = set()
        def __len__(self):
            return len(frames)
        def __getitem__(self, i):
            self.touched.add(i)
            return frames[i]

    for factor, interval, mode in ((3, 1, "reference"), (1, 2, "reference"), (3, 1, "recursive")):
        fi = FrameInterpolator(model, factor, interval, batch_pairs=2, mode=mode)
        whole = list(fi.run(frames))
        cat = []
        for rank in range(3):
            lazy = Lazy()
            cat += list(fi.run(lazy, rank=rank, world=3))
            _, lo, hi, _ = FrameInterpolator.segment(len(frames), interval, rank, 3)
            assert lazy.touched <= set(range(lo, hi))
        assert len(cat) == len(whole) == fi.count_outputs(len(frames))
        for k, (a, b) in enumerate(zip(cat, whole)):
            assert np.array_equal(a, b), (factor, interval, mode, k)


@pytest.mark.gpu
def test_preprocess_and_postprocess_bit_exact():
    g = np.random.default_rng(0)
    u8 = g.integers(0, 256, (3, 37, 53, 3), dtype=np.uint8)
    got = lib.preprocess_u8(torch.from_numpy(u8).cuda()).cpu()
    ref = torch.stack([ref_preprocess(f) for f in u8])
    assert torch.equal(got, ref)
    # H*W % 4 == 0 takes the 4-pixels-per-thread kernels; a pinned host buffer is read / written in place
    v8 = g.integers(0, 256, (2, 36, 52, 3), dtype=np.uint8)
    pin = torch.from_numpy(v8).pin_memory()
    for src in (torch.from_numpy(v8).cuda(), pin):
        assert torch.equal(lib.preprocess_u8(src, device="cuda:0").cpu(), torch.stack([ref_preprocess(f) for f in v8]))
    y = torch.from_numpy(g.normal(0.4, 0.6, (2, 3, 36, 52)).astype(np.float32))
    out_pin = torch.empty(2, 36, 52, 3, dtype=torch.uint8).pin_memory()
    lib.postprocess_u8(y.cuda(), denormalize=True, out=out_pin)
    torch.cuda.synchronize()
    assert np.array_equal(out_pin.numpy(), np.stack([ref_denormalize(t) for t in y]))
    assert np.array_equal(lib.postprocess_u8(y.cuda(), denormalize=True).cpu().numpy(), out_pin.numpy())
    x = torch.from_numpy(g.normal(0.4, 0.6, (2, 3, 19, 31)).astype(np.float32))
    x[0, 0, 0, 0], x[0, 1, 0, 0] = 5.0, -5.0
    for denorm in (True, False):
        got = lib.postprocess_u8(x.cuda(), denormalize=denorm).cpu().numpy()
        if denorm:
            ref = np.stack([ref_denormalize(t) for t in x])
        else:
            ref = (np.clip(np.transpose(x.numpy(), (0, 2, 3, 1)).astype(np.float64), 0, 1) * 255).astype(np.uint8)
        assert np.array_equal(got, ref)
    # the 16-byte-quad kernels (round 5: whole batch a multiple of 16 bytes): quads that straddle pixels AND frames (420-byte frames)
    w8 = g.integers(0, 256, (4, 10, 14, 3), dtype=np.uint8)
    for src in (torch.from_numpy(w8).cuda(), torch.from_numpy(w8).pin_memory()):
        assert torch.equal(lib.preprocess_u8(src, device="cuda:0").cpu(), torch.stack([ref_preprocess(f) for f in w8]))
    z = torch.from_numpy(g.normal(0.4, 0.6, (4, 3, 10, 14)).astype(np.float32))
    z[1, 2, 9, 13], z[2, 0, 0, 0] = float("nan"), 7.0
    for denorm in (True, False):
        zp = torch.empty(4, 10, 14, 3, dtype=torch.uint8).pin_memory()
        lib.postprocess_u8(z.cuda(), denormalize=denorm, out=zp)
        torch.cuda.synchronize()
        ref = np.stack([ref_denormalize(t) for t in z]) if denorm else (np.clip(np.nan_to_num(np.transpose(z.numpy(), (0, 2, 3, 1)).astype(np.float64), nan=0.0), 0, 1) * 255).astype(np.uint8)
        keep = ~np.isnan(np.transpose(z.numpy(), (0, 2, 3, 1)))      # (a NaN has no defined uint8 image in numpy; the kernels write 0)
        assert np.array_equal(zp.numpy()[keep], ref[keep]) and zp.numpy()[1, 9, 13, 2] == 0
        assert np.array_equal(zp.numpy(), lib.postprocess_u8(z.cuda(), denormalize=denorm).cpu().numpy())
    # round trip of a source frame: what the reference writes for frame1 (inference.py:187-188)
    rt = lib.postprocess_u8(lib.preprocess_u8(torch.from_numpy(u8).cuda())).cpu().numpy()
    assert np.array_equal(rt, np.stack([ref_denormalize(ref_preprocess(f)) for f in u8]))


@pytest.mark.gpu
@pytest.mark.parametrize("factor,interval,batch", [(1, 1, 8), (3, 1, 2), (2, 2, 3)])
def test_stream_matches_reference_loop(factor, interval, batch):
    sd = synth.synthetic_state_dict(seed=21, mid_channels=8)
    f1, _ = synth.synthetic_frames_u8(5, 7, 24, 32, "natural")
    frames = [f for f in f1]
    model = EMA_VFI(mid_channels=8, compute_dtype="fp32").cuda().eval()
    model.load_state_dict(sd)
    got = list(FrameInterpolator(model, factor, interval, batch_pairs=batch).run(frames))
    # the two transports of the harness - SDMA copies + device kernels (default, round 5) and kernels that read / write the pinned
    # buffers themselves (zero_copy=True, rounds 1-4) - emit the same bytes; so does a longer stream, whose first batch is half-size
    # (ramp-up) and whose source-frame round trips leave from the pre lane
    zc = list(FrameInterpolator(model, factor, interval, batch_pairs=batch, zero_copy=True).run(frames))
    assert len(zc) == len(got) and all(np.array_equal(a, b) for a, b in zip(zc, got))
    if interval == 1:
        long_frames = frames * 3
        a = list(FrameInterpolator(model, factor, interval, batch_pairs=2).run(long_frames))
        b = list(FrameInterpolator(model, factor, interval, batch_pairs=len(long_frames), zero_copy=True).run(long_frames))
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    ref = ref_loop(frames, lambda a, b: oracle.forward(sd, a, b), factor, interval)
    assert len(got) == len(ref)
    assert len(got) == FrameInterpolator(model, factor, interval).count_outputs(len(frames))
    for a, b in zip(got, ref):
        assert a.shape == b.shape and a.dtype == np.uint8
        # predictions may differ by one count where a ~1e-6 difference crosses a truncation boundary
        assert np.abs(a.astype(np.int16) - b.astype(np.int16)).max() <= 1
    # the last frame: raw when the reference's loop ends in its pair branch, round-tripped (bit-exact pre/post-processing)
    # when it ends in the skip branch (inference.py:198-201) - content equal to the reference loop's either way
    assert np.array_equal(got[-1], ref[-1])
    if interval == 1:
        assert np.array_equal(got[-1], frames[-1])


@pytest.mark.gpu
def test_stream_without_quirks_passes_sources_through():
    sd = synth.synthetic_state_dict(seed=22, mid_channels=8)
    frames = [f for f in synth.synthetic_frames_u8(6, 4, 16, 32, "stress")[0]]
    model = EMA_VFI(mid_channels=8, compute_dtype="fp32").cuda().eval()
    model.load_state_dict(sd)
    got = list(FrameInterpolator(model, 1, 1, reference_quirks=False).run(frames))
    assert len(got) == 2 * 3 + 1
    for k in range(3):
        assert np.array_equal(got[2 * k + 1], frames[k])


@pytest.mark.gpu
def test_recursive_mode_emits_distinct_midpoints():
    """Opt-in mode that the reference lacks (it has no timestep input): factor 3 -> quarter points."""
    sd = synth.synthetic_state_dict(seed=23, mid_channels=8)
    frames = [f for f in synth.synthetic_frames_u8(7, 5, 24, 32, "natural")[0]]
    model = EMA_VFI(mid_channels=8, compute_dtype="fp32").cuda().eval()
    model.load_state_dict(sd)
    fi = FrameInterpolator(model, 3, 1, batch_pairs=2, reference_quirks=False, mode="recursive")
    got = list(fi.run(frames))
    assert len(got) == fi.count_outputs(len(frames)) == 4 * 4 + 1
    # oracle replay of the recursion for the first pair
    mean = torch.tensor(MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(1, 3, 1, 1)
    a, b = ref_preprocess(frames[0])[None], ref_preprocess(frames[1])[None]
    mid = oracle.forward(sd, a, b)
    q1 = oracle.forward(sd, a, (mid - mean) / std)
    q3 = oracle.forward(sd, (mid - mean) / std, b)
    for g, r in zip(got[:3], (q1, mid, q3)):
        want = (np.clip(np.transpose(r[0].numpy(), (1, 2, 0)).astype(np.float64), 0, 1) * 255).astype(np.uint8)
        assert np.abs(g.astype(np.int16) - want.astype(np.int16)).max() <= 1
    assert np.array_equal(got[3], frames[0])
    with pytest.raises(ValueError):
        FrameInterpolator(model, 2, mode="recursive")


@pytest.mark.gpu
def test_copy_out_false_yields_the_same_frames_as_views():
    """copy_out=False hands out views into the pinned result buffers (valid until the generator is advanced):
    consumed one at a time they are the frames copy_out=True returns."""
    sd = synth.synthetic_state_dict(seed=3, mid_channels=8)
    m = EMA_VFI(mid_channels=8, compute_dtype="fp32").cuda().eval()
    m.load_state_dict(sd)
    a, _ = synth.synthetic_frames_u8(5, 1, 24, 40, "natural")
    frames = [np.roll(a[0], 2 * i, axis=1) for i in range(7)]
    for quirks in (True, False):
        ref = list(FrameInterpolator(m, interpolation_factor=2, batch_pairs=2, reference_quirks=quirks).run(frames))
        n = 0
        for k, f in enumerate(FrameInterpolator(m, interpolation_factor=2, batch_pairs=2, reference_quirks=quirks,
                                                copy_out=False).run(frames)):
            assert np.array_equal(f, ref[k]), (quirks, k)
            n += 1
        assert n == len(ref)
