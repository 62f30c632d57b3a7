"""Host-side contracts of the drop-in boundary on a real GPU (SURVEY.md section 8b "re-entrant ... safe for one
process per GPU", 8e "one weight broadcast, shards equal the single-GPU result"): per-stream workspaces, captured
graphs surviving workspace growth, installed (broadcast) weight blobs, and a two-process shard run."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT
from emavfi import EMA_VFI, lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_model(sd, mid=64, dtype="fp32"):
    m = EMA_VFI(mid_channels=mid, compute_dtype=dtype).to(DEV).eval()
    m.load_state_dict(sd, strict=True)
    return m


def test_captured_graph_survives_workspace_growth():
    """A hipGraph bakes the workspace address in.  Growing the workspace of the same stream afterwards must not free
    the captured buffer: the replay still has to produce the eager result while other tensors churn the allocator."""
    lib.release_workspaces()
    sd = synth.synthetic_state_dict(seed=0)
    m = make_model(sd, dtype="bf16")
    a1, a2 = (t.to(DEV) for t in synth.synthetic_frames(41, 1, 128, 160, "natural"))
    big1, big2 = (t.to(DEV) for t in synth.synthetic_frames(42, 3, 200, 264, "natural"))
    side = torch.cuda.Stream()
    with torch.no_grad():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                eager_small = m(a1, a2).clone()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = m(a1, a2)
        key = (0, side.cuda_stream)
        captured_ptr = lib._ws_cache[key].data_ptr()
        # same stream, larger problem: the cached workspace must grow
        with torch.cuda.stream(side):
            eager_big = m(big1, big2).clone()
        side.synchronize()
        assert lib._ws_cache[key].data_ptr() != captured_ptr, "the larger forward was expected to re-allocate"
        assert any(b.data_ptr() == captured_ptr for b in lib._ws_graph_held), "captured workspace must be kept alive"
        # churn: anything the allocator hands out now must not alias the captured workspace
        junk = [torch.full((64 << 20,), 1.0e30, device=DEV) for _ in range(4)]
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager_small)
        with torch.cuda.stream(side):
            again_big = m(big1, big2)
        side.synchronize()
        assert torch.equal(again_big, eager_big)
        del junk
    del graph
    lib.release_workspaces()


def test_workspace_cache_is_bounded_over_transient_streams(monkeypatch):
    """A caller that runs every forward on a fresh stream must not pin one workspace per stream handle for ever: the cache keeps
    the EMAVFI_WS_CACHE_MAX most recently used buffers (buffers a captured graph may point at are exempt)."""
    lib.release_workspaces()
    monkeypatch.setenv("EMAVFI_WS_CACHE_MAX", "3")
    sd = synth.synthetic_state_dict(seed=0, mid_channels=8)
    m = make_model(sd, mid=8, dtype="fp32")
    x = [t.to(DEV) for t in synth.synthetic_frames(53, 1, 48, 64, "natural")]
    with torch.no_grad():
        ref = m(*x).clone()
        streams = [torch.cuda.Stream() for _ in range(7)]
        outs = []
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(m(*x))
            assert len(lib._ws_cache) <= 3
        torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert list(lib._ws_cache)[-1][1] == streams[-1].cuda_stream      # most recently used last
    lib.release_workspaces()


def test_two_streams_run_forwards_concurrently():
    """Two streams enqueue forwards back to back without host synchronisation; each stream has its own workspace, so
    the interleaved results equal the serial ones bit for bit (a shared scratch buffer would be a data race)."""
    lib.release_workspaces()
    sd = synth.synthetic_state_dict(seed=0)
    m = make_model(sd, dtype="bf16")
    xa = [t.to(DEV) for t in synth.synthetic_frames(51, 2, 192, 256, "natural")]
    xb = [t.to(DEV) for t in synth.synthetic_frames(52, 1, 360, 640, "stress")]
    with torch.no_grad():
        ref_a, ref_b = m(*xa).clone(), m(*xb).clone()
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs_a, outs_b = [], []
        for _ in range(6):
            with torch.cuda.stream(s1):
                outs_a.append(m(*xa))
            with torch.cuda.stream(s2):
                outs_b.append(m(*xb))
        torch.cuda.synchronize()
    assert len({k for k in lib._ws_cache if k[1] in (s1.cuda_stream, s2.cuda_stream)}) == 2
    def same(outs, ref, tag):
        for i, o in enumerate(outs):
            d = (o - ref).abs()
            assert torch.equal(o, ref), (f"{tag}[{i}]: {int((d > 0).sum())} of {d.numel()} elements differ, max {d.max().item():.3e}, "
                                         f"first at {(d > 0).nonzero()[:4].tolist()}")

    same(outs_a, ref_a, "a")
    same(outs_b, ref_b, "b")
    lib.release_workspaces()


def test_forward_is_exact_beside_a_foreign_gemm_stream():
    """Round 2's two-stream corruption: a packed f32 VALU operation with op_sel[1] = 1 reads src1's high
    half as zero in lanes 48-63 while a kernel on another stream executes MFMAs on the same SIMDs; the warp kernel, built with such operations,
    lost one bilinear term in every forward that overlapped a torch bf16 GEMM on another stream (`make TAG=_pk NOPK=` still
    does).  The library is built without packed f32 operations: every tap of every forward, and the standalone C-ABI warp,
    must equal the serial result bit for bit beside the GEMM stream."""
    lib.release_workspaces()
    sd = synth.synthetic_state_dict(seed=0)
    m = make_model(sd, dtype="bf16")
    xb = [t.to(DEV) for t in synth.synthetic_frames(52, 1, 360, 640, "stress")]     # flows that leave the warp's LDS window
    A = torch.randn(2048, 2048, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        ref, taps = m(*xb, return_taps=True)
        ref, taps = ref.clone(), {k: v.clone() for k, v in taps.items()}
        flow, f2 = taps["flow"].float().contiguous(), xb[1].contiguous()
        ref_warp = lib.warp(f2, flow).clone()
        A @ A
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(4):
            runs, warps = [], []
            for _ in range(10):
                with torch.cuda.stream(s1):
                    for _ in range(40):
                        A @ A
                with torch.cuda.stream(s2):
                    runs.append(m(*xb, return_taps=True))
                    warps += [lib.warp(f2, flow) for _ in range(10)]
            torch.cuda.synchronize()
            for o, tp in runs:
                for k in taps:
                    assert torch.equal(tp[k], taps[k]), f"stage {k}: {int((tp[k] != taps[k]).sum())} elements differ beside the GEMM stream"
                assert torch.equal(o, ref)
            for w in warps:
                assert torch.equal(w, ref_warp), f"emavfi_warp: {int((w != ref_warp).sum())} elements differ beside the GEMM stream"
    lib.release_workspaces()


def test_installed_blob_is_pinned_until_a_state_dict_is_loaded():
    """share_model_weights() installs rank 0's packed blob on the other ranks, whose own nn.Parameters stay random:
    the blob must survive .to() / dtype-preserving moves, an in-place edit must raise (it cannot be honoured), and
    load_state_dict() makes the parameters authoritative again."""
    sd = synth.synthetic_state_dict(seed=3, mid_channels=8)
    sd2 = synth.synthetic_state_dict(seed=4, mid_channels=8)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(9, 1, 24, 40, "natural"))
    src = make_model(sd, mid=8, dtype="bf16")
    with torch.no_grad():
        want = src(f1, f2)
        blob = src.packed_weights(lib.BF16, torch.device(DEV)).clone()
        other = EMA_VFI(mid_channels=8, compute_dtype="bf16").to(DEV).eval()     # random parameters, never loaded
        other.load_packed_weights("bf16", blob)
        assert torch.equal(other(f1, f2), want)
        other.to(DEV)
        other.float()
        assert torch.equal(other(f1, f2), want)                                 # still the installed blob
        with pytest.raises(RuntimeError, match="no packed weights"):
            other.compute_dtype = "fp32"
            other(f1, f2)
        other.compute_dtype = "bf16"
        other.feat_ext_conv1[0].bias.add_(1.0)    # (under no_grad; an edit through .data bypasses the version counter)
        with pytest.raises(RuntimeError, match="modified in place"):
            other(f1, f2)
        other.load_state_dict(sd2)
        got2 = other(f1, f2)
        assert torch.equal(got2, make_model(sd2, mid=8, dtype="bf16")(f1, f2))
        with pytest.raises(ValueError, match="uint8 blob"):
            other.load_packed_weights("bf16", blob[:-16])


_RANK_SCRIPT = r"""
import os, sys
sys.path[:0] = [r"%(pkg)s"]
import numpy as np, torch
from emavfi import EMA_VFI, synth, dist as vdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
vdist.init("gloo", dev)
f1, f2 = synth.synthetic_frames(77, %(n)d, %(h)d, %(w)d, "natural")
for dt in ("fp32", "bf16"):
    model = EMA_VFI(compute_dtype=dt).to(dev).eval()      # every rank starts from its own random init
    if rank == 0:
        model.load_state_dict(synth.synthetic_state_dict(seed=0))
    vdist.share_model_weights(model, dt, dev)             # the path's one collective
    lo, hi = vdist.shard_range(%(n)d, rank, world)
    with torch.no_grad():
        out = model(f1[lo:hi].to(dev), f2[lo:hi].to(dev)).cpu().numpy()
    np.save(os.path.join(r"%(out)s", f"shard_{dt}_{rank}.npy"), out)
    np.save(os.path.join(r"%(out)s", f"range_{dt}_{rank}.npy"), np.array([lo, hi]))
    if dt == "bf16":   # segment sharding of a frame STREAM (configs[4] logic): this rank's segment through the harness
        from emavfi import FrameInterpolator
        u8, _ = synth.synthetic_frames_u8(78, 1, %(h)d, %(w)d, "natural")
        stream = [np.roll(u8[0], 5 * i, axis=1) for i in range(6)]
        seg = list(FrameInterpolator(model, interpolation_factor=3, batch_pairs=2).run(stream, rank=rank, world=world))
        np.save(os.path.join(r"%(out)s", f"stream_{rank}.npy"), np.stack(seg))
vdist.barrier()
torch.distributed.destroy_process_group()
"""


def test_two_rank_shards_equal_the_single_process_forward(tmp_path):
    """BASELINE configs[3] logic on one GPU: two fresh processes (gloo, both on cuda:0), rank 0 packs and broadcasts the
    blob, rank 1 runs its shard_range slice from the RECEIVED blob; the concatenated shards equal a single-process
    forward of all pairs bit for bit.  (RCCL itself needs one device per rank; the driver's multi-GPU run covers it.)"""
    n, h, w = 5, 96, 160
    code = _RANK_SCRIPT % {"pkg": PKG, "n": n, "h": h, "w": w, "out": str(tmp_path)}
    port = 29500 + (os.getpid() % 400)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
    f1, f2 = synth.synthetic_frames(77, n, h, w, "natural")
    sd = synth.synthetic_state_dict(seed=0)
    for dt in ("fp32", "bf16"):
        with torch.no_grad():
            whole = make_model(sd, dtype=dt)(f1.to(DEV), f2.to(DEV)).cpu().numpy()
        ranges = [tuple(np.load(tmp_path / f"range_{dt}_{r}.npy")) for r in range(2)]
        assert ranges == [(0, 3), (3, 5)]
        got = np.concatenate([np.load(tmp_path / f"shard_{dt}_{r}.npy") for r in range(2)])
        assert got.shape == whole.shape and np.array_equal(got, whole), dt
    # the frame stream: rank 0 took pairs 0..2 (frames 0..3), rank 1 pairs 3..4 (frames 3..5: the boundary frame is shared)
    # and the final frame; concatenated, the two segments are the single-process stream frame for frame
    from emavfi import FrameInterpolator
    u8, _ = synth.synthetic_frames_u8(78, 1, h, w, "natural")
    stream = [np.roll(u8[0], 5 * i, axis=1) for i in range(6)]
    whole = np.stack(list(FrameInterpolator(make_model(sd, dtype="bf16"), interpolation_factor=3, batch_pairs=2).run(stream)))
    parts = [np.load(tmp_path / f"stream_{r}.npy") for r in range(2)]
    assert [len(x) for x in parts] == [12, 9] and np.array_equal(np.concatenate(parts), whole)


def test_packed_blob_is_cached_on_disk_by_content(tmp_path, monkeypatch):
    """the packed blob is cached keyed by a content hash of the state_dict + dtype + library build; a
    second construction with the same weights performs NO emavfi_pack_weights call, other weights or another dtype do."""
    monkeypatch.setenv("EMAVFI_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("EMAVFI_CACHE", "1")
    L = lib.load()
    calls = []
    real = L.emavfi_pack_weights

    def counting(*a):
        calls.append(1)
        return real(*a)

    monkeypatch.setattr(L, "emavfi_pack_weights", counting)
    sd = synth.synthetic_state_dict(seed=5, mid_channels=8)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(9, 1, 24, 40, "natural"))
    with torch.no_grad():
        a = make_model(sd, mid=8, dtype="bf16")(f1, f2)
        assert len(calls) == 1 and len(list(tmp_path.glob("packed_*.bin"))) == 1
        b = make_model(sd, mid=8, dtype="bf16")(f1, f2)               # fresh module, same content: cache hit
        assert len(calls) == 1 and torch.equal(a, b)
        make_model(sd, mid=8, dtype="fp32")(f1, f2)                   # another dtype: its own entry
        assert len(calls) == 2
        make_model(synth.synthetic_state_dict(seed=6, mid_channels=8), mid=8, dtype="bf16")(f1, f2)   # other weights
        assert len(calls) == 3 and len(list(tmp_path.glob("packed_*.bin"))) == 3
        monkeypatch.setenv("EMAVFI_CACHE", "0")                      # disabled: packs again, writes nothing
        make_model(sd, mid=8, dtype="bf16")(f1, f2)
        assert len(calls) == 4 and len(list(tmp_path.glob("packed_*.bin"))) == 3
        monkeypatch.setenv("EMAVFI_CACHE", "1")                      # a corrupted entry is not run: re-packed and rewritten
        for f in tmp_path.glob("packed_*.bin"):
            raw = bytearray(f.read_bytes()); raw[64] ^= 0x10; f.write_bytes(bytes(raw))
        c = make_model(sd, mid=8, dtype="bf16")(f1, f2)
        assert len(calls) == 5 and torch.equal(a, c)
        make_model(sd, mid=8, dtype="bf16")(f1, f2)
        assert len(calls) == 5


def test_foreign_blob_is_refused_or_poisoned():
    """VERDICT r3 item 8 / ADVICE r3: the packed blob carries a header (magic, library version, model, dtype, layout tag, size, payload
    checksum).  emavfi_packed_check names what is wrong with a foreign blob (a code, after a device-to-host copy); emavfi_forward, which
    must not synchronise, refuses a buffer that is too short and otherwise compares the header ON THE DEVICE: a blob of another dtype
    (same size: only the header can tell) or without a header yields an all-NaN frame, never plausible garbage."""
    import ctypes
    L = lib.load()
    sd = synth.synthetic_state_dict(seed=0)
    blobs = {}
    for mode in ("bf16", "fp16"):
        m = EMA_VFI(compute_dtype=mode).to(DEV).eval()
        m.load_state_dict(sd, strict=True)
        blobs[mode] = m.packed_weights(lib.dtype_code(mode), torch.device(DEV)).clone()
        lib.packed_check(3, 64, 3, lib.dtype_code(mode), blobs[mode])                     # what pack_weights wrote verifies
    assert blobs["bf16"].numel() == blobs["fp16"].numel()
    with pytest.raises(RuntimeError, match="dtype"):
        lib.packed_check(3, 64, 3, lib.BF16, blobs["fp16"])
    bad = blobs["bf16"].clone()
    bad[123456] ^= 0x20
    with pytest.raises(RuntimeError, match="checksum"):
        lib.packed_check(3, 64, 3, lib.BF16, bad)
    with pytest.raises(RuntimeError, match="dtype"):                                      # installing a foreign blob is refused
        EMA_VFI(compute_dtype="bf16").to(DEV).load_packed_weights("bf16", blobs["fp16"])
    headerless = blobs["bf16"].clone()
    headerless[:256] = 0
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(3, 1, 48, 64, "natural"))
    ws = torch.empty(L.emavfi_workspace_bytes(3, 64, 3, 1, 48, 64, lib.BF16), dtype=torch.uint8, device=DEV)
    out = torch.empty_like(f1)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def forward(blob, nbytes=None):
        rc = L.emavfi_forward(3, 64, 3, blob.data_ptr(), blob.numel() if nbytes is None else nbytes, f1.data_ptr(), f2.data_ptr(), out.data_ptr(),
                              ws.data_ptr(), ws.numel(), 1, 48, 64, lib.BF16, None, stream)
        torch.cuda.synchronize()
        return rc, out.clone()
    rc, good = forward(blobs["bf16"])
    assert rc == 0 and torch.isfinite(good).all() and good.min() >= 0 and good.max() <= 1
    rc, _ = forward(blobs["bf16"], nbytes=blobs["bf16"].numel() - 256)
    assert rc == -1 and "packed blob has" in lib.last_error()
    for name, blob in (("another dtype's blob", blobs["fp16"]), ("a blob without a header", headerless)):
        rc, got = forward(blob)
        assert rc == 0 and torch.isnan(got).all(), name
    rc, again = forward(blobs["bf16"])                                                    # (a corrupted payload byte is the checksum's business:
    assert rc == 0 and torch.equal(again, good)                                           #  emavfi_packed_check, above)


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("pieces,stagger", [(2, 0), (4, 0), (3, 1), (2, -1)])
def test_pipelined_forward_is_bit_identical(dtype, pieces, stagger):
    """the batch as `pieces` slices pipelined over the caller's stream and a side stream (EMA_VFI.pipeline;
    emavfi_forward_staged records the stage event that releases the next piece) - frame pairs are independent (ema_vfi.py:110-147 has
    no cross-sample op) and every kernel is batch-invariant per sample, so the frame must equal the one-sequence forward BIT FOR BIT,
    also on a non-default caller stream, for an uneven split (B = 5) and when the same model runs twice back to back (event reuse)."""
    sd = synth.synthetic_state_dict(seed=0)
    m = make_model(sd, dtype=dtype)
    f1, f2 = (t.to(DEV) for t in synth.synthetic_frames(61, 5, 76, 132, "stress"))   # (a slice must start 16-byte aligned: C*H*W % 4 == 0)
    lib.release_workspaces()
    with torch.no_grad():
        m.pipeline = 1
        ref = m(f1, f2).clone()
        m.pipeline, m.pipeline_stagger = pieces, stagger
        got = m(f1, f2).clone()
        assert (0, lib.side_stream(DEV).cuda_stream) in lib._ws_cache, "the side stream never ran a piece"
        odd = [t.to(DEV) for t in synth.synthetic_frames(62, 3, 23, 37, "natural")]   # odd sample size: runs as one sequence, same frame
        odd_got = m(*odd).clone()
        m.pipeline = 1
        assert torch.equal(m(*odd), odd_got)
        m.pipeline = pieces
        again = m(f1, f2).clone()
        caller = torch.cuda.Stream()
        caller.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(caller):
            on_side = m(f1, f2).clone()
        caller.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.equal(again, ref) and torch.equal(on_side, ref)
    assert torch.isfinite(ref).all()
