#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native EMA-VFI forward (BASELINE.json metric).

One "step" = one forward pass of the hot path over one batch of synthetic frame pairs already
resident in HBM.  Workload at any N: BASELINE.json configs[2] per GPU - batch 8 of 1280x720
pairs, bf16 convolutions + fp32 warp (weak scaling: every rank gets its own 8 pairs; the only
collective is one RCCL broadcast of the packed weights before the timed region).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (largest share of device
time), measured with HIP events recorded around every launch on the launch stream during the
timed steps; `cpu_baseline` times the CPU oracle on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
import hashlib
import json
import os
import statistics
import sys
import time

# RCCL on this driver needs dmabuf IPC; the variable is read when HSA initialises, i.e. before the first GPU call
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "video-frame-interpolation_amd"))
sys.path.insert(0, ROOT)



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per GPU (configs[2]: 8)")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="bf16 = BASELINE configs[2]; fp16 = the autocast arithmetic of the reference; fp32 = parity mode")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed CPU-oracle forwards of one full frame (0 = skip the CPU baseline)")
    ap.add_argument("--cpu-rows", type=int, default=None, help="deprecated: 0 = skip the CPU baseline")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (fp32 / fp16 / 256x256 / warp / CPU)")
    ap.add_argument("--no-events", action="store_true", help="time the plain entry point (no per-launch events)")
    ap.add_argument("--rehearse", action="store_true",
                    help="no GPU: exercise launch, rendezvous, the weight-blob broadcast, barrier / MAX timing and the JSON relay with a "
                         "stand-in step (CPU test of the N > 1 plumbing; `value` is null)")
    args = ap.parse_args()
    if args.cpu_rows == 0:
        args.cpu_reps = 0
    return args


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves.  The parent touches neither
    torch nor the GPU (a process that has initialised HIP must not exec, and a fresh child per rank is what torchrun gives anyway):
    it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
    CHILD process, relays rank 0's single JSON line on stdout and exits with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)   # stderr passes through
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)   # anything else a rank printed is diagnostics, not the result
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


ARGS = None
if __name__ == "__main__":
    ARGS = parse_args()
    if ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(ARGS)

import torch  # noqa: E402

PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3,   # dense MFMA TFLOP/s (MI355X_MICROARCH.md, chip-level table)
        "fp32x3": 2500.0 / 3.0}                            # the three-term f16 split: three f16 MFMAs per fp32-accurate product
PEAK_HBM_GBS = 8000.0                     # HBM3E spec peak (same table)
FLOP_PER_PX = 1054908.0                   # whole forward, unfolded
DCN_FLOP_PER_PX = 2.0 * 9.0 * 67.0 * 67.0  # one deform_conv2d 67 -> 67 (the offset conv is an ATen conv2d)


class Hip:
    """hipEvent_* from the HIP runtime already loaded into this process (torch's)."""

    def __init__(self):
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        self.lib = ctypes.CDLL(path or "libamdhip64.so")
        self.lib.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.lib.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.lib.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.lib.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

    def events(self, n):
        arr = (ctypes.c_void_p * n)()
        for i in range(n):
            e = ctypes.c_void_p()
            rc = self.lib.hipEventCreate(ctypes.byref(e))
            if rc != 0:
                raise RuntimeError(f"hipEventCreate failed ({rc})")
            arr[i] = e
        return arr

    def elapsed_ms(self, a, b):
        ms = ctypes.c_float()
        rc = self.lib.hipEventElapsedTime(ctypes.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return ms.value

    def destroy(self, arr):
        for e in arr:
            self.lib.hipEventDestroy(e)


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr per leg (never stdout: that carries the ONE JSON line): a long default run stays visibly alive."""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_sha():
    """sha256 over the HIP sources: profiles/traffic.json carries the value it was measured at, so a kernel change
    that keeps its label cannot silently report stale PMC bytes."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "video-frame-interpolation_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inl", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def granted_cpu_threads(cgroup_file="/sys/fs/cgroup/cpu.max"):
    """Every core this job is GRANTED: the affinity mask, a cgroup CPU quota if one is set, and the pool's stated share for a one-GPU
    job (16; EMAVFI_CPU_THREADS overrides the share).  The GPU box lists ALL 256 host cores in the affinity mask against a 16-CPU
    quota: taking them all oversubscribes the quota 16 x and the oracle crawls (round 6: a 7-minute silence in the first bench run)."""
    try:
        granted = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        granted = os.cpu_count() or 1
    try:
        quota, period = open(cgroup_file).read().split()
        if quota != "max":
            granted = min(granted, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(granted, int(os.environ.get("EMAVFI_CPU_THREADS", "16"))))


def cpu_baseline(sd, height, width, reps, dev=None):
    """The oracle (CPU restatement, kind "port") per SURVEY.md section 8(d): fp32, ONE full-size warm-up + `reps` (default 3) timed
    forwards, median - one 1280x720 pair (B = 1: the unit `value` counts), one 256x256 pair, and BASELINE configs[1]'s batch of 16
    256x256 pairs.  About 70 s of CPU work on the GPU box's 16-core share; the whole default bench stays within a few minutes.
    The full frame then goes through the HIP path in all three arithmetic modes for the accuracy fields."""
    from emavfi import synth
    from oracle import emavfi_oracle as oracle
    threads = granted_cpu_threads()
    torch.set_num_threads(threads)
    cpu_sd = {k: v.float().cpu() for k, v in sd.items()}
    # the split VERDICT r5 asks for: the three deformable convolutions are a restatement in Python-level tensor ops (torchvision's C++
    # kernel is not on the image), everything else is the ATen kernels the reference itself calls - time the restated op inside the
    # very forwards that are timed
    dcn_s = [0.0]
    plain_dcn = oracle.deform_conv2d

    def timed_dcn(*a, **k):
        t0 = time.perf_counter()
        out = plain_dcn(*a, **k)
        dcn_s[0] += time.perf_counter() - t0
        return out

    def timed(f1, f2, n):
        ref = oracle.forward(cpu_sd, f1, f2)  # full-size warm-up (thread pool, allocator, page faults of every intermediate)
        ts, ds = [], []
        oracle.deform_conv2d = timed_dcn
        try:
            for _ in range(n):
                dcn_s[0] = 0.0
                t0 = time.perf_counter()
                ref = oracle.forward(cpu_sd, f1, f2)
                ts.append(time.perf_counter() - t0)
                ds.append(dcn_s[0])
        finally:
            oracle.deform_conv2d = plain_dcn
        timed.dcn = ds
        return ts, ref

    s1, s2 = synth.synthetic_frames(7, 1, 256, 256, "natural")
    t256, _ = timed(s1, s2, reps)
    b1, b2 = synth.synthetic_frames(8, 16, 256, 256, "natural")
    t256b, _ = timed(b1, b2, reps)
    f1, f2 = synth.synthetic_frames(7, 1, height, width, "natural")
    tfull, ref = timed(f1, f2, reps)
    dfull = list(timed.dcn)
    accuracy = None
    if dev is not None:
        import math
        from emavfi import EMA_VFI
        accuracy = {"sample": f"the cpu_baseline frame (1 pair, {width}x{height}), HIP path vs CPU oracle"}
        for mode in ("fp32", "fp32x3", "bf16", "fp16"):
            m = EMA_VFI(compute_dtype=mode).to(dev).eval()
            m.load_state_dict(sd, strict=True)
            with torch.no_grad():
                got = m(f1.to(dev), f2.to(dev)).cpu()
            mse = (got.double() - ref.double()).pow(2).mean().item()
            accuracy[mode] = {"max_abs": float(f"{(got - ref).abs().max().item():.3e}"),
                              "psnr_db": round(99.0 if mse == 0 else 10.0 * math.log10(1.0 / mse), 2)}
    cpu_baseline.accuracy = accuracy
    med = statistics.median(tfull)
    m256b = statistics.median(t256b)
    return {"value": round(1.0 / med, 5), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"1 pair {width}x{height} (one full frame of the benchmarked size), fp32, oracle.forward: 1 full-size warm-up + "
                      f"{len(tfull)} timed, median {med:.2f} s (all: {', '.join(f'{t:.2f}' for t in tfull)}); "
                      "restated deform conv, not torchvision's C++ kernel",
            "also_256x256": {"value": round(1.0 / statistics.median(t256), 3), "unit": "frames/s",
                             "median_s": round(statistics.median(t256), 4), "timed": len(t256)},
            "also_config1_b16_256x256": {"value": round(16.0 / m256b, 3), "unit": "frames/s", "median_s": round(m256b, 3), "timed": len(t256b),
                                         "sample": "BASELINE configs[1]: one batch of 16 pairs 256x256"},
            "gflops": round(FLOP_PER_PX * height * width / med / 1e9, 1),
            # the split of the timed frame: the restated deform_conv2d x 3 (oracle/emavfi_oracle.py:124-145) against the ATen ops
            "dcn_restatement_share": round(statistics.median(d / t for d, t in zip(dfull, tfull)), 4),
            "dcn_restatement_s": round(statistics.median(dfull), 3),
            "gflops_aten_only": round((FLOP_PER_PX - 3 * DCN_FLOP_PER_PX) * height * width / max(1e-9, med - statistics.median(dfull)) / 1e9, 1),
            "gflops_dcn_restatement": round(3 * DCN_FLOP_PER_PX * height * width / max(1e-9, statistics.median(dfull)) / 1e9, 1),
            "cpu_model": cpu_model_name(), "torch": torch.__version__}


def warp_roofline(hip, B, H, W, steps=20):
    """The C-ABI warp kernel alone (NCHW fp32, 32 algorithmic bytes per pixel)."""
    from emavfi import lib
    g = torch.Generator().manual_seed(0)
    f2 = torch.randn(B, 3, H, W, generator=g).cuda()
    flow = (torch.randn(B, 2, H, W, generator=g) * 4).cuda()
    for _ in range(3):
        lib.warp(f2, flow)
    ev = hip.events(2 * steps)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty_like(f2)
    L = lib.load()
    for i in range(steps):
        hip.lib.hipEventRecord(ev[2 * i], stream)
        L.emavfi_warp(f2.data_ptr(), flow.data_ptr(), out.data_ptr(), B, 3, H, W, stream)
        hip.lib.hipEventRecord(ev[2 * i + 1], stream)
    torch.cuda.synchronize()
    ms = sorted(hip.elapsed_ms(ev[2 * i], ev[2 * i + 1]) for i in range(steps))[steps // 2]
    hip.destroy(ev)
    gbs = 32.0 * B * H * W / (ms * 1e-3) / 1e9
    return {"kernel": "warp_tiled_kernel<3> (emavfi_warp, NCHW fp32)", "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "median_us": round(ms * 1e3, 2),
            "algorithmic_bytes_per_px": 32, "pixels": B * H * W}


def stream_legs(sd, dev, B, H, W):
    """Side measurements through the streaming harness (uint8 HWC frames in host memory in, uint8 host frames out: PCIe
    included; reference inference.py:157-205).  Never part of `value`.
      also_stream_pcie: 64 pairs of 720p frames, interpolation_factor 1, batch 8 (the benchmarked size through the harness).
      also_1080p_4x:    BASELINE configs[4] on one GPU - 1920x1080, interpolation_factor 3: the HBM-resident forward rate at
                        B = 4, and emitted frames/s for 16 pairs in both harness modes (reference: one forward per pair, emitted
                        three times, inference.py:173-184; recursive: three distinct midpoints = three forwards per pair)."""
    import numpy as np
    from emavfi import EMA_VFI, FrameInterpolator, synth
    model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
    model.load_state_dict(sd, strict=True)

    def run(frames, warm, reps=3, **kw):
        """frames out, MEDIAN wall time of `reps` runs of the whole stream (and all of them)"""
        fi = FrameInterpolator(model, **kw)
        sum(1 for _ in fi.run(frames[:warm]))
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = sum(1 for _ in fi.run(frames))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        run.all = ts
        return n, statistics.median(ts)

    out = {}
    u8, _ = synth.synthetic_frames_u8(3, 1, H, W, "natural")
    frames = [np.roll(u8[0], 3 * i, axis=1) for i in range(65)]
    n, el = run(frames, 25, interpolation_factor=1, batch_pairs=B, copy_out=False)   # (warm-up long enough to take the half-size first batch once)
    out["also_stream_pcie"] = {"value": round(64 / el, 2), "unit": "interpolated frames/s", "emitted_frames_per_sec": round(n / el, 2),
                               "pairs": 64, "frames_out": n, "height": H, "width": W, "batch_pairs": B, "dtype": "bf16",
                               "runs_frames_per_sec": [round(64 / t, 1) for t in run.all],
                               "note": "uint8 host frames in and out through FrameInterpolator (reference loop order and bytes; hipMemcpyAsync "
                                       "pinned <-> HBM + device pre/post-processing kernels on two high-priority side streams, events to the "
                                       "compute stream); median of three runs of the 64-pair stream; PCIe-inclusive, never part of `value`"}
    # a long stream, where pipeline fill / drain no longer counts: the steady-state cost of the harness
    long_frames = [frames[i % 65] for i in range(257)]
    n2, el2 = run(long_frames, 25, reps=3, interpolation_factor=1, batch_pairs=B, copy_out=False)
    out["also_stream_pcie"]["long_stream_256_pairs"] = round(256 / el2, 2)
    a, b = synth.fast_frames(7, 4, 1080, 1920, device=dev)
    with torch.no_grad():
        for _ in range(2):
            model(a, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            model(a, b)
        torch.cuda.synchronize()
        fw = (time.perf_counter() - t0) / 8
    del a, b
    u8, _ = synth.synthetic_frames_u8(3, 1, 1080, 1920, "natural")
    frames = [np.roll(u8[0], 3 * i, axis=1) for i in range(17)]
    leg = {"forward_passes_per_sec": round(4 / fw, 2), "forward_ms_per_step": round(fw * 1e3, 3), "pairs_per_step": 4, "height": 1080,
           "width": 1920, "dtype": "bf16", "interpolation_factor": 3, "stream_pairs": 16, "batch_pairs": 4}
    for mode in ("reference", "recursive"):
        n, el = run(frames, 5, interpolation_factor=3, batch_pairs=4, mode=mode, reference_quirks=(mode == "reference"), copy_out=False)
        leg[f"emitted_frames_per_sec_{mode}"] = round(n / el, 2)
        leg[f"interpolated_frames_per_sec_{mode}"] = round(16 * 3 / el, 2)
    leg["note"] = ("reference = the reference loop's output (three identical predictions per pair, computed once: parity-mode de-dup); "
                   "recursive = three distinct midpoints per pair; host uint8 frames in and out (PCIe included)")
    out["also_1080p_4x"] = leg
    return out


def stream_leg_one_rank(model, rank, world, B, H, W, pairs_per_rank=64, reps=2):
    """N > 1: THIS rank's shard of one global stream through the streaming harness - uint8 host frames in pinned
    buffers in, uint8 host frames out, PCIe included (reference inference.py:146-205 feeds ONE process; SURVEY 8(e) / section 7 name host
    feeding as the 8-GPU risk).  The stream has world x pairs_per_rank pairs (BASELINE configs[3]: 64 pairs per 8 GPUs x 8; here 64 per
    rank so that pipeline fill / drain does not dominate); FrameInterpolator.run(frames, rank, world) takes this rank's contiguous
    segment (dist.shard_range).  All ranks stream at the same time (the caller puts a barrier in front).  Returns (pairs, frames out,
    best-of-`reps` seconds)."""
    import numpy as np
    from emavfi import FrameInterpolator, synth
    u8, _ = synth.synthetic_frames_u8(3, 1, H, W, "natural")
    base = [np.roll(u8[0], 3 * i, axis=1) for i in range(65)]
    frames = [base[i % 65] for i in range(world * pairs_per_rank + 1)]
    fi = FrameInterpolator(model, interpolation_factor=1, batch_pairs=B, copy_out=False)
    mine, _, _, _ = FrameInterpolator.segment(len(frames), 1, rank, world)
    sum(1 for _ in fi.run(frames[:25]))                       # warm-up: buffers, the half-size first batch
    best, n = None, 0
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(1 for _ in fi.run(frames, rank, world))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        best = el if best is None else min(best, el)
    return len(mine), n, best


def stream_leg_summary(rows, value):
    """rows: [rank, pairs, frames out, seconds] per rank (gathered) -> the keys of the bench line"""
    rows = sorted(rows)
    per_rank = [round(r[1] / r[3], 2) for r in rows]
    agg = sum(r[1] for r in rows) / max(r[3] for r in rows)
    return {"stream_pcie_per_rank": per_rank, "stream_pcie_aggregate": round(agg, 2), "unit": "interpolated frames/s",
            "pairs_per_rank": [int(r[1]) for r in rows], "frames_out_per_rank": [int(r[2]) for r in rows],
            "seconds_per_rank": [round(r[3], 4) for r in rows],
            "fraction_of_resident_value": None if not value else round(agg / value, 4),
            "note": "every rank streams its contiguous segment of ONE global stream through FrameInterpolator at the same time: uint8 "
                    "host frames in and out (pinned buffers, hipMemcpyAsync + device pre / post kernels on side streams); aggregate = all "
                    "pairs / the slowest rank's time; PCIe- and host-feeding-inclusive, never part of `value`"}


def fixup_census(raw, H, W):
    """Which deformable samples leave the pack kernel's staged window (deform_pack3.inl: 16 x 16-pixel tiles, a 23 x 23-pixel window
    = the tile + 1 tap + R = 2 px of offset reach + 1 bilinear neighbour), restated from the kernel's own test: the sample's
    top-left corner, floor(clamp(position, -2, size + 1)), must lie in window rows / columns [0, 21].  A wave (4 rows x 16 columns of
    the tile) whose ANY lane fails for tap k runs the fix-up loop for tap k (global gathers, 24 MFMAs + the tail).
    raw: [n, 27, H, W] = offset_conv's output (ema_vfi.py:56-58: offsets = raw[:, 0:9] ++ raw[:, 18:27]; (dy, dx) of tap k = channels (2k, 2k+1))."""
    n = raw.shape[0]
    off = torch.cat([raw[:, 0:9], raw[:, 18:27]], 1)
    yy = torch.arange(H, device=raw.device, dtype=torch.float32).view(1, H, 1)
    xx = torch.arange(W, device=raw.device, dtype=torch.float32).view(1, 1, W)
    ty0 = (torch.arange(H, device=raw.device) // 16 * 16 - 3).view(1, H, 1)
    tx0 = (torch.arange(W, device=raw.device) // 16 * 16 - 3).view(1, 1, W)
    out = torch.zeros(n, 9, H, W, dtype=torch.bool, device=raw.device)
    for k in range(9):
        i, j = divmod(k, 3)
        py = ((yy - 1 + i) + off[:, 2 * k]).clamp(-2.0, H + 1.0)
        px = ((xx - 1 + j) + off[:, 2 * k + 1]).clamp(-2.0, W + 1.0)
        ly = torch.floor(py).long() - ty0
        lx = torch.floor(px).long() - tx0
        out[:, k] = (ly < 0) | (ly > 21) | (lx < 0) | (lx > 21)
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    pad = torch.zeros(n, 9, Hp, Wp, dtype=torch.bool, device=raw.device)
    pad[:, :, :H, :W] = out
    groups = pad.view(n, 9, Hp // 4, 4, Wp // 16, 16).any(dim=5).any(dim=3)   # (wave, tap): 4 rows x 16 columns
    halves = pad.view(n, 9, Hp // 4, 2, 2, Wp // 16, 16).any(dim=6).any(dim=4)   # the wave's two fragment rows (2 image rows x 16 columns each)
    one_row = (halves.sum(dim=3) == 1).float().sum().item() / max(1.0, groups.float().sum().item())
    a = off.abs().flatten()
    q = a[torch.randint(0, a.numel(), (1 << 20,), device=a.device, generator=torch.Generator(device=a.device).manual_seed(0))].float()
    return {"samples_outside_window": round(out.float().mean().item(), 5), "wave_taps_in_fixup_loop": round(groups.float().mean().item(), 5),
            "fixup_wave_taps_with_one_fragment_row_parked": round(one_row, 4),
            "abs_offset_px_p50": round(q.quantile(0.5).item(), 3), "abs_offset_px_p99": round(q.quantile(0.99).item(), 3),
            "abs_offset_px_max": round(a.max().item(), 2)}


def pack_vs_offset_spread(hip, sd, dev, B, H, W, spreads=(0, 1, 2, 3, 4, 8), reps=5):
    """what the dominant kernel costs when the offsets leave its window.  The reference puts no bound on the offsets
    (ema_vfi.py:55-60) and ships no trained weights (.MISSING_LARGE_BLOBS), so the headline holds for the synthetic recipe's range;
    this leg runs ONE ModulatedDeformConvPack (emavfi_mdcn_profiled = the forward's attention_block() on the second block's real input,
    B x 720p, bf16) with that block's offset_conv rescaled so that the offsets span about +-s px: the data-dependent part (weights)
    to a standard deviation of s / 2, the bias to U(+-s / 2).  Mask channels and the DCN weights stay the recipe's."""
    from emavfi import EMA_VFI, lib
    model = EMA_VFI(compute_dtype="bf16").to(dev).eval()
    model.load_state_dict(sd, strict=True)
    f1, f2 = __import__("emavfi").synth.fast_frames(100, B, H, W, device=dev)
    with torch.no_grad():
        _, taps = model(f1, f2, return_taps=True)
    x = taps["fused_0"].clone()           # the second attention block's input [B, 67, H, W]
    del taps, f1, f2, model
    torch.cuda.empty_cache()
    ow = sd["attention_blocks.1.offset_conv.weight"].to(dev)
    ob = sd["attention_blocks.1.offset_conv.bias"].to(dev)
    dw = sd["attention_blocks.1.dcn_v2.weight"].to(dev)
    db = sd["attention_blocks.1.dcn_v2.bias"].to(dev)
    offch = torch.tensor(list(range(0, 9)) + list(range(18, 27)), device=dev)
    raw0 = lib.conv3x3(x[:1], ow, torch.zeros_like(ob), dtype="fp32")
    sigma0 = raw0[:, offch].std().item()
    px = float(B) * H * W
    flops = 2.0 * 9.0 * 67.0 * (67.0 + 27.0) * px
    ev = hip.events(2 * reps)
    # the forward hands its second pack IEEE f16 bit patterns and takes f16 back (the packs' hand-off: DESIGN.md 3.3 item 1); `rows` times
    # the stage in THAT form; `rows_bf16_input` in the form rounds 4-5 timed (bf16 in, converted to f16 on chip: +40-60 us at +-0 and one
    # conversion pass per fix-up round) for continuity with BENCH_r05.json
    f16_flags = lib.MDCN_IN_F16 | lib.MDCN_OUT_F16
    rows, rows_bf16 = [], []
    for s_px in spreads:
        ow_s, ob_s = ow.clone(), ob.clone()
        ow_s[offch] *= (0.5 * s_px / sigma0)
        ob_s[offch] *= 0.5 * s_px             # the recipe's offset bias is U(+-1)
        census = fixup_census(lib.conv3x3(x[:1], ow_s, ob_s, dtype="fp32"), H, W)
        for flags, dest in ((f16_flags, rows), (0, rows_bf16)):
            lib.mdcn(x, ow_s, ob_s, dw, db, dtype="bf16", flags=flags)     # warm-up
            for i in range(reps):
                off = ctypes.cast(ctypes.addressof(ev) + 2 * i * ctypes.sizeof(ctypes.c_void_p), ctypes.c_void_p)
                lib.mdcn(x, ow_s, ob_s, dw, db, dtype="bf16", flags=flags, _events=(off, 2))
            torch.cuda.synchronize()
            us = sorted(hip.elapsed_ms(ev[2 * i], ev[2 * i + 1]) for i in range(reps))[reps // 2] * 1e3
            row = {"spread_px": s_px, "pack_us": round(us, 1), "frac": round(flops / (us * 1e-6) / 1e12 / PEAK["bf16"], 4)}
            if flags:
                kc = lib.mdcn_census(B, 67, H, W, dtype="bf16", flags=flags, device=dev)[0]   # what the kernel counted in the last timed launch (all B samples)
                row.update(census)
                if kc:
                    row["kernel_census"] = {"fixup_share": round(kc["fixup_share"], 5), "samples_outside_share": round(kc["samples_outside_share"], 6),
                                            "abs_offset_px_max": round(kc["abs_offset_px_max"], 2)}
            dest.append(row)
    hip.destroy(ev)
    return {"kernel": "deform_pack3<bf16,fused> via emavfi_mdcn_profiled (attention_blocks.1 on its real input, f16 in / out as inside the forward)",
            "pairs": B, "height": H, "width": W,
            "window": "16x16 tile, R = 2 px of offset reach beyond the tap (23 x 23 pixels staged)", "rows": rows, "rows_bf16_input": rows_bf16,
            "note": "offsets = s/2 x unit-variance data term + U(+-s/2) bias; census (one sample, fp32 offset_conv) restates the kernel's in-window "
                    "test, kernel_census is what the kernel itself counted over all samples; never part of `value`"}


def board_under_load(model, a1, a2, seconds=2.0):
    """Package power and clocks (rocm-smi, polled from a thread) while the measured forward runs back to back for about two
    seconds AFTER the timed region: the roofline fractions are taken against a 2.4 GHz peak the board does not sustain on this
    workload.  Never part of `value`; {} when rocm-smi is unavailable."""
    import subprocess
    import threading

    def smi():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            c = json.loads(out)
            c = c[sorted(c)[0]]

            def num(*keys):
                for k, v in c.items():
                    if all(x in k.lower() for x in keys):
                        try:
                            return float(str(v).replace("Mhz", "").strip("() "))
                        except ValueError:
                            pass
                return None
            return {"power_w": num("socket", "power") or num("average", "power"), "cap_w": num("max", "power"), "sclk_mhz": num("sclk", "speed"),
                    "mclk_mhz": num("mclk", "speed")}
        except Exception:  # noqa: BLE001 - a side measurement must never fail the bench
            return None

    samples, stop = [], []

    def poll():
        while not stop:
            r = smi()
            if r:
                samples.append((time.perf_counter(), r))
            time.sleep(0.05)

    th = threading.Thread(target=poll, daemon=True)
    t0 = time.perf_counter()
    th.start()
    with torch.no_grad():
        while time.perf_counter() - t0 < seconds:
            for _ in range(10):
                model(a1, a2)
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    stop.append(1)
    th.join(timeout=15)
    load = [r for (t, r) in samples if t0 + 0.7 < t < t1]
    out = {}
    for key in ("power_w", "sclk_mhz", "mclk_mhz", "cap_w"):
        v = sorted(r[key] for r in load if r.get(key) is not None)
        if v:
            out[key] = v[len(v) // 2]
    if out:
        out["samples"] = len(load)
        out["note"] = "medians of rocm-smi readings while the forward runs back to back (after the timed region); peak sclk 2400 MHz"
    return out


def per_kernel_table(hip, ev, launches, steps, dtype):
    """Aggregate the HIP events recorded around every launch of `steps` forwards: per-kernel table (sorted by device time) and the
    roofline object of the dominant kernel - `achieved` = ALGORITHMIC flops (bytes) per launch / average launch duration."""
    nl = len(launches)
    agg = {}
    for i in range(steps):
        for j, (name, fl, by) in enumerate(launches):
            k = name.split(" ")[0]
            a = agg.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
            base = 2 * (nl * i + j)
            a["ms"] += hip.elapsed_ms(ev[base], ev[base + 1])
            a["flops"] += fl
            a["bytes"] += by
            a["n"] += 1
    total_ms = sum(a["ms"] for a in agg.values())
    table = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        table.append({"kernel": k, "launches_per_step": a["n"] // steps, "avg_us": round(a["ms"] / a["n"] * 1e3, 1),
                      "share": round(a["ms"] / total_ms, 4),
                      "tflops": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 2) if a["flops"] else 0.0,
                      "gbs": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)})
    dom = table[0]
    a = agg[dom["kernel"]]
    is_mfma = a["flops"] / max(a["bytes"], 1.0) > PEAK[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    if is_mfma:
        roof = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"], "peak": PEAK[dtype], "unit": "TFLOP/s",
                "frac": round(dom["tflops"] / PEAK[dtype], 4), "avg_launch_us": dom["avg_us"], "share_of_device_time": dom["share"],
                "algorithmic_flops_per_launch": a["flops"] / a["n"]}
    else:
        roof = {"kernel": dom["kernel"], "bound": "hbm", "achieved": dom["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(dom["gbs"] / PEAK_HBM_GBS, 4), "avg_launch_us": dom["avg_us"], "share_of_device_time": dom["share"],
                "algorithmic_bytes_per_launch": a["bytes"] / a["n"]}
    return table, roof, agg, total_ms


def warp_in_forward(table, agg, dtype, px):
    """The warp kernel the forward actually runs (row W inside the timed region).  Two accountings, both printed: the ALGORITHMIC
    bytes (8 B flow + 12 B frame2 read + 3 channels of the storage type written: 26 B/px in the 16-bit modes, 32 B/px in fp32) and the
    bytes the layout forces it to move (16-bit modes: the three channels leave as a four-channel tail record, 8 + 12 + 8 = 28 B/px -
    round 6; 36 with the 16-byte record of rounds 3-5; fp32: the warp fills channels 64..79 of the 80-channel fusion pixels,
    8 + 12 + 64 = 84 B/px)."""
    wk = [t for t in table if t["kernel"].startswith("warp_fused")]
    if not wk:
        return None
    wa = agg[wk[0]["kernel"]]
    moved = (28.0 if dtype != "fp32" else 84.0) * px
    return {"kernel": wk[0]["kernel"], "bound": "hbm", "achieved": wk[0]["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(wk[0]["gbs"] / PEAK_HBM_GBS, 4), "avg_launch_us": wk[0]["avg_us"],
            "algorithmic_bytes_per_launch": wa["bytes"] / wa["n"], "layout_bytes_per_launch": moved,
            "frac_on_layout_bytes": round(moved / (wk[0]["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}


def add_sustained_clock(obj, sclk_mhz, peak_mhz=2400.0):
    """Every MFMA-bound `frac` is quoted against the dense peak at 2.4 GHz; the board sustains less under this workload (it runs at its
    1 400 W cap: board_under_load).  frac_at_sustained_clock = frac x 2400 / sclk - the fraction of the peak the chip can actually
    reach at the clock it holds, i.e. how much a better kernel could still gain.  HBM-bound objects are left alone (the memory clock
    does not move)."""
    if isinstance(obj, dict):
        if obj.get("bound") == "mfma" and isinstance(obj.get("frac"), float):
            obj["frac_at_sustained_clock"] = round(obj["frac"] * peak_mhz / sclk_mhz, 4)
            # against what this board's matrix pipe was MEASURED to deliver at its 1 400 W cap with convolution-like operands (a bare
            # v_mfma_f32_32x32x16_bf16 loop: 1 855 TFLOP/s at 1.90 GHz; with constant operands the same loop reaches 94-97 % of the
            # 2.5 PFLOP/s peak: the MFMA's energy is in its data - tools/microbench/mfma_duty_power.hip, profiles/r05_mfma_duty_power.txt)
            if obj.get("unit") == "TFLOP/s" and obj.get("peak") == PEAK["bf16"] and isinstance(obj.get("achieved"), float):
                obj["frac_of_measured_bf16_matrix_ceiling_1855_tflops"] = round(obj["achieved"] / 1855.0, 4)
        if "frac_of_mfma_peak" in obj:
            obj["frac_of_mfma_peak_at_sustained_clock"] = round(obj["frac_of_mfma_peak"] * peak_mhz / sclk_mhz, 4)
        for v in obj.values():
            add_sustained_clock(v, sclk_mhz, peak_mhz)
    elif isinstance(obj, list):
        for v in obj:
            add_sustained_clock(v, sclk_mhz, peak_mhz)


def stored_traffic(kernel, fname="traffic.json"):
    """PMC HBM bytes per launch of `kernel` from profiles/<fname> (measured with tools/profile_round.sh on the builder's box,
    stamped with the sha of the kernel sources it was measured on: refused when the sources have changed since)."""
    tpath = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(tpath):
        return None, f"profiles/{fname} missing"
    tj = json.load(open(tpath))
    if tj.get("_kernel_source_sha") != kernel_source_sha():
        return None, f"profiles/{fname} was measured on other kernel sources: refused as stale"
    return tj.get(kernel), f"PMC FETCH_SIZE x2 + WRITE_SIZE, {tj.get('_measured', '')}"


def profiled_mode(hip, sd, dev, dtype, B, H, W, steps):
    """One arithmetic mode timed with per-launch events (the side legs: exact fp32 at the benchmarked size and at BASELINE
    configs[1]): frames/s, the per-kernel table and the roofline object of ITS dominant kernel."""
    from emavfi import EMA_VFI, lib, synth
    alt = EMA_VFI(compute_dtype=dtype).to(dev).eval()
    alt.load_state_dict(sd, strict=True)
    a1, a2 = synth.fast_frames(300, B, H, W, device=dev)
    launches = lib.forward_launches(3, 64, 3, B, H, W, dtype)
    nl = len(launches)
    ev = hip.events(2 * nl * steps)
    with torch.no_grad():
        for _ in range(2):
            alt(a1, a2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            off = ctypes.cast(ctypes.addressof(ev) + 2 * nl * i * ctypes.sizeof(ctypes.c_void_p), ctypes.c_void_p)
            alt(a1, a2, _events=(off, 2 * nl))
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
    table, roof, agg, total_ms = per_kernel_table(hip, ev, launches, steps, dtype)
    hip.destroy(ev)
    roof["traffic"], roof["traffic_source"] = stored_traffic(roof["kernel"], f"traffic_{dtype}_{B}x{H}x{W}.json")
    res = {"value": round(B * steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
           "pairs_per_step": B, "height": H, "width": W, "dtype": dtype, "roofline": roof, "kernels": table[:12],
           "device_ms_per_step_sum_of_kernels": round(total_ms / steps, 3)}
    warp = warp_in_forward(table, agg, dtype, B * H * W)
    if warp:
        res["roofline_warp_in_forward"] = warp
    return res


def rehearse(args, rank, world):
    """--rehearse: the N > 1 plumbing without a GPU (CPU test, gloo): rendezvous, broadcast of a blob-sized buffer from rank 0,
    barrier + MAX-over-ranks timing around K stand-in steps, the rank census, ONE JSON line on rank 0.  `value` is null."""
    from emavfi import dist as vdist
    backend = os.environ.get("EMAVFI_DIST_BACKEND", "gloo")
    vdist.init(backend, None)
    g = torch.Generator().manual_seed(1)
    blob = torch.randint(0, 256, (3 << 20,), dtype=torch.uint8, generator=g) if rank == 0 else torch.zeros(3 << 20, dtype=torch.uint8)
    vdist.broadcast_packed(blob, 0)
    want = torch.randint(0, 256, (3 << 20,), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    assert torch.equal(blob, want), "broadcast did not deliver rank 0's bytes"
    vdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    vdist.barrier()
    mine = time.perf_counter() - t0
    elapsed = vdist.max_over_ranks(mine, None)
    seen = vdist.all_gather_floats([float(rank), mine / args.steps * 1e3], None)
    # the per-rank stream leg's plumbing (segment sharding + gather + summary) with a stand-in for the harness: every rank "streams"
    # ITS segment of a (world x 64 + 1)-frame stream - the host logic FrameInterpolator.segment is the real one
    from emavfi import FrameInterpolator
    pairs_mine, lo, hi, tail = FrameInterpolator.segment(world * 64 + 1, 1, rank, world)
    vdist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.0005 * len(pairs_mine) * (1 + 0.1 * rank))
    rows = vdist.all_gather_floats([float(rank), float(len(pairs_mine)), float(2 * len(pairs_mine) + (1 if tail else 0)), time.perf_counter() - t0], None)
    if rank == 0:
        print(json.dumps({"metric": "interpolated_frames_per_sec_720p_2x", "value": None, "unit": "frames/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "rehearsal": True, "backend": backend,
                          "ranks_seen": sorted(int(r[0]) for r in seen), "ms_per_step_per_rank": [round(r[1], 3) for r in seen],
                          "also_stream_pcie_per_rank": stream_leg_summary(rows, None)}), flush=True)
    if world > 1:
        vdist.barrier()
        torch.distributed.destroy_process_group()


def main():
    args = ARGS if ARGS is not None else parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse:
        return rehearse(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # one process per GPU; the modulo only matters for rehearsals with more ranks than GPUs
    # (EMAVFI_DIST_BACKEND=gloo, e.g. 2 ranks sharing the single GPU of a test box - RCCL itself
    # refuses two ranks on one device)
    local_dev = local % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    from emavfi import EMA_VFI, lib, synth, dist as vdist
    backend = os.environ.get("EMAVFI_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
    vdist.init(backend, dev)

    B, H, W = args.batch, args.height, args.width
    model = EMA_VFI(compute_dtype=args.dtype).to(dev).eval()
    sd = synth.synthetic_state_dict(seed=0)
    if rank == 0:
        model.load_state_dict(sd, strict=True)
    # the path's one collective: RCCL broadcast of the packed weights over xGMI (no-op at N=1); the receiving ranks verify the
    # blob's header and checksum (emavfi_packed_check) when they install it
    vdist.share_model_weights(model, args.dtype, dev)

    f1, f2 = synth.fast_frames(100 + rank, B, H, W, device=dev)
    launches = lib.forward_launches(3, 64, 3, B, H, W, args.dtype)
    nl = len(launches)
    hip = Hip()
    use_events = not args.no_events
    ev = hip.events(2 * nl * args.steps) if use_events else None

    def step(i):
        with torch.no_grad():
            if use_events:
                off = ctypes.cast(ctypes.addressof(ev) + 2 * nl * i * ctypes.sizeof(ctypes.c_void_p), ctypes.c_void_p)
                return model(f1, f2, _events=(off, 2 * nl))
            return model(f1, f2)

    def fence():
        vdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        with torch.no_grad():
            model(f1, f2)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0       # this rank's own K steps (reported per rank)
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = vdist.max_over_ranks(elapsed, dev)
    # the rank census over the benchmark's own backend (RCCL when it is "nccl"): which ranks took part, each one's step time, and whether
    # its last frame is finite - gathered, THEN raised on every rank: a rank-local assert between two collectives would leave the others
    # blocked in the all-gather until the RCCL timeout (a mismatched broadcast blob yields exactly that: the guard's all-NaN frame)
    finite = bool(torch.isfinite(out).all().item())
    census = vdist.all_gather_floats([float(rank), mine / args.steps * 1e3, float(local_dev), 1.0 if finite else 0.0], dev)
    bad = sorted(int(r[0]) for r in census if r[3] != 1.0)
    if bad:
        raise SystemExit(f"bench.py: non-finite output frame on rank(s) {bad} (a foreign / corrupted packed blob yields an all-NaN frame)")

    # N > 1: the PCIe-inclusive leg per rank (at N = 1 the extras below carry also_stream_pcie); gathered INSIDE a collective like the
    # rank census, so that no rank-local failure leaves the others blocked
    stream_rows = None
    if world > 1 and not args.no_extras and args.dtype == "bf16":
        fence()
        try:
            pairs, n_out, secs = stream_leg_one_rank(model, rank, world, B, H, W)
        except Exception as e:   # noqa: BLE001 - reported through the gather, raised on every rank
            print(f"bench.py: rank {rank}: stream leg failed: {e}", file=sys.stderr, flush=True)
            pairs, n_out, secs = -1, -1, 1.0
        stream_rows = vdist.all_gather_floats([float(rank), float(pairs), float(n_out), float(secs)], dev)
        if any(r[1] < 0 for r in stream_rows):
            raise SystemExit(f"bench.py: the stream leg failed on rank(s) {sorted(int(r[0]) for r in stream_rows if r[1] < 0)}")

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        frames = world * B * args.steps
        value = frames / elapsed
        res = {"metric": "interpolated_frames_per_sec_720p_2x", "value": round(value, 2), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp32": "f32", "fp16": "f16"}.get(args.dtype, args.dtype),
               "data": "synthetic",
               "config": {"workload": f"BASELINE configs[2]: batch {B} x {W}x{H} frame pairs per GPU, "
                                      f"{args.dtype + ' convs + fp32 warp' if args.dtype != 'fp32' else 'fp32 MFMA convs + fp32 warp'}, "
                                      "EMA_VFI(3,64,3), synthetic non-degenerate weights",
                          "pairs_per_gpu": B, "height": H, "width": W, "parallelism": f"replica-dp{world}",
                          "collectives": f"one {backend} broadcast of packed weights before timing" if world > 1 else "none"},
               "backend": backend if world > 1 else None,
               "ranks_seen": sorted(int(r[0]) for r in census),
               "ms_per_step_per_rank": [round(r[1], 3) for r in sorted(census)],
               "device_per_rank": [int(r[2]) for r in sorted(census)],
               "frames_per_sec_per_gpu": round(value / world, 2),
               "forward_passes_per_sec": round(value, 2),
               "whole_forward": {"tflops_algorithmic": round(FLOP_PER_PX * B * H * W / (ms_step * 1e-3) / 1e12, 2),
                                 "frac_of_mfma_peak": round(FLOP_PER_PX * B * H * W / (ms_step * 1e-3) / 1e12 / PEAK[args.dtype], 4),
                                 "flop_per_px": FLOP_PER_PX}}
        if use_events:
            table, roof, agg, total_ms = per_kernel_table(hip, ev, launches, args.steps, args.dtype)
            traffic, traffic_note = stored_traffic(roof["kernel"])
            res["roofline"] = dict(roof, traffic=traffic, traffic_source=traffic_note)
            # where the headline sits on the pack's own offset curve: what the three one-launch packs of the LAST
            # TIMED forward counted while they ran - share of (wave, tap) groups that took the fix-up pass, samples outside the staged
            # window, largest |offset| (also_pack_vs_offset_spread below prices the rows of that curve)
            try:
                rows = model.pack_census()
                if any(r is not None for r in rows):
                    res["roofline"]["fixup_share"] = [None if r is None else round(r["fixup_share"], 5) for r in rows]
                    res["roofline"]["offset_census"] = {
                        "per_pack": [None if r is None else {"fixup_wave_taps": r["fixup_wave_taps"], "wave_taps": r["wave_taps"],
                                                             "samples_outside_share": round(r["samples_outside_share"], 6),
                                                             "abs_offset_px_max": round(r["abs_offset_px_max"], 2)} for r in rows],
                        "window": "16x16 tile + 1 tap + R = 2 px + 1 bilinear neighbour (23 x 23 px staged)",
                        "source": "device counters of deform_pack3_kernel in the last timed forward (emavfi_forward_census)"}
            except RuntimeError as e:   # a side read-out must never fail the bench
                res["roofline"]["offset_census"] = {"error": str(e)[:200]}
            res["kernels"] = table
            res["device_ms_per_step_sum_of_kernels"] = round(total_ms / args.steps, 3)
            warp = warp_in_forward(table, agg, args.dtype, B * H * W)
            if warp:
                res["roofline_warp_in_forward"] = warp
        if stream_rows is not None:
            res["also_stream_pcie_per_rank"] = stream_leg_summary(stream_rows, value)
        if world == 1 and not args.no_extras:
            def timed_alt(dtype, b, h, w, steps):
                alt = EMA_VFI(compute_dtype=dtype).to(dev).eval()
                alt.load_state_dict(sd, strict=True)
                a1, a2 = synth.fast_frames(300, b, h, w, device=dev)
                with torch.no_grad():
                    for _ in range(2):
                        alt(a1, a2)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(steps):
                        alt(a1, a2)
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t1
                return {"value": round(b * steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
                        "pairs_per_step": b, "height": h, "width": w, "dtype": dtype}

            progress("timed region done; side legs: warp, board power, fp16, amp16, fp32, configs[1], stream, spread, CPU oracle")
            res["roofline_warp"] = warp_roofline(hip, B, H, W)
            res["board_under_load"] = board_under_load(model, f1, f2)
            # reported beside, never part of `value`
            if args.dtype != "fp16":  # the fast half-precision mode (every contraction in fp16)
                res["also_fp16_fast"] = timed_alt("fp16", B, H, W, args.steps)
            # what the reference's torch.cuda.amp.autocast() computes on a GPU (inference.py:159), op policy restated:
            # fp16 convolutions, fp32 grid_sample and fp32 deform_conv2d on an fp32 fusion tensor (EMAVFI_AMP16)
            progress("amp16 leg")
            res["also_amp16_autocast_policy"] = timed_alt("amp16", B, H, W, max(3, args.steps // 2))
            # round 6: fp32-ACCURATE contractions on the f16 matrix pipe (three-term hi / lo split, exact fp32 DCN and warp);
            # passes the fp32 mode's parity gates on every reference-run fixture (tests/test_gpu_fp32x3.py) but is NOT the parity mode:
            # `fp32` in roofline_fp32 / accuracy_vs_cpu_oracle stays the exact one
            progress("fp32x3 leg")
            res["also_fp32_split16"] = timed_alt("fp32x3", B, H, W, max(3, args.steps // 4))
            if args.dtype != "fp32":  # the parity mode (exact fp32 MFMA; the only mode north_star's 1e-3 bound applies to): with ITS roofline
                progress("fp32 leg")
                res["also_fp32_exact"] = profiled_mode(hip, sd, dev, "fp32", B, H, W, max(3, args.steps // 4))
                res["roofline_fp32"] = dict(res["also_fp32_exact"]["roofline"], workload=f"B={B} x {W}x{H}, exact fp32")
                if "roofline_warp_in_forward" in res["also_fp32_exact"]:   # the mode north_star's 1e-3 clause is about
                    res["roofline_warp_in_forward_fp32"] = dict(res["also_fp32_exact"]["roofline_warp_in_forward"], workload=f"B={B} x {W}x{H}, exact fp32")
            # BASELINE.json configs[1]: batch 16 of 256x256 pairs, fp32 (with its roofline) and bf16
            progress("configs[1] legs")
            c1 = profiled_mode(hip, sd, dev, "fp32", 16, 256, 256, args.steps)
            res["config1_256"] = {"fp32": c1, "bf16": timed_alt("bf16", 16, 256, 256, 4 * args.steps)}
            res["roofline_fp32_config1"] = dict(c1["roofline"], workload="BASELINE configs[1]: B=16 x 256x256, exact fp32")
            if args.dtype == "bf16" and (H, W) == (720, 1280):   # the harness legs (PCIe-inclusive; BASELINE configs[4] size)
                progress("stream legs (PCIe-inclusive)")
                res.update(stream_legs(sd, dev, B, H, W))
                res["also_stream_pcie"]["fraction_of_resident_value"] = round(res["also_stream_pcie"]["value"] / value, 4)
                progress("pack vs offset spread")
                res["also_pack_vs_offset_spread"] = pack_vs_offset_spread(hip, sd, dev, B, H, W)
            sclk = (res.get("board_under_load") or {}).get("sclk_mhz")
            if sclk:
                add_sustained_clock(res, float(sclk))
            if args.cpu_reps > 0:
                progress("CPU oracle baseline (1 warm-up + timed forwards at three sizes)")
                res["cpu_baseline"] = cpu_baseline(sd, H, W, args.cpu_reps, dev)
                progress("done")
                res["accuracy_vs_cpu_oracle"] = cpu_baseline.accuracy
        print(json.dumps(res), flush=True)
    if ev is not None:
        hip.destroy(ev)
    if world > 1:
        vdist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
