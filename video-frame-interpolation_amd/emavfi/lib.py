"""ctypes binding of libemavfi.so (C-ABI: include/emavfi.h).

There is deliberately no fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes
import os
import collections
import threading
from ctypes import c_int, c_size_t, c_void_p, c_char_p, POINTER

F32, BF16, F16, AMP16, F32X3 = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_TANH01 = 0, 1, 2
MDCN_IN_F16, MDCN_OUT_F16, MDCN_SPLIT_TAIL = 1, 2, 4
# emavfi_debug_switches bits (include/emavfi.h)
SW_NO_CONV_FIRST, SW_NO_FIRSTRING, SW_NO_HEAD, SW_NO_TAILFUSE, SW_NO_CONV_LIGHT, SW_NO_PERSISTENT_CONV, SW_NO_RING2, SW_NO_POOLFUSE, SW_RING_ONE_WG, SW_NO_RING_CHUNK = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512
DTYPES = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16,
          "fp16": F16, "f16": F16, "float16": F16, "half": F16,
          # the reference's forward under torch.cuda.amp.autocast(), op policy restated (include/emavfi.h, EMAVFI_AMP16)
          "amp16": AMP16, "autocast": AMP16, "autocast16": AMP16,
          # fp32-accurate three-term f16 split on the 16-bit matrix pipe, exact fp32 DCN (include/emavfi.h, EMAVFI_F32X3)
          "fp32x3": F32X3, "f32x3": F32X3}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMAVFI_LIB", os.path.join(_HERE, "lib", "libemavfi.so"))

# every symbol include/emavfi.h declares: (restype, argtypes)
_PROTOTYPES = {
    "emavfi_version": (c_int, []),
    "emavfi_last_error": (c_char_p, []),
    "emavfi_supported": (c_int, [c_int] * 4),
    "emavfi_param_count": (c_int, [c_int]),
    "emavfi_packed_bytes": (c_size_t, [c_int] * 4),
    "emavfi_pack_weights": (c_int, [c_int] * 3 + [POINTER(c_void_p), c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "emavfi_workspace_bytes": (c_size_t, [c_int] * 7),
    "emavfi_layout_tag": (c_int, []),
    "emavfi_packed_check": (c_int, [c_int] * 4 + [c_void_p, c_size_t]),
    "emavfi_forward": (c_int, [c_int] * 3 + [c_void_p, c_size_t] + [c_void_p] * 4 + [c_size_t] + [c_int] * 4 + [POINTER(c_void_p), c_void_p]),
    "emavfi_forward_launches": (c_int, [c_int] * 7 + [c_char_p, c_size_t, POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_int]),
    "emavfi_forward_profiled": (c_int, [c_int] * 3 + [c_void_p, c_size_t] + [c_void_p] * 4 + [c_size_t] + [c_int] * 4 + [POINTER(c_void_p), c_int, c_void_p]),
    "emavfi_forward_staged": (c_int, [c_int] * 3 + [c_void_p, c_size_t] + [c_void_p] * 4 + [c_size_t] + [c_int] * 4 + [POINTER(c_void_p), POINTER(c_void_p), c_int, c_void_p]),
    "emavfi_warp": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "emavfi_preprocess_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [POINTER(ctypes.c_float), POINTER(ctypes.c_float), c_void_p]),
    "emavfi_postprocess_u8": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [POINTER(ctypes.c_double), POINTER(ctypes.c_double), c_int, c_void_p]),
    "emavfi_conv3x3_workspace_bytes": (c_size_t, [c_int] * 7),
    "emavfi_conv3x3": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "emavfi_deform_conv2d_workspace_bytes": (c_size_t, [c_int] * 6),
    "emavfi_deform_conv2d": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "emavfi_mdcn_workspace_bytes": (c_size_t, [c_int] * 6),
    "emavfi_mdcn": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "emavfi_mdcn_profiled": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t, POINTER(c_void_p), c_int, c_void_p]),
    "emavfi_forward_census": (c_int, [c_int] * 7 + [c_void_p, c_size_t, c_void_p, c_void_p]),
    "emavfi_mdcn_census": (c_int, [c_int] * 6 + [c_void_p, c_size_t, c_void_p, c_void_p]),
    "emavfi_context_workspace_bytes": (c_size_t, [c_int] * 5),
    "emavfi_context": (c_int, [c_void_p, POINTER(c_void_p), c_void_p] + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "emavfi_reconstruct_workspace_bytes": (c_size_t, [c_int] * 5),
    "emavfi_reconstruct": (c_int, [c_void_p, POINTER(c_void_p), c_void_p] + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "emavfi_debug_switches": (c_int, [c_int, c_int]),
}
SYMBOLS = tuple(_PROTOTYPES)

_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Load libemavfi.so once; raise loudly if it is not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"libemavfi.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                        "g.build()'` or `make -C video-frame-interpolation_amd/csrc` (hipcc, gfx950). "
                        "There is no CPU or PyTorch fallback for this path.")
                lib = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in _PROTOTYPES.items():
                    fn = getattr(lib, name)
                    fn.restype, fn.argtypes = res, args
                _lib = lib
    return _lib


_fingerprint = None


def fingerprint() -> str:
    """sha256 of the loaded libemavfi.so: part of the key of the on-disk packed-weight cache (the packed layout is a
    property of the build, not of emavfi_version())."""
    global _fingerprint
    if _fingerprint is None:
        import hashlib
        load()
        with open(LIB_PATH, "rb") as f:
            _fingerprint = hashlib.sha256(f.read()).hexdigest()
    return _fingerprint


def layout_switches() -> str:
    """The process-wide switches the PACKED LAYOUT depends on, as the LIBRARY latched them at first use (emavfi_layout_tag(),
    include/emavfi.h) - not os.environ, which may have changed since (two sources of truth).  Part of the in-memory and
    on-disk cache keys of the packed weights; the blob header carries the same tag and emavfi_packed_check compares it."""
    return f"layout_tag={load().emavfi_layout_tag()}"


def packed_check(in_channels, mid_channels, num_blocks, dt, blob) -> None:
    """emavfi_packed_check on a uint8 tensor (device or host): header fields and payload checksum; RuntimeError names the mismatch.
    Synchronises - call when a blob ARRIVES (cache file, broadcast, load_packed_weights), never per frame."""
    if not blob.is_contiguous():
        raise ValueError("packed_check: contiguous uint8 tensor expected")
    check(load().emavfi_packed_check(in_channels, mid_channels, num_blocks, dt, blob.data_ptr(), blob.numel()), "emavfi_packed_check")


def debug_switches(and_mask=-1, or_mask=0) -> int:
    """Test hook (include/emavfi.h, emavfi_debug_switches): returns the previous switch word."""
    return load().emavfi_debug_switches(and_mask, or_mask)


def last_error() -> str:
    return (load().emavfi_last_error() or b"").decode("utf-8", "replace")


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {last_error()}")


def dtype_code(name) -> int:
    if isinstance(name, int):
        return name
    try:
        return DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"compute dtype must be one of {sorted(DTYPES)}, got {name!r}") from None


def _stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("emavfi: tensors must live on a ROCm device (MI355X); "
                               "this build has no CPU path - use the reference or the oracle for CPU runs")


def _f32c(t):
    import torch
    return t.detach().to(torch.float32).contiguous()


# Scratch memory (caller-owned, as the C-ABI requires): one grow-only buffer per (device, stream).
#   * two streams (or two threads on different streams) never share scratch - forwards on different streams may
#     run concurrently;
#   * a buffer is allocated while its stream is current, so the caching allocator's stream-ordered reuse makes
#     dropping the old buffer on growth safe: it was only ever used on that stream;
#   * a buffer that was handed out while its stream was being captured into a hipGraph has its address baked into the
#     graph: it is never freed on growth but parked in _ws_graph_held (until release_workspaces()), so replaying the
#     graph after a later, larger forward still writes to memory nobody else owns;
#   * the cache holds at most EMAVFI_WS_CACHE_MAX (default 8) buffers: a caller that runs forwards on transient streams would
#     otherwise pin one workspace (GBs at B = 8 x 720p) per stream handle it ever used.  The least recently used buffer that no
#     captured graph may point at is dropped first (the caching allocator keeps a freed block on its allocation stream's pool,
#     so work still queued on that stream is not disturbed).  Captured buffers are only released by release_workspaces().
_ws_cache = collections.OrderedDict()   # (device index, stream handle) -> uint8 tensor, least recently used first
_ws_captured = set()  # keys whose current buffer a captured graph may point at
_ws_graph_held = []
_ws_lock = threading.Lock()


def workspace(nbytes: int, device):
    import torch
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, torch.cuda.current_stream(device).cuda_stream)
    capturing = torch.cuda.is_current_stream_capturing()
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None and key in _ws_captured:
                _ws_graph_held.append(buf)
                _ws_captured.discard(key)
            with torch.cuda.device(index):
                buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=torch.device("cuda", index))
            _ws_cache[key] = buf
            cap = max(1, int(os.environ.get("EMAVFI_WS_CACHE_MAX", "8")))
            for old in [k for k in _ws_cache if k != key and k not in _ws_captured][:max(0, len(_ws_cache) - cap)]:
                del _ws_cache[old]
        _ws_cache.move_to_end(key)
        if capturing:
            _ws_captured.add(key)
    return buf


def release_workspaces():
    """Drop every cached workspace, including those kept alive for captured graphs (destroy the graphs first)."""
    with _ws_lock:
        _ws_cache.clear()
        _ws_captured.clear()
        del _ws_graph_held[:]


# ---------------------------------------------------------------- HIP events / side streams for the pipelined forward (model.py)
class _Hip:
    """hipEvent* / hipStreamWaitEvent of the HIP runtime torch has already loaded into this process (plain handles: what the C-ABI's
    stage events are).  Events are created once per (device, purpose) and re-recorded: a hipStreamWaitEvent captures the record
    that precedes it in host call order."""

    def __init__(self):
        path = None
        for line in open("/proc/self/maps"):
            if "libamdhip64" in line:
                path = line.split()[-1]
                break
        self.rt = ctypes.CDLL(path or "libamdhip64.so")
        self.rt.hipEventCreateWithFlags.argtypes = [POINTER(c_void_p), ctypes.c_uint]
        self.rt.hipEventRecord.argtypes = [c_void_p, c_void_p]
        self.rt.hipStreamWaitEvent.argtypes = [c_void_p, c_void_p, ctypes.c_uint]
        self._events = {}

    def event(self, key):
        """One event per key, created on first use and kept for the life of the process (keys are (device, caller stream, purpose): a
        handful per stream that ever ran a pipelined forward)."""
        e = self._events.get(key)
        if e is None:
            with _lock:
                e = self._events.get(key)
                if e is None:
                    h = c_void_p()
                    rc = self.rt.hipEventCreateWithFlags(ctypes.byref(h), 2)   # hipEventDisableTiming
                    if rc != 0:
                        raise RuntimeError(f"hipEventCreateWithFlags failed ({rc})")
                    e = self._events[key] = h
        return e

    def record(self, event, stream):
        rc = self.rt.hipEventRecord(event, c_void_p(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed ({rc})")

    def wait(self, stream, event):
        rc = self.rt.hipStreamWaitEvent(c_void_p(stream.cuda_stream), event, 0)
        if rc != 0:
            raise RuntimeError(f"hipStreamWaitEvent failed ({rc})")


_hip = None
_side_streams = {}


def hip() -> _Hip:
    global _hip
    if _hip is None:
        with _lock:
            if _hip is None:
                _hip = _Hip()
    return _hip


def side_stream(device, which=0, priority=0):
    """One cached side stream per (device, which): the pipelined forward's second lane, the harness's pre / post lanes."""
    import torch
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    key = (index, which, priority)
    s = _side_streams.get(key)
    if s is None:
        with _lock:
            s = _side_streams.get(key)
            if s is None:
                s = _side_streams[key] = torch.cuda.Stream(device=torch.device("cuda", index), priority=priority)
    return s


# ---------------------------------------------------------------- operator-level wrappers
def warp(frame2, flow):
    """EMA_VFI.warp (reference ema_vfi.py:149-171) on the GPU."""
    import torch
    _require_cuda(frame2, flow)
    f2, fl = _f32c(frame2), _f32c(flow)
    B, C, H, W = f2.shape
    if fl.shape != (B, 2, H, W):
        raise ValueError(f"flow must be [B,2,H,W] = {(B, 2, H, W)}, got {tuple(fl.shape)}")
    out = torch.empty_like(f2)
    with torch.cuda.device(f2.device):
        check(load().emavfi_warp(f2.data_ptr(), fl.data_ptr(), out.data_ptr(), B, C, H, W, _stream()), "emavfi_warp")
    return out


def conv3x3(x, weight, bias, stride=1, act=ACT_NONE, dtype="fp32"):
    """One conv / conv_block of the reference (ema_vfi.py:7-14)."""
    import torch
    _require_cuda(x, weight, bias)
    dt = dtype_code(dtype)
    x, w = _f32c(x), _f32c(weight)
    b = _f32c(bias) if bias is not None else torch.zeros(w.shape[0], device=x.device)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3):
        raise ValueError(f"weight must be [Cout,{Cin},3,3], got {tuple(w.shape)}")
    L = load()
    n = L.emavfi_conv3x3_workspace_bytes(B, Cin, Cout, H, W, stride, dt)
    if n == 0:
        raise RuntimeError(f"emavfi_conv3x3: {last_error()}")
    ws = workspace(n, x.device)
    y = torch.empty(B, Cout, (H + stride - 1) // stride, (W + stride - 1) // stride, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(L.emavfi_conv3x3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Cin, Cout, H, W, stride, act, dt,
                               ws.data_ptr(), ws.numel(), _stream()), "emavfi_conv3x3")
    return y


def deform_conv2d(x, offset, mask, weight, bias, dtype="fp32"):
    """torchvision.ops.deform_conv2d as configured at reference ema_vfi.py:45-51."""
    import torch
    _require_cuda(x, offset, mask, weight, bias)
    dt = dtype_code(dtype)
    x, off, msk, w = _f32c(x), _f32c(offset), _f32c(mask), _f32c(weight)
    B, C, H, W = x.shape
    O = w.shape[0]
    b = _f32c(bias) if bias is not None else torch.zeros(O, device=x.device)
    if tuple(off.shape) != (B, 18, H, W) or tuple(msk.shape) != (B, 9, H, W) or tuple(w.shape) != (O, C, 3, 3):
        raise ValueError("deform_conv2d: offset [B,18,H,W], mask [B,9,H,W], weight [O,C,3,3] expected")
    L = load()
    n = L.emavfi_deform_conv2d_workspace_bytes(B, C, O, H, W, dt)
    if n == 0:
        raise RuntimeError(f"emavfi_deform_conv2d: {last_error()}")
    ws = workspace(n, x.device)
    y = torch.empty(B, O, H, W, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        check(L.emavfi_deform_conv2d(x.data_ptr(), off.data_ptr(), msk.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                     B, C, O, H, W, dt, ws.data_ptr(), ws.numel(), _stream()), "emavfi_deform_conv2d")
    return y


def mdcn(x, offset_weight, offset_bias, dcn_weight, dcn_bias, dtype="fp32", flags=0, _events=None):
    """ModulatedDeformConvPack.forward (reference ema_vfi.py:53-60) as ONE stage, routed as a block of the forward is
    (include/emavfi.h, emavfi_mdcn): the one-launch kernel in the 16-bit modes at the reference width.
    `_events` (bench.py): (ctypes array of hipEvent_t, count) bracketing the stage's own launches (emavfi_mdcn_profiled)."""
    import torch
    _require_cuda(x, offset_weight, offset_bias, dcn_weight, dcn_bias)
    dt = dtype_code(dtype)
    x, ow, ob, dw = _f32c(x), _f32c(offset_weight), _f32c(offset_bias), _f32c(dcn_weight)
    db = _f32c(dcn_bias) if dcn_bias is not None else None
    B, C, H, W = x.shape
    if tuple(ow.shape) != (27, C, 3, 3) or tuple(ob.shape) != (27,) or tuple(dw.shape) != (C, C, 3, 3) or (db is not None and tuple(db.shape) != (C,)):
        raise ValueError(f"mdcn: offset_conv [27,{C},3,3] + [27] and dcn_v2 [{C},{C},3,3] + [{C}] expected")
    L = load()
    n = L.emavfi_mdcn_workspace_bytes(B, C, H, W, dt, flags)
    if n == 0:
        raise RuntimeError(f"emavfi_mdcn: {last_error()}")
    ws = workspace(n, x.device)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        if _events is not None:
            check(L.emavfi_mdcn_profiled(x.data_ptr(), ow.data_ptr(), ob.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None,
                                         y.data_ptr(), B, C, H, W, dt, flags, ws.data_ptr(), ws.numel(), ctypes.cast(_events[0], POINTER(c_void_p)),
                                         _events[1], _stream()), "emavfi_mdcn_profiled")
        else:
            check(L.emavfi_mdcn(x.data_ptr(), ow.data_ptr(), ob.data_ptr(), dw.data_ptr(), db.data_ptr() if db is not None else None, y.data_ptr(),
                                B, C, H, W, dt, flags, ws.data_ptr(), ws.numel(), _stream()), "emavfi_mdcn")
    return y


def _census_rows(raw, samples):
    """[blocks][4] u64 (device) -> list of dicts (one blocking copy)"""
    import struct
    rows = []
    for fix, total, parked, mx in raw.cpu().tolist():
        if total == 0:
            rows.append(None)   # this block did not run the one-launch kernel
            continue
        rows.append({"fixup_wave_taps": fix, "wave_taps": total, "fixup_share": fix / total, "samples_outside_window": parked,
                     "samples_outside_share": parked / samples, "abs_offset_px_max": struct.unpack("<f", struct.pack("<I", mx & 0xffffffff))[0]})
    return rows


def mdcn_census(B, C, H, W, dtype="bf16", flags=0, device=None):
    """What the one-launch pack kernel counted during the LAST lib.mdcn(...) call of this shape on the current stream
    (include/emavfi.h, emavfi_mdcn_census): [{fixup_wave_taps, wave_taps, fixup_share, samples_outside_window, ...}] or [None]."""
    import torch
    L = load()
    dt = dtype_code(dtype)
    dev = torch.device(device if device is not None else "cuda")
    n = L.emavfi_mdcn_workspace_bytes(B, C, H, W, dt, flags)
    if n == 0:
        raise RuntimeError(f"emavfi_mdcn_census: {last_error()}")
    ws = workspace(n, dev)
    out = torch.zeros(1, 4, dtype=torch.int64, device=ws.device)
    with torch.cuda.device(ws.device):
        check(L.emavfi_mdcn_census(B, C, H, W, dt, flags, ws.data_ptr(), ws.numel(), out.data_ptr(), _stream()), "emavfi_mdcn_census")
    return _census_rows(out, B * H * W * 9)


def _stage(entry, ws_entry, x, params, out_shape, mid, dtype, what):
    import torch
    _require_cuda(x, *params)
    dt = dtype_code(dtype)
    x = _f32c(x)
    ps = [_f32c(p) for p in params]
    B, _, H, W = x.shape
    L = load()
    n = getattr(L, ws_entry)(B, mid, H, W, dt)
    if n == 0:
        raise RuntimeError(f"{what}: {last_error()}")
    ws = workspace(n, x.device)
    out = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    arr = (c_void_p * len(ps))(*[p.data_ptr() for p in ps])
    with torch.cuda.device(x.device):
        check(getattr(L, entry)(x.data_ptr(), ctypes.cast(arr, POINTER(c_void_p)), out.data_ptr(), B, mid, H, W, dt, ws.data_ptr(), ws.numel(), _stream()), what)
    return out


def context(feat, params, dtype="fp32"):
    """context_encoding(feat) -> ctx [B, mid] (reference ema_vfi.py:79-86, :120); params: the 8 tensors of the Sequential in
    registration order.  The launches the forward runs for the stage (include/emavfi.h, emavfi_context)."""
    mid = feat.shape[1]
    shapes = [(2 * mid, mid, 3, 3), (2 * mid,), (4 * mid, 2 * mid, 3, 3), (4 * mid,), (4 * mid, 4 * mid, 3, 3), (4 * mid,), (mid, 4 * mid), (mid,)]
    if len(params) != 8 or [tuple(p.shape) for p in params] != shapes:
        raise ValueError(f"context: 8 tensors of shapes {shapes} expected")
    return _stage("emavfi_context", "emavfi_context_workspace_bytes", feat, params, (feat.shape[0], mid), mid, dtype, "emavfi_context")


def reconstruct(fused, params, dtype="fp32"):
    """reconstruction(fused) -> frame [B, 3, H, W] in [0, 1] (reference ema_vfi.py:102-107, :144-146); params: the 6 tensors of the
    Sequential in registration order.  The launches the forward runs for the stage (include/emavfi.h, emavfi_reconstruct)."""
    mid = fused.shape[1] - 3
    shapes = [(mid, mid + 3, 3, 3), (mid,), (mid // 2, mid, 3, 3), (mid // 2,), (3, mid // 2, 3, 3), (3,)]
    if len(params) != 6 or [tuple(p.shape) for p in params] != shapes:
        raise ValueError(f"reconstruct: 6 tensors of shapes {shapes} expected")
    return _stage("emavfi_reconstruct", "emavfi_reconstruct_workspace_bytes", fused, params, (fused.shape[0], 3, fused.shape[2], fused.shape[3]), mid, dtype,
                  "emavfi_reconstruct")


def forward_launches(in_channels, mid_channels, num_blocks, B, H, W, dtype):
    """[(name, algorithmic_flops, algorithmic_bytes)] for every kernel launch of one forward."""
    L = load()
    dt = dtype_code(dtype)
    n = L.emavfi_forward_launches(in_channels, mid_channels, num_blocks, B, H, W, dt, None, 0, None, None, 0)
    if n < 0:
        raise RuntimeError(f"emavfi_forward_launches: {last_error()}")
    names = ctypes.create_string_buffer(128 * n)
    fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)()
    check(min(0, L.emavfi_forward_launches(in_channels, mid_channels, num_blocks, B, H, W, dt, names, len(names), fl, by, n)),
          "emavfi_forward_launches")
    labels = names.value.decode().strip().split("\n")
    return [(labels[i], fl[i], by[i]) for i in range(n)]


IMAGENET_MEAN = (0.485, 0.456, 0.406)  # reference inference.py:40
IMAGENET_STD = (0.229, 0.224, 0.225)


def _stats(mean, std, C, ctype=ctypes.c_float):
    if len(mean) != C or len(std) != C:
        raise ValueError(f"mean/std must have {C} entries")
    return (ctype * C)(*mean), (ctype * C)(*std)


def _pinned_or_cuda(t, what):
    if not (t.is_cuda or (t.device.type == "cpu" and t.is_pinned())):
        raise RuntimeError(f"{what}: a ROCm tensor or a PINNED host tensor (read / written by the kernel over PCIe) is expected")


def preprocess_u8(frames_hwc, mean=IMAGENET_MEAN, std=IMAGENET_STD, device=None, out=None):
    """ToTensor + Normalize of reference inference.py:38-41 on the GPU: uint8 [B,H,W,C] -> fp32 [B,C,H,W].
    `frames_hwc` may be a pinned host tensor: the kernel then reads it in place over PCIe (no separate H2D copy;
    `device` names the GPU) - the caller keeps it unchanged until the kernel has run.  `out`: a contiguous fp32 [B,C,H,W] device
    tensor to fill (the streaming harness keeps one per buffer slot, so nothing is allocated on its side streams)."""
    import torch
    _pinned_or_cuda(frames_hwc, "preprocess_u8")
    if frames_hwc.dtype != torch.uint8 or frames_hwc.dim() != 4:
        raise ValueError("preprocess_u8: uint8 [B,H,W,C] tensor expected")
    x = frames_hwc if frames_hwc.is_contiguous() else frames_hwc.contiguous()
    dev = x.device if x.is_cuda else torch.device(device if device is not None else "cuda")
    B, H, W, C = x.shape
    m, s = _stats(mean, std, C)
    if out is None:
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
    elif not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (B, C, H, W) and out.is_contiguous()):
        raise ValueError("preprocess_u8: out must be a contiguous fp32 [B,C,H,W] device tensor")
    with torch.cuda.device(dev):
        check(load().emavfi_preprocess_u8(x.data_ptr(), out.data_ptr(), B, H, W, C, m, s, _stream()), "emavfi_preprocess_u8")
    return out


def postprocess_u8(frames_nchw, denormalize=True, mean=IMAGENET_MEAN, std=IMAGENET_STD, out=None):
    """denormalize_frame of reference inference.py:51-58 on the GPU: fp32 [B,C,H,W] -> uint8 [B,H,W,C].
    `out` may be a pinned host tensor [B,H,W,C]: the kernel then writes the frames straight into host memory."""
    import torch
    _require_cuda(frames_nchw)
    x = _f32c(frames_nchw)
    if x.dim() != 4:
        raise ValueError("postprocess_u8: [B,C,H,W] tensor expected")
    B, C, H, W = x.shape
    m, s = _stats(mean, std, C, ctypes.c_double)  # numpy's float64 constants (inference.py:55)
    if out is None:
        out = torch.empty(B, H, W, C, dtype=torch.uint8, device=x.device)
    else:
        _pinned_or_cuda(out, "postprocess_u8(out=)")
        if out.dtype != torch.uint8 or tuple(out.shape) != (B, H, W, C) or not out.is_contiguous():
            raise ValueError("postprocess_u8: out must be a contiguous uint8 [B,H,W,C] tensor")
    with torch.cuda.device(x.device):
        check(load().emavfi_postprocess_u8(x.data_ptr(), out.data_ptr(), B, H, W, C, m, s, 1 if denormalize else 0, _stream()),
              "emavfi_postprocess_u8")
    return out
