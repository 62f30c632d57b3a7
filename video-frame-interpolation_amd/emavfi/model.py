"""Drop-in mirror of the reference model API (reference ``src/models/ema_vfi.py``).

Same names, constructor arguments, parameter names / shapes (so reference
checkpoints load with ``strict=True``) and call signatures as the reference:

* ``conv`` / ``conv_block``                      (ema_vfi.py:7-14)
* ``ModulatedDeformConvPack``                     (ema_vfi.py:23-60)
* ``EMA_VFI(in_channels=3, mid_channels=64, num_blocks=3)`` with
  ``forward(frame1, frame2)`` and ``warp(frame2, feature, flow)``  (ema_vfi.py:63-171)

The modules below only HOLD parameters (``nn.Conv2d`` gives the reference's
initialisation and state_dict keys for free); every operator of the model's
arithmetic executes in ``libemavfi.so``.  What torch still does around a call:
contiguity / dtype normalisation of the inputs (no-ops for the fp32 contiguous
frames the reference passes) and, under ``amp16`` only, the final cast of the
fp32-stored, fp16-valued frame to an fp16 tensor.  Inference only.  There is no CPU or
PyTorch fallback: CPU tensors or a missing library raise ``RuntimeError``.
"""
from __future__ import annotations

import hashlib
import math
import os
import tempfile
from collections import OrderedDict
import ctypes
from ctypes import POINTER, c_void_p, cast

import torch
import torch.nn as nn

from . import lib as _lib


def conv(in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True,
         padding_mode='zeros'):
    """Same signature as the reference helper (ema_vfi.py:7-8)."""
    return nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)


def conv_block(in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True,
               padding_mode='zeros', act=None):
    """Same signature as the reference helper (ema_vfi.py:10-14): Sequential(conv, act)."""
    return nn.Sequential(
        conv(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation,
             groups=groups, bias=bias, padding_mode=padding_mode),
        act if act is not None else nn.ReLU())


class DeformConv2d(nn.Module):
    """Parameter holder with ``torchvision.ops.DeformConv2d``'s constructor and
    initialisation (the class the reference imports at ema_vfi.py:18);
    ``forward(x, offset, mask)`` runs the HIP kernel."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True):
        super().__init__()
        if (kernel_size, stride, padding, dilation, groups) != (3, 1, 1, 1, 1):
            raise NotImplementedError("emavfi DeformConv2d: only the reference's configuration "
                                      "(3x3, stride 1, padding 1, dilation 1, groups 1) is built")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_channels * 9)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, offset, mask=None, dtype="fp32"):
        if mask is None:
            mask = torch.ones(x.shape[0], 9, x.shape[2], x.shape[3], device=x.device, dtype=x.dtype)
        return _lib.deform_conv2d(x, offset, mask, self.weight, self.bias, dtype=dtype).to(x.dtype)


class ModulatedDeformConvPack(nn.Module):
    """Mirror of ema_vfi.py:23-60 (note the reference ignores ``out_channels``, :27)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, bias=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.groups, self.bias = groups, bias
        self.offset_conv = nn.Conv2d(self.in_channels, self.groups * 3 * kernel_size * kernel_size,
                                     kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=True)
        nn.init.constant_(self.offset_conv.weight, 0.)
        nn.init.constant_(self.offset_conv.bias, 0.)
        self.dcn_v2 = DeformConv2d(self.in_channels, self.out_channels, kernel_size=kernel_size, stride=stride,
                                   padding=padding, dilation=dilation, bias=bias)

    def forward(self, x, dtype="fp32", flags=0):
        """ema_vfi.py:53-60 through the stage-level entry emavfi_mdcn: offset_conv, the chunk / cat / sigmoid routing and dcn_v2 run
        exactly as one attention block of EMA_VFI.forward does (16-bit modes at 67 channels: ONE kernel launch)."""
        return _lib.mdcn(x, self.offset_conv.weight, self.offset_conv.bias, self.dcn_v2.weight, self.dcn_v2.bias, dtype=dtype,
                         flags=flags).to(x.dtype)


class EMA_VFI(nn.Module):
    """MI355X-native EMA-VFI.  ``compute_dtype``:

    * ``"fp32"``  parity mode: exact-fp32 MFMA, <= 1e-3 max-abs vs the reference CPU forward;
    * ``"bf16"``  bf16 convolutions and tensors, fp32 warp / offsets / accumulation (BASELINE.json configs[2]);
    * ``"fp16"``  the same data flow in IEEE half, deformable convolution included - the FAST half mode;
    * ``"amp16"`` the reference's forward under ``torch.cuda.amp.autocast()`` (inference.py:159) with the autocast op
      policy restated: fp16 Conv2d / Linear (weight and bias cast too), but ``grid_sample`` and torchvision's
      ``deform_conv2d`` in fp32 on an fp32 fusion tensor with the fp32 master weights, fp16 roundings after
      sigmoid / tanh / ``(t + 1) / 2``; returns an fp16 tensor as the reference does there;
    * ``None``    fp32, unless autocast is active: ``torch.autocast("cuda", torch.float16)`` (the reference's own
      ``torch.cuda.amp.autocast()``) selects ``"amp16"``, ``torch.autocast("cuda", torch.bfloat16)`` selects ``"bf16"``."""

    def __init__(self, in_channels=3, mid_channels=64, num_blocks=3, compute_dtype=None):
        super().__init__()
        self.in_channels = in_channels
        self.mid_channels = mid_channels
        self.num_blocks = num_blocks
        self.deformable_groups = 8  # present but unused in the reference too (ema_vfi.py:70)
        self.compute_dtype = compute_dtype if compute_dtype is not None else os.environ.get("EMAVFI_DTYPE")
        # pieces of a batch pipelined over two streams (forward(): _forward_pipelined); 1 = the whole batch as one sequence of launches
        self.pipeline = int(os.environ.get("EMAVFI_PIPELINE", "1"))
        # which stage event of piece k releases piece k + 1: 0 = its front (default), 1 = its attention blocks, -1 = none (pieces start together)
        self.pipeline_stagger = int(os.environ.get("EMAVFI_PIPELINE_STAGGER", "0"))
        m = mid_channels
        self.feat_ext_conv1 = conv_block(in_channels * 2, m)
        self.feat_ext_blocks = nn.Sequential(OrderedDict(
            [(f'conv_block_{i}', conv_block(m, m)) for i in range(num_blocks)]))
        self.context_encoding = nn.Sequential(
            conv_block(m, m * 2, stride=2), conv_block(m * 2, m * 4, stride=2), conv_block(m * 4, m * 4),
            nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(m * 4, m))
        self.motion_estimation = nn.Sequential(conv_block(m * 2, m), conv_block(m, m), conv(m, 2))
        self.attention_blocks = nn.ModuleList(
            [ModulatedDeformConvPack(m + 3, m + 3, kernel_size=3, padding=1, groups=1) for _ in range(num_blocks)])
        self.reconstruction = nn.Sequential(conv_block(m + 3, m), conv_block(m, m // 2), conv(m // 2, in_channels), nn.Tanh())
        self._packed = {}      # dtype code -> (key, packed uint8 tensor)
        self._installed = {}   # dtype code -> parameter versions when load_packed_weights() installed the blob
        self._params_loaded = True   # False once a foreign blob is installed and no state_dict has been loaded since

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """``nn.Module.load_state_dict`` plus the checkpoint shapes met in practice for this model
        (SURVEY.md section 8f row 4): a ``{"state_dict": ...}`` / ``{"model": ...}`` wrapper, keys saved
        from ``nn.DataParallel`` (``module.`` prefix), half-precision tensors (cast to fp32, the dtype
        ``train.py:182,190`` saves).  Shapes are verified against the reference's parameter table and a
        mismatch raises with the offending key."""
        sd = state_dict
        for wrapper in ("state_dict", "model", "model_state_dict"):
            if isinstance(sd, dict) and wrapper in sd and isinstance(sd[wrapper], dict) and \
                    all(isinstance(k, str) for k in sd[wrapper]):
                sd = sd[wrapper]
                break
        clean = OrderedDict()
        for k, v in sd.items():
            k2 = k[len("module."):] if k.startswith("module.") else k
            clean[k2] = v.float() if torch.is_tensor(v) and v.is_floating_point() and v.dtype != torch.float32 else v
        own = dict(self.named_parameters())
        for k, v in clean.items():
            if k in own and torch.is_tensor(v) and tuple(v.shape) != tuple(own[k].shape):
                raise RuntimeError(f"EMA_VFI.load_state_dict: {k} has shape {tuple(v.shape)}, "
                                   f"expected {tuple(own[k].shape)} for EMA_VFI({self.in_channels}, "
                                   f"{self.mid_channels}, {self.num_blocks})")
        res = super().load_state_dict(clean, strict=strict, assign=assign)
        self._installed.clear()          # the parameters are authoritative again: blobs re-pack from them
        self._packed.clear()
        self._params_loaded = True
        return res

    def _ordered_params(self):
        """Tensors in the order emavfi_pack_weights expects = the reference's registration order."""
        return [p for _, p in self.named_parameters()]

    def _weights_key(self, device):
        return (str(device), _lib.layout_switches()) + tuple((p.data_ptr(), p._version) for p in self._ordered_params())

    def _versions(self):
        return tuple(p._version for p in self._ordered_params())

    def packed_weights(self, dt: int, device):
        """Packed blob for this dtype, re-packed lazily after load_state_dict / .to() / in-place edits."""
        inst = self._installed.get(dt)
        if inst is not None:
            # a blob installed by load_packed_weights() (another rank's weights): it stays in force across .to() /
            # device moves; this rank's nn.Parameters are NOT its source, so an in-place edit cannot be honoured
            if inst != self._versions():
                raise RuntimeError("EMA_VFI: parameters were modified in place, but this model runs on packed weights "
                                   "installed by load_packed_weights() and never loaded a state_dict - call "
                                   "load_state_dict() (then the blob is re-packed from the parameters)")
            blob = self._packed[dt][1]
            if blob.device != torch.device(device):
                blob = blob.to(device)
                self._packed[dt] = (None, blob)
            return blob
        if not self._params_loaded:
            raise RuntimeError(f"EMA_VFI: no packed weights for dtype code {dt}: this model received its weights through "
                               "load_packed_weights() for another dtype and its nn.Parameters were never loaded")
        key = self._weights_key(device)
        hit = self._packed.get(dt)
        if hit is not None and hit[0] == key:
            return hit[1]
        L = _lib.load()
        nbytes = L.emavfi_packed_bytes(self.in_channels, self.mid_channels, self.num_blocks, dt)
        if nbytes == 0:
            raise RuntimeError(f"EMA_VFI: {_lib.last_error()}")
        params = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in self._ordered_params()]
        n = L.emavfi_param_count(self.num_blocks)
        if len(params) != n:
            raise RuntimeError(f"EMA_VFI: expected {n} parameter tensors, module has {len(params)}")
        # on-disk cache keyed by CONTENT (the reference reloads and re-prepares its checkpoint at every
        # process start, inference.py:69): state_dict bytes + dtype + model shape + the library build
        path = self._cache_path(params, dt, nbytes)
        blob = self._cache_read(path, nbytes, device)
        if blob is not None:
            try:   # the header says what the file IS (version, model, dtype, layout tag, checksum): a foreign file is re-packed, never run
                _lib.packed_check(self.in_channels, self.mid_channels, self.num_blocks, dt, blob)
            except RuntimeError:
                blob = None
        if blob is None:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
            arr = (c_void_p * n)(*[p.data_ptr() for p in params])
            with torch.cuda.device(device):
                _lib.check(L.emavfi_pack_weights(self.in_channels, self.mid_channels, self.num_blocks,
                                                 cast(arr, POINTER(c_void_p)), n, blob.data_ptr(), nbytes, dt, _lib._stream()),
                           "emavfi_pack_weights")
            self._cache_write(path, blob)
        self._packed[dt] = (key, blob)
        return blob

    # ------------------------------------------------------------------ packed-weight cache on disk
    @staticmethod
    def cache_dir():
        """Directory of the packed-weight cache, or None when disabled (EMAVFI_CACHE=0)."""
        if os.environ.get("EMAVFI_CACHE", "1") == "0":
            return None
        return os.environ.get("EMAVFI_CACHE_DIR") or os.path.join(
            os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "emavfi")

    def _cache_path(self, params, dt, nbytes):
        d = self.cache_dir()
        if d is None:
            return None
        h = hashlib.sha256()
        h.update(f"{_lib.fingerprint()}|{_lib.layout_switches()}|{dt}|{self.in_channels}|{self.mid_channels}|{self.num_blocks}|{nbytes}".encode())
        for name, p in zip((k for k, _ in self.named_parameters()), params):
            h.update(name.encode())
            h.update(p.cpu().numpy().tobytes())
        return os.path.join(d, f"packed_{h.hexdigest()[:32]}.bin")

    @staticmethod
    def _cache_read(path, nbytes, device):
        """The file is the blob followed by the sha256 of the blob: a truncated, foreign or bit-rotten file is ignored (and
        overwritten by the re-pack), never run."""
        if path is None or not os.path.isfile(path) or os.path.getsize(path) != nbytes + 32:
            return None
        try:
            import numpy as np
            raw = np.fromfile(path, dtype=np.uint8)
        except OSError:
            return None
        if raw.size != nbytes + 32 or hashlib.sha256(raw[:nbytes].tobytes()).digest() != raw[nbytes:].tobytes():
            return None
        return torch.from_numpy(raw[:nbytes].copy()).to(device)

    @staticmethod
    def _cache_write(path, blob):
        if path is None:
            return
        try:  # best effort: a read-only or full cache directory must never fail a forward
            os.makedirs(os.path.dirname(path), exist_ok=True)
            fd, tmp = tempfile.mkstemp(dir=os.path.dirname(path), suffix=".tmp")
            data = blob.cpu().numpy().tobytes()
            with os.fdopen(fd, "wb") as f:
                f.write(data)
                f.write(hashlib.sha256(data).digest())
            os.replace(tmp, path)  # atomic: concurrent ranks write identical bytes
        except OSError:
            pass

    def load_packed_weights(self, dt, blob, params_are_source=False):
        """Install an already-packed blob (e.g. received by an RCCL broadcast from rank 0).  The blob is pinned: it is
        used as is until load_state_dict() is called.  ``params_are_source=True`` (the broadcasting rank, whose own
        parameters produced the blob) keeps the normal re-pack-on-change behaviour."""
        dt = _lib.dtype_code(dt)
        nbytes = _lib.load().emavfi_packed_bytes(self.in_channels, self.mid_channels, self.num_blocks, dt)
        if blob.dtype != torch.uint8 or blob.numel() != nbytes:
            raise ValueError(f"load_packed_weights: a uint8 blob of {nbytes} bytes is expected for this model / dtype")
        # the blob's header must say this model, this dtype, this library version and this process's layout switches
        _lib.packed_check(self.in_channels, self.mid_channels, self.num_blocks, dt, blob)
        if params_are_source:
            self._packed[dt] = (self._weights_key(blob.device), blob)
            return
        self._packed[dt] = (None, blob)
        self._installed[dt] = self._versions()
        self._params_loaded = False

    def _resolve_dtype(self) -> int:
        if self.compute_dtype is not None:
            return _lib.dtype_code(self.compute_dtype)
        if not torch.is_autocast_enabled():
            return _lib.F32
        get = getattr(torch, "get_autocast_dtype", None)
        auto = get("cuda") if get is not None else torch.get_autocast_gpu_dtype()
        return _lib.BF16 if auto == torch.bfloat16 else _lib.AMP16

    # ------------------------------------------------------------------ forward
    def forward(self, frame1, frame2, return_taps=False, _events=None):
        if frame1.shape != frame2.shape or frame1.dim() != 4 or frame1.shape[1] != self.in_channels:
            raise ValueError(f"EMA_VFI.forward: two [B,{self.in_channels},H,W] tensors expected, got "
                             f"{tuple(frame1.shape)} and {tuple(frame2.shape)}")
        if frame1.device != frame2.device:
            raise ValueError("EMA_VFI.forward: frame1 and frame2 are on different devices")
        if self.in_channels != 3:
            # the reference's fusion width is the literal mid_channels + 3 (ema_vfi.py:97): any other in_channels
            # fails there with a channel mismatch at the first attention block
            raise RuntimeError(f"EMA_VFI: in_channels={self.in_channels} is not runnable (the reference's fusion width "
                               "is mid_channels + 3, ema_vfi.py:97); only in_channels=3 is supported")
        _lib._require_cuda(frame1, frame2)
        if torch.is_grad_enabled() and (frame1.requires_grad or frame2.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("EMA_VFI (MI355X-native) is inference-only: call it under torch.no_grad() "
                               "as the reference's inference.py:158 does")
        dev, dt = frame1.device, self._resolve_dtype()
        f1, f2 = _lib._f32c(frame1), _lib._f32c(frame2)
        B, C, H, W = f1.shape
        L = _lib.load()
        packed = self.packed_weights(dt, dev)
        out = torch.empty_like(f1)
        taps_arg, taps = None, None
        if return_taps:
            m = self.mid_channels
            taps = OrderedDict(feat=torch.empty(B, m, H, W, device=dev), ctx=torch.empty(B, m, device=dev),
                               flow=torch.empty(B, 2, H, W, device=dev), warped=torch.empty(B, C, H, W, device=dev))
            ptrs = [taps["feat"].data_ptr(), taps["ctx"].data_ptr(), taps["flow"].data_ptr(), taps["warped"].data_ptr(), None]
            for i in range(self.num_blocks):
                taps[f"fused_{i}"] = torch.empty(B, m + 3, H, W, device=dev)
                ptrs.append(taps[f"fused_{i}"].data_ptr())
            taps_arg = cast((c_void_p * len(ptrs))(*ptrs), POINTER(c_void_p))
        pieces = min(int(self.pipeline), B)
        if (C * H * W) % 4 != 0:
            pieces = 1   # a slice of the batch must start 16-byte aligned (include/emavfi.h): odd sample sizes run as one sequence
        if pieces >= 2 and not return_taps and not torch.cuda.is_current_stream_capturing():
            self._forward_pipelined(L, packed, f1, f2, out, dt, pieces, _events)
        else:
            nws = L.emavfi_workspace_bytes(C, self.mid_channels, self.num_blocks, B, H, W, dt)
            if nws == 0:
                raise RuntimeError(f"EMA_VFI: {_lib.last_error()}")
            ws = _lib.workspace(nws, dev)
            with torch.cuda.device(dev):
                if _events is not None:  # bench.py: (ctypes array of hipEvent_t, count) bracketing every launch
                    _lib.check(L.emavfi_forward_profiled(C, self.mid_channels, self.num_blocks, packed.data_ptr(), packed.numel(), f1.data_ptr(),
                                                         f2.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, dt,
                                                         cast(_events[0], POINTER(c_void_p)), _events[1], _lib._stream()),
                               "emavfi_forward_profiled")
                else:
                    _lib.check(L.emavfi_forward(C, self.mid_channels, self.num_blocks, packed.data_ptr(), packed.numel(), f1.data_ptr(),
                                                f2.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, dt, taps_arg,
                                                _lib._stream()), "emavfi_forward")
            self._last_call = (dev, torch.cuda.current_stream(dev).cuda_stream, B, H, W, dt, nws)
        # under autocast the reference's reconstruction tail is fp16, so its frame is an fp16 tensor (the values computed
        # here are fp16-representable: the conversion is exact)
        out = out.half() if dt == _lib.AMP16 else out.to(frame1.dtype)
        if return_taps:
            taps["out"] = out
            return out, taps
        return out

    def pack_census(self):
        """What the one-launch ModulatedDeformConvPack kernels of the LAST one-sequence forward on the current stream counted while
        they ran (include/emavfi.h, emavfi_forward_census): per attention block a dict {fixup_wave_taps, wave_taps, fixup_share,
        samples_outside_window, samples_outside_share, abs_offset_px_max}, or None for a block that ran other kernels (fp32 / autocast
        modes, other widths).  The reference bounds its offsets nowhere (ema_vfi.py:55-60); the kernel's staged window holds offsets up
        to +-2 px beyond the tap and pays for every (4 x 16 pixels, tap) group with a sample outside it.  One blocking D2H copy."""
        last = getattr(self, "_last_call", None)
        if last is None:
            raise RuntimeError("EMA_VFI.pack_census: no one-sequence forward has run yet")
        dev, stream, B, H, W, dt, nws = last
        if torch.cuda.current_stream(dev).cuda_stream != stream:
            raise RuntimeError("EMA_VFI.pack_census: call it on the stream the forward ran on")
        ws = _lib.workspace(nws, dev)
        out = torch.zeros(self.num_blocks, 4, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().emavfi_forward_census(self.in_channels, self.mid_channels, self.num_blocks, B, H, W, dt, ws.data_ptr(), ws.numel(),
                                                         out.data_ptr(), _lib._stream()), "emavfi_forward_census")
        return _lib._census_rows(out, B * H * W * 9)

    def _forward_pipelined(self, L, packed, f1, f2, out, dt, pieces, _events=None):
        """The batch as `pieces` contiguous slices alternating between the caller's stream and one side stream, piece k + 1 starting
        when piece k's FRONT (feature extraction, context, motion, warp: ema_vfi.py:112-130) has been enqueued-and-finished, so that one
        piece's attention blocks (the LDS-window pack kernel: gather / blend / MFMA per sample) run beside another piece's plain
        convolutions (LDS-ring kernels: MFMA-dense) on the same CUs - the two kernel families fit beside each other (80.8 KB + <= 80 KiB
        of LDS, <= 256 VGPRs each).  Frame pairs are independent (no cross-sample op in ema_vfi.py:110-147) and every kernel is
        batch-invariant per sample, so the result is bit-identical to the one-sequence forward
        (tests/test_gpu_runtime.py::test_pipelined_forward_is_bit_identical).  Joins back onto the caller's stream before returning.
        `_events`: per-launch event pairs, piece k's launches in [k * count / pieces, (k + 1) * count / pieces)."""
        B, C, H, W = f1.shape
        dev = f1.device
        hip = _lib.hip()
        cur = torch.cuda.current_stream(dev)
        side = _lib.side_stream(dev)
        if side.cuda_stream == cur.cuda_stream:
            side = _lib.side_stream(dev, which=1)
        cuts = [B * k // pieces for k in range(pieces + 1)]
        key = (dev.index, cur.cuda_stream)
        per = (_events[1] // pieces) if _events is not None else 0
        if _events is not None:
            # every piece runs the FULL launch list: an array sized for one sequence would leave most of each piece's launches
            # unbracketed and the caller reading never-recorded events
            nl = L.emavfi_forward_launches(C, self.mid_channels, self.num_blocks, 1, H, W, dt, None, 0, None, None, 0)
            if per < 2 * nl:
                raise ValueError(f"EMA_VFI: a pipelined forward of {pieces} pieces needs {pieces} x 2 x {nl} events, got {_events[1]}")
        front_prev = None
        # hipEventCreateWithFlags binds an event to the CURRENT device: everything that creates, records or waits on one runs under
        # the model's device (a model on cuda:1 while cuda:0 is current got hipErrorInvalidHandle here)
        with torch.cuda.device(dev):
            fork, join = hip.event(key + ("fork",)), hip.event(key + ("join",))
            hip.record(fork, cur)          # the frames (and the packed blob) are ready on the caller's stream
            hip.wait(side, fork)
            try:
                for k in range(pieces):
                    b0, b1 = cuts[k], cuts[k + 1]
                    s = cur if k % 2 == 0 else side
                    if front_prev is not None:
                        hip.wait(s, front_prev)
                    front = hip.event(key + ("stage", k))
                    with torch.cuda.stream(s):
                        nws = L.emavfi_workspace_bytes(C, self.mid_channels, self.num_blocks, b1 - b0, H, W, dt)
                        if nws == 0:
                            raise RuntimeError(f"EMA_VFI: {_lib.last_error()}")
                        ws = _lib.workspace(nws, dev)
                        stage = (c_void_p * 3)(*[front if i == self.pipeline_stagger else None for i in range(3)])
                        evp = None
                        if _events is not None:
                            evp = cast(c_void_p(_events[0].value + k * per * ctypes.sizeof(c_void_p)), POINTER(c_void_p))
                        _lib.check(L.emavfi_forward_staged(C, self.mid_channels, self.num_blocks, packed.data_ptr(), packed.numel(),
                                                           f1[b0:b1].data_ptr(), f2[b0:b1].data_ptr(), out[b0:b1].data_ptr(), ws.data_ptr(), ws.numel(),
                                                           b1 - b0, H, W, dt, cast(stage, POINTER(c_void_p)), evp, per, c_void_p(s.cuda_stream)),
                                   "emavfi_forward_staged")
                    front_prev = front if self.pipeline_stagger >= 0 else None
            finally:
                # also when a piece failed to enqueue: `out` and the frames are known to the caching allocator on the caller's stream
                # only, so whatever the side stream already holds must be ordered in front of their reuse
                hip.record(join, side)
                hip.wait(cur, join)
        # the caching allocator knows `out` (and the frames) only on the caller's stream; the side stream's work on them is ordered
        # before everything the caller enqueues from here on by the join above

    def warp(self, frame2, feature, flow):
        """Same signature as the reference (ema_vfi.py:149); ``feature`` is only a device hint there."""
        return _lib.warp(frame2, flow).to(frame2.dtype)
