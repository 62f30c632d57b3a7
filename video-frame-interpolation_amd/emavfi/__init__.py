"""MI355X-native EMA-VFI inference path (host side).

``emavfi.model.EMA_VFI`` mirrors the reference class at
``src/models/ema_vfi.py:63`` (same constructor, state_dict keys and
``forward(frame1, frame2)``); all arithmetic happens in ``lib/libemavfi.so``
(hand-written HIP for gfx950) reached through the C-ABI of ``include/emavfi.h``.
"""
from .model import EMA_VFI, ModulatedDeformConvPack, DeformConv2d, conv, conv_block  # noqa: F401
from . import lib, synth, dist  # noqa: F401
from .stream import FrameInterpolator  # noqa: F401

__all__ = ["EMA_VFI", "ModulatedDeformConvPack", "DeformConv2d", "conv", "conv_block", "lib", "synth", "dist", "FrameInterpolator"]
