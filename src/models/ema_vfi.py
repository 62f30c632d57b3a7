"""Drop-in for the reference's ``src/models/ema_vfi.py``.

``from src.models.ema_vfi import EMA_VFI`` (reference inference.py:4, train.py:6)
resolves to the MI355X-native implementation: same class name, constructor,
state_dict keys and ``forward(frame1, frame2)``; the arithmetic runs in
``libemavfi.so`` (hand-written HIP for gfx950).  See INTEGRATION.md.
"""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "video-frame-interpolation_amd")
_PKG = os.path.abspath(_PKG)
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from emavfi.model import EMA_VFI, ModulatedDeformConvPack, DeformConv2d, conv, conv_block  # noqa: E402,F401

__all__ = ["EMA_VFI", "ModulatedDeformConvPack", "DeformConv2d", "conv", "conv_block"]
