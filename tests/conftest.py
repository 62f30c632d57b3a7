import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "video-frame-interpolation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# packed-weight cache: a per-session directory, so the suite neither reads nor leaves files under ~/.cache
import tempfile  # noqa: E402
os.environ.setdefault("EMAVFI_CACHE_DIR", tempfile.mkdtemp(prefix="emavfi_cache_"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs on torch's intra-op pool: cap it, a pool wider than the cores this process is granted
    # (container CPU shares) stalls for tens of milliseconds per op
    import torch
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))


def have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def oracle_c():
    """ctypes handle on the plain-C oracle restatement (oracle/deform_warp_ref.c)."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.oracle_deform_conv2d.argtypes = [fp] * 6 + [ctypes.c_int] * 5
    lib.oracle_deform_conv2d.restype = None
    lib.oracle_warp.argtypes = [fp] * 3 + [ctypes.c_int] * 4
    lib.oracle_warp.restype = None

    def ptr(a):
        return a.ctypes.data_as(fp) if a is not None else None

    class C:
        @staticmethod
        def deform(x, offset, mask, weight, bias):
            x, offset, mask, weight = (np.ascontiguousarray(t, dtype=np.float32) for t in (x, offset, mask, weight))
            bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
            B, Cc, H, W = x.shape
            O = weight.shape[0]
            out = np.empty((B, O, H, W), dtype=np.float32)
            lib.oracle_deform_conv2d(ptr(x), ptr(offset), ptr(mask), ptr(weight), ptr(bias), ptr(out), B, Cc, O, H, W)
            return out

        @staticmethod
        def warp(frame2, flow):
            frame2, flow = (np.ascontiguousarray(t, dtype=np.float32) for t in (frame2, flow))
            B, Cc, H, W = frame2.shape
            out = np.empty_like(frame2)
            lib.oracle_warp(ptr(frame2), ptr(flow), ptr(out), B, Cc, H, W)
            return out

    return C


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name} missing")
    return np.load(path)
