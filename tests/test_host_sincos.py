"""geom.hpp's restatement of the host's sinf / cosf against the libm of the machine running the tests.

fp32 box geometry (crop masks, IoU, NMS) is bit-compared with an oracle that calls the host's sinf / cosf; the device
evaluates the same polynomials in the same order (d3d_amd/csrc/geom.hpp, HostSinCos).  The library exports the HOST build
of that routine so the claim can be checked without a GPU: every float it returns must equal libm's.
"""
import ctypes
import ctypes.util

import numpy as np
import pytest

from d3d_amd import _lib


def _libm():
    m = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for f in (m.sinf, m.cosf):
        f.argtypes = [ctypes.c_float]
        f.restype = ctypes.c_float
    return m


def _ours(angles):
    lib = _lib.load()
    fn = lib.d3d_internal_host_sincosf
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    fn.restype = ctypes.c_int64
    s = np.empty_like(angles)
    c = np.empty_like(angles)
    covered = fn(angles.ctypes.data, angles.size, s.ctypes.data, c.ctypes.data)
    return s, c, covered


def _fma_libm():
    """glibc picks its sinf build by cpu feature; the restatement follows the FMA one (every x86-64 host since 2013)."""
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read().replace("\n", " ")
    except OSError:
        return False


@pytest.mark.parametrize("span", [0.8, 8.0, 100.0, 119.9])
def test_restated_sinf_cosf_equal_the_hosts(span):
    if not _fma_libm():
        pytest.skip("host libm is not the FMA build the restatement follows")
    rng = np.random.default_rng(int(span * 10))
    angles = rng.uniform(-span, span, 200_000).astype(np.float32)
    angles[:8] = np.float32([0.0, -0.0, 1e-5, -1e-5, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi])[:8]
    s, c, covered = _ours(angles)
    assert covered == angles.size
    m = _libm()
    want_s = np.float32([m.sinf(float(a)) for a in angles])
    want_c = np.float32([m.cosf(float(a)) for a in angles])
    assert np.array_equal(s.view(np.uint32), want_s.view(np.uint32))
    assert np.array_equal(c.view(np.uint32), want_c.view(np.uint32))


def test_large_angles_leave_the_restated_path():
    angles = np.float32([120.0, -1e6, np.inf, np.nan, 3e38])
    _, _, covered = _ours(angles)
    assert covered == 0
