"""ctypes front-end of oracle/rng.c (MT19937 streams N and P of SURVEY.md 8c-bis).

TEST INFRASTRUCTURE ONLY.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle_rng.so')


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    vp, i64, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double
    lib.orc_mt_seed.argtypes = [vp, ctypes.c_uint32]
    lib.orc_np_random_sample.argtypes = [vp, i64, vp]
    lib.orc_np_standard_exponential.argtypes = [vp, i64, vp]
    lib.orc_skewed_levy.argtypes = [vp, dbl, i64, vp, vp]
    lib.orc_cms_from_uw.argtypes = [dbl, i64, vp, vp, vp]
    lib.orc_torch_randn.argtypes = [vp, i64, vp]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


class MT:
    """One MT19937 stream (state layout = struct orc_mt in rng.c)."""

    def __init__(self, seed):
        self.buf = ctypes.create_string_buffer(624 * 4 + 4 + 4 + 8)
        lib().orc_mt_seed(self.buf, seed & 0xFFFFFFFF)

    def random_sample(self, n):
        out = np.empty(n, np.float64)
        lib().orc_np_random_sample(self.buf, n, out.ctypes.data)
        return out

    def standard_exponential(self, n):
        out = np.empty(n, np.float64)
        lib().orc_np_standard_exponential(self.buf, n, out.ctypes.data)
        return out

    def skewed_levy(self, alpha, n):
        """float64 draws of scipy.stats.levy_stable.rvs(alpha/2, 1, 0, 2cos(pi a/4)^(2/a), n)."""
        out = np.empty(n, np.float64)
        scratch = np.empty(2 * n, np.float64)
        lib().orc_skewed_levy(self.buf, float(alpha), n, out.ctypes.data, scratch.ctypes.data)
        return out

    def torch_randn(self, n):
        out = np.empty(n, np.float32)
        lib().orc_torch_randn(self.buf, n, out.ctypes.data)
        return out


def cms_from_uw(alpha, U, W):
    U = np.ascontiguousarray(U, np.float64)
    W = np.ascontiguousarray(W, np.float64)
    out = np.empty_like(U)
    lib().orc_cms_from_uw(float(alpha), U.size, U.ctypes.data, W.ctypes.data, out.ctypes.data)
    return out
