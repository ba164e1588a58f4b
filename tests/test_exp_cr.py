"""csrc/exp_cr.h: the exponential of the Gaussian weights is CORRECTLY ROUNDED -- checked against Python's decimal exp at 60
digits (float(Decimal) rounds to nearest), on the host (the header compiled with gcc) and on the device (glx_exp_cr)."""
import ctypes
import os
import subprocess
import sys
from decimal import Decimal, getcontext

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _arguments():
    rng = np.random.default_rng(0)
    return np.concatenate([-4 * rng.random(20000),                 # what weightmatrix.knn's gaussian kernel evaluates: -4 D/eps in [-4, 0]
                           rng.uniform(-700, 700, 4000), rng.normal(size=3000) * 1e-3, -rng.exponential(30, size=3000),
                           [0.0, -0.0, 1.0, -1.0, 709.0, -708.0, 1e-300, -1e-300, -4.0, np.log(2.0), -np.log(2.0)]])


def _reference(x):
    getcontext().prec = 60
    return np.array([float(Decimal(float(v)).exp()) for v in x])


def test_exp_cr_host_build_is_correctly_rounded(tmp_path):
    src = tmp_path / 'exp_host.c'
    src.write_text('#include "%s"\nvoid exp_cr_array(const double* x, double* out, long n) { for (long i = 0; i < n; ++i) out[i] = exp_cr(x[i]); }\n'
                   % os.path.join(ROOT, 'graphlearning_amd', 'csrc', 'exp_cr.h'))
    so = tmp_path / 'libexpcr.so'
    subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', str(so), str(src), '-lm'], check=True)
    lib = ctypes.CDLL(str(so))
    x = _arguments()
    out = np.empty_like(x)
    lib.exp_cr_array(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(x)))
    assert np.array_equal(out, _reference(x))
    special = np.array([np.nan, np.inf, -np.inf, 800.0, -800.0])
    o2 = np.empty_like(special)
    lib.exp_cr_array(special.ctypes.data_as(ctypes.c_void_p), o2.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(special)))
    assert np.isnan(o2[0]) and o2[1] == np.inf and o2[2] == 0.0 and o2[3] == np.inf and o2[4] == 0.0


@pytest.mark.gpu
def test_exp_cr_device_is_correctly_rounded():
    sys.path.insert(0, ROOT)
    from graphlearning_amd import _hip
    x = _arguments()
    assert np.array_equal(_hip.exp_cr(x), _reference(x))
