"""The product's deterministic math header (mvsdf_amd/csrc/det_math.h, host compilation) agrees BIT FOR BIT with the
oracle's independent copy (oracle/det_math.h) -- the basis of HIP == oracle bit-exactness on the tracing path."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from conftest import ROOT

SHIM = r'''
#include "%s"
extern "C" {
void p_softplus100(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = dm_softplus100(x[i]); }
void p_expneg(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = dm_expneg(x[i]); }
void p_log1p01(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = dm_log1p01(x[i]); }
void p_sincos(const float* x, int n, float* s, float* c) { for (int i = 0; i < n; ++i) dm_sincos(x[i], s + i, c + i); }
void p_div(const float* x, int n, float* a, float* b) { for (int i = 0; i < n; ++i) { a[i] = dm_div100(x[i]); b[i] = dm_div_sqrt2(x[i]); } }
void p_sigmoid100(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = dm_sigmoid100(x[i]); }
}
'''


def test_product_header_bitwise_equals_oracle(oracle):
    hdr = os.path.join(ROOT, 'mvsdf_amd', 'csrc', 'det_math.h')
    with tempfile.TemporaryDirectory() as td:
        src, so = os.path.join(td, 'shim.cpp'), os.path.join(td, 'shim.so')
        open(src, 'w').write(SHIM % hdr)
        subprocess.check_call(['g++', '-O2', '-ffp-contract=off', '-mfma', '-shared', '-fPIC', '-o', so, src])
        L = C.CDLL(so)
        rs = np.random.RandomState(1)

        def run(name, x, nout=1):
            x = np.ascontiguousarray(x, np.float32)
            outs = [np.empty_like(x) for _ in range(nout)]
            getattr(L, name)(x.ctypes.data_as(C.c_void_p), C.c_int(x.size), *[o.ctypes.data_as(C.c_void_p) for o in outs])
            return outs[0] if nout == 1 else outs
        z = np.concatenate([rs.uniform(-0.5, 0.5, 300000), rs.uniform(-0.02, 0.02, 100000), [0, 0.2, -2, 3]]).astype(np.float32)
        assert np.array_equal(run('p_softplus100', z), oracle.softplus100(z))
        x = -rs.uniform(0, 110, 300000).astype(np.float32)
        assert np.array_equal(run('p_expneg', x), oracle.expneg(x))
        u = rs.uniform(0, 1, 300000).astype(np.float32)
        assert np.array_equal(run('p_log1p01', u), oracle.log1p01(u))
        a = rs.uniform(-100, 100, 300000).astype(np.float32)
        s, c = run('p_sincos', a, 2)
        so_, co_ = oracle.sincos(a)
        assert np.array_equal(s, so_) and np.array_equal(c, co_)
        v = rs.uniform(-50, 50, 300000).astype(np.float32)
        d0, d1 = run('p_div', v, 2)
        o0, o1 = oracle.div_consts(v)
        assert np.array_equal(d0, o0) and np.array_equal(d1, o1)


def test_oracle_array_softplus_equals_the_scalar_function(oracle):
    """oracle_mvsdf.c evaluates the hidden activations with a branch-free array form of Softplus(100) (eight lanes per instruction on the
    host CPU); it must be dm_softplus100 bit for bit: a dense grid through the interesting range, random values, the thresholds."""
    rs = np.random.RandomState(3)
    x = np.concatenate([np.linspace(-1.5, 1.5, 3000001), rs.normal(size=400000) * 0.3, rs.normal(size=100000) * 5.0,
                        np.array([0.0, -0.0, 0.2, 0.20000002, 0.19999999, -0.2, -0.86, -0.8600001, -5.0, 5.0, 1e-8, -1e-8, 3e38, -3e38])]).astype(np.float32)
    a, b = oracle.softplus100(x), oracle.softplus100_arr(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
