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
void p_softplus100_lean(const float* x, int n, float* y) { for (int i = 0; i < n; ++i) y[i] = dm_softplus100_lean(x[i]); }
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
        zl = np.concatenate([z, rs.uniform(-3, 3, 100000).astype(np.float32), np.float32([-0.0, 0.2, 0.20000002, 0.19999999, -0.2, 1e-30, -1e-30, 50.0, -50.0])])
        assert np.array_equal(run('p_softplus100_lean', zl), oracle.softplus100_lean(zl))
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


def test_lean_softplus_accuracy_and_range(oracle):
    """dm_softplus100_lean (the activation of the default tracing arithmetic 'f32x3'): Softplus(beta=100, threshold=20) of idr.py:75 to the ABSOLUTE accuracy of
    an fp32 rounding of the exact result (7.5e-9 on results up to 0.3; rms 1.1e-9 -- dm_softplus100, built for relative accuracy, measures 2.0e-8 / 2.7e-9),
    z > 0.2 returns z exactly (the reference's threshold branch), and no result ever falls below 2e-11 (above the engine's 2^-40 flush: no denormal term)."""
    z = np.concatenate([np.linspace(-0.4, 0.4, 4000001), np.linspace(-0.01, 0.01, 400001)]).astype(np.float32)
    y = 100.0 * z.astype(np.float64)
    exact = np.where(y > 20, z.astype(np.float64), np.log1p(np.exp(np.minimum(y, 20.0))) / 100.0)
    lean = oracle.softplus100_lean(z).astype(np.float64)
    det = oracle.softplus100(z).astype(np.float64)
    e_lean, e_det = np.abs(lean - exact), np.abs(det - exact)
    print('lean: max %.3g rms %.3g; dm_softplus100: max %.3g rms %.3g' % (e_lean.max(), np.sqrt((e_lean ** 2).mean()), e_det.max(), np.sqrt((e_det ** 2).mean())))
    assert e_lean.max() < 8e-9 and np.sqrt((e_lean ** 2).mean()) < 1.3e-9
    assert e_lean.max() <= e_det.max() and (e_lean ** 2).mean() <= (e_det ** 2).mean()
    near0 = np.abs(z) < 0.01                                                       # results ~0.007: within 4 ulp (4.7e-10 each)
    assert e_lean[near0].max() < 1.9e-9
    big = np.float32([0.2000001, 0.25, 1.0, 37.5, 1e6, 1e30])
    assert np.array_equal(oracle.softplus100_lean(big), big)
    neg = oracle.softplus100_lean(np.float32([-0.2, -0.21, -1.0, -1e6, -1e30]))
    assert (neg > 2e-11).all() and (neg < 2.2e-11).all() and len(set(neg.tolist())) == 1
    w = oracle.softplus100_lean(np.linspace(-0.3, 0.3, 200001).astype(np.float32))
    assert (np.diff(w) >= -1e-9).all()                                             # monotone up to rounding
