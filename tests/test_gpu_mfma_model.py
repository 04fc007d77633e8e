"""The bit-level model of v_mfma_f32_16x16x32_bf16 behind the default tracing arithmetic 'f32x3' (oracle/oracle_mvsdf.c::mfma_step8 and its eight-column AVX2
form; derivation: tools/micro/mfma_bf16_model/README.md) re-verified on whatever GPU runs the suite:
  * the ORACLE's C model against the bare instruction on random tiles (exponent windows of 1 .. 24 octaves, zeros, accumulators far above / below the
    products, and the `low` domain: operands down to 2^-49 -- the engine flushes below 2^-40, so its smallest product is 2^-112);
  * a bounded run of the on-GPU fuzz (the model in 64-bit integers beside the instruction): ~1e9 outputs in a few seconds, two seeds + one `low` seed.
Test infrastructure only: tests/native/mfma_check.hip is built into its own library, the product never loads it."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def chk():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'native'))
    import build_native as nb
    L = C.CDLL(nb.build())
    L.mfma_fuzz_run.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.mfma_exec_tiles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return L


def _tiles(rs, n, low):
    """random bf16 operand tiles + accumulators in the regimes of mfma_check.hip"""
    def operands():
        spread = rs.choice([1, 2, 3, 4, 6, 8, 24], size=(n, 1, 1))
        base = 127 - (rs.randint(20, 36, size=(n, 1, 1)) if low else rs.choice([0, 6], size=(n, 1, 1)))
        if low:
            spread = np.minimum(spread, 8)
        e = base - (rs.randint(0, 1 << 16, size=(n, 16, 32)) % spread)
        v = (rs.randint(0, 2, size=(n, 16, 32)) << 15) | (e << 7) | rs.randint(0, 128, size=(n, 16, 32))
        v[rs.randint(0, 64, size=(n, 16, 32)) == 0] = 0
        return v.astype(np.uint16)
    A, B = operands(), operands()
    ce = (127 - 100 + rs.randint(0, 104, size=(n, 16, 16))) if low else (127 - 40 + rs.randint(0, 94, size=(n, 16, 16)))
    Cb = (rs.randint(0, 2, size=(n, 16, 16)).astype(np.uint32) << 31) | (ce.astype(np.uint32) << 23) | rs.randint(0, 1 << 23, size=(n, 16, 16)).astype(np.uint32)
    Cb[rs.randint(0, 32, size=(n, 16, 16)) == 0] = 0
    Cb[rs.randint(0, 64, size=(n, 16, 16)) == 0] = 0x80000000                       # signed zeros
    return A, B, Cb.view(np.float32)


@pytest.mark.parametrize('low', [False, True])
def test_oracle_instruction_model_equals_the_hardware(oracle, chk, low):
    rs = np.random.RandomState(11 + low)
    n = 24000
    A, B, Cin = _tiles(rs, n, low)
    D = np.empty_like(Cin)
    assert chk.mfma_exec_tiles(A.ctypes.data, B.ctypes.data, Cin.ctypes.data, D.ctypes.data, n) == 0
    for vector in (False, True):
        M = oracle.mfma_tiles(A, B, Cin, vector=vector)
        same = (M.view(np.uint32) == D.view(np.uint32)) | ((M == 0) & (D == 0))
        bad = np.argwhere(~same)
        assert bad.shape[0] == 0, ('vector' if vector else 'scalar', bad.shape[0], bad[:3], M[~same][:3], D[~same][:3])
    assert np.isfinite(D).all() and (D != Cin).mean() > 0.5                           # (the instruction did something)


def test_on_gpu_fuzz_of_the_instruction_model(chk):
    out = (C.c_ulonglong * 3)()
    total = 0
    for seed, mode in ((101, 0), (102, 0), (103, 1)):
        assert chk.mfma_fuzz_run(seed, 8192, 160, mode, out) == 0
        assert out[1] == 8192 * 160 * 256 and out[0] == 0, (seed, mode, out[0], out[1], (out[2] & ~(1 << 63)) // 256)
        total += out[1]
    print('%d outputs of v_mfma_f32_16x16x32_bf16 compared with the model on the GPU: 0 differ' % total)
