import numpy as np, sys
from fractions import Fraction
D0 = '/root/repo/gpurun_out/mfma/'
SH = 200  # fixed point: value * 2^SH as python int

def load(pre, s):
    A = np.fromfile(D0 + '%s_s%d_A.bin' % (pre, s), dtype=np.uint16).reshape(-1, 16, 32)
    B = np.fromfile(D0 + '%s_s%d_B.bin' % (pre, s), dtype=np.uint16).reshape(-1, 16, 32)
    C = np.fromfile(D0 + '%s_s%d_C.bin' % (pre, s), dtype=np.float32).reshape(-1, 16, 16)
    D = np.fromfile(D0 + '%s_s%d_D.bin' % (pre, s), dtype=np.float32).reshape(-1, 16, 16)
    return A, B, C, D

def bf_parts(h):
    """bf16 bits -> (sign, unbiased exponent e, mantissa m in [128, 255]) with value = (-1)^s * m * 2^(e - 7); zero -> None"""
    s = (h >> 15) & 1; eb = (h >> 7) & 0xff; m = h & 0x7f
    if eb == 0: return None
    return s, int(eb) - 127, int(m) | 0x80

def f32_fix(f):
    """float32 -> exact int scaled by 2^SH"""
    u = int(np.float32(f).view(np.uint32))
    s = u >> 31; eb = (u >> 23) & 0xff; m = u & 0x7fffff
    if eb == 0: return 0
    v = (m | 0x800000) << (eb - 127 - 23 + SH)
    return -v if s else v

def fix_to_f32(x):
    """exact int (scaled 2^SH) -> float32 by round-to-nearest-even"""
    if x == 0: return np.float32(0.0)
    neg = x < 0; ax = -x if neg else x
    hb = ax.bit_length() - 1
    drop = hb - 23
    if drop > 0:
        mant = ax >> drop; rem = ax & ((1 << drop) - 1); half = 1 << (drop - 1)
        if rem > half or (rem == half and (mant & 1)): mant += 1
    else:
        mant = ax << (-drop)
    v = np.float64(mant) * np.float64(2.0) ** (drop - SH)
    return np.float32(-v if neg else v)

def trunc_mag(x, cut):
    if cut <= 0: return x
    return -((-x >> cut) << cut) if x < 0 else (x >> cut) << cut
