// det_math_pk.h -- two-wide forms of the deterministic elementary functions of det_math.h for the GPU epilogues.
// gfx950 issues an fp32 FMA / MUL / ADD on a PAIR of lanes-values per instruction (v_pk_fma_f32 ...): the polynomial chains, which are
// ~3/4 of the softplus, run on two activations at once.  Every element sees exactly the operation sequence of the scalar function
// (IEEE fma / mul / add per element), so results are bit-identical to det_math.h -- checked on the device by mvsdf_det_math op 6.
#pragma once
#include "det_math.h"

typedef float dm_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ dm_f2 dm2_fma(dm_f2 a, dm_f2 b, dm_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dm_f2 dm2_s(float v) { return dm_f2{v, v}; }

// exp(x) for x <= 0 (dm_expneg)
__device__ __forceinline__ dm_f2 dm2_expneg(dm_f2 x) {
    x = dm_f2{fmaxf(x.x, -86.0f), fmaxf(x.y, -86.0f)};
    const dm_f2 magic = dm2_s(12582912.0f);
    const dm_f2 t = dm2_fma(x, dm2_s(1.4426950216293335f), magic);
    const dm_f2 n = t - magic;
    dm_f2 r = dm2_fma(n, dm2_s(-0.693359375f), x);
    r = dm2_fma(n, dm2_s(2.12194440e-4f), r);
    dm_f2 p = dm2_s(1.9875691500e-4f);
    p = dm2_fma(p, r, dm2_s(1.3981999507e-3f));
    p = dm2_fma(p, r, dm2_s(8.3334519073e-3f));
    p = dm2_fma(p, r, dm2_s(4.1665795894e-2f));
    p = dm2_fma(p, r, dm2_s(1.6666665459e-1f));
    p = dm2_fma(p, r, dm2_s(5.0000001201e-1f));
    const dm_f2 e = dm2_fma(p, r * r, r) + dm2_s(1.0f);
    return dm_f2{dm_from_bits(dm_to_bits(e.x) + (dm_to_bits(t.x) << 23)), dm_from_bits(dm_to_bits(e.y) + (dm_to_bits(t.y) << 23))};
}

// log1p(t) for t in [0, 1] (dm_log1p01)
__device__ __forceinline__ dm_f2 dm2_log1p01(dm_f2 t) {
    const bool k0 = !(t.x < 0.4142135679721832f), k1 = !(t.y < 0.4142135679721832f);
    const dm_f2 up = dm2_fma(t, dm2_s(0.5f), dm2_s(-0.5f));
    const dm_f2 f = dm_f2{k0 ? up.x : t.x, k1 ? up.y : t.y};
    dm_f2 q = dm2_s(7.1513607744e-02f);
    q = dm2_fma(q, f, dm2_s(-1.1573007339e-01f));
    q = dm2_fma(q, f, dm2_s(1.1661760853e-01f));
    q = dm2_fma(q, f, dm2_s(-1.2410829558e-01f));
    q = dm2_fma(q, f, dm2_s(1.4249891856e-01f));
    q = dm2_fma(q, f, dm2_s(-1.6668487893e-01f));
    q = dm2_fma(q, f, dm2_s(2.0000708849e-01f));
    q = dm2_fma(q, f, dm2_s(-2.4999988981e-01f));
    q = dm2_fma(q, f, dm2_s(3.3333331185e-01f));
    const dm_f2 f2 = f * f;
    const dm_f2 res = dm2_fma(f2 * f, q, dm2_fma(dm2_s(-0.5f), f2, f));
    return res + dm_f2{k0 ? 0.6931471805599453f : 0.0f, k1 ? 0.6931471805599453f : 0.0f};
}

// Softplus(beta=100, threshold=20) (dm_softplus100)
__device__ __forceinline__ dm_f2 dm2_softplus100(dm_f2 z) {
    const dm_f2 y = z * dm2_s(100.0f);
    const dm_f2 t = dm2_expneg(dm_f2{-fabsf(y.x), -fabsf(y.y)});
    const dm_f2 s = dm_f2{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)} + dm2_log1p01(t);
    const dm_f2 r = s * dm2_s(0.009999999776482582f);            // dm_div100
    return dm_f2{y.x > 20.0f ? z.x : r.x, y.y > 20.0f ? z.y : r.y};
}

// Softplus(beta=100, threshold=20), the LEAN form of the f32x3 tracing arithmetic (dm_softplus100_lean: same operations per element, bit-identical;
// mvsdf_det_math op 8 checks it on the device).  Per activation: one v_med3 (|z| clamped), one v_med3 (max(z, 0)), one v_lshl_add_u32 and sixteen
// two-wide fma / add halves = 11 instructions (dm2_softplus100: 27).
__device__ __forceinline__ dm_f2 dm2_softplus100_lean(dm_f2 z) {
    const dm_f2 za = dm_f2{__builtin_amdgcn_fmed3f(fabsf(z.x), 0.0f, 0.2f), __builtin_amdgcn_fmed3f(fabsf(z.y), 0.0f, 0.2f)};
    const dm_f2 magic = dm2_s(12582912.0f), c = dm2_s(-144.26950073242188f);
    const dm_f2 tm = dm2_fma(za, c, magic);
    const dm_f2 nf = tm - magic;
    const dm_f2 fr = dm2_fma(za, c, -nf);
    dm_f2 p = dm2_s(1.341362135e-03f);
    p = dm2_fma(p, fr, dm2_s(9.671657346e-03f));
    p = dm2_fma(p, fr, dm2_s(5.550281703e-02f));
    p = dm2_fma(p, fr, dm2_s(2.402223945e-01f));
    p = dm2_fma(p, fr, dm2_s(6.931472421e-01f));
    p = dm2_fma(p, fr, dm2_s(1.0f));
    const dm_f2 t = dm_f2{dm_from_bits(dm_to_bits(p.x) + (dm_to_bits(tm.x) << 23)), dm_from_bits(dm_to_bits(p.y) + (dm_to_bits(tm.y) << 23))};
    dm_f2 g = dm2_s(-6.453247624e-05f);
    g = dm2_fma(g, t, dm2_s(3.608817351e-04f));
    g = dm2_fma(g, t, dm2_s(-9.533045813e-04f));
    g = dm2_fma(g, t, dm2_s(1.676565735e-03f));
    g = dm2_fma(g, t, dm2_s(-2.407359425e-03f));
    g = dm2_fma(g, t, dm2_s(3.317999188e-03f));
    g = dm2_fma(g, t, dm2_s(-4.998743068e-03f));
    g = dm2_fma(g, t, dm2_s(9.999964386e-03f));
    return dm2_fma(g, t, dm_f2{__builtin_amdgcn_fmed3f(z.x, 0.0f, 3.0e38f), __builtin_amdgcn_fmed3f(z.y, 0.0f, 3.0e38f)});
}
