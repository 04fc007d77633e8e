/* det_math.h -- deterministic fp32 elementary functions (ORACLE COPY).
 *
 * TEST INFRASTRUCTURE: this file belongs to oracle/ and is only used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product keeps its own copy
 * (mvsdf_amd/csrc/det_math.h); tests/test_det_math.py checks the two agree bit for bit.
 *
 * Why: the tracer takes discrete decisions (sdf > 5e-5, sdf < 0, acc_start < acc_end;
 * reference code/model/ray_tracing.py:41,143,150,173,193,227) on fp32 MLP outputs.  To make
 * "hit masks bit-exact" checkable, every function below is built ONLY from IEEE-754
 * correctly-rounded primitives (+, -, *, fmaf, compare/select, int<->float bit casts), so the
 * CPU oracle and the gfx950 kernels produce identical bits.  Accuracy is ~1-2 ulp, the same
 * class as the Sleef routines behind torch.exp/log1p/sin/cos on CPU.
 *
 * Restates: nn.Softplus(beta=100) (idr.py:75; torch: x*beta > 20 ? x : log1p(exp(x*beta))/beta),
 * Embedder sin/cos (embedder.py:24-30), and the `/ np.sqrt(2)` of idr.py:87.
 * Compile with -ffp-contract=off (explicit fmaf only).
 */
#ifndef MVSDF_ORACLE_DET_MATH_H
#define MVSDF_ORACLE_DET_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef DM_FN
#define DM_FN static inline
#endif

DM_FN float dm_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
DM_FN uint32_t dm_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* round to nearest integer (ties to even), valid for |x| < 2^22 */
DM_FN float dm_rint(float x) {
    const float magic = 12582912.0f; /* 1.5 * 2^23 */
    float t = x + magic;
    return t - magic;
}

/* exp(x) for x <= 0, clamped at x = -86 (exp(-86) = 4.4e-38, still a normal float: the 2^n scaling below is a plain
 * exponent add).  The magic-number rounding leaves n in the low mantissa bits of t, so (bits(t) << 23) == n << 23. */
DM_FN float dm_expneg(float x) {
    x = fmaxf(x, -86.0f);
    const float magic = 12582912.0f; /* 1.5 * 2^23 */
    float t = fmaf(x, 1.4426950216293335f, magic);
    float n = t - magic;
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    return dm_from_bits(dm_to_bits(e) + (dm_to_bits(t) << 23));
}

/* log1p(t) for t in [0, 1] */
DM_FN float dm_log1p01(float t) {
    int k = !(t < 0.4142135679721832f);
    float f = k ? fmaf(t, 0.5f, -0.5f) : t;          /* 1 + t = 2 (1 + f) on the upper branch */
    float q = 7.1513607744e-02f;
    q = fmaf(q, f, -1.1573007339e-01f);
    q = fmaf(q, f, 1.1661760853e-01f);
    q = fmaf(q, f, -1.2410829558e-01f);
    q = fmaf(q, f, 1.4249891856e-01f);
    q = fmaf(q, f, -1.6668487893e-01f);
    q = fmaf(q, f, 2.0000708849e-01f);
    q = fmaf(q, f, -2.4999988981e-01f);
    q = fmaf(q, f, 3.3333331185e-01f);
    float f2 = f * f;
    float res = fmaf(f2 * f, q, fmaf(-0.5f, f2, f));
    return res + (k ? 0.6931471805599453f : 0.0f);
}

/* x / 100 and x / fl32(sqrt(2)) as one multiplication by the rounded reciprocal (<= 1 ulp from the true quotient). */
DM_FN float dm_div100(float x) { return x * 0.009999999776482582f; }
DM_FN float dm_div_sqrt2(float x) { return x * 0.7071067690849304f; }

/* Softplus(beta=100, threshold=20) */
DM_FN float dm_softplus100(float z) {
    float y = z * 100.0f;
    if (y > 20.0f) return z;
    float t = dm_expneg(-fabsf(y));
    float s = fmaxf(y, 0.0f) + dm_log1p01(t);
    return dm_div100(s);
}

/* Softplus(beta=100, threshold=20) of the `f32x3` tracing arithmetic (trace_dtype 5; csrc/tile_engine_bf16s.h, oracle sdf_row_f32x3): the LEAN form.
 *     softplus(100 z) / 100 = max(z, 0) + t G(t),   t = exp(-100 |z|) = 2^(-n) P(fr),   u = 100 log2(e) |z| = n - fr,  n = rint(u),  fr in [-1/2, 1/2]
 * P = degree-5 fit of 2^fr (9.5e-8), G = degree-7 fit of ln(1 + t) / (100 t) on [0, 1] weighted by t (3.2e-10 in t G): the ABSOLUTE error of the
 * result stays near 1e-9 (asserted in tests/test_det_math.py) -- what the next layer's fp32 sums can see; dm_softplus100 spends more than twice the
 * operations on 1-2 ulp RELATIVE accuracy of a term that is at most 0.007.  |z| is clamped at 0.2 (u <= 28.9): beyond it t G(t) stays 2.1e-11, i.e. the
 * result is max(z, 0) + 2.1e-11 -- for z > 0.2 exactly z (the reference's threshold branch, idr.py:75), for z < -0.2 a positive number that is never
 * denormal and never below the engine's 2^-40 flush.  IEEE operations only (fma, add, min / max, one integer shift-add): eleven instructions per
 * activation on gfx950 with the polynomial chains two-wide (det_math_pk.h::dm2_softplus100_lean), against 27 for dm_softplus100. */
DM_FN float dm_softplus100_lean(float z) {
    const float za = fminf(fabsf(z), 0.2f);
    const float magic = 12582912.0f; /* 1.5 * 2^23 */
    const float tm = fmaf(za, -144.26950073242188f, magic);       /* low mantissa bits: -n */
    const float nf = tm - magic;                                   /* -n */
    const float fr = fmaf(za, -144.26950073242188f, -nf);         /* n - u */
    float p = 1.341362135e-03f;
    p = fmaf(p, fr, 9.671657346e-03f);
    p = fmaf(p, fr, 5.550281703e-02f);
    p = fmaf(p, fr, 2.402223945e-01f);
    p = fmaf(p, fr, 6.931472421e-01f);
    p = fmaf(p, fr, 1.0f);
    const float t = dm_from_bits(dm_to_bits(p) + (dm_to_bits(tm) << 23));
    float g = -6.453247624e-05f;
    g = fmaf(g, t, 3.608817351e-04f);
    g = fmaf(g, t, -9.533045813e-04f);
    g = fmaf(g, t, 1.676565735e-03f);
    g = fmaf(g, t, -2.407359425e-03f);
    g = fmaf(g, t, 3.317999188e-03f);
    g = fmaf(g, t, -4.998743068e-03f);
    g = fmaf(g, t, 9.999964386e-03f);
    return fmaf(g, t, fmaxf(z, 0.0f));
}

/* sin(a), cos(a) for |a| < ~1e4 (positional encoding arguments are < 64) */
DM_FN void dm_sincos(float a, float *s, float *c) {
    float j = dm_rint(a * 0.6366197466850281f);
    float r = fmaf(j, -1.5703125f, a);
    r = fmaf(j, -4.837512969970703125e-4f, r);
    r = fmaf(j, -7.549790126404332e-08f, r);
    float r2 = r * r;
    float ps = -1.9515295891e-4f;
    ps = fmaf(ps, r2, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    float sr = fmaf(ps * r2, r, r);
    float pc = 2.443315711809948e-5f;
    pc = fmaf(pc, r2, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    float cr = fmaf(pc, r2 * r2, fmaf(-0.5f, r2, 1.0f));
    int q = ((int)j) & 3;
    float ss = (q & 1) ? cr : sr;
    float cc = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    *s = ss;
    *c = cc;
}

/* sigmoid(100 z) as used by the softplus derivative (1 where 100 z > 20); tolerance domain. */
DM_FN float dm_sigmoid100(float z) {
    float y = z * 100.0f;
    if (y > 20.0f) return 1.0f;
    float t = dm_expneg(-fabsf(y));
    return (y >= 0.0f ? 1.0f : t) / (1.0f + t);
}

#endif
