// What does v_mfma_f32_16x16x32_bf16 compute, bit for bit?  D = C + A B^T with A, B bf16 (products exact in fp32) -- in which order, with how many roundings?
// Runs the instruction on random tiles and compares every output with host models built on exact integer arithmetic.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_model mfma_model.hip ; run: ./mfma_model [spread]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void k_mfma(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C, float* __restrict__ D) {
    const int lane = threadIdx.x, t = blockIdx.x, r = lane & 15, q = lane >> 4;
    const uint4 a = *(const uint4*)(A + ((size_t)t * 16 + r) * 32 + 8 * q);
    const uint4 b = *(const uint4*)(B + ((size_t)t * 16 + r) * 32 + 8 * q);
    f32x4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[((size_t)t * 16 + 4 * q + i) * 16 + r];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((size_t)t * 16 + 4 * q + i) * 16 + r] = c[i];
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

// exact value as (integer mantissa) * 2^E0 with a common exponent E0 (all inputs are chosen so that everything fits 128 bits)
typedef __int128 i128;
static const int E0 = -100;
static i128 to_fix(double v) {                 // v is exactly representable with <= 53 bits; |v| < 2^60, multiple of 2^E0... (checked by range of inputs)
    if (v == 0) return 0;
    int e; double m = frexp(v, &e);            // v = m 2^e, 0.5 <= |m| < 1
    long long mi = (long long)ldexp(m, 53);    // 53-bit integer
    int sh = e - 53 - E0;
    return sh >= 0 ? (i128)mi << sh : (i128)mi >> (-sh);
}
// round an exact fixed value to fp32: mode 0 = nearest even, 1 = toward zero
static float round_fix(i128 x, int mode) {
    if (x == 0) return 0.0f;
    const bool neg = x < 0;
    unsigned __int128 u = neg ? (unsigned __int128)(-x) : (unsigned __int128)x;
    int hb = 127; while (!((u >> hb) & 1)) --hb;          // highest bit
    int drop = hb - 23;                                    // bits below the 24-bit mantissa
    unsigned __int128 mant = drop > 0 ? u >> drop : u << (-drop);
    if (drop > 0 && mode == 0) {
        const unsigned __int128 rem = u & (((unsigned __int128)1 << drop) - 1), half = (unsigned __int128)1 << (drop - 1);
        if (rem > half || (rem == half && (mant & 1))) ++mant;
    }
    double v = ldexp((double)(uint64_t)mant, drop + E0);
    return (float)(neg ? -v : v);                          // mant <= 2^24: exact
}
static i128 fix_of_float(float f) { return to_fix((double)f); }

int main(int argc, char** argv) {
    const int spread = argc > 1 ? atoi(argv[1]) : 6;       // exponents of the inputs drawn from [-spread, spread]
    const int T = 4096;
    std::vector<uint16_t> A((size_t)T * 16 * 32), B(A.size());
    std::vector<float> C((size_t)T * 256), D(C.size());
    srand(7);
    auto rnd = [&]() { return rand() / (double)RAND_MAX; };
    auto val = [&]() { return (float)((rnd() * 2 - 1) * ldexp(1.0, (int)(rnd() * (2 * spread + 1)) - spread)); };
    for (auto& v : A) v = f2bf(val());
    for (auto& v : B) v = f2bf(val());
    for (auto& v : C) v = val() * (rand() % 4 == 0 ? 0.0f : 1.0f);
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    // models
    enum { M_EXACT, M_EXACT_RZ, M_G8, M_G8_RZ, M_G4, M_G16, M_CHAIN, M_G8_INTER, M_G16_INTER, M_PAIR, NM };
    const char* names[NM] = {"one rounding of c + all 32 products (RNE)", "same, toward zero", "4 sequential groups k=8g..8g+7 (RNE each)", "same, toward zero",
                             "8 sequential groups of 4", "2 sequential groups of 16", "32-step fmaf chain k ascending", "4 groups {k: k%4==g... interleaved by 4}",
                             "2 groups: lane-group pairs {q0,q2} then {q1,q3}", "groups of 2"};
    long long miss[NM] = {0};
    const int NF = 20, NV = 4;
    static long long missA[20][4];
    long long total = 0; int shown = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                i128 p[32];
                for (int k = 0; k < 32; ++k) p[k] = to_fix((double)bf2f(A[((size_t)t * 16 + i) * 32 + k]) * (double)bf2f(B[((size_t)t * 16 + j) * 32 + k]));
                const float c = C[((size_t)t * 16 + i) * 16 + j], d = D[((size_t)t * 16 + i) * 16 + j];
                float m[NM];
                { i128 s = fix_of_float(c); for (int k = 0; k < 32; ++k) s += p[k]; m[M_EXACT] = round_fix(s, 0); m[M_EXACT_RZ] = round_fix(s, 1); }
                auto grouped = [&](int G, int mode) { float acc = c; for (int g = 0; g < 32 / G; ++g) { i128 s = fix_of_float(acc); for (int k = g * G; k < (g + 1) * G; ++k) s += p[k]; acc = round_fix(s, mode); } return acc; };
                m[M_G8] = grouped(8, 0); m[M_G8_RZ] = grouped(8, 1); m[M_G4] = grouped(4, 0); m[M_G16] = grouped(16, 0); m[M_CHAIN] = grouped(1, 0); m[M_PAIR] = grouped(2, 0);
                { float acc = c; for (int g = 0; g < 4; ++g) { i128 s = fix_of_float(acc); for (int k = 0; k < 32; ++k) if (((k >> 1) & 3) == g) s += p[k]; acc = round_fix(s, 0); } m[M_G8_INTER] = acc; }
                { float acc = c; for (int g = 0; g < 2; ++g) { i128 s = fix_of_float(acc); for (int k = 0; k < 32; ++k) if ((((k >> 3) & 1)) == g) s += p[k]; acc = round_fix(s, 0); } m[M_G16_INTER] = acc; }
                for (int h = 0; h < NM; ++h) miss[h] += memcmp(&m[h], &d, 4) != 0;
                if (memcmp(&m[M_G8], &d, 4) != 0 && shown < 12) {
                    ++shown;
                    int32_t ud, um; memcpy(&ud, &d, 4); memcpy(&um, &m[M_G8], 4);
                    printf("  case t=%d i=%d j=%d: c=%.9g hw=%.9g model=%.9g (ulp diff %d)\n", t, i, j, c, d, m[M_G8], ud - um);
                    float acc = c;
                    for (int g = 0; g < 4; ++g) { i128 s2 = 0; for (int k = 8 * g; k < 8 * g + 8; ++k) s2 += p[k]; i128 tot = fix_of_float(acc) + s2;
                        double gs = (double)(long long)(s2 >> 40) * ldexp(1.0, 40 + E0), ex = (double)(long long)(tot >> 40) * ldexp(1.0, 40 + E0);
                        float na = round_fix(tot, 0); printf("     group %d: acc %.9g + sum %.12g = %.12g -> %.9g ; products:", g, acc, gs, ex, na);
                        for (int k = 8 * g; k < 8 * g + 8; ++k) printf(" %.4g", (double)(long long)(p[k] >> 40) * ldexp(1.0, 40 + E0)); printf("\n"); acc = na; }
                }
                // aligned models: per group of 8, every term (accumulator included) is aligned to the largest exponent among them and cut below 2^(emax - F)
                for (int F = 22; F < 22 + NF; ++F)
                    for (int var = 0; var < NV; ++var) {
                        float acc = c;
                        for (int g = 0; g < 4; ++g) {
                            i128 terms[9]; terms[0] = fix_of_float(acc);
                            for (int k = 0; k < 8; ++k) terms[1 + k] = p[8 * g + k];
                            int emax = -1000;
                            if (terms[0] != 0) { i128 x = terms[0] < 0 ? -terms[0] : terms[0]; int hb = 126; while (!((x >> hb) & 1)) --hb; emax = hb; }
                            for (int k = 0; k < 8; ++k) {                  // a product's exponent: ea + eb (mantissa product in [1, 4), not normalised)
                                const float fa = bf2f(A[((size_t)t * 16 + i) * 32 + 8 * g + k]), fb = bf2f(B[((size_t)t * 16 + j) * 32 + 8 * g + k]);
                                if (fa == 0 || fb == 0) continue;
                                int ea, eb; frexp(fa, &ea); frexp(fb, &eb);
                                const int hb = (ea - 1) + (eb - 1) - E0;    // bit position of 2^(ea + eb) in the fixed format
                                if (hb > emax) emax = hb;
                            }
                            i128 ssum = 0;
                            const int cut = emax - F;                      // bits below this position are dropped
                            for (int u = 0; u < 9; ++u) {
                                i128 x = terms[u];
                                if (cut > 0) {
                                    if (var & 1) { x = x >> cut; x = x << cut; }                 // floor (two's complement truncation)
                                    else { const bool ng = x < 0; i128 ax = ng ? -x : x; ax = (ax >> cut) << cut; x = ng ? -ax : ax; }   // toward zero (sign-magnitude)
                                }
                                ssum += x;
                            }
                            acc = round_fix(ssum, (var & 2) ? 1 : 0);
                        }
                        missA[F - 22][var] += memcmp(&acc, &d, 4) != 0;
                    }
                ++total;
            }
    printf("spread 2^+-%d, %lld outputs:\n", spread, total);
    for (int h = 0; h < NM; ++h) printf("  %-70s mismatches %lld (%.4f)\n", names[h], miss[h], miss[h] / (double)total);
    const char* vn[4] = {"terms cut toward zero, final RNE", "terms floored, final RNE", "terms cut toward zero, final RZ", "terms floored, final RZ"};
    for (int var = 0; var < NV; ++var) { printf("  aligned, %s: F -> mismatches:", vn[var]); for (int F = 0; F < NF; ++F) printf(" %d:%lld", F + 22, missA[F][var]); printf("\n"); }
    // a few raw cases for the best model
    int best = 0; for (int h = 1; h < NM; ++h) if (miss[h] < miss[best]) best = h;
    printf("best: %s\n", names[best]);
    return 0;
}
