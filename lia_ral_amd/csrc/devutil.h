// devutil.h -- device-side helpers shared by the gfx950 kernels (wave64, fp64 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: D[16x16] += A[16x4] * B[4x16], one f64 per lane for A and B.
//   A: lane l holds A[i = l&15][k = l>>4]      B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r holds D[row = (l>>4) + 4r][col = l&15]   (f64 layout, NOT the f32 one)
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define GMMIV_NEG_BIG (-1.0e300)
#define GMMIV_PAD_LOGIT (-1.0e9) // constant term of padded / zero-weight Gaussians in the packed MFMA model

// exp(x) for x <= ~700, branch-free, no special cases: arguments below -750 give ~0.
// n = rint(x log2 e); r = x - n ln2 (Cody-Waite, two steps); exp(r) by a degree-13 Taylor
// polynomial in Horner form (|r| <= 0.347 -> truncation 4e-18); result scaled with ldexp.
__device__ __forceinline__ double gexp(double x)
{
    x = fmax(x, -750.0);
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;              // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);   // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);  // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);  // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);   // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);  // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// Table-driven variant: exp(x) = 2^n * T[j] * (1 + p(r)), k = rint(x 32/ln2), j = k & 31, n = k >> 5,
// r = x - k ln2/32 (|r| <= 0.0109), p = r + r^2/2 + ... + r^7/5040 (truncation 3e-18).  `tab` is the
// 32-entry table 2^(j/32) in LDS (filled by gexp_table_init).  13 fp64 ops instead of 20: on gfx950
// every fp64 VALU op takes issue time away from the fp64 MFMA pipe, so this is MFMA throughput.
__device__ __forceinline__ void gexp_table_init(double *tab, int tid)
{
    if (tid < 32) tab[tid] = exp2((double)tid * 0.03125);
}
__device__ __forceinline__ double gexp_t(double x, const double *tab)
{
    x = fmax(x, -750.0);
    const double k = __builtin_rint(x * 46.16624130844683);
    double r = __builtin_fma(k, -0.021660849219188094, x);
    r = __builtin_fma(k, -1.733101967801894e-10, r);
    const int ki = (int)k;
    const double tj = tab[ki & 31];
    double p = 1.984126984126984e-04;               // 1/5040
    p = __builtin_fma(p, r, 1.388888888888889e-03);  // 1/720
    p = __builtin_fma(p, r, 8.333333333333333e-03);  // 1/120
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/24
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/6
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    return __builtin_ldexp(__builtin_fma(tj, p, tj), ki >> 5);
}

// exp(x) for x <= ~0 with a 64-entry table of 2^(j/64) (tab64[j], j < 64): |r| <= ln2/128, so the
// degree-5 series is exact to 4e-17; ONE fma reduces the argument -- the representation error of
// ln2/64 (<= 2^-60) times |k| = 92 |x| is below 3e-15 |x|/36 relative to exp(x), i.e. invisible in a
// posterior that small.  15 VALU instructions.
__device__ __forceinline__ void gexp_table64_init(double *tab, int tid)
{
    if (tid < 64) tab[tid] = exp2((double)tid * 0.015625);
}
__device__ __forceinline__ double gexp_t64(double x, const double *tab)
{
    x = fmax(x, -750.0);
    const double k = __builtin_rint(x * 92.33248261689366);
    const double r = __builtin_fma(k, -0.010830424696249145, x);
    const int ki = (int)k;
    const double tj = tab[ki & 63];
    double p = 8.333333333333333e-03;               // 1/120
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/24
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/6
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    return __builtin_ldexp(__builtin_fma(tj, p, tj), ki >> 6);
}

// exp(x) * 2^-E with the binary exponent applied in ONE ldexp, so x may be far outside exp()'s range
// (logits of -5000 against a running exponent E of -7200 are fine).  x >= -4e7 (int32 range of k).
__device__ __forceinline__ double gexp_scaled(double x, int E, const double *tab)
{
    x = fmax(x, -4.0e7);
    const double k = __builtin_rint(x * 46.16624130844683);
    double r = __builtin_fma(k, -0.021660849219188094, x);
    r = __builtin_fma(k, -1.733101967801894e-10, r);
    const int ki = (int)k;
    const double tj = tab[ki & 31];
    double p = 1.984126984126984e-04;
    p = __builtin_fma(p, r, 1.388888888888889e-03);
    p = __builtin_fma(p, r, 8.333333333333333e-03);
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    int n = (ki >> 5) - E;
    n = n < -2000 ? -2000 : n;
    return __builtin_ldexp(__builtin_fma(tj, p, tj), n);
}
// gexp_scaled with the 64-entry table (gexp_table64_init): |r| <= ln2/128, degree-5 series (4e-17), two-step
// argument reduction kept because x is a raw logit (|x| up to 1e4 and more).  The binary exponent is ki >> 6.
// NO clamp: the callers' logits are >= GMMIV_PAD_LOGIT - |quadratic terms| (the packed model pads with -1e9, not with
// -1e300); below -2.3e7 the float-to-int conversion saturates, the exponent stays hugely negative and the result is 0.
__device__ __forceinline__ double gexp_scaled64(double x, int E, const double *tab)
{
    const double k = __builtin_rint(x * 92.33248261689366);
    double r = __builtin_fma(k, -0.010830424609594047, x);   // ln2/64 high part (trailing bits zero)
    r = __builtin_fma(k, -8.665509839009470e-11, r);         // ln2/64 low part
    const int ki = (int)k;
    const double tj = tab[ki & 63];
    double p = 8.333333333333333e-03;
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    return __builtin_ldexp(__builtin_fma(tj, p, tj), (ki >> 6) - E);
}
// gexp_scaled64 in two halves, so that a caller that also needs the binary exponent of exp(x) (ki >> 6)
// gets it from the argument reduction it has to do anyway: phase 1 -> (ki, r), phase 2 -> exp(x) 2^-E
__device__ __forceinline__ void gexp64_reduce(double x, int &ki, double &r)
{
    const double k = __builtin_rint(x * 92.33248261689366);
    r = __builtin_fma(k, -0.010830424609594047, x);
    r = __builtin_fma(k, -8.665509839009470e-11, r);
    ki = (int)k;
}
__device__ __forceinline__ double gexp64_finish(int ki, double r, int E, const double *tab)
{
    const double tj = tab[ki & 63];
    double p = 8.333333333333333e-03;
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    return __builtin_ldexp(__builtin_fma(tj, p, tj), (ki >> 6) - E);
}
// the binary exponent gexp_scaled64 assigns to exp(x): same clamp, same rounding
__device__ __forceinline__ int gexp_exponent64(double x)
{
    return ((int)__builtin_rint(x * 92.33248261689366)) >> 6;
}
// floor(x log2 e) as used by gexp_scaled (the binary exponent of exp(x))
__device__ __forceinline__ int gexp_exponent(double x)
{
    x = fmax(x, -4.0e7);
    return ((int)__builtin_rint(x * 46.16624130844683)) >> 5;
}

// all-reduce over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1): every lane gets the result
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row_max_i32(int v)
{
    int o;
    o = dpp_i32<0x128>(v); v = o > v ? o : v;
    o = dpp_i32<0x124>(v); v = o > v ? o : v;
    o = dpp_i32<0x122>(v); v = o > v ? o : v;
    o = dpp_i32<0x121>(v); v = o > v ? o : v;
    return v;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor_f64(v, o);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, shfl_xor_f64(v, o));
    return v;
}

// XOR swizzle of the column index inside one row of the frame tile held in LDS (row length a
// multiple of 32 doubles).  g(t) = ((2t) ^ (16 (t&1))) & 31 makes both access patterns of the
// statistics kernel conflict-free for ds_read_b64 (64 banks x 4 B, two 32-lane groups):
//   (L) 16 consecutive rows x 2 adjacent columns   (MFMA A operand of the logit GEMM)
//   (S) 2 consecutive rows x 16 consecutive columns (MFMA B operand of the statistics GEMM)
__device__ __forceinline__ int xswz(int t) { return ((t << 1) ^ ((t & 1) << 4)) & 31; }

// Additive variant used by the statistics kernel: element (t, col) of the frame tile lives at
// t * RLp + xrot(t) + col with RLp = RL + 32 and RLp % 32 == 0.  Same two conflict-free patterns
// (rows t, t+1 sit 16 bank-pairs apart; 16 consecutive rows cover all 32 bank-pairs two by two),
// but the column enters the address as a plain sum, so every LDS read of the inner loops is
// "lane base + compile-time immediate" and costs no VALU address arithmetic.
__device__ __forceinline__ int xrot(int t) { return ((t & 1) << 4) + (((t >> 1) & 7) << 1); }

template <typename T> struct feat_load;
template <> struct feat_load<float> {
    static __device__ __forceinline__ double get(const void *p, long i) { return (double)((const float *)p)[i]; }
};
template <> struct feat_load<double> {
    static __device__ __forceinline__ double get(const void *p, long i) { return ((const double *)p)[i]; }
};
