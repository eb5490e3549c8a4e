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
#define GMMIV_ZERO_LLK (-745.1332191019412) // log 2^-1075: a likelihood below it is 0 in fp64 -- include/gmmiv.h, "degenerate inputs"
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
// Table-driven exp for the MFMA log-likelihood kernel: N = 2^GEXP_TAB_BITS entries 2^(j/N) in LDS (gexp_tab_init),
// x = (k/N) ln2 + r, |r| <= ln2/(2N).  With 2048 entries (16 KB) the degree-3 series is exact to r^4/24 = 3.5e-17, two
// fp64 operations fewer per exponential than 64 entries + degree 5 -- fp64 VALU work is what this kernel pays for next to
// its MFMAs (it does not overlap with them, not even from the other wave of the SIMD).  The argument reduction is a
// single fma (see gexp_tab_reduce; -DGEXP_TAB_TWO_STEP restores the Cody-Waite pair, high part with 26 trailing zeros).
// NO clamp: the callers' logits are >= GMMIV_PAD_LOGIT - |quadratic terms| (the packed model pads with -1e9, not with
// -1e300); below -2^31/(N/ln2) the float-to-int conversion saturates, the exponent stays hugely negative, the result is 0.
#ifndef GEXP_TAB_BITS
#define GEXP_TAB_BITS 11
#endif
#define GEXP_TAB_N (1 << GEXP_TAB_BITS)
#if GEXP_TAB_BITS == 11
#define GEXP_TAB_SCALE 2954.639443740597       /* N / ln2 */
#define GEXP_TAB_HI 0.00033845076904981397     /* ln2 / N, high part */
#define GEXP_TAB_LO 2.7079718246904592e-12
#define GEXP_TAB_LN2N 0.00033845077175778578    /* ln2 / N rounded to nearest */
#elif GEXP_TAB_BITS == 6
#define GEXP_TAB_SCALE 92.33248261689366
#define GEXP_TAB_HI 0.010830424609594047
#define GEXP_TAB_LO 8.66550983900947e-11
#define GEXP_TAB_LN2N 0.010830424696249145
#else
#error "GEXP_TAB_BITS must be 6 or 11"
#endif
__device__ __forceinline__ void gexp_tab_init(double *tab, int tid, int nthreads)
{
    for (int j = tid; j < GEXP_TAB_N; j += nthreads) tab[j] = exp2((double)j * (1.0 / GEXP_TAB_N));
}
// exp(r) - 1 for the reduced argument
__device__ __forceinline__ double gexp_tab_poly(double r)
{
#if GEXP_TAB_BITS == 11
    double p = 1.6666666666666666e-01;
#else
    double p = 8.333333333333333e-03;
    p = __builtin_fma(p, r, 4.1666666666666664e-02);
    p = __builtin_fma(p, r, 1.6666666666666666e-01);
#endif
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    return p * r;
}
// phase 1 -> (ki, r); a caller that also needs the binary exponent of exp(x) takes ki >> GEXP_TAB_BITS
__device__ __forceinline__ void gexp_tab_reduce(double x, int &ki, double &r)
{
    const double k = __builtin_rint(x * GEXP_TAB_SCALE);
#ifdef GEXP_TAB_TWO_STEP
    r = __builtin_fma(k, -GEXP_TAB_HI, x);
    r = __builtin_fma(k, -GEXP_TAB_LO, r);
#else
    // ONE fma: the representation error of ln2/N (<= 2^-53 ln2/N = 3.8e-20) times |k| = 2955 |x| puts <= 1.1e-16 |x|
    // (absolute) on r, i.e. 1e-12 on exp(x) at |x| = 1e4 -- the size of the rounding error the expanded logit itself
    // carries, and one fp64 instruction less per exponential
    r = __builtin_fma(k, -GEXP_TAB_LN2N, x);
#endif
    ki = (int)k;
}
// phase 2 -> exp(x) 2^-E
__device__ __forceinline__ double gexp_tab_finish(int ki, double r, int E, const double *tab)
{
    const double tj = tab[ki & (GEXP_TAB_N - 1)];
    // (v_ldexp_f64 beats inserting 2^n into the exponent field with integer instructions: measured +0.4 % kernel time)
    return __builtin_ldexp(__builtin_fma(tj, gexp_tab_poly(r), tj), (ki >> GEXP_TAB_BITS) - E);
}
__device__ __forceinline__ double gexp_tab_scaled(double x, int E, const double *tab)
{
    int ki;
    double r;
    gexp_tab_reduce(x, ki, r);
    return gexp_tab_finish(ki, r, E, tab);
}
// the binary exponent gexp_tab_scaled assigns to exp(x): same rounding
__device__ __forceinline__ int gexp_tab_exponent(double x)
{
    return ((int)__builtin_rint(x * GEXP_TAB_SCALE)) >> GEXP_TAB_BITS;
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

__device__ __forceinline__ int row_min_i32(int v)
{
    int o;
    o = dpp_i32<0x128>(v); v = o < v ? o : v;
    o = dpp_i32<0x124>(v); v = o < v ? o : v;
    o = dpp_i32<0x122>(v); v = o < v ? o : v;
    o = dpp_i32<0x121>(v); v = o < v ? o : v;
    return v;
}

// Wave-wide arg-max of (value, index) pairs, larger value first, smaller index on ties; every lane gets the winner.
// DPP row rotations inside the 16-lane rows (plain VALU instructions), then the 4 row results through v_readlane and scalar
// compares -- a ds_bpermute butterfly is 6 dependent LDS-pipe round trips of 3 permutes each (~2000 cycles per arg-max).
__device__ __forceinline__ void wave_argmax_f64(double &v, int &c)
{
#define GMMIV_ARGMAX_STEP(CTRL)                                                                    \
    {                                                                                              \
        const int ohi = dpp_i32<CTRL>(__double2hiint(v)), olo = dpp_i32<CTRL>(__double2loint(v)); \
        const int oc = dpp_i32<CTRL>(c);                                                           \
        const double ov = __hiloint2double(ohi, olo);                                              \
        if (ov > v || (ov == v && oc < c)) { v = ov; c = oc; }                                     \
    }
    GMMIV_ARGMAX_STEP(0x128)
    GMMIV_ARGMAX_STEP(0x124)
    GMMIV_ARGMAX_STEP(0x122)
    GMMIV_ARGMAX_STEP(0x121)
#undef GMMIV_ARGMAX_STEP
    // row 0's own result takes part through lane 0
    const double v0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const int c0 = __builtin_amdgcn_readlane(c, 0);
    double wv = v0;
    int wc = c0;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const double ov = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16 * r), __builtin_amdgcn_readlane(__double2loint(v), 16 * r));
        const int oc = __builtin_amdgcn_readlane(c, 16 * r);
        if (ov > wv || (ov == wv && oc < wc)) { wv = ov; wc = oc; }
    }
    v = wv;
    c = wc;
}

// DPP versions of the wave reductions (row rotations + 4 v_readlane): no LDS-pipe round trips.  Every lane gets the result.
__device__ __forceinline__ double dpp_f64_0x128(double v) { return __hiloint2double(dpp_i32<0x128>(__double2hiint(v)), dpp_i32<0x128>(__double2loint(v))); }
__device__ __forceinline__ double dpp_f64_0x124(double v) { return __hiloint2double(dpp_i32<0x124>(__double2hiint(v)), dpp_i32<0x124>(__double2loint(v))); }
__device__ __forceinline__ double dpp_f64_0x122(double v) { return __hiloint2double(dpp_i32<0x122>(__double2hiint(v)), dpp_i32<0x122>(__double2loint(v))); }
__device__ __forceinline__ double dpp_f64_0x121(double v) { return __hiloint2double(dpp_i32<0x121>(__double2hiint(v)), dpp_i32<0x121>(__double2loint(v))); }
__device__ __forceinline__ double readlane_f64u(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v)
{
    v += dpp_f64_0x128(v); v += dpp_f64_0x124(v); v += dpp_f64_0x122(v); v += dpp_f64_0x121(v);
    return (readlane_f64u(v, 0) + readlane_f64u(v, 16)) + (readlane_f64u(v, 32) + readlane_f64u(v, 48));
}
// the same per HALF wave (lanes 0..31 / 32..63 each get their own half's result); hi = lane >= 32
__device__ __forceinline__ double half_sum_f64_dpp(double v, bool hi)
{
    v += dpp_f64_0x128(v); v += dpp_f64_0x124(v); v += dpp_f64_0x122(v); v += dpp_f64_0x121(v);
    const double a = readlane_f64u(v, 0) + readlane_f64u(v, 16), b = readlane_f64u(v, 32) + readlane_f64u(v, 48);
    return hi ? b : a;
}
__device__ __forceinline__ double half_max_f64_dpp(double v, bool hi)
{
    v = fmax(v, dpp_f64_0x128(v)); v = fmax(v, dpp_f64_0x124(v)); v = fmax(v, dpp_f64_0x122(v)); v = fmax(v, dpp_f64_0x121(v));
    const double a = fmax(readlane_f64u(v, 0), readlane_f64u(v, 16)), b = fmax(readlane_f64u(v, 32), readlane_f64u(v, 48));
    return hi ? b : a;
}
__device__ __forceinline__ double half_min_f64_dpp(double v, bool hi)
{
    v = fmin(v, dpp_f64_0x128(v)); v = fmin(v, dpp_f64_0x124(v)); v = fmin(v, dpp_f64_0x122(v)); v = fmin(v, dpp_f64_0x121(v));
    const double a = fmin(readlane_f64u(v, 0), readlane_f64u(v, 16)), b = fmin(readlane_f64u(v, 32), readlane_f64u(v, 48));
    return hi ? b : a;
}
__device__ __forceinline__ double wave_max_f64_dpp(double v)
{
    v = fmax(v, dpp_f64_0x128(v)); v = fmax(v, dpp_f64_0x124(v)); v = fmax(v, dpp_f64_0x122(v)); v = fmax(v, dpp_f64_0x121(v));
    return fmax(fmax(readlane_f64u(v, 0), readlane_f64u(v, 16)), fmax(readlane_f64u(v, 32), readlane_f64u(v, 48)));
}
__device__ __forceinline__ double wave_min_f64_dpp(double v)
{
    v = fmin(v, dpp_f64_0x128(v)); v = fmin(v, dpp_f64_0x124(v)); v = fmin(v, dpp_f64_0x122(v)); v = fmin(v, dpp_f64_0x121(v));
    return fmin(fmin(readlane_f64u(v, 0), readlane_f64u(v, 16)), fmin(readlane_f64u(v, 32), readlane_f64u(v, 48)));
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

// floor(e / D) for e < 2^16 by reciprocal multiplication: magic = 2^32 / D + 1 (host side: gmmiv_div_magic).  vectSize 1 has no
// 32-bit magic (2^32 + 1 wraps to 1 and every quotient came out 0 -- the statistics of a ONE-dimensional model were garbage until
// round 6; tests/golden's EnergyDetector case, C = 2, D = 1, found it): magic 0 stands for "divide by one".
__device__ __forceinline__ int div_by_magic(unsigned e, unsigned magic) { return magic ? (int)__umulhi(e, magic) : (int)e; }
static inline unsigned gmmiv_div_magic(int D) { return D == 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)D + 1); }

// DEGENERATE INPUTS, kind (1) (include/gmmiv.h): a feature value that is NaN, infinite or beyond 1e18 in magnitude makes its frame a
// zero-likelihood frame.  Every kernel that reads features reads them through feat_sane: such a value is READ AS 1e10 -- finite, so no
// 0 x NaN can poison a statistic, and far enough from any mean that every logit of the frame lies near -0.5e20 / variance: the frame
// then IS a zero-likelihood frame of kind (2) for every kernel, on the device, with no host decision (no flag read back, no
// compaction, no synchronisation).  Holds for variances in (1e-17, 1e17); 1e10 is exact in float.  Usable values pass unchanged
// (one compare + select per element loaded; in the MFMA log-likelihood kernel that is once per frame and workgroup, outside its loop).
#define GMMIV_UNUSABLE_BOUND 1e18
#define GMMIV_UNUSABLE_READ_AS 1e10
#ifndef GMMIV_FEAT_SANE_OFF   // (tools/feat_sane_ab.sh builds a second library without the select to price it; never defined in the product build)
__device__ __forceinline__ float feat_sane(float v) { return __builtin_fabsf(v) <= (float)GMMIV_UNUSABLE_BOUND ? v : (float)GMMIV_UNUSABLE_READ_AS; }
__device__ __forceinline__ double feat_sane(double v) { return __builtin_fabs(v) <= GMMIV_UNUSABLE_BOUND ? v : GMMIV_UNUSABLE_READ_AS; }
#else
__device__ __forceinline__ float feat_sane(float v) { return v; }
__device__ __forceinline__ double feat_sane(double v) { return v; }
#endif
template <typename T> struct feat_load;
template <> struct feat_load<float> {
    static __device__ __forceinline__ double get(const void *p, long i) { return (double)feat_sane(((const float *)p)[i]); }
    static __device__ __forceinline__ double raw(const void *p, long i) { return (double)((const float *)p)[i]; } // FrameAccGD: not screened (gmmiv.h)
};
template <> struct feat_load<double> {
    static __device__ __forceinline__ double get(const void *p, long i) { return feat_sane(((const double *)p)[i]); }
    static __device__ __forceinline__ double raw(const void *p, long i) { return ((const double *)p)[i]; }
};
