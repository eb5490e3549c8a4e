// em_fused.hip -- single-pass EM statistics: every frame x Gaussian logit is computed ONCE.
//
// The two-kernel path (k_llk_mfma, then k_stats_mfma) evaluates every logit twice because the
// posterior of (frame t, Gaussian c) needs log sum_c' exp(z_tc') over ALL Gaussians, while a
// workgroup can only keep accumulators for 128 Gaussians.  Here the 16 workgroups that own the 16
// Gaussian groups of a model form a TEAM that walks the same frame tiles in step:
//   phase A (tile k)    logits z for 32 frames x 128 Gaussians (31 MFMAs / 16x16), kept in registers;
//                       per-frame partial (max, sum exp) over the workgroup's Gaussians (DPP row
//                       reductions + one LDS pass), published to the team through global memory
//   phase C (tile k-1)  gather the 16 partials of the tile -> lse_t; gamma = exp(z - lse_t) from the
//                       registers of phase A; statistics MFMAs exactly as in k_stats_mfma
// so 63 instead of 93 MFMAs per 256 pairs (47 instead of 77 without the x^2 statistics).
// The hand-off uses self-validating 8-byte words like the LL protocol of RCCL: every word carries
// 32 bits of payload and the 32-bit sequence number (tile + 1) of the tile it belongs to, written
// and read with agent-scope relaxed 8-byte atomics (sc1, L1-bypassing).  A reader simply re-loads a
// word until its sequence number is the one it wants -- no separate flag, no store drain, no fence
// (8-byte atomics cannot tear, and every word validates itself).  Per (frame, group): the partial
// maximum as f32 (it is only a reference exponent) and the f64 partial sum in two words.
// One tile of look-ahead hides the hand-off latency: the loads of tile k-1's partials are issued
// before the logit MFMAs of tile k and checked after them.  Correctness does not depend on
// placement; liveness needs all workgroups of the grid resident (grid <= CUs x resident workgroups
// per CU) -- every spin is bounded and reports through `err`, and the host then falls back.
#include "devutil.h"
#include "lds_attr.h"
#include "gmm_kernels.h"

#define EMF_NBUF 4      // hand-off slots (>= 2 * lookahead + 2 with lookahead 1)
#define EMF_FT 32       // frames per tile
#define EMF_MAXGRP 16   // Gaussian groups per team (one DPP row of the gathering wave)

template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1): every lane gets the result
__device__ __forceinline__ float row_max_f32(float v)
{
    v = fmaxf(v, dpp_f32<0x128>(v));
    v = fmaxf(v, dpp_f32<0x124>(v));
    v = fmaxf(v, dpp_f32<0x122>(v));
    v = fmaxf(v, dpp_f32<0x121>(v));
    return v;
}
__device__ __forceinline__ double row_sum_f64(double v)
{
    v += dpp_f64<0x128>(v);
    v += dpp_f64<0x124>(v);
    v += dpp_f64<0x122>(v);
    v += dpp_f64<0x121>(v);
    return v;
}

typedef unsigned long long u64;
__device__ __forceinline__ void ll_store(u64 *p, unsigned data, unsigned seq)
{
    __hip_atomic_store(p, ((u64)seq << 32) | (u64)data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ll_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// SQ: also accumulate sum gamma x^2 (EM); without it only N and F (Baum-Welch statistics).
// mode 0: segment sg -> partial block out0[sg] (summed later); mode 1: segment sg = utterance, its
// statistics go straight to N = out0[sg][C], F = out1[sg][C*D].  A team walks segments team,
// team + nteams, ...; the tile counter of the exchange keeps running across segments (kbase).
template <int KS, typename XT, bool SQ>
__global__ __launch_bounds__(512, 2) void k_em_fused(const void *__restrict__ x, long ldx, int D, int C, const double *__restrict__ Pt,
                                                     int nct, double lse_shift, const long *__restrict__ seg_begin, int nseg,
                                                     int nteams, int ngrp, int mode, double *__restrict__ out0,
                                                     double *__restrict__ out1, double *__restrict__ lse_out,
                                                     u64 *__restrict__ slots, unsigned *__restrict__ err, unsigned magicD,
                                                     int dbg)
{
    // dbg (timing experiments, wrong results): 1 = no inter-workgroup exchange
    constexpr int NR = 2 * KS + 2;
    constexpr int Dp = 4 * KS;
    constexpr int RL = ((Dp + 2 + 31) / 32) * 32;
    constexpr int RLp = RL + 32;
    constexpr int JT = RL / 16;
    constexpr int FT = EMF_FT;
    constexpr int NT = 512;
    constexpr int NLD = (FT * Dp + NT - 1) / NT;
    constexpr int NXB = 3; // frame-tile buffers: tile k (phase A), k-1 (phase C), k+1 (being staged)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xb = (double *)smem;                   // NXB x FT x RLp
    double *red = xb + NXB * FT * RLp;             // 8 waves x FT x (m, s)
    double *lse_t = red + 8 * FT * 2;              // FT
    double *etab = lse_t + FT;                     // 32
    gexp_table_init(etab, threadIdx.x);

    // XCD-aware decode (see k_stats_mfma): the workgroups of a team share an XCD when b % 8 is the XCD
    const int b = blockIdx.x;
    const int xcd = b & 7, rest = b >> 3;
    const int grp = rest % ngrp;
    const int team = (rest / ngrp) * 8 + xcd;
    if (team >= nteams) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int ct = grp * 8 + wave;
    const bool active = ct < nct;

    double Pr[2 * KS + 1];
#pragma unroll
    for (int s = 0; s < 2 * KS + 1; ++s) {
        const int row = s < 2 * KS ? s : 2 * KS + 1;
        Pr[s] = active ? Pt[((size_t)ct * NR + row) * 64 + lane] : 0.0;
    }
    long f0 = 0, f1 = 0;
    int ntiles = 0, kbase = 0;
    d4 S[JT], S2[SQ ? JT : 1];

    // staging plan (see k_stats_mfma)
    XT stg[NLD];
    unsigned pk[NLD]; // rows are contiguous (ldx == D): the host falls back to the two-kernel path otherwise
    const int npad = FT * (RL - D);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NT * i;
        const int fr = div_by_magic((unsigned)e, magicD), d = e - fr * D;
        pk[i] = fr < FT ? ((unsigned)fr << 16) | (unsigned)((fr * RLp + xrot(fr) + d) * 8) : 0xffff0000u;
    }
    auto load_tile = [&](int tl) {
        const long fb = f0 + (long)tl * FT;
        const long rem = f1 - fb;
        const unsigned lim = rem >= FT ? (unsigned)FT << 16 : (rem > 0 ? (unsigned)rem << 16 : 0u);
        const XT *xt = (const XT *)x + fb * ldx;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            XT v = 0;
            if (pk[i] < lim) v = xt[tid + NT * i];
            stg[i] = v;
        }
    };
    auto write_tile = [&](int tl) {
        double *dst = xb + (tl % NXB) * FT * RLp;
        const long rem = f1 - (f0 + (long)tl * FT);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (pk[i] < ((unsigned)FT << 16)) *(double *)((char *)dst + (pk[i] & 0xffffu)) = (double)feat_sane(stg[i]);
        for (int e = tid; e < npad; e += NT) { // pad columns: const 1 at Dp for existing rows, zeros elsewhere
            const int fr = e / (RL - D), d = D + (e - fr * (RL - D));
            dst[fr * RLp + xrot(fr) + d] = (d == Dp && fr < rem) ? 1.0 : 0.0;
        }
    };

    const int offL = i16 * RLp + xrot(i16) + q;
    const int offS = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + i16;
    u64 *my_slots = slots + (size_t)team * EMF_NBUF * ngrp * 3 * FT; // [NBUF][ngrp][3 words][FT frames]

    // ---- exchange, step 1 (start of a step): wave w gathers the partials of ALL groups for its
    // four frames 4w..4w+3 of tile k (lane = 16 * frame + group) and only ISSUES the loads here;
    // they are checked after this step's logit MFMAs, which hide the latency.
    u64 g0 = 0, g1 = 0, g2 = 0;
    auto gather_issue = [&](int k) {
        if (dbg & 1) return;
        if (i16 < ngrp) {
            const u64 *sl = my_slots + ((size_t)((kbase + k) % EMF_NBUF) * ngrp + i16) * 3 * FT + 4 * wave + q;
            g0 = ll_load(sl); g1 = ll_load(sl + FT); g2 = ll_load(sl + 2 * FT);
        }
    };
    // ---- exchange, step 3 (after the logit MFMAs): validate / re-load, combine the groups -> lse_t
    auto combine_lse = [&](int k) {
        double m = GMMIV_NEG_BIG, sg = 0.0;
        if (!(dbg & 1)) {
            if (i16 < ngrp) {
                const unsigned want = (unsigned)(kbase + k + 1);
                const u64 *sl = my_slots + ((size_t)((kbase + k) % EMF_NBUF) * ngrp + i16) * 3 * FT + 4 * wave + q;
                unsigned spins = 0;
                while ((unsigned)(g0 >> 32) != want || (unsigned)(g1 >> 32) != want || (unsigned)(g2 >> 32) != want) {
                    __builtin_amdgcn_s_sleep(1);
                    g0 = ll_load(sl); g1 = ll_load(sl + FT); g2 = ll_load(sl + 2 * FT);
                    if (++spins > (1u << 22)) { atomicOr(err, 1u); break; } // bounded: a non-resident peer must not hang the GPU
                }
                m = (double)__uint_as_float((unsigned)g0);
                sg = __longlong_as_double((long long)(((g2 & 0xffffffffull) << 32) | (g1 & 0xffffffffull)));
            }
        } else if (i16 == 0) { m = 0.0; sg = 1.0; }
        const double M = (double)row_max_f32((float)fmax(m, -3.0e38));
        const double Ssum = row_sum_f64(sg * gexp_t(m - M, etab));
        if (i16 == 0) {
            const double lse = M + log(Ssum);
            // a zero-likelihood frame (include/gmmiv.h, "degenerate inputs"): the largest w_c lk_c rounds to 0 in fp64, or the sum is not finite
            const bool ok = M > GMMIV_ZERO_LLK && lse > -__builtin_inf() && lse < __builtin_inf();
            const int tl = 4 * wave + q;
            lse_t[tl] = ok ? lse + lse_shift : 1.0e300;
            const long fr = f0 + (long)k * FT + tl;
            if (lse_out && grp == 0 && fr < f1) lse_out[fr] = ok ? lse : -__builtin_inf();
        }
    };

    // ---- phase A: logits of tile k (kept in z), per-wave partial log-sum-exp into LDS -------------
    auto phaseA = [&](int k, double (&z)[2][4]) {
        const double *cur = xb + (k % NXB) * FT * RLp;
        const double *pL = cur + offL;
        if (active) {
#pragma unroll
            for (int fs = 0; fs < 2; ++fs) {
                d4 zx = (d4){0, 0, 0, 0}, zq = (d4){0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = pL[fs * 16 * RLp + 4 * s];
                    zx = MFMA_F64(a, Pr[s], zx);
                    zq = MFMA_F64(a * a, Pr[KS + s], zq);
                }
                const double a = pL[fs * 16 * RLp + Dp]; // (1, 0, 0, 0): adds a_c
                zx = MFMA_F64(a, Pr[2 * KS], zx);
#pragma unroll
                for (int r = 0; r < 4; ++r) z[fs][r] = zx[r] + zq[r];
            }
        } else {
#pragma unroll
            for (int fs = 0; fs < 2; ++fs)
#pragma unroll
                for (int r = 0; r < 4; ++r) z[fs][r] = GMMIV_NEG_BIG;
        }
        // per wave and frame row: reference = f32 row maximum (any value within a few hundred of the
        // true maximum works), sum of exp(z - reference) over the wave's 16 Gaussians
#pragma unroll
        for (int fs = 0; fs < 2; ++fs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double mw = (double)row_max_f32((float)fmax(z[fs][r], -3.0e38));
                const double sw = row_sum_f64(gexp_t(z[fs][r] - mw, etab));
                if (i16 == 0) {
                    const int row = fs * 16 + q + 4 * r;
                    red[(wave * FT + row) * 2] = mw;
                    red[(wave * FT + row) * 2 + 1] = sw;
                }
            }
    };

    // ---- exchange, step 2 (after the barrier that completes `red`): wave 0 publishes this
    // workgroup's partials of tile k.  The per-wave maxima are f32 values, so M is exact in f32.
    auto publish = [&](int k) {
        if (wave == 0 && !(dbg & 1) && lane < FT) {
            double M = red[lane * 2];
#pragma unroll
            for (int w = 1; w < 8; ++w) M = fmax(M, red[(w * FT + lane) * 2]);
            double Ssum = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) Ssum += red[(w * FT + lane) * 2 + 1] * gexp_t(red[(w * FT + lane) * 2] - M, etab);
            const unsigned seq = (unsigned)(kbase + k + 1);
            u64 *sl = my_slots + ((size_t)((kbase + k) % EMF_NBUF) * ngrp + grp) * 3 * FT + lane;
            const u64 sb = (u64)__double_as_longlong(Ssum);
            ll_store(sl, __float_as_uint((float)M), seq);
            ll_store(sl + FT, (unsigned)sb, seq);
            ll_store(sl + 2 * FT, (unsigned)(sb >> 32), seq);
        }
    };

    // ---- phase C: posteriors of tile k from z and lse_t, statistics MFMAs -------------------------
    auto phaseC = [&](int k, const double (&z)[2][4]) {
        if (active) {
            const double *cur = xb + (k % NXB) * FT * RLp;
            const double *pS = cur + offS;
#pragma unroll
            for (int fs = 0; fs < 2; ++fs)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double gam = gexp_t(z[fs][r] - lse_t[fs * 16 + q + 4 * r], etab);
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const double bv = pS[(fs * 16 + 4 * r) * RLp + 4 * r + 16 * j];
                        S[j] = MFMA_F64(gam, bv, S[j]);
                        if (SQ) S2[j] = MFMA_F64(gam, bv * bv, S2[j]);
                    }
                }
        }
    };

    // one pipeline step: A(ka) into za, C(kc = ka - 1) from zc.  Two barriers:
    //   B1 closes `red` (phase A partials of tile ka) and lse_t (tile kc), and frees frame buffer ka+1
    //      (= ka-2, last read by the phase C of the previous step);
    //   B2 closes the freshly staged frame tile ka+1 and frees red / lse_t for the next step.
    auto step = [&](int ka, double (&za)[2][4], double (&zc)[2][4]) {
        const int kc = ka - 1;
        const bool doA = ka < ntiles, doC = kc >= 0 && kc < ntiles;
        if (ka + 1 < ntiles) load_tile(ka + 1);
        if (doC) gather_issue(kc);
        if (doA) phaseA(ka, za);
        if (doC) combine_lse(kc);
        __syncthreads();                       // B1
        if (doA) publish(ka);
        if (doC) phaseC(kc, zc);
        if (ka + 1 < ntiles) write_tile(ka + 1);
        __syncthreads();                       // B2
    };

    // ---- software pipeline over the tiles (look-ahead 1) ------------------------------------------
    double zA[2][4], zB[2][4];
#pragma unroll
    for (int fs = 0; fs < 2; ++fs)
#pragma unroll
        for (int r = 0; r < 4; ++r) { zA[fs][r] = 0.0; zB[fs][r] = 0.0; }
    for (int sg = team; sg < nseg; sg += nteams) {
        f0 = seg_begin[sg]; f1 = seg_begin[sg + 1];
        ntiles = (int)((f1 - f0 + FT - 1) / FT);
#pragma unroll
        for (int j = 0; j < JT; ++j) S[j] = (d4){0, 0, 0, 0};
        if (SQ) {
#pragma unroll
            for (int j = 0; j < JT; ++j) S2[j] = (d4){0, 0, 0, 0};
        }
        if (ntiles > 0) { load_tile(0); write_tile(0); }
        __syncthreads();
        for (int k = 0; k <= ntiles; k += 2) {
            step(k, zA, zB);
            step(k + 1, zB, zA);
        }
        __syncthreads();
        kbase += ntiles;

        if (!active) continue;
        if (mode == 0) {
            const size_t Cp = (size_t)nct * 16;
            double *o = out0 + (size_t)sg * Cp * (2 * RL);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t c = (size_t)ct * 16 + q + 4 * r;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    o[c * (2 * RL) + 16 * j + i16] = S[j][r];
                    if (SQ) o[c * (2 * RL) + RL + 16 * j + i16] = S2[j][r];
                }
            }
        } else {
            double *N = out0 + (size_t)sg * C;
            double *F = out1 + (size_t)sg * C * D;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + q + 4 * r;
                if (c >= C) continue;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int col = 16 * j + i16;
                    if (col < D) F[(size_t)c * D + col] = S[j][r];
                    else if (col == Dp) N[c] = S[j][r];
                }
            }
        }
    }
}

#define HIPCHK(e)                                                         \
    do {                                                                  \
        hipError_t _e = (e);                                              \
        if (_e != hipSuccess) return (int)_e;                             \
    } while (0)

// hand-off words (8 bytes each, zeroed by the caller on the stream before every launch) + error word
size_t gmmk_em_fused_slot_words(int nteams, int ngrp) { return (size_t)nteams * EMF_NBUF * ngrp * 3 * EMF_FT + 2; }

template <int KS, typename XT, bool SQ>
static int launch_fused(hipStream_t st, const void *x, long ldx, int D, int C, const double *Pt, int nct, double lse_shift,
                        const long *seg_begin, int nseg, int nteams, int ngrp, int mode, double *out0, double *out1,
                        double *lse_out, double *slots, int n_cu, int dbg, int *query_blocks)
{
    constexpr int RL = ((4 * KS + 2 + 31) / 32) * 32;
    const size_t lds = ((size_t)3 * EMF_FT * (RL + 32) + 8 * EMF_FT * 2 + EMF_FT + 32) * sizeof(double);
    HIPCHK((gmmiv_lds_attr<k_em_fused<KS, XT, SQ>>(lds))); // per (device, kernel): lds_attr.h
    // resident workgroups per CU: asked per call (a host-side computation of the runtime, microseconds) -- a process-wide
    // cache would hand device 0's answer to a context on another device
    int blocks_per_cu = 0;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_em_fused<KS, XT, SQ>, 512, lds));
    if (query_blocks) { *query_blocks = blocks_per_cu; return 0; }
    if (blocks_per_cu < 1 || nteams * ngrp > n_cu * blocks_per_cu) return (int)hipErrorCooperativeLaunchTooLarge;
    const unsigned grid = (unsigned)(8 * ngrp * ((nteams + 7) / 8));
    const unsigned magicD = gmmiv_div_magic(D);
    k_em_fused<KS, XT, SQ><<<grid, 512, lds, st>>>(x, ldx, D, C, Pt, nct, lse_shift, seg_begin, nseg, nteams, ngrp, mode, out0, out1,
                                                    lse_out, (u64 *)slots,
                                                    (unsigned *)((u64 *)slots + gmmk_em_fused_slot_words(nteams, ngrp) - 2), magicD, dbg);
    return (int)hipGetLastError();
}

// slots: gmmk_em_fused_slot_words() 8-byte words, zeroed by the caller on the stream before every
// launch; the last two hold the error word.  sq = 1: EM statistics (occ, sum x, sum x^2); sq = 0: N and F.
// query_blocks != NULL: only report the resident workgroups per CU of the instantiation.
int gmmk_em_fused(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, const double *Pt, int nct,
                  double lse_shift, const long *seg_begin, int nseg, int nteams, int ngrp, int mode, double *out0, double *out1,
                  double *lse_out, double *slots, int n_cu, int dbg, int *query_blocks)
{
    if (!query_blocks && (nteams <= 0 || nseg <= 0)) return 0;
    if (ngrp > EMF_MAXGRP) return (int)hipErrorCooperativeLaunchTooLarge; // a DPP row gathers the groups
#define ARGS st, x, ldx, D, C, Pt, nct, lse_shift, seg_begin, nseg, nteams, ngrp, mode, out0, out1, lse_out, slots, n_cu, dbg, query_blocks
#define CASE(K)                                                                                           \
    case K:                                                                                               \
        if (sq) return x_f64 ? launch_fused<K, double, true>(ARGS) : launch_fused<K, float, true>(ARGS);  \
        return x_f64 ? launch_fused<K, double, false>(ARGS) : launch_fused<K, float, false>(ARGS);
    switch (KS) {
        CASE(4) CASE(8) CASE(15)
    }
#undef CASE
#undef ARGS
    return (int)hipErrorInvalidValue;
}
