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
// so 63 instead of 93 MFMAs per 256 pairs.  The hand-off is the write-through payload + flag recipe
// of cdna_hip_programming.md Guideline 16 (8-byte agent-scope relaxed atomic stores = sc1 stores,
// every storing wave drains vmcnt before ONE lane stores the flag; consumers poll relaxed, then one
// agent-scope acquire).  One tile of look-ahead hides the hand-off latency.  Correctness does not
// depend on placement; liveness needs all workgroups of the grid resident (grid <= CUs, one 512-thread
// workgroup with ~100 KB LDS per CU) -- every spin is bounded and reports through `err`.
#include "devutil.h"
#include "gmm_kernels.h"

#define EMF_NBUF 4      // hand-off slots (>= 2 * lookahead + 2 with lookahead 1)
#define EMF_FT 32       // frames per tile
#define EMF_MAXGRP 64   // Gaussian groups per team (lanes of the polling wave)

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
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int KS, typename XT>
__global__ __launch_bounds__(512, 2) void k_em_fused(const void *__restrict__ x, long ldx, int D, const double *__restrict__ Pt,
                                                     int nct, double lse_shift, const long *__restrict__ seg_begin, int nteams,
                                                     int ngrp, double *__restrict__ part, double *__restrict__ lse_out,
                                                     double *__restrict__ slots, unsigned *__restrict__ flags,
                                                     unsigned *__restrict__ err, unsigned magicD, int dbg)
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
    double *gat = etab + 32;                       // 16 groups x FT x (m, s): the team's partials of one tile
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
    const long f0 = seg_begin[team], f1 = seg_begin[team + 1];
    const int ntiles = (int)((f1 - f0 + FT - 1) / FT);

    d4 S[JT], S2[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) { S[j] = (d4){0, 0, 0, 0}; S2[j] = (d4){0, 0, 0, 0}; }

    // staging plan (see k_stats_mfma)
    XT stg[NLD];
    unsigned pk[NLD]; // rows are contiguous (ldx == D): the host falls back to the two-kernel path otherwise
    const int npad = FT * (RL - D);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NT * i;
        const int fr = (int)__umulhi((unsigned)e, magicD), d = e - fr * D;
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
            if (pk[i] < ((unsigned)FT << 16)) *(double *)((char *)dst + (pk[i] & 0xffffu)) = (double)stg[i];
        for (int e = tid; e < npad; e += NT) { // pad columns: const 1 at Dp for existing rows, zeros elsewhere
            const int fr = e / (RL - D), d = D + (e - fr * (RL - D));
            dst[fr * RLp + xrot(fr) + d] = (d == Dp && fr < rem) ? 1.0 : 0.0;
        }
    };

    const int offL = i16 * RLp + xrot(i16) + q;
    const int offS = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + i16;
    double *my_slots = slots + (size_t)team * EMF_NBUF * ngrp * FT * 2;
    unsigned *my_flags = flags + (size_t)team * EMF_NBUF * EMF_MAXGRP;

    // ---- exchange, step 1 (start of a step): every wave polls the flags of the two groups it
    // gathers for tile k (published by the peers one step ago) and ISSUES the loads of their
    // (max, sum) pairs; the values are consumed after this step's logit MFMAs, which hide the latency.
    double gm = GMMIV_NEG_BIG, gs = 0.0;
    auto gather_issue = [&](int k) {
        if (dbg & 1) return;
        const int gsel = 2 * wave + (lane >> 5); // group gathered by this half-wave
        if ((lane & 31) == 0 && gsel < ngrp) {
            const unsigned want = (unsigned)(k + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(&my_flags[(k % EMF_NBUF) * EMF_MAXGRP + gsel], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { atomicOr(err, 1u); break; } // bounded: a non-resident peer must not hang the GPU
            }
        }
        // no acquire fence: the payload is written with sc1 stores and read with sc1 (L1-bypassing)
        // loads, the form Guideline 16 allows in place of an agent-scope acquire (saves ~1.7 us per step)
        gm = GMMIV_NEG_BIG; gs = 0.0;
        if (gsel < ngrp) {
            const double *sl = my_slots + (((size_t)(k % EMF_NBUF) * ngrp + gsel) * FT + (lane & 31)) * 2;
            gm = ld_agent(sl);
            gs = ld_agent(sl + 1);
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
    // workgroup's partials of tile k; every wave drops the pairs it gathered for tile kprev into LDS
    auto publish_and_stash = [&](int k, bool do_pub, bool do_stash) {
        if (do_pub && wave == 0 && !(dbg & 1)) {
            if (lane < FT) {
                double M = red[lane * 2];
#pragma unroll
                for (int w = 1; w < 8; ++w) M = fmax(M, red[(w * FT + lane) * 2]);
                double Ssum = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) Ssum += red[(w * FT + lane) * 2 + 1] * gexp_t(red[(w * FT + lane) * 2] - M, etab);
                double *sl = my_slots + (((size_t)(k % EMF_NBUF) * ngrp + grp) * FT + lane) * 2;
                st_agent(sl, M);
                st_agent(sl + 1, Ssum);
            }
        }
        if (do_stash) {
            const int gsel = 2 * wave + (lane >> 5);
            gat[(gsel * FT + (lane & 31)) * 2] = gm;
            gat[(gsel * FT + (lane & 31)) * 2 + 1] = gs;
        }
    };

    // the flag follows the payload: the storing wave drains its stores, then ONE lane raises the flag.
    // Called after the next barrier, so the drain has usually nothing left to wait for.
    auto publish_flag = [&](int k) {
        if (wave == 0 && !(dbg & 1)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&my_flags[(k % EMF_NBUF) * EMF_MAXGRP + grp], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- exchange, step 3: wave w combines the 16 group partials of frames 4w..4w+3 -> lse_t ------
    auto combine_lse = [&](int k) {
        if (lane < 32) {
            const int tl = 4 * wave + (lane >> 3), p = lane & 7;
            const double m0 = gat[((2 * p) * FT + tl) * 2], s0 = gat[((2 * p) * FT + tl) * 2 + 1];
            const double m1 = gat[((2 * p + 1) * FT + tl) * 2], s1 = gat[((2 * p + 1) * FT + tl) * 2 + 1];
            double M = fmax(m0, m1);
            // 8-lane all-reduce (lanes of one frame are contiguous): xor 1, 2, 4
            M = fmax(M, shfl_xor_f64(M, 1)); M = fmax(M, shfl_xor_f64(M, 2)); M = fmax(M, shfl_xor_f64(M, 4));
            double Ssum = s0 * gexp_t(m0 - M, etab) + s1 * gexp_t(m1 - M, etab);
            Ssum += shfl_xor_f64(Ssum, 1); Ssum += shfl_xor_f64(Ssum, 2); Ssum += shfl_xor_f64(Ssum, 4);
            if (p == 0) {
                const double lse = M + log(Ssum);
                lse_t[tl] = lse + lse_shift;
                const long fr = f0 + (long)k * FT + tl;
                if (grp == 0 && fr < f1) lse_out[fr] = lse;
            }
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
                        S2[j] = MFMA_F64(gam, bv * bv, S2[j]);
                    }
                }
        }
    };

    // one pipeline step: A(ka) into za, C(kc = ka - 1) from zc.  Two barriers:
    //   B1 closes `red` (phase A partials) and `gat` (gathered team partials of tile kc);
    //   B2 closes lse_t and the freshly staged frame tile ka+1.
    // Reuse is safe without a third barrier: red/gat are next written after B2, lse_t and the frame
    // buffer of tile kc are next written after the following B1, which every wave reaches only after
    // finishing its phase C.
    auto step = [&](int ka, double (&za)[2][4], double (&zc)[2][4]) {
        const int kc = ka - 1;
        const bool doA = ka < ntiles, doC = kc >= 0 && kc < ntiles;
        if (ka + 1 < ntiles) load_tile(ka + 1);
        if (doC) gather_issue(kc);
        if (doA) phaseA(ka, za);
        if (doC && !(dbg & 1)) {
            const int gsel = 2 * wave + (lane >> 5);
            gat[(gsel * FT + (lane & 31)) * 2] = gm;
            gat[(gsel * FT + (lane & 31)) * 2 + 1] = gs;
        }
        __syncthreads();                       // B1
        publish_and_stash(ka, doA, false);
        if (doC) combine_lse(kc);
        if (ka + 1 < ntiles) write_tile(ka + 1);
        __syncthreads();                       // B2
        if (doA) publish_flag(ka);
        if (doC) phaseC(kc, zc);
    };

    // ---- software pipeline over the tiles (look-ahead 1) ------------------------------------------
    double zA[2][4], zB[2][4];
#pragma unroll
    for (int fs = 0; fs < 2; ++fs)
#pragma unroll
        for (int r = 0; r < 4; ++r) { zA[fs][r] = 0.0; zB[fs][r] = 0.0; }
    if (ntiles > 0) { load_tile(0); write_tile(0); }
    __syncthreads();
    for (int k = 0; k <= ntiles; k += 2) {
        step(k, zA, zB);
        step(k + 1, zB, zA);
    }
    __syncthreads();

    if (!active) return;
    const size_t Cp = (size_t)nct * 16;
    double *o = part + (size_t)team * Cp * (2 * RL);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t c = (size_t)ct * 16 + q + 4 * r;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            o[c * (2 * RL) + 16 * j + i16] = S[j][r];
            o[c * (2 * RL) + RL + 16 * j + i16] = S2[j][r];
        }
    }
}

#define HIPCHK(e)                                                         \
    do {                                                                  \
        hipError_t _e = (e);                                              \
        if (_e != hipSuccess) return (int)_e;                             \
    } while (0)

size_t gmmk_em_fused_slot_doubles(int nteams, int ngrp) { return (size_t)nteams * EMF_NBUF * ngrp * EMF_FT * 2; }
size_t gmmk_em_fused_flag_words(int nteams) { return (size_t)nteams * EMF_NBUF * EMF_MAXGRP + 16; }

template <int KS, typename XT>
static int launch_fused(hipStream_t st, const void *x, long ldx, int D, const double *Pt, int nct, double lse_shift,
                        const long *seg_begin, int nteams, int ngrp, double *part, double *lse_out, double *slots,
                        unsigned *flags, int n_cu, int dbg)
{
    constexpr int RL = ((4 * KS + 2 + 31) / 32) * 32;
    const size_t lds = ((size_t)3 * EMF_FT * (RL + 32) + 8 * EMF_FT * 2 + EMF_FT + 32 + 16 * EMF_FT * 2) * sizeof(double);
    static int blocks_per_cu = -1;
    if (blocks_per_cu < 0) {
        HIPCHK(hipFuncSetAttribute((const void *)k_em_fused<KS, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int nb = 0;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_em_fused<KS, XT>, 512, lds));
        blocks_per_cu = nb;
    }
    if (blocks_per_cu < 1 || nteams * ngrp > n_cu * blocks_per_cu) return (int)hipErrorCooperativeLaunchTooLarge;
    const unsigned grid = (unsigned)(8 * ngrp * ((nteams + 7) / 8));
    const unsigned magicD = (unsigned)((1ULL << 32) / (unsigned)D + 1);
    k_em_fused<KS, XT><<<grid, 512, lds, st>>>(x, ldx, D, Pt, nct, lse_shift, seg_begin, nteams, ngrp, part, lse_out, slots,
                                                flags, flags + gmmk_em_fused_flag_words(nteams) - 16, magicD, dbg);
    return (int)hipGetLastError();
}

// flags (zeroed by the caller on the stream before every launch) hold the hand-off flags and, in the
// last 16 words, the error word.
int gmmk_em_fused(hipStream_t st, int KS, int x_f64, const void *x, long ldx, int D, const double *Pt, int nct,
                  double lse_shift, const long *seg_begin, int nteams, int ngrp, double *part, double *lse_out, double *slots,
                  unsigned *flags, int n_cu, int dbg)
{
    if (nteams <= 0) return 0;
    if (ngrp > 16) return (int)hipErrorCooperativeLaunchTooLarge; // each wave gathers two groups
#define CASE(K)                                                                                                              \
    case K:                                                                                                                  \
        return x_f64 ? launch_fused<K, double>(st, x, ldx, D, Pt, nct, lse_shift, seg_begin, nteams, ngrp, part, lse_out, slots, flags, n_cu, dbg) \
                     : launch_fused<K, float>(st, x, ldx, D, Pt, nct, lse_shift, seg_begin, nteams, ngrp, part, lse_out, slots, flags, n_cu, dbg);
    switch (KS) {
        CASE(4) CASE(8) CASE(15)
    }
#undef CASE
    return (int)hipErrorInvalidValue;
}
