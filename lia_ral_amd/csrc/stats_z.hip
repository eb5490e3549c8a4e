// stats_z.hip -- posterior statistics from STORED logits.
//
// k_llk_mfma (WZ) evaluates every frame x Gaussian logit once and leaves it in HBM (fp64, in the
// MFMA register layout); this kernel streams the logits back, turns them into posteriors
// gamma = exp(z - lse_t) and runs only the statistics MFMAs.  The two-kernel path that recomputes the
// logits in the statistics kernel issues 31 + (31 + 32) fp64 MFMAs per 16 frames x 16 Gaussians, this
// one 31 + 32 -- the MFMA pipe is the bound (MI355X_MICROARCH: 78.6 TFLOP/s fp64), HBM is not: the
// logit stream is 2 KB per 32 MFMAs = 1 byte per cycle per SIMD (~2.4 TB/s at full MFMA rate).
//
// Workgroup = 8 waves; wave w owns TWO Gaussian tiles (32 Gaussians), so every x operand read from
// LDS (and every x^2) feeds 4 MFMAs.  Frames stream through the same rotated LDS tile as
// k_stats_mfma (64 rows [x_0..x_{D-1}, 0.., 1, lse_t, 0..], double buffered, register staged).
//   mode 0 (EM):  out0[seg][c][2 RL] partial sums (cols: x | x^2 halves; col Dp = occupancy);
//                 accum != 0 adds to what is there (frame chunks processed by successive launches)
//   mode 1 (TV):  N = out0[seg][C], F = out1[seg][C*D] written directly
// Segment bounds are frame indices relative to x / lse / zbuf block 0; a segment may start anywhere:
// its first tile starts at the 16-frame block holding f0 and rows before f0 are masked (lse = 1e300).
#include "devutil.h"
#include "gmm_kernels.h"

// ABL (timing experiments only, wrong results): 1 = no exp, 2 = no logit loads, 3 = no staging / barrier per tile
template <int KS, bool SQ, typename XT, bool PRUNE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_stats_z(const void *__restrict__ x, long ldx, int D, int C, int nct,
                                                    const double *__restrict__ zbuf, long nfb,
                                                    const double *__restrict__ lse, double lse_shift,
                                                    const long *__restrict__ seg_begin, int nseg, int ngrp,
                                                    double *__restrict__ out0, double *__restrict__ out1, int mode, int accum,
                                                    unsigned magicD, double prune_arg)
{
    constexpr int Dp = 4 * KS;
    constexpr int RL = ((Dp + 2 + 31) / 32) * 32;
    constexpr int JT = RL / 16;
    constexpr int FT = 64;
    constexpr int NT = 512;
    constexpr int NLD = (FT * Dp + NT - 1) / NT;
    constexpr int RLp = RL + 32; // padded row: additive rotation xrot(t) < 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *buf0 = (double *)smem;
    double *buf1 = buf0 + FT * RLp;
    double *etab = buf1 + FT * RLp; // 64-entry exp table
    gexp_table64_init(etab, threadIdx.x);

    // XCD-aware decode (see k_stats_mfma): all Gaussian groups of one segment share an XCD
    const int b = blockIdx.x;
    const int seg_lo = b & 7;
    const int rest = b >> 3;
    const int grp = rest % ngrp;
    const int seg = (rest / ngrp) * 8 + seg_lo;
    if (seg >= nseg) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int ct0 = (grp * 8 + wave) * 2; // nct is even (host pads the packed model to pairs of tiles)
    const bool active = ct0 < nct;

    const long f0 = seg_begin[seg], f1 = seg_begin[seg + 1];
    const long fa = f0 & ~15L;                                  // first tile starts on a logit block
    const int nblk = f1 > f0 ? (int)((f1 - fa + 15) >> 4) : 0;  // 16-frame blocks
    const int ntiles = (nblk + 3) >> 2;

    d4 S[2][JT], S2[2][SQ ? JT : 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int j = 0; j < JT; ++j) S[t][j] = (d4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < (SQ ? JT : 1); ++j) S2[t][j] = (d4){0, 0, 0, 0};
    }

    // staging plan (see k_stats_mfma)
    XT stg[NLD];
    double stg_lse = 0.0;
    const int npad = FT * (RL - D);
    unsigned pk[NLD];
    const bool contig = (ldx == D);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NT * i;
        const int fr = (int)__umulhi((unsigned)e, magicD), d = e - fr * D;
        pk[i] = fr < FT ? ((unsigned)fr << 16) | (unsigned)((fr * RLp + xrot(fr) + d) * 8) : 0xffff0000u;
    }
    auto load_tile = [&](int tl) {
        const long fb = fa + (long)tl * FT;
        const long rem = f1 - fb;
        const unsigned lim = rem >= FT ? (unsigned)FT << 16 : (unsigned)rem << 16; // rows fr < lim >> 16 exist
        const XT *xt = (const XT *)x + fb * ldx;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            XT v = 0;
            if (pk[i] < lim) {
                if (contig) v = xt[tid + NT * i];
                else { const unsigned fr = pk[i] >> 16; v = xt[(long)fr * ldx + (tid + NT * i - (int)fr * D)]; }
            }
            stg[i] = v;
        }
        // rows outside [f0, f1) get lse = +1e300 -> posterior exp(z - lse) = 0
        if (tid < FT) { const long t = fb + tid; stg_lse = (t >= f0 && t < f1) ? lse[t] + lse_shift : 1.0e300; }
    };
    auto write_tile = [&](double *dst) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (pk[i] < ((unsigned)FT << 16)) *(double *)((char *)dst + (pk[i] & 0xffffu)) = (double)stg[i];
        if (tid < FT) dst[tid * RLp + xrot(tid) + Dp + 1] = stg_lse;
    };
    // pad columns (1.0 at Dp, zeros elsewhere; Dp + 1 is the lse column) never change: written once
    for (int e = tid; e < 2 * npad; e += NT) {
        double *dst = e < npad ? buf0 : buf1;
        const int ee = e < npad ? e : e - npad;
        const int fr = ee / (RL - D), d = D + (ee - fr * (RL - D));
        if (d != Dp + 1) dst[fr * RLp + xrot(fr) + d] = (d == Dp) ? 1.0 : 0.0;
    }

    // per-lane LDS offsets (doubles): statistics B operand (row q, col i16) and the lse column of row q
    const int offS = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + i16;
    const int offE = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + Dp + 1;

    // logit stream: block n of tile t is 2 KB at zp[t] + n * 256 doubles, 32 bytes per lane
    const double *zp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) zp[t] = zbuf + ((((size_t)(active ? ct0 + t : 0)) * nfb + (fa >> 4)) * 64 + lane) * 4;
    // two register sets, alternating between even and odd blocks (the tile loop is unrolled by 4)
    d4 zA[2], zB[2];
    zA[0] = (d4){0, 0, 0, 0}; zA[1] = zA[0]; zB[0] = zA[0]; zB[1] = zA[0];
    if (active && nblk > 0) { zA[0] = __builtin_nontemporal_load((const d4 *)zp[0]); zA[1] = __builtin_nontemporal_load((const d4 *)zp[1]); }

    if (ntiles > 0) {
        load_tile(0);
        write_tile(buf0);
    }
    __syncthreads();
    for (int tl = 0; tl < ntiles; ++tl) {
        const double *cur = (tl & 1) ? buf1 : buf0;
        double *nxt = (tl & 1) ? buf0 : buf1;
        if (ABL == 3) { cur = buf0; nxt = buf1; }
        if (tl + 1 < ntiles && ABL != 3) load_tile(tl + 1);
        if (active) {
            const double *pS = cur + offS, *pE = cur + offE;
            auto block = [&](int fs, const d4 (&zc)[2], d4 (&zn)[2]) {
                const int n = tl * 4 + fs;
                if (n + 1 < nblk && ABL != 2) { // logits of the next block: streamed once, keep them out of the caches
                    zn[0] = __builtin_nontemporal_load((const d4 *)(zp[0] + (size_t)(n + 1) * 256));
                    zn[1] = __builtin_nontemporal_load((const d4 *)(zp[1] + (size_t)(n + 1) * 256));
                }
                // register r holds rows (frames) fs*16 + 4r + q of 16 Gaussians: already the A operand
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double le = pE[(fs * 16 + 4 * r) * RLp + 4 * r];
                    const double a0 = zc[0][r] - le, a1 = zc[1][r] - le;
                    if (PRUNE && __builtin_amdgcn_ballot_w64(a0 > prune_arg || a1 > prune_arg) == 0) continue;
                    const double g0 = ABL == 1 ? a0 : gexp_t64(a0, etab), g1 = ABL == 1 ? a1 : gexp_t64(a1, etab);
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const double bv = pS[(fs * 16 + 4 * r) * RLp + 4 * r + 16 * j];
                        S[0][j] = MFMA_F64(g0, bv, S[0][j]);
                        S[1][j] = MFMA_F64(g1, bv, S[1][j]);
                        if (SQ) {
                            const double b2 = bv * bv;
                            S2[0][j] = MFMA_F64(g0, b2, S2[0][j]);
                            S2[1][j] = MFMA_F64(g1, b2, S2[1][j]);
                        }
                    }
                }
            };
            const int nb = nblk - tl * 4; // blocks of this tile that exist (wave-uniform)
            block(0, zA, zB);
            if (nb > 1) block(1, zB, zA);
            if (nb > 2) block(2, zA, zB);
            if (nb > 3) block(3, zB, zA);
        }
        if (tl + 1 < ntiles && ABL != 3) write_tile(nxt);
        if (ABL != 3) __syncthreads();
    }
    if (!active) return;
    // D layout: lane holds column j = 16 jt + i16, rows (Gaussians) q + 4 r
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ct = ct0 + t;
        if (mode == 0) {
            const size_t Cp = (size_t)nct * 16;
            double *o = out0 + (size_t)seg * Cp * (2 * RL);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t c = (size_t)ct * 16 + q + 4 * r;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    double *o1 = o + c * (2 * RL) + 16 * j + i16;
                    *o1 = (accum ? *o1 : 0.0) + S[t][j][r];
                    if (SQ) { double *o2 = o1 + RL; *o2 = (accum ? *o2 : 0.0) + S2[t][j][r]; }
                }
            }
        } else {
            double *N = out0 + (size_t)seg * C;
            double *F = out1 + (size_t)seg * C * D;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = ct * 16 + q + 4 * r;
                if (c >= C) continue;
#pragma unroll
                for (int j = 0; j < JT; ++j) {
                    const int col = 16 * j + i16;
                    if (col < D) F[(size_t)c * D + col] = S[t][j][r];
                    else if (col == Dp) N[c] = S[t][j][r];
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

static int g_stats_z_abl = 0;
void gmmk_stats_z_set_ablation(int a) { g_stats_z_abl = a; }

template <int KS, bool SQ, typename XT, bool PRUNE, int ABL = 0>
static int launch_z(hipStream_t st, const void *x, long ldx, int D, int C, int nct, const double *zbuf, long nfb, const double *lse,
                    double lse_shift, const long *seg_begin, int nseg, double *out0, double *out1, int mode, int accum,
                    double prune_arg)
{
    constexpr int RL = ((4 * KS + 2 + 31) / 32) * 32;
    const size_t lds = (2 * 64 * (RL + 32) + 64) * sizeof(double); // two frame tiles + the 64-entry exp table
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute((const void *)k_stats_z<KS, SQ, XT, PRUNE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int ngrp = gmmk_stats_z_groups(nct);
    const unsigned grid = (unsigned)(ngrp * 8 * ((nseg + 7) / 8));
    const unsigned magicD = (unsigned)((1ULL << 32) / (unsigned)D + 1);
    k_stats_z<KS, SQ, XT, PRUNE, ABL><<<grid, 512, lds, st>>>(x, ldx, D, C, nct, zbuf, nfb, lse, lse_shift, seg_begin, nseg, ngrp, out0, out1,
                                                          mode, accum, magicD, prune_arg);
    return (int)hipGetLastError();
}

int gmmk_stats_z_groups(int nct) { return (nct + 15) / 16; }

#define ZARGS st, x, ldx, D, C, nct, zbuf, nfb, lse, lse_shift, seg_begin, nseg, out0, out1, mode, accum, prune_arg
template <int KS, bool SQ, typename XT>
static int launch_z_p(hipStream_t st, const void *x, long ldx, int D, int C, int nct, const double *zbuf, long nfb, const double *lse,
                      double lse_shift, const long *seg_begin, int nseg, double *out0, double *out1, int mode, int accum,
                      double prune_arg)
{
    if (prune_arg > -1.0e300) return launch_z<KS, SQ, XT, true>(ZARGS);
    if (KS == 15 && SQ && sizeof(XT) == 4 && g_stats_z_abl) { // ablation builds exist for the bench shape only
        if (g_stats_z_abl == 1) return launch_z<15, true, float, false, 1>(ZARGS);
        if (g_stats_z_abl == 2) return launch_z<15, true, float, false, 2>(ZARGS);
        if (g_stats_z_abl == 3) return launch_z<15, true, float, false, 3>(ZARGS);
    }
    return launch_z<KS, SQ, XT, false>(ZARGS);
}

int gmmk_stats_z(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, int nct, const double *zbuf,
                 long nfb, const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0, double *out1,
                 int mode, int accum, double prune_arg)
{
    if (nseg <= 0) return 0;
#define CASE(K)                                                                                      \
    case K:                                                                                          \
        if (sq) return x_f64 ? launch_z_p<K, true, double>(ZARGS) : launch_z_p<K, true, float>(ZARGS); \
        return x_f64 ? launch_z_p<K, false, double>(ZARGS) : launch_z_p<K, false, float>(ZARGS);
    switch (KS) {
        CASE(4) CASE(8) CASE(15)
    }
#undef CASE
    return -1;
}
