// stats_z.hip -- posterior statistics from STORED scaled likelihoods.
//
// k_llk_mfma (WZ) evaluates every frame x Gaussian logit once; its log-sum-exp needs
// e = exp(z) 2^-E of every pair anyway, and it leaves those in HBM (fp64, in the MFMA register
// layout) together with the running exponent E of each (frame, tile pair) and, per frame, the final
// exponent Efin and 1 / S_t (sum_c exp(z_tc) = S_t 2^Efin).  This kernel streams them back, forms the
// posterior with ONE multiply, gamma = e * [2^(E - Efin) / S_t], and runs only the statistics MFMAs.
// The path that recomputes the logits in the statistics kernel issues 31 + (31 + 32) fp64 MFMAs per
// 16 frames x 16 Gaussians and two exponentials per pair; this one 31 + 32 MFMAs and one exponential.
// The MFMA pipe is the bound (MI355X_MICROARCH: 78.6 TFLOP/s fp64), HBM is not: the stream is 2 KB
// per 32 MFMAs = 1 byte per cycle per SIMD (~2.4 TB/s at full MFMA rate).
//
// Workgroup = 8 waves; wave w owns TWO Gaussian tiles (32 Gaussians = one tile pair of k_llk_mfma), so
// every x operand read from LDS (and every x^2) feeds 4 MFMAs.  Frames stream through the same rotated
// LDS tile as k_stats_mfma (64 rows [x_0..x_{D-1}, 0.., 1, 0..], double buffered, register staged);
// the per-frame posterior factors f = scale / S_t * 2^(E - Efin) of each wave ride along in LDS.
//   mode 0 (EM):  out0[seg][c][2 RL] partial sums (cols: x | x^2 halves; col Dp = occupancy);
//                 accum != 0 adds to what is there (frame chunks processed by successive launches)
//   mode 1 (TV):  N = out0[seg][C], F = out1[seg][C*D] written directly
// Segment bounds are frame indices relative to x / inv / efin / zbuf block 0; a segment may start
// anywhere: its first tile starts at the 16-frame block holding f0 and rows before f0 are masked
// (1 / S_t = 0).
#include <atomic>
#include "devutil.h"
#include "lds_attr.h"
#include "gmm_kernels.h"

// NW = 8: one 8-wave workgroup per CU, 64-frame tiles; NW = 4: two independent 4-wave workgroups per
// CU with 32-frame tiles (their barriers and staging phases drift apart).
// Shapes (NW waves, TPW Gaussian tiles per wave, FT frames per LDS tile):
//   <8, 2, 64>   one workgroup per CU, 2 waves per SIMD, 256 VGPRs (default for the EM statistics)
//   <8, 4, 64>   the same with 4 tiles per wave: default for N / F statistics (SQ = false: no x^2 accumulators)
//   <4, 2, 32>   two independent workgroups per CU
//   <16, 1, 64>  one workgroup per CU, 4 waves per SIMD at 128 VGPRs: more waves to cover a stalled one,
//                half the operand reuse (every x operand feeds 2 MFMAs)
// ZD: register sets of the likelihood stream = prefetch distance + 1.  2: block n + 1 is requested when block n starts.  4: block n + 3
// -- the N / F shape does 32 MFMAs per 16-frame block and wave (the EM shape 64): one block of cover is less than the HBM latency under
// load, the kernel stalled on every block (0.70 of the MFMA peak at 3.4 TB/s of the 7 TB/s a plain stream reaches).  ZD divides the blocks
// per tile, so the set of a block is a compile-time index without unrolling the tile loop; everything in flight is collected at the end of
// a tile (the compiler's vmcnt bookkeeping does not survive the back-edge), i.e. the first block of a tile still has one block of cover.
template <int KS, bool SQ, typename XT, bool PRUNE, int NW = 8, int TPW = 2, int FT = 64, int ZD = 2>
__global__ __launch_bounds__(NW * 64, (TPW == 1 ? 4 : 2)) void k_stats_z(const void *__restrict__ x, long ldx, int D, int C, int nct,
                                                    const double *__restrict__ zbuf, long nfb, const int *__restrict__ eit,
                                                    const double *__restrict__ inv, const int *__restrict__ efin, double scale,
                                                    const long *__restrict__ seg_begin, int nseg, int ngrp,
                                                    double *__restrict__ out0, double *__restrict__ out1, int mode, int accum,
                                                    unsigned magicD, double prune_arg)
{
    constexpr int Dp = 4 * KS;
    constexpr int RL = ((Dp + 2 + 31) / 32) * 32;
    constexpr int JT = RL / 16;
    constexpr int NT = NW * 64;
    constexpr int BPT = FT / 16; // logit blocks per frame tile
    constexpr int NLD = (FT * Dp + NT - 1) / NT;
    constexpr int RLp = RL + 32; // padded row: additive rotation xrot(t) < 32
    constexpr int PF = TPW > 2 ? TPW / 2 : 1; // tile pairs of a wave (a running exponent belongs to a PAIR of tiles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *buf0 = (double *)smem;
    double *buf1 = buf0 + FT * RLp;

    // XCD-aware decode (see k_stats_mfma): all Gaussian groups of one segment share an XCD
    const int b = blockIdx.x;
    const int seg_lo = b & 7;
    const int rest = b >> 3;
    const int grp = rest % ngrp;
    const int seg = (rest / ngrp) * 8 + seg_lo;
    if (seg >= nseg) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int ct0 = (grp * NW + wave) * TPW; // nct is even (host pads the packed model to pairs of tiles)
    const bool active = ct0 < nct;

    const long f0 = seg_begin[seg], f1 = seg_begin[seg + 1];
    const long fa = f0 & ~15L;                                  // first tile starts on a logit block
    const int nblk = f1 > f0 ? (int)((f1 - fa + 15) >> 4) : 0;  // 16-frame blocks
    const int ntiles = (nblk + BPT - 1) / BPT;

    d4 S[TPW][JT], S2[TPW][SQ ? JT : 1];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int j = 0; j < JT; ++j) S[t][j] = (d4){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < (SQ ? JT : 1); ++j) S2[t][j] = (d4){0, 0, 0, 0};
    }

    // ---- vector memory: nothing in flight across the loop back-edge --------------------------------
    // hipcc (ROCm 7.2) keeps exact vmcnt counts only inside straight-line code: a load that is issued
    // in one iteration of the tile loop and consumed in the next makes it wait vmcnt(0..3) at the top
    // of every block -- for the prefetch it has just issued (measured: -15 % on this kernel).  So the
    // per-block operands that used to come from global memory (the running exponents) are staged
    // through LDS with the frame tile, and the one prefetch that spans the back-edge (the first
    // likelihood block of the next tile) is pinned at the END of the tile: it was issued a whole block
    // earlier, so the wait is free, and the loop head sees an empty VMEM queue.
    typedef double d2 __attribute__((ext_vector_type(2)));
    XT stg[NLD];
    double stg_inv = 0.0;
    int stg_ef = 0, stg_e[PF];
    const int npad = FT * (RL - D);
    // staging plan of element i of this thread: pk = (tile row << 16 | LDS byte offset), goff = offset in the frame block.
    // Kept in registers (2 x NLD) except in the 4-tile shape, whose 128 accumulator + 64 stream registers leave no room:
    // there it is recomputed at every stage (a dozen integer instructions per element and 64-frame tile).
    constexpr bool PLAN_IN_REGS = TPW < 4;
    const bool contig = (ldx == D);
    auto plan_pk = [&](int i) -> unsigned {
        const int e = tid + NT * i;
        const int fr = div_by_magic((unsigned)e, magicD), d = e - fr * D;
        return fr < FT ? ((unsigned)fr << 16) | (unsigned)((fr * RLp + xrot(fr) + d) * 8) : 0xffff0000u;
    };
    auto plan_goff = [&](int i) -> unsigned {
        const int e = tid + NT * i;
        const int fr = div_by_magic((unsigned)e, magicD), d = e - fr * D;
        return fr < FT ? (contig ? (unsigned)e : (unsigned)(fr * (int)ldx + d)) : 0u;
    };
    unsigned pk_r[PLAN_IN_REGS ? NLD : 1], goff_r[PLAN_IN_REGS ? NLD : 1];
    if constexpr (PLAN_IN_REGS) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) { pk_r[i] = plan_pk(i); goff_r[i] = plan_goff(i); }
    }
#define PK(i) (PLAN_IN_REGS ? pk_r[PLAN_IN_REGS ? (i) : 0] : plan_pk(i))
#define GOFF(i) (PLAN_IN_REGS ? goff_r[PLAN_IN_REGS ? (i) : 0] : plan_goff(i))
    // ftile[2][NW][PF][FT]: per frame of the tile, per wave and per tile pair of the wave, the factor that turns a stored likelihood into a
    // posterior, f = scale / S_t * 2^(E - Efin) with E the running exponent of the wave's tile pair (0 outside [f0, f1))
    double *ftile = buf1 + FT * RLp;
    const int *epair[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) epair[p] = eit + (size_t)(active ? (ct0 >> 1) + p : 0) * (nfb * 16) + fa;
    const int srow = lane & (FT - 1);
    auto issue_stage = [&](int tl) { // exactly SL loads
        const long fb = fa + (long)tl * FT;
        const long rem = f1 - fb;
        const unsigned lim = rem >= FT ? (unsigned)FT << 16 : (unsigned)rem << 16; // rows fr < lim >> 16 exist
        const XT *xt = (const XT *)x + fb * ldx;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            stg[i] = xt[PK(i) < lim ? GOFF(i) : 0u]; // unconditional (clamped, masked at use): no branch per element
        }
        const long t = fb + srow < f1 ? fb + srow : f1 - 1;
        stg_inv = inv[t];
        stg_ef = efin[t];
#pragma unroll
        for (int p = 0; p < PF; ++p) stg_e[p] = epair[p][(long)tl * FT + srow];
    };
    auto finish_stage = [&](double *dst, int buf, int tl) {
        const long fb = fa + (long)tl * FT;
        const long rem = f1 - fb;
        const unsigned lim = rem >= FT ? (unsigned)FT << 16 : (unsigned)rem << 16;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
        {
            const unsigned pki = PK(i);
            if (pki < ((unsigned)FT << 16)) *(double *)((char *)dst + (pki & 0xffffu)) = pki < lim ? (double)feat_sane(stg[i]) : 0.0;
        }
        if (lane < FT) { // rows outside [f0, f1) get f = 0 -> posterior 0
            const long t = fb + lane;
#pragma unroll
            for (int p = 0; p < PF; ++p)
                ftile[((buf * NW + wave) * PF + p) * FT + lane] = (t >= f0 && t < f1) ? __builtin_ldexp(stg_inv * scale, stg_e[p] - stg_ef) : 0.0;
        }
    };
    // pad columns (1.0 at Dp, zeros elsewhere) never change: written once
    for (int e = tid; e < 2 * npad; e += NT) {
        double *dst = e < npad ? buf0 : buf1;
        const int ee = e < npad ? e : e - npad;
        const int fr = ee / (RL - D), d = D + (ee - fr * (RL - D));
        dst[fr * RLp + xrot(fr) + d] = (d == Dp) ? 1.0 : 0.0;
    }

    // per-lane LDS offset (doubles) of the statistics B operand (row q, col i16)
    const int offS = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + i16;

    // likelihood stream: block n of tile t is 2 KB at zp[t] + n * 256 doubles, 32 bytes per lane.
    // Two register sets, alternating between even and odd blocks (the tile loop is unrolled).
    const double *zp0 = zbuf + ((((size_t)(active ? ct0 : 0)) * nfb + (fa >> 4)) * 64 + lane) * 4;
    const size_t ztile = (size_t)nfb * 256; // doubles between consecutive tiles (uniform)
    static_assert((FT / 16) % ZD == 0, "the stream sets must divide the blocks of a tile");
    d2 zs[ZD][TPW][2];
#pragma unroll
    for (int b = 0; b < ZD; ++b)
#pragma unroll
        for (int t = 0; t < TPW; ++t) { zs[b][t][0] = (d2){0, 0}; zs[b][t][1] = zs[b][t][0]; }
    auto issue_z = [&](d2 (&z)[TPW][2], int n) { // streamed once, kept out of the caches (nt)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const d2 *pz = (const d2 *)(zp0 + (size_t)t * ztile + (size_t)n * 256);
            z[t][0] = __builtin_nontemporal_load(pz);
            z[t][1] = __builtin_nontemporal_load(pz + 1);
        }
    };
    // "this value is needed now": the compiler places the (exactly counted) wait for its load here
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "operand list of PIN_Z");
#define PIN_Z(z)                                                                                              \
    do {                                                                                                      \
        if constexpr (TPW == 1) asm volatile("" : "+v"(z[0][0]), "+v"(z[0][1]));                             \
        else if constexpr (TPW == 2) asm volatile("" : "+v"(z[0][0]), "+v"(z[0][1]), "+v"(z[1][0]), "+v"(z[1][1])); \
        else asm volatile("" : "+v"(z[0][0]), "+v"(z[0][1]), "+v"(z[1][0]), "+v"(z[1][1]), "+v"(z[TPW / 2][0]),     \
                          "+v"(z[TPW / 2][1]), "+v"(z[TPW - 1][0]), "+v"(z[TPW - 1][1]));                      \
    } while (0)

    if (ntiles > 0) {
        issue_stage(0);
        if (active) {
#pragma unroll
            for (int b = 0; b + 1 < ZD; ++b) issue_z(zs[b], b < nblk ? b : nblk - 1);
        }
        finish_stage(buf0, 0, 0);
#pragma unroll
        for (int b = 0; b + 1 < ZD; ++b) PIN_Z(zs[b]);
    }
    __syncthreads();
    // one tile; `full` = not the last tile of the segment: all BPT blocks exist and so does every prefetch,
    // the body is straight-line code (the compiler schedules LDS reads and waits across the blocks)
    auto tile = [&](int tl, bool full) {
        const double *cur = (tl & 1) ? buf1 : buf0;
        double *nxt = (tl & 1) ? buf0 : buf1;
        const bool staged = full || tl + 1 < ntiles;
        if (staged) issue_stage(tl + 1);
        if (active) {
            const double *pS = cur + offS;
            const double *pF = ftile + ((tl & 1) * NW + wave) * PF * FT + q;
            auto block = [&](int fs, d2 (&zc)[TPW][2], d2 (&zn)[TPW][2]) {
                const int n = tl * BPT + fs;
                // (full tile, more than two sets: the request may lie beyond the segment's last block -- clamped, never consumed)
                if (full) issue_z(zn, (ZD == 2 || n + ZD - 1 < nblk) ? n + ZD - 1 : nblk - 1);
                else if (n + ZD - 1 < nblk) issue_z(zn, n + ZD - 1);
                // register r holds rows (frames) fs*16 + 4r + q of 16 Gaussians: already the A operand
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double f[PF]; // 2^(E - Efin) / S_t (x EM weight), 0 for a masked frame
#pragma unroll
                    for (int p = 0; p < PF; ++p) f[p] = pF[p * FT + fs * 16 + 4 * r];
                    double gm[TPW];
                    bool keep = false;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) { gm[t] = zc[t][r >> 1][r & 1] * f[PF > 1 ? t >> 1 : 0]; keep |= gm[t] > prune_arg; }
                    if (PRUNE && __builtin_amdgcn_ballot_w64(keep) == 0) continue;
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const double bv = pS[(fs * 16 + 4 * r) * RLp + 4 * r + 16 * j];
#pragma unroll
                        for (int t = 0; t < TPW; ++t) S[t][j] = MFMA_F64(gm[t], bv, S[t][j]);
                        if (SQ) {
                            const double b2 = bv * bv;
#pragma unroll
                            for (int t = 0; t < TPW; ++t) S2[t][j] = MFMA_F64(gm[t], b2, S2[t][j]);
                        }
                    }
                }
            };
            const int nb = nblk - tl * BPT; // blocks of this tile that exist (wave-uniform)
            // block fs consumes set fs % ZD and requests block n + ZD - 1 into the set block n - 1 has just left
            block(0, zs[0], zs[ZD - 1]);
            if (full || nb > 1) block(1, zs[1 % ZD], zs[0]);
            if (BPT > 2) {
                if (full || nb > 2) block(2, zs[2 % ZD], zs[1 % ZD]);
                if (full || nb > 3) block(3, zs[3 % ZD], zs[2 % ZD]);
            }
            // the next tile's first blocks (issued by the last blocks above) have landed: nothing in flight at the back-edge
#pragma unroll
            for (int b = 0; b + 1 < ZD; ++b) PIN_Z(zs[b]);
        }
        if (staged) finish_stage(nxt, (tl + 1) & 1, tl + 1);
        __syncthreads();
    };
    for (int tl = 0; tl + 1 < ntiles; ++tl) tile(tl, true);
    if (ntiles > 0) tile(ntiles - 1, false);
#undef PIN_Z
#undef PK
#undef GOFF
    if (!active) return;
    // D layout: lane holds column j = 16 jt + i16, rows (Gaussians) q + 4 r
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
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

// per host thread: a context belongs to one thread at a time and pushes its own option right before the launch,
// so contexts driven from different threads (one per GPU) cannot see each other's value
// workgroup shape / stream depth: options of the calling context (ctx.h: gmmiv_kopts, bound per call)
#define g_stats_z_waves (gmmiv_kopts_cur().z_waves)
#define g_stats_z_tv4 (gmmiv_kopts_cur().z_tv4)
#define g_stats_z_depth_em (gmmiv_kopts_cur().z_depth_em)
#define g_stats_z_depth_tv (gmmiv_kopts_cur().z_depth_tv)
int gmmk_stats_z_groups(int nct) { const int tpg = g_stats_z_waves == 4 ? 8 : 16; return (nct + tpg - 1) / tpg; }
int gmmk_stats_z_wg_per_cu(void) { return g_stats_z_waves == 4 ? 2 : 1; }

template <int KS, bool SQ, typename XT, bool PRUNE, int NW, int TPW, int FT, int ZD = 2>
static int launch_z(hipStream_t st, const void *x, long ldx, int D, int C, int nct, const double *zbuf, long nfb, const int *eit,
                    const double *inv, const int *efin, double scale, const long *seg_begin, int nseg, double *out0, double *out1,
                    int mode, int accum, double prune_thr)
{
    constexpr int RL = ((4 * KS + 2 + 31) / 32) * 32;
    constexpr int PF = TPW > 2 ? TPW / 2 : 1;
    const size_t lds = (size_t)2 * FT * (RL + 32) * sizeof(double) + (size_t)2 * NW * PF * FT * sizeof(double); // two frame tiles + posterior factors
    HIPCHK((gmmiv_lds_attr<k_stats_z<KS, SQ, XT, PRUNE, NW, TPW, FT, ZD>>(lds))); // per (device, kernel): lds_attr.h
    const int ngrp = (nct + TPW * NW - 1) / (TPW * NW);
    const unsigned grid = (unsigned)(ngrp * 8 * ((nseg + 7) / 8));
    const unsigned magicD = gmmiv_div_magic(D);
    k_stats_z<KS, SQ, XT, PRUNE, NW, TPW, FT, ZD><<<grid, NW * 64, lds, st>>>(x, ldx, D, C, nct, zbuf, nfb, eit, inv, efin, scale, seg_begin, nseg, ngrp,
                                                              out0, out1, mode, accum, magicD, prune_thr);
    return (int)hipGetLastError();
}

#define ZARGS st, x, ldx, D, C, nct, zbuf, nfb, eit, inv, efin, scale, seg_begin, nseg, out0, out1, mode, accum, prune_thr
template <int KS, bool SQ, typename XT>
static int launch_z_p(hipStream_t st, const void *x, long ldx, int D, int C, int nct, const double *zbuf, long nfb, const int *eit,
                      const double *inv, const int *efin, double scale, const long *seg_begin, int nseg, double *out0, double *out1,
                      int mode, int accum, double prune_thr)
{
    if (prune_thr > 0.0) return launch_z<KS, SQ, XT, true, 8, 2, 64>(ZARGS);
    if (g_stats_z_waves == 4) return launch_z<KS, SQ, XT, false, 4, 2, 32>(ZARGS);
    if (g_stats_z_waves == 16) return launch_z<KS, SQ, XT, false, 16, 1, 64>(ZARGS);
    // N / F statistics only (no x^2 accumulators): FOUR tiles per wave in the same 128 accumulator registers, so that
    // every x operand read from LDS feeds 4 MFMAs here too
    // (a wave's four tiles must all exist: the host pads the packed model to PAIRS of tiles only)
    if constexpr (!SQ) {
        if (g_stats_z_depth_tv == 4) return launch_z<KS, SQ, XT, false, 8, 2, 64, 4>(ZARGS);
        if (g_stats_z_tv4 && nct % 4 == 0) return launch_z<KS, SQ, XT, false, 8, 4, 64>(ZARGS);
    } else if (g_stats_z_depth_em == 4) return launch_z<KS, SQ, XT, false, 8, 2, 64, 4>(ZARGS);
    return launch_z<KS, SQ, XT, false, 8, 2, 64>(ZARGS);
}

// scale multiplies every posterior (the EM frame weight); prune_thr > 0: groups of 4 frames x 32
// Gaussians whose posteriors are all below prune_thr are skipped (opt-in, see ctx.h)
int gmmk_stats_z(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, int nct, const double *zbuf,
                 long nfb, const int *eit, const double *inv, const int *efin, double scale, const long *seg_begin, int nseg,
                 double *out0, double *out1, int mode, int accum, double prune_thr)
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
