// gmm_kernels.hip -- diagonal-GMM kernels for gfx950 (MI355X): log-likelihood, top-C selection,
// full-posterior sufficient statistics (EM and Baum-Welch N/F), frame moments.
//
// The frame x Gaussian logit
//     z_tc = log w_c + log cst_c - 1/2 sum_d (x_td - mu_cd)^2 iv_cd
//          = a_c + sum_d x_td (mu_cd iv_cd) + sum_d x_td^2 (-iv_cd / 2)
// is a [T x 2D] . [2D x C] contraction; the statistics sum_t g_tc [1 | x_t | x_t^2] are a second
// one, [C x T] . [T x (1+2D)].  Both run on v_mfma_f64_16x16x4_f64 -- the D-layout of the first
// (lane = Gaussian column, registers = frame rows) is exactly the A-operand layout of the second,
// so posteriors never leave registers and the T x C matrix is never materialised.
// The top-C kernels evaluate the direct form (x-mu)^2 iv on the VALU like the reference does.
#include <atomic>
#include "devutil.h"
#include "lds_attr.h"
#include "gmm_kernels.h"

typedef double d2v __attribute__((ext_vector_type(2)));

// -------------------------------------------------------------------------------------------
// Model packing
// -------------------------------------------------------------------------------------------
// a_c = log w_c - D/2 log 2pi + 1/2 sum log iv - 1/2 sum mu^2 iv   (log(w cst) - mu'S^-1mu/2)
__global__ void k_gmm_const(int C, int Cp, int D, const double *__restrict__ w,
                            const double *__restrict__ mean, const double *__restrict__ iv,
                            double *__restrict__ a, double *__restrict__ logwcst)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cp) return;
    if (c >= C) { a[c] = GMMIV_NEG_BIG; logwcst[c] = GMMIV_NEG_BIG; return; }
    double sl = 0.0, sm = 0.0;
    for (int d = 0; d < D; ++d) {
        double v = iv[(size_t)c * D + d], m = mean[(size_t)c * D + d];
        sl += log(v);
        sm += m * m * v;
    }
    double lw = (w[c] > 0.0) ? log(w[c]) : GMMIV_NEG_BIG;
    double lc = lw - 0.5 * D * 1.8378770664093454836 + 0.5 * sl; // log(2 pi)
    lc = fmax(lc, GMMIV_NEG_BIG);
    logwcst[c] = lc;
    a[c] = fmax(lc - 0.5 * sm, GMMIV_NEG_BIG);
}

// Pt[ct][row][lane]: B operands of the logit GEMM in MFMA lane order (lane = 16 q + j:
// Gaussian c = 16 ct + j, contraction index k = 4 s + q).
//   rows 0..KS-1      mu iv      (x part)        rows KS..2KS-1   -iv/2   (x^2 part)
//   row 2KS           a_c replicated over q (accumulator init of the LLK kernel)
//   row 2KS+1         const step of the statistics kernel: q=0 -> a_c, q=1 -> -1, else 0
__global__ void k_gmm_pack(int C, int D, int KS, int nct, const double *__restrict__ mean,
                           const double *__restrict__ iv, const double *__restrict__ a,
                           double *__restrict__ Pt)
{
    const int NR = 2 * KS + 2;
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)nct * NR * 64;
    if (e >= total) return;
    int lane = e & 63;
    int row = (e >> 6) % NR;
    int ct = (e >> 6) / NR;
    int j = lane & 15, q = lane >> 4;
    int c = ct * 16 + j;
    double v = 0.0;
    if (row < 2 * KS) {
        int s = row < KS ? row : row - KS;
        int k = 4 * s + q;
        if (c < C && k < D) {
            double ivv = iv[(size_t)c * D + k];
            v = row < KS ? mean[(size_t)c * D + k] * ivv : -0.5 * ivv;
        }
    } else if (row == 2 * KS) {
        v = c < C ? fmax(a[c], GMMIV_PAD_LOGIT) : GMMIV_PAD_LOGIT;
    } else {
        v = q == 0 ? (c < C ? fmax(a[c], GMMIV_PAD_LOGIT) : GMMIV_PAD_LOGIT) : (q == 1 ? -1.0 : 0.0);
    }
    Pt[e] = v;
}

// transposed copies for the VALU top-C kernel: meanT[d][Cp], ivT[d][Cp] (pads: mean 0, iv 0)
__global__ void k_gmm_transpose(int C, int Cp, int D, const double *__restrict__ mean,
                                const double *__restrict__ iv, double *__restrict__ meanT,
                                double *__restrict__ ivT)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)D * Cp) return;
    int c = e % Cp, d = e / Cp;
    meanT[e] = c < C ? mean[(size_t)c * D + d] : 0.0;
    ivT[e] = c < C ? iv[(size_t)c * D + d] : 0.0;
}

// -------------------------------------------------------------------------------------------
// K1: per-frame log-sum-exp over all Gaussians (MFMA).  One workgroup = 8 waves x 32 frames.
// Frame operands stay in registers; the packed model streams through a double-buffered LDS
// tile (2 c-tiles = 32 Gaussians per stage), by LDS-DMA when use_glds.
// -------------------------------------------------------------------------------------------
// WZ: also leave every SCALED LIKELIHOOD e[t][c] = exp(z[t][c]) * 2^-E in zbuf for the statistics
// kernel that follows (stats_z.hip) -- the log-sum-exp needs these exponentials anyway, so the
// statistics kernel gets its posteriors with one multiply: gamma = e * 2^(E - Efin) / S_t.  E is the
// running binary exponent of the frame's row at that tile pair (eit[tile pair][frame]); Efin and
// 1 / S_t (the sum is S_t 2^Efin) are written per frame at the end.  zbuf keeps the register layout
// of the MFMA result = A-operand layout of the statistics MFMA: 2 KB blocks
// [Gaussian tile ct][16-frame block fb][lane][4 rows], so both sides move 32 contiguous bytes per lane.
// MODE 2 (TC), the world pass of ComputeTest with the top-C' candidates collected IN the epilogue (no likelihood round trip
// through HBM): every logit that reaches the frame's running threshold th is appended, as a 16-byte record (logit, Gaussian),
// to the frame's candidate list cand[t][0..TOPC_CAP) through a per-frame LDS counter; th = the smallest of the 16 per-lane
// running maxima of the frame's row (16 distinct Gaussians reach it, so the C' <= 16 largest logits of the frame all do), as a
// float rounded DOWN, refreshed after every stage with four v_min_f32 DPP steps per row; it only ever rises, so every logit
// >= the FINAL threshold is in the list.  The sum of the likelihoods that were NOT appended is kept like the log-sum-exp of the
// other modes (sacc 2^E, all exponentials evaluated); k_topc_rank (topc_z.hip) finishes: filter by the final threshold,
// direct-form logits of the survivors, ranking, remainder.  zbuf = the record array, eit = the per-frame candidate counts,
// inv_out = the non-appended sums (2^-Efin), lse_out = the final thresholds.
#define TOPC_CAP 256
// A frame whose largest w_c lk_c = t 2^E (t in [1, 2)) has E < -1075 is a zero-likelihood frame: every term of the reference's
// linear-domain sum rounds to 0 in fp64 (largest logit below GMMIV_ZERO_LLK = log 2^-1075)
#define GMMIV_MIN_FRAME_EXP (-1076)
// K1_ABL (compile-time, timing experiments only -- tools/k1_ablate.sh; results are wrong when != 0): 1 = no epilogue, 2 = exp table read
// from one address, 4 = no likelihood stores, 8 = no DPP row maximum, 16 = no staging / barrier, 32 = no stores of the running exponents; TC mode: 64 = no hit test / append,
// 128 = no lane maxima / threshold refresh, 256 = x^2 operands not recomputed (x fed twice)
#ifndef K1_ABL
#define K1_ABL 0
#endif
// K1_PLAIN_SHARED 1: the plain log-likelihood (MODE 0) runs the epilogue of the stored-likelihood variant without its stores -- one
// exponent per frame ROW (DPP row maximum), every exponential evaluated, no skip logic; 0: per-lane exponents, pairs 57 binades
// below the row's maximum skipped bit-exactly
#ifndef K1_PLAIN_SHARED
#define K1_PLAIN_SHARED 1
#endif
template <int KS, typename XT, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void k_llk_mfma(const void *__restrict__ x, long T, long ldx, int D,
                                                  const double *__restrict__ Pt, int nct,
                                                  double *__restrict__ lse_out, int use_glds, int dbg,
                                                  double *__restrict__ zbuf, long nfb, int *__restrict__ eit,
                                                  double *__restrict__ inv_out, int *__restrict__ efin_out)
{
    constexpr bool WZ = MODE == 1, TC = MODE == 2;
    // dbg (timing experiments only, results are wrong when != 0): 1 = no log-sum-exp epilogue,
    // 2 = additionally no per-tile staging / barrier, 3 = additionally B operands not re-read from LDS
    constexpr int NR = 2 * KS + 2;
    constexpr int GT = 2;
    constexpr int TILE_D = GT * NR * 64;       // doubles per LDS stage
    constexpr int PIECES = TILE_D * 8 / 1024;  // 1 KiB pieces per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *buf0 = (double *)smem;
    double *buf1 = buf0 + TILE_D;
    double *etab = buf1 + TILE_D; // exp table, GEXP_TAB_N entries
    int *ccnt = (int *)(etab + GEXP_TAB_N); // TC: candidates appended per frame of the workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const long tb = (long)blockIdx.x * (NW * 32) + wave * 32;
    gexp_tab_init(etab, tid, NW * 64);
    if (TC) ccnt[tid >> 1] = 0; // NW * 32 counters

    // TC keeps only x in registers and squares it inside the MFMA phase (60 extra v_mul_f64 per stage, ~4 % of the MFMA time):
    // its epilogue state (thresholds, lane maxima) does not fit next to 120 operand registers -- with them the compiler spilled
    // 140 VGPRs and reloaded operands from scratch inside the MFMA loop (2x slower)
    constexpr int NA = TC ? KS : 2 * KS;
    double A[2][NA];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long t = tb + h * 16 + i16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + q;
            double v = 0.0;
            if (t < T && k < D) v = feat_load<XT>::get(x, t * ldx + k);
            A[h][s] = v;
            if (!TC) A[h][NA - KS + s] = v * v;
        }
    }

    const int ntiles = nct / GT; // nct is padded to a multiple of GT by the host
    auto stage = [&](double *dst, int tile) {
        const char *src = (const char *)(Pt + (size_t)tile * TILE_D);
        if (use_glds) {
            if (PIECES % NW == 0) { // fixed trip count: straight-line code in the tile loop
#pragma unroll
                for (int j = 0; j < PIECES / NW; ++j) {
                    const int p = wave + NW * j;
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void *)((char *)dst + p * 1024), 16, 0, 0);
                }
            } else {
                for (int p = wave; p < PIECES; p += NW)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                        (__attribute__((address_space(3))) void *)((char *)dst + p * 1024), 16, 0, 0);
            }
        } else {
            for (int p = wave; p < PIECES; p += NW) {
                uint4 v = *(const uint4 *)(src + p * 1024 + lane * 16);
                *(uint4 *)((char *)dst + p * 1024 + lane * 16) = v;
            }
        }
    };

    // running sum per (lane, frame row): sum_c exp(z_c) = sacc * 2^E, nref = largest binary exponent seen
    double sacc[2][4];
    int E[2][4], nref[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sacc[h][r] = 0.0; E[h][r] = -(1 << 30); nref[h][r] = -(1 << 30); }

    // The two waves that share a SIMD (w and w + 4 of an 8-wave workgroup) run the two phases of a
    // tile in opposite order: waves 0..3 do the logit MFMAs of tile tl and then its log-sum-exp
    // epilogue, waves 4..7 first do the epilogue of the PREVIOUS tile (their accumulators are still in
    // registers) and then the MFMAs of tile tl.  Between two barriers every SIMD therefore has one wave
    // in its MFMA phase while the other is in its VALU phase: VALU work of a different wave overlaps
    // the fp64 MFMA pipe partially (tools/mfma_probe), work of the same wave never does.
    const bool late = NW == 8 && ((wave >> 1) & 1);
    // TC: threshold (row-uniform) and running maximum (per lane) of every frame row
    float thf[2][4], lmaxf[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) { thf[h][r] = -__builtin_inff(); lmaxf[h][r] = -__builtin_inff(); }
    d4 acc[GT][2];
    auto mfma_phase = [&](const double *cur) __attribute__((always_inline)) {
    // TC: x^2 = fma(x, x, zv) with zv an OPAQUE zero redefined in every stage -- a plain x * x is loop invariant and gets hoisted
    // out of the tile loop (60 more live registers: spills); an inline-asm multiply stays put but hides its VALU -> MFMA
    // operand hazard from the compiler (wrong logits in the second half of the wave: the parity tests caught it)
    double zv = 0.0;
    if (TC) asm volatile("" : "+v"(zv));
#pragma unroll
    for (int g = 0; g < GT; ++g) {
        const double a = cur[(g * NR + 2 * KS) * 64 + lane];
        acc[g][0] = (d4){a, a, a, a};
        acc[g][1] = acc[g][0];
    }
#pragma unroll
    for (int s = 0; s < 2 * KS; ++s) {
        const double b0 = cur[(0 * NR + s) * 64 + lane];
        const double b1 = cur[(1 * NR + s) * 64 + lane];
        double a0, a1;
        if (TC && s >= KS && !(K1_ABL & 256)) { a0 = __builtin_fma(A[0][s - KS], A[0][s - KS], zv); a1 = __builtin_fma(A[1][s - KS], A[1][s - KS], zv); }
        else if (TC && s >= KS) { a0 = A[0][s - KS]; a1 = A[1][s - KS]; }
        else { a0 = A[0][s < NA ? s : 0]; a1 = A[1][s < NA ? s : 0]; }
        acc[0][0] = MFMA_F64(a0, b0, acc[0][0]);
        acc[1][0] = MFMA_F64(a0, b1, acc[1][0]);
        acc[0][1] = MFMA_F64(a1, b0, acc[0][1]);
        acc[1][1] = MFMA_F64(a1, b1, acc[1][1]);
    }
    };
    auto epi_phase = [&](int te) __attribute__((always_inline)) {
        if (K1_ABL & 1) {
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(acc[g][h]));
            return;
        }
        // Online log-sum-exp per (lane, frame row) with an INTEGER reference: the sum is kept as
        // sacc * 2^E.  exp(z) = t * 2^n (t in [1,2)) is added as ldexp(t, n - E); when a logit's n
        // exceeds E by 64 or more the reference moves with one ldexp (no exp, no fp64 compare
        // chain).  A pair whose n is 57 or more below the largest n seen in its row is skipped:
        // sacc * 2^E >= 2^nref already, so such a term (both of its terms together) is < 2^-54 of the sum -- skipping is bit-exact.
        // All branches are wave-uniform (ballots): every VALU instruction here is MFMA time.
        if (TC) {
            // refresh schedule (wave-uniform): after stages 0, 1, 3, 7, 15, 31, ... and the last.  A refresh is ~400 VALU instructions per
            // wave (eight rows x six knock-out rounds), an appended record far less: round 2 refreshed 21 times in 64 stages (every stage
            // up to 8, every 2nd up to 16, every 4th up to 32, every 8th after: 70 records per frame, TC_SCHED 0); measured per 10^6
            // frames, log-likelihood kernel + ranking: 21 refreshes 11.19 + 1.11 ms (70 records), 15: 10.93 + 1.12 (73), 12: 10.80 + 1.12
            // (74), these 7: 10.50 + 1.17 (87), the same without stage 0: 11.3 + 1.2 (111), 5 refreshes: 13.9 + 1.4 (134).
#ifndef TC_SCHED
#define TC_SCHED 1
#endif
            const bool refresh = TC_SCHED == 0 ? (te < 8 || (te < 16 && (te & 1) == 1) || (te < 32 && (te & 3) == 3) || (te & 7) == 7 || te == ntiles - 1)
                                               : ((te & (te + 1)) == 0 || te == ntiles - 1);
            const int kth = dbg > 0 ? dbg : 16;          // TC: the launcher passes ctop here
            // one half (16 frames) at a time: the temporaries of 4 rows, not 8, are live
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int nm[4], k0[4], k1[4];
                bool hit0[4], hit1[4];
                bool grow = false;
                if (!(K1_ABL & 128))
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lmaxf[h][r] = fmaxf(lmaxf[h][r], (float)fmax(acc[0][h][r], acc[1][h][r])); // rounding to nearest: the push-down covers it
                // Threshold refresh (wave-uniform schedule above), BEFORE
                // this stage's logits are tested: th = the ctop-th largest of the row's 16 lane maxima -- ctop distinct Gaussians
                // reach it, so the ctop largest logits of the frame do -- pushed down by 2^-17 (relative): that covers the float
                // roundings and leaves a gap of about 4e-6 |z| to the weakest of them, which k_topc_rank's margin check needs.
                // Between refreshes the stale (lower) threshold only lets more candidates in.
                if (refresh && !(K1_ABL & 128)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // Order-preserving integer keys (negative floats: magnitude bits flipped), made unique inside the row (low 4
                        // bits <- lane, at most 15 ulps either way): selection by knock-out is exact and runs on v_min/max_i32 DPP.
                        const int fb = __float_as_int(lmaxf[h][r]);
                        int key = ((fb ^ ((fb >> 31) & 0x7fffffff)) & ~15) | i16;
                        if (kth <= 8) { // the kth largest: knock out the maximum kth - 1 times
                            for (int i = 1; i < kth; ++i) {
                                const int mx = row_max_i32(key);
                                key = key == mx ? (int)0x80000000 : key;
                            }
                            key = row_max_i32(key);
                        } else {        // = the (17 - kth)-th smallest of the 16: knock out the minimum 16 - kth times
                            for (int i = kth; i < 16; ++i) {
                                const int mn = row_min_i32(key);
                                key = key == mn ? 0x7fffffff : key;
                            }
                            key = row_min_i32(key);
                        }
                        const float f = __int_as_float(key ^ ((key >> 31) & 0x7fffffff));
                        thf[h][r] = __builtin_fmaf(-__builtin_fabsf(f), 0x1p-17f, f); // 15 ulps of key + rounding, then the margin
                    }
                }
                // hits of the four rows, ONE counter update per row (both tiles), all four in flight before the first is used
                int slot[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double thd = (double)thf[h][r];
                    hit0[r] = !(K1_ABL & 64) && acc[0][h][r] >= thd;
                    hit1[r] = !(K1_ABL & 64) && acc[1][h][r] >= thd;
                    slot[r] = 0;
                    if (hit0[r] || hit1[r])
                        slot[r] = __hip_atomic_fetch_add(ccnt + (wave * 32 + q) + (h * 16 + 4 * r), (hit0[r] ? 1 : 0) + (hit1[r] ? 1 : 0), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (hit0[r] || hit1[r]) { // append (logit, Gaussian) records to the frame's list
                        double *rp = zbuf + 2 * ((size_t)(tb + q + h * 16 + 4 * r) * TOPC_CAP + slot[r]);
                        if (hit0[r] && slot[r] < TOPC_CAP) {
                            d2v rec; rec[0] = acc[0][h][r]; rec[1] = __longlong_as_double((long long)(32 * te + i16));
                            *(d2v *)rp = rec;
                        }
                        const int s1 = slot[r] + (hit0[r] ? 1 : 0);
                        if (hit1[r] && s1 < TOPC_CAP) {
                            d2v rec; rec[0] = acc[1][h][r]; rec[1] = __longlong_as_double((long long)(32 * te + 16 + i16));
                            *(d2v *)(rp + (hit0[r] ? 2 : 0)) = rec;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double r0, r1;
                    gexp_tab_reduce(acc[0][h][r], k0[r], r0);
                    gexp_tab_reduce(acc[1][h][r], k1[r], r1);
                    acc[0][h][r] = r0;
                    acc[1][h][r] = r1;
                    const int km = k0[r] > k1[r] ? k0[r] : k1[r];
                    nm[r] = km >> GEXP_TAB_BITS; // the row maximum only behind the branch (see the WZ epilogue below)
                    grow |= nm[r] - E[h][r] >= 64;
                }
                if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        nm[r] = row_max_i32(nm[r]);
                        if (nm[r] - E[h][r] >= 64) {
                            int sh = E[h][r] - nm[r];
                            sh = sh < -2000 ? -2000 : sh;
                            sacc[h][r] = __builtin_ldexp(sacc[h][r], sh);
                            E[h][r] = nm[r];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double e0 = gexp_tab_finish(k0[r], acc[0][h][r], E[h][r], etab);
                    const double e1 = gexp_tab_finish(k1[r], acc[1][h][r], E[h][r], etab);
                    sacc[h][r] += (hit0[r] ? 0.0 : e0) + (hit1[r] ? 0.0 : e1); // appended values leave the remainder
                }
            }
            return;
        }
        if (WZ || (MODE == 0 && K1_PLAIN_SHARED)) {
            // stored-likelihood variant: E is shared by the 16 lanes of a frame row (DPP row maximum
            // of the exponents), every pair's exponential is evaluated and kept
            // The argument reduction of the 16 exponentials comes first: its integer part is the binary
            // exponent of exp(z), so the row maximum needs no separate evaluation; the reduced arguments
            // replace the logits in the accumulator registers.
            int nm[2][4], k0[2][4], k1[2][4];
            bool grow = false;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double r0, r1;
                    gexp_tab_reduce(acc[0][h][r], k0[h][r], r0);
                    gexp_tab_reduce(acc[1][h][r], k1[h][r], r1);
                    acc[0][h][r] = r0;
                    acc[1][h][r] = r1;
                    const int km = k0[h][r] > k1[h][r] ? k0[h][r] : k1[h][r];
                    // (round 5) the row maximum of the exponents is only needed when SOME lane's exponent has outgrown its row's E by 64:
                    // "any row maximum - E >= 64" is "any lane's own exponent - E >= 64", so the eight DPP reductions (32 v_max_i32_dpp
                    // per stage and wave -- MFMA time like every VALU instruction, section 3.2 of DESIGN.md) move behind the
                    // wave-uniform branch that is taken in the first stages only.  Same E, same sums, bit for bit.  (ISA of the hot path per stage
                    // and wave after this: 120 MFMAs, 293 VALU instructions -- 160 fp64, 133 integer / move -- against 357 before; measured
                    // 26.03 -> 25.97 ms per 3.07 M frames: the reductions were not on the critical path of a 2-waves-per-SIMD kernel.)
                    nm[h][r] = km >> GEXP_TAB_BITS;
                    grow |= nm[h][r] - E[h][r] >= 64;
                }
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!(K1_ABL & 8)) nm[h][r] = row_max_i32(nm[h][r]);
                        if (nm[h][r] - E[h][r] >= 64) {
                            int sh = E[h][r] - nm[h][r];
                            sh = sh < -2000 ? -2000 : sh;
                            sacc[h][r] = __builtin_ldexp(sacc[h][r], sh);
                            E[h][r] = nm[h][r];
                        }
                    }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double e0 = gexp_tab_finish(k0[h][r], acc[0][h][r], E[h][r], (K1_ABL & 2) ? etab - (k0[h][r] & (GEXP_TAB_N - 1)) : etab);
                    const double e1 = gexp_tab_finish(k1[h][r], acc[1][h][r], E[h][r], (K1_ABL & 2) ? etab - (k1[h][r] & (GEXP_TAB_N - 1)) : etab);
                    sacc[h][r] += e0 + e1;
                    acc[0][h][r] = e0;
                    acc[1][h][r] = e1;
                }
            // The LDS-DMA of the next tile (issued at the top of this iteration) and the stores of the
            // previous iteration have had a whole iteration to land: wait for them HERE, before this
            // iteration's stores are issued, so that those stay in flight across the barrier.
            // (__syncthreads() would wait vmcnt(0) after the stores: an HBM write round trip per tile.)
            if (!WZ) return; // plain log-likelihood with the shared row exponent: nothing is stored
            if (!late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // unconditional stores: zbuf / eit cover the whole grid (nfb = 16 blocks per workgroup)
            double *zw = zbuf + ((((size_t)(te * GT)) * nfb + (tb >> 4)) * 64 + lane) * 4;
            if (K1_ABL & 4) {
#pragma unroll
                for (int g = 0; g < GT; ++g)
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(acc[g][h]));
                return;
            }
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) __builtin_nontemporal_store(acc[g][h], (d4 *)(zw + ((size_t)g * nfb + h) * 256)); // streamed: keep the model tiles in L2
            if (i16 == 0 && !(K1_ABL & 32)) {
                int *ew = eit + (size_t)te * (nfb * 16) + tb + q;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ew[h * 16 + 4 * r] = E[h][r];
            }
            return;
        }
        int nm[2][4];
        bool grow = false;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nm[h][r] = gexp_tab_exponent(fmax(acc[0][h][r], acc[1][h][r]));
                grow |= nm[h][r] - E[h][r] >= 64;
            }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nm[h][r] - E[h][r] >= 64) {
                        int sh = E[h][r] - nm[h][r];
                        sh = sh < -2000 ? -2000 : sh;
                        sacc[h][r] = __builtin_ldexp(sacc[h][r], sh);
                        E[h][r] = nm[h][r];
                    }
        }
        unsigned need = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nref[h][r] = nm[h][r] > nref[h][r] ? nm[h][r] : nref[h][r];
                need |= (__builtin_amdgcn_ballot_w64(nm[h][r] > nref[h][r] - 57) != 0 ? 1u : 0u) << (h * 4 + r);
            }
        if (__builtin_popcount(need) > 3) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sacc[h][r] += gexp_tab_scaled(acc[0][h][r], E[h][r], etab) + gexp_tab_scaled(acc[1][h][r], E[h][r], etab);
        } else if (need) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (need & (1u << (h * 4 + r)))
                        sacc[h][r] += gexp_tab_scaled(acc[0][h][r], E[h][r], etab) + gexp_tab_scaled(acc[1][h][r], E[h][r], etab);
        }
    };
    stage(buf0, 0);
    __syncthreads();
    // One tile step; the loops below call it with compile-time-known flags so that the hot body carries no
    // wave-uniform conditionals (hipcc ends a basic block -- and its instruction scheduling -- at each of them).
    auto step = [&](int tl, bool staged, bool is_late, bool first) __attribute__((always_inline)) {
        double *cur = (tl & 1) ? buf1 : buf0;
        double *nxt = (tl & 1) ? buf0 : buf1;
        if (staged && (TC || dbg < 2) && !(K1_ABL & 16)) stage(nxt, tl + 1);
        if (!is_late) {
            mfma_phase(cur);
            epi_phase(tl);
        } else {
            if (!first) epi_phase(tl - 1);
            mfma_phase(cur);
        }
        if (WZ) {
            // The LDS-DMA of the next tile must have landed before the barrier; the stores of the
            // epilogue should stay in flight across it (__syncthreads() would drain them: an HBM write
            // round trip per tile).  Early waves wait inside their epilogue, BEFORE they issue the
            // stores (the DMA was issued a whole MFMA phase earlier); late waves issued their stores a
            // whole MFMA phase ago, so draining everything here costs them nothing.
            if (is_late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(K1_ABL & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            __syncthreads();
        }
    };
    if (!late) {
        for (int tl = 0; tl + 1 < ntiles; ++tl) step(tl, true, false, false);
        if (ntiles > 0) step(ntiles - 1, false, false, false);
    } else {
        if (ntiles > 1) step(0, true, true, true);
        for (int tl = 1; tl + 1 < ntiles; ++tl) step(tl, true, true, false);
        if (ntiles > 0) step(ntiles - 1, false, true, ntiles == 1);
    }
    if (late && ntiles > 0) epi_phase(ntiles - 1);
    // combine the 16 lanes (Gaussian columns) that share a frame row: common exponent, then sum
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int Em = E[h][r];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const int oe = __shfl_xor(Em, o, 64); Em = oe > Em ? oe : Em; }
            int sh = E[h][r] - Em;
            sh = sh < -2000 ? -2000 : sh;
            double sv = __builtin_ldexp(sacc[h][r], sh);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sv += shfl_xor_f64(sv, o);
            const long t = tb + h * 16 + q + 4 * r;
            if (TC) {
                if (i16 == 0 && t < T) { lse_out[t] = (double)thf[h][r]; inv_out[t] = sv; efin_out[t] = Em; }
                continue;
            }
            if (i16 == 0 && t < T) {
                // A ZERO-LIKELIHOOD frame (include/gmmiv.h, "degenerate inputs"): the scaled sum is not a positive finite number, or
                // even the largest w_c lk_c is below 2^-1075 -- 0 in the fp64 arithmetic of the reference (far below that the integer
                // exponent of gexp_tab_reduce saturates and nothing in the row is trustworthy).  log-likelihood -inf, scale 0: every
                // posterior the statistics kernels form from it is 0.
                const bool ok = sv > 0.0 && sv < __builtin_inf() && Em > GMMIV_MIN_FRAME_EXP;
                lse_out[t] = ok ? log(sv) + (double)Em * 0.693147180559945309417 : -__builtin_inf();
                if (WZ) { inv_out[t] = ok ? 1.0 / sv : 0.0; efin_out[t] = ok ? Em : 0; }
            }
        }
    if (TC) { // candidate counts of the workgroup's frames (each wave appended to its own 32 rows only)
        __syncthreads();
        const long t = (long)blockIdx.x * (NW * 32) + tid;
        if (tid < NW * 32 && t < T) eit[t] = ccnt[tid];
    }
}

// clamp + sums: out[t] = clamp(lse[t]); partial[b] = {sum clamped, sum raw, frames with a finite raw value}.  A zero-likelihood
// frame (lse = -inf / NaN) clamps to lo like log(0) does in the reference; the RAW sum -- the EM accumulator's sum of log-likelihoods --
// and the frame count beside it skip it.
__global__ __launch_bounds__(256) void k_llk_finalize(const double *__restrict__ lse, long T, double lo,
                                                     double hi, double *__restrict__ llk_out,
                                                     double *__restrict__ partial)
{
    __shared__ double red[3][4];
    double sc = 0.0, sr = 0.0, sn = 0.0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x) {
        const double v = lse[t];
        const double c = fmin(fmax(v, lo), hi); // fmax(NaN, lo) = lo
        if (llk_out) llk_out[t] = c;
        sc += c;
        const bool fin = v > -__builtin_inf() && v < __builtin_inf();
        sr += fin ? v : 0.0;
        sn += fin ? 1.0 : 0.0;
    }
    sc = wave_sum_f64(sc);
    sr = wave_sum_f64(sr);
    sn = wave_sum_f64(sn);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sc; red[1][wave] = sr; red[2][wave] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[3 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        partial[3 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        partial[3 * blockIdx.x + 2] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

// frames of kind (2) of include/gmmiv.h "DEGENERATE INPUTS": the log-likelihood kernel has left lse = -inf (or a NaN) for them.
// Counted into the context's device counter (option "zero_llk_frames"); one atomic per block that found any -- none on ordinary data.
__global__ __launch_bounds__(256) void k_count_dead(const double *__restrict__ lse, long T, unsigned long long *__restrict__ cnt)
{
    unsigned n = 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x) {
        const double v = lse[t];
        n += (v > -__builtin_inf() && v < __builtin_inf()) ? 0u : 1u;
    }
    const unsigned long long b = __ballot(n > 0);
    if (b == 0) return; // wave-uniform
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(cnt, (unsigned long long)n);
}
int gmmk_count_dead(hipStream_t st, const double *lse, long T, unsigned long long *cnt)
{
    if (T <= 0 || !cnt) return 0;
    long nb = (T + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    k_count_dead<<<(unsigned)nb, 256, 0, st>>>(lse, T, cnt);
    return (int)hipGetLastError();
}

// dst_i += s_i * sum_b partial[b*n + i], i < 3   (single block, deterministic order)
__global__ void k_reduce_partials(const double *__restrict__ partial, int nb, int n, double s0, double s1, double s2,
                                  double *__restrict__ dst0, double *__restrict__ dst1, double *__restrict__ dst2)
{
    if (threadIdx.x < n) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partial[(size_t)b * n + threadIdx.x];
        if (threadIdx.x == 0 && dst0) *dst0 += s0 * s;
        if (threadIdx.x == 1 && dst1) *dst1 += s1 * s;
        if (threadIdx.x == 2 && dst2) *dst2 += s2 * s;
    }
}

__global__ void k_add_scalar(double *dst, double v) { *dst += v; }

// -------------------------------------------------------------------------------------------
// K2/K3: posterior statistics (MFMA).  Workgroup = 8 waves = 8 c-tiles (128 Gaussians) x one
// frame segment; frames stream through a double-buffered, XOR-swizzled LDS tile of 64 rows
// [x_0..x_{D-1}, 0.., 1, lse_t, 0..]; packed model operands live in registers.
//   mode 0 (EM):  out[seg][c][2 RL] partial sums  (cols: x | x^2 halves; col Dp = occupancy)
//   mode 1 (TV):  N[seg][c], F[seg][c][D] written directly
// -------------------------------------------------------------------------------------------
template <int KS, bool SQ, typename XT, int NW, bool PRUNE>
__global__ __launch_bounds__(NW * 64, 2) void k_stats_mfma(const void *__restrict__ x, long ldx, int D, int C,
                                                    const double *__restrict__ Pt, int nct,
                                                    const double *__restrict__ lse, double lse_shift,
                                                    const long *__restrict__ seg_begin, int nseg, int ngrp,
                                                    double *__restrict__ out0, double *__restrict__ out1,
                                                    int mode, unsigned magicD, double prune_arg)
{
    constexpr int NR = 2 * KS + 2;
    constexpr int Dp = 4 * KS;
    constexpr int RL = ((Dp + 2 + 31) / 32) * 32;
    constexpr int JT = RL / 16;
    constexpr int FT = 64;
    constexpr int NT = NW * 64;
    constexpr int NLD = (FT * Dp + NT - 1) / NT;
    constexpr int RLp = RL + 32; // padded row: additive rotation xrot(t) < 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *buf0 = (double *)smem;
    double *buf1 = buf0 + FT * RLp;
    double *etab = buf1 + FT * RLp; // 32-entry exp table
    gexp_table_init(etab, threadIdx.x);

    // XCD-aware decode: the 8 consecutive block ids that land on the 8 XCDs carry 8 different
    // segments, and all Gaussian groups of one segment share an XCD (its L2 then serves the
    // frame stream to every group).
    const int b = blockIdx.x;
    const int seg_lo = b & 7;
    const int rest = b >> 3;
    const int grp = rest % ngrp;
    const int seg = (rest / ngrp) * 8 + seg_lo;
    if (seg >= nseg) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int ct = grp * NW + wave;
    const bool active = ct < nct;

    double Pr[2 * KS + 1];
#pragma unroll
    for (int s = 0; s < 2 * KS + 1; ++s) {
        const int row = s < 2 * KS ? s : 2 * KS + 1;
        Pr[s] = active ? Pt[((size_t)ct * NR + row) * 64 + lane] : 0.0;
    }

    const long f0 = seg_begin[seg], f1 = seg_begin[seg + 1];
    const int ntiles = (int)((f1 - f0 + FT - 1) / FT);

    d4 S[JT], S2[SQ ? JT : 1];
#pragma unroll
    for (int j = 0; j < JT; ++j) S[j] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < (SQ ? JT : 1); ++j) S2[j] = (d4){0, 0, 0, 0};

    XT stg[NLD];
    double stg_lse = 0.0;
    const int npad = FT * (RL - D); // pad entries per tile (cols D..RL-1)
    // Staging plan, fixed for the whole kernel: element i of this thread is tile entry e = tid + NT i
    // = (row fr, column d).  pk[i] = fr << 16 | byte offset of (fr, d) in the LDS tile (< 65536), so
    // the per-tile work per element is one compare, one address add and the load / convert / store:
    // every VALU instruction issued here is time the fp64 MFMA pipe of this SIMD stands still.
    unsigned pk[NLD];
    const bool contig = (ldx == D);
    unsigned goff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + NT * i;
        const int fr = div_by_magic((unsigned)e, magicD), d = e - fr * D; // e / D by reciprocal
        pk[i] = fr < FT ? ((unsigned)fr << 16) | (unsigned)((fr * RLp + xrot(fr) + d) * 8) : 0xffff0000u;
        goff[i] = (unsigned)(fr * (int)ldx + d);
    }
    auto load_tile = [&](int tl) {
        const long fb = f0 + (long)tl * FT;
        const long rem = f1 - fb;
        const unsigned lim = rem >= FT ? (unsigned)FT << 16 : (unsigned)rem << 16; // rows fr < lim >> 16 exist
        const XT *xt = (const XT *)x + fb * ldx;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            XT v = 0;
            if (pk[i] < lim) v = contig ? xt[tid + NT * i] : xt[goff[i]];
            stg[i] = v;
        }
        // out-of-range rows get lse = +1e300 -> posterior exp(z - lse) = 0
        if (tid < FT) {
            const double l = (fb + tid < f1) ? lse[fb + tid] : 1.0e300;
            stg_lse = (l > -__builtin_inf() && l < 1.0e299) ? l + lse_shift : 1.0e300; // a zero-likelihood frame (lse -inf / NaN): posteriors 0
        }
    };
    auto write_tile = [&](double *dst) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (pk[i] < ((unsigned)FT << 16)) *(double *)((char *)dst + (pk[i] & 0xffffu)) = (double)feat_sane(stg[i]);
        for (int e = tid; e < npad; e += NT) { // pad columns: 1.0 at Dp, zeros elsewhere
            const int fr = e / (RL - D), d = D + (e - fr * (RL - D));
            if (d != Dp + 1) dst[fr * RLp + xrot(fr) + d] = (d == Dp) ? 1.0 : 0.0;
        }
        if (tid < FT) dst[tid * RLp + xrot(tid) + Dp + 1] = stg_lse;
    };

    // per-lane LDS offsets (doubles); everything else in the inner loops is a compile-time immediate
    const int offL = i16 * RLp + xrot(i16) + q;                         // logit GEMM A operand: row i16, col q
    const int offS = q * RLp + ((q & 1) << 4) + ((q >> 1) << 1) + i16;  // statistics GEMM B operand: row q, col i16
    if (ntiles > 0) {
        load_tile(0);
        write_tile(buf0);
    }
    __syncthreads();
    for (int tl = 0; tl < ntiles; ++tl) {
        const double *cur = (tl & 1) ? buf1 : buf0;
        double *nxt = (tl & 1) ? buf0 : buf1;
        if (tl + 1 < ntiles) load_tile(tl + 1);
        if (active) {
            const double *pL = cur + offL, *pS = cur + offS;
#pragma unroll
            for (int fs = 0; fs < FT / 16; ++fs) {
                // logits z_tc - lse_t for 16 frames x 16 Gaussians: two independent MFMA chains
                d4 zx = (d4){0, 0, 0, 0}, zq = (d4){0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = pL[fs * 16 * RLp + 4 * s];
                    zx = MFMA_F64(a, Pr[s], zx);
                    zq = MFMA_F64(a * a, Pr[KS + s], zq);
                }
                {   // const step: + a_c - lse_t
                    const double a = pL[fs * 16 * RLp + Dp];
                    zx = MFMA_F64(a, Pr[2 * KS], zx);
                }
                // statistics: S[c][j] += sum_t gamma[t][c] * row_t[j]; gamma is already in A layout.
                // row t = fs*16 + 4r + q: xrot(t) = 16 (q&1) + 2 (q>>1) + 4r  (lane part in offS).
                // Register r holds 4 frames x 16 Gaussians; when all 64 posteriors are below 2^-100
                // (prune_arg = -100 ln 2; wave-uniform test) the group contributes < 2^-100 per pair
                // to any sum -- its exp and its 8 MFMAs are skipped.  Opt-in (template PRUNE): the
                // default instantiation accumulates every pair and carries no branch.
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double arg = zx[r] + zq[r];
                    if (PRUNE && __builtin_amdgcn_ballot_w64(arg > prune_arg) == 0) continue;
                    const double gam = gexp_t(arg, etab);
#pragma unroll
                    for (int j = 0; j < JT; ++j) {
                        const double bv = pS[(fs * 16 + 4 * r) * RLp + 4 * r + 16 * j];
                        S[j] = MFMA_F64(gam, bv, S[j]);
                        if (SQ) S2[j] = MFMA_F64(gam, bv * bv, S2[j]);
                    }
                }
            }
        }
        if (tl + 1 < ntiles) write_tile(nxt);
        __syncthreads();
    }
    if (!active) return;
    // D layout: lane holds column j = 16 jt + i16, rows (Gaussians) q + 4 r
    if (mode == 0) {
        const size_t Cp = (size_t)nct * 16;
        double *o = out0 + (size_t)seg * Cp * (2 * RL);
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
        double *N = out0 + (size_t)seg * C;
        double *F = out1 + (size_t)seg * C * D;
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

// acc[occ | sx | sxx] += sum_seg partial[seg][c][...]
__global__ void k_em_reduce(const double *__restrict__ part, int nseg, int C, int Cp, int D, int Dp, int RL,
                            double *__restrict__ acc)
{
    const long n = (long)C * (1 + 2 * D);
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    int c, col;
    if (e < C) { c = (int)e; col = Dp; }
    else if (e < (long)C * (1 + D)) { long k = e - C; c = (int)(k / D); col = (int)(k % D); }
    else { long k = e - (long)C * (1 + D); c = (int)(k / D); col = RL + (int)(k % D); }
    double s = 0.0;
    for (int g = 0; g < nseg; ++g) s += part[((size_t)g * Cp + c) * (2 * RL) + col];
    acc[e] += s;
}

// MixtureStat::getEM on the device copy of the accumulator
__global__ void k_em_get(int C, int D, const double *__restrict__ acc, const double *__restrict__ prev_mean,
                         const double *__restrict__ prev_cov, double *__restrict__ w,
                         double *__restrict__ mean, double *__restrict__ cov)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * D) return;
    const int c = (int)(e / D);
    const double occ = acc[c];
    const double count = acc[(size_t)C * (1 + 2 * D) + 1];
    if (e % D == 0) w[c] = occ / count;
    if (occ > 0.0) {
        const double m = acc[C + e] / occ;
        mean[e] = m;
        cov[e] = acc[(size_t)C * (1 + D) + e] / occ - m * m;
    } else {
        mean[e] = prev_mean[e];
        cov[e] = prev_cov[e];
    }
}

// out[i][0..D) = x[idx[i]][0..D)   (frame selection: label segments / bagging, on the device)
template <typename XT>
__global__ __launch_bounds__(256) void k_gather_frames(const XT *__restrict__ x, long ldx, int D,
                                                       const long *__restrict__ idx, long n, XT *__restrict__ out)
{
    const long tot = n * D;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long i = e / D;
        const int d = (int)(e - i * D);
        out[e] = x[idx[i] * ldx + d];
    }
}

// Frame selection by RUNS: run r copies frames [src, src + len) of x to rows [dst, dst + len) of out (ld = D).  One wave per
// run (a bagged chunk is 3..7 frames, GeneralTools.cpp:455-510; the host cuts longer runs into pieces of <= 64 frames), the
// rows of a run are adjacent in both buffers, so the wave moves len * D contiguous elements -- 16 bytes per lane when both
// sides allow it.  The run table costs 24 bytes per run instead of 8 bytes per frame of an index list.
template <typename XT>
__global__ __launch_bounds__(256) void k_gather_runs(const XT *__restrict__ x, long ldx, int D, const long *__restrict__ runs,
                                                     long nrun, XT *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long nw = (long)gridDim.x * (blockDim.x >> 6);
    constexpr int V = 16 / (int)sizeof(XT);
    for (long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < nrun; r += nw) {
        const long src = runs[3 * r], dst = runs[3 * r + 1], len = runs[3 * r + 2];
        if (ldx == D) {
            const XT *s = x + src * D;
            XT *d = out + dst * D;
            const long tot = len * D;
            if ((((size_t)s | (size_t)d) & 15) == 0) {
                const long nv = tot / V;
                for (long e = lane; e < nv; e += 64) ((float4 *)d)[e] = ((const float4 *)s)[e];
                for (long e = nv * V + lane; e < tot; e += 64) d[e] = s[e];
            } else
                for (long e = lane; e < tot; e += 64) d[e] = s[e];
        } else
            for (long i = 0; i < len; ++i)
                for (int e = lane; e < D; e += 64) out[(dst + i) * D + e] = x[(src + i) * ldx + e];
    }
}

// varianceControl (TrainTools.cpp:567-587): floor first, then ceiling; counts[0/1] += hits
__global__ void k_variance_control(int C, int D, double *__restrict__ cov, double flooring, double ceiling,
                                   const double *__restrict__ cov_signal, unsigned long long *__restrict__ counts)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * D) return;
    const double cs = cov_signal[e % D];
    double cv = cov[e];
    if (cv <= flooring * cs) { cv = flooring * cs; if (counts) atomicAdd(&counts[0], 1ULL); }
    if (cv >= ceiling * cs) { cv = ceiling * cs; if (counts) atomicAdd(&counts[1], 1ULL); }
    cov[e] = cv;
}
__global__ void k_reciprocal(long n, const double *__restrict__ in, double *__restrict__ out)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) out[e] = 1.0 / in[e];
}

// -------------------------------------------------------------------------------------------
// K1t: DETERMINE_TOP_DISTRIBS (VALU, direct form like DistribGD::computeLK).  One workgroup =
// FT frames; the 4 waves split the Gaussians for the evaluation, then each wave selects for
// FT/4 frames with wave-wide arg-max rounds over the logits kept in LDS.
// -------------------------------------------------------------------------------------------
template <int FT, typename XT>
__global__ __launch_bounds__(256) void k_topc_determine(
    const void *__restrict__ x, long T, long ldx, int D, int C, int Cp, const double *__restrict__ meanT,
    const double *__restrict__ ivT, const double *__restrict__ lwc, const double *__restrict__ w, int ctop,
    int complete, double lo, double hi, int *__restrict__ idx_out, double *__restrict__ lk_out,
    double *__restrict__ nontop_lk, double *__restrict__ nontop_llk, double *__restrict__ nontop_w,
    double *__restrict__ llk_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *zs = (double *)smem;          // [FT][Cp]
    double *xs = zs + (size_t)FT * Cp;    // [FT][D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long tb = (long)blockIdx.x * FT;

    for (int e = tid; e < FT * D; e += 256) {
        const int f = e / D, d = e - f * D;
        xs[e] = (tb + f < T) ? feat_load<XT>::get(x, (tb + f) * ldx + d) : 0.0;
    }
    __syncthreads();
    for (int c = tid; c < Cp; c += 256) {
        double acc[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = 0.0;
        for (int d = 0; d < D; ++d) {
            const double mu = meanT[(size_t)d * Cp + c], iv = ivT[(size_t)d * Cp + c];
#pragma unroll
            for (int f = 0; f < FT; ++f) {
                const double dx = xs[f * D + d] - mu;
                acc[f] = __builtin_fma(dx * dx, iv, acc[f]);
            }
        }
        const double a = lwc[c];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const double z = __builtin_fma(-0.5, acc[f], a);
            zs[(size_t)f * Cp + c] = z == z ? z : -__builtin_inf(); // a NaN logit (non-finite feature) = likelihood 0
        }
    }
    __syncthreads();

    const double NINF = -__builtin_inf();
    for (int f = wave; f < FT; f += 4) {
        const long t = tb + f;
        if (t >= T) break;
        double *z = zs + (size_t)f * Cp;
        double topv = 0.0; // lane j keeps the j-th selected logit
        int topi = 0;
        double M = 0.0;
        const double TAKEN = __builtin_nan(""); // marks a selected entry: never compares as a candidate again
        for (int k = 0; k < ctop; ++k) {
            double bv = NINF;
            int bc = 0x7fffffff;
            for (int c = lane; c < C; c += 64) { // real Gaussians only: padding can never be selected
                const double v = z[c];
                if (v > bv || (v == bv && c < bc)) { bv = v; bc = c; } // likelihood 0 (-inf) is a candidate too: lowest index first
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = shfl_xor_f64(bv, o);
                const int oc = __shfl_xor(bc, o, 64);
                if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            }
            if (k == 0) M = bv;
            if (lane == k) { topv = bv; topi = bc; }
            if (lane == (bc & 63)) z[bc] = TAKEN;
        }
        // non-selected remainder and selected sum, both relative to the largest logit M (M = -inf: a zero-likelihood frame)
        const bool dead = !(M > GMMIV_ZERO_LLK); // zero-likelihood frame: the lowest indices, lk 0 (the selection loop has put them there only when every logit is -inf: do it here for all)
        double sr = 0.0;
        for (int c = lane; c < Cp; c += 64) { const double v = z[c]; sr += (v == v && !dead) ? gexp(v - M) : 0.0; }
        sr = wave_sum_f64(sr);
        double st = (lane < ctop && !dead) ? gexp(topv - M) : 0.0;
        st = wave_sum_f64(st);
        if (dead) topi = lane;
        if (lane < ctop) {
            idx_out[t * ctop + lane] = topi;
            if (lk_out) lk_out[t * ctop + lane] = dead ? 0.0 : exp(topv);
        }
        if (lane == 0) {
            const double rest_llk = (sr > 0.0 && !dead) ? M + log(sr) : NINF;
            if (nontop_llk) nontop_llk[t] = rest_llk;
            if (nontop_lk) nontop_lk[t] = (sr > 0.0 && !dead) ? exp(rest_llk) : 0.0;
            if (llk_out) {
                const double tot = complete ? st + sr : st;
                llk_out[t] = dead ? lo : fmin(fmax(M + log(tot), lo), hi);
            }
        }
        if (nontop_w) { // 1 - sum of selected weights, subtracted in selection order (TopGauss.cpp:183-186)
            double snsw = 1.0;
            for (int k = 0; k < ctop; ++k) {
                const int c = __shfl(topi, k, 64);
                snsw -= w[c];
            }
            if (lane == 0) nontop_w[t] = snsw;
        }
    }
}

// The SAME selection for shapes the LDS kernel above does not serve -- any mixtureDistribCount, vectSize and topDistribsCount (all
// free configuration keys of the reference: ComputeTest.cpp:129-215, TrainWorld.cfg): the logits of a frame live in a GLOBAL scratch
// row (zs: 4 rows of Cp doubles per workgroup, L2-resident while the frame is worked on) instead of LDS, the selected Gaussians
// are written as they are found (no per-lane result registers, so no bound of 64 on ctop; the row is never modified: round k takes
// the first entry after the previous pick in the order (logit descending, index ascending)), sums relative to the largest logit M.
// One wave per frame, four frames per workgroup.  Same rules as k_topc_determine: direct form, ties to the lowest index, a NaN
// logit is likelihood 0, a zero-likelihood frame gets the lowest indices and lk 0.  A fallback (37 G pairs/s class), not a hot path.
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_determine_big(
    const void *__restrict__ x, long T, long ldx, int D, int C, int Cp, const double *__restrict__ meanT,
    const double *__restrict__ ivT, const double *__restrict__ lwc, const double *__restrict__ w, int ctop,
    int complete, double lo, double hi, int *__restrict__ idx_out, double *__restrict__ lk_out,
    double *__restrict__ nontop_lk, double *__restrict__ nontop_llk, double *__restrict__ nontop_w,
    double *__restrict__ llk_out, double *__restrict__ zs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xs = (double *)smem; // [4][D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long tb = (long)blockIdx.x * 4;
    double *zb = zs + (size_t)blockIdx.x * 4 * Cp;
    for (int e = tid; e < 4 * D; e += 256) {
        const int f = e / D, d = e - f * D;
        xs[e] = (tb + f < T) ? feat_load<XT>::get(x, (tb + f) * ldx + d) : 0.0;
    }
    __syncthreads();
    for (int c = tid; c < Cp; c += 256) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int d = 0; d < D; ++d) {
            const double mu = meanT[(size_t)d * Cp + c], iv = ivT[(size_t)d * Cp + c];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const double dx = xs[f * D + d] - mu;
                acc[f] = __builtin_fma(dx * dx, iv, acc[f]);
            }
        }
        const double a = lwc[c];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const double z = __builtin_fma(-0.5, acc[f], a);
            zb[(size_t)f * Cp + c] = z == z ? z : -__builtin_inf();
        }
    }
    __threadfence_block();
    __syncthreads();
    const long t = tb + wave;
    if (t >= T) return;
    const double *z = zb + (size_t)wave * Cp;
    const double NINF = -__builtin_inf();
    // The selection walks the total order (logit descending, index ascending): round k takes the first entry that comes AFTER the
    // previous pick (pv, pc) -- nothing is marked in memory, the logit row is only ever read.
    double M = 0.0, st = 0.0, snsw = 1.0, pv = __builtin_inf();
    int pc = -1;
    bool dead = false;
    for (int k = 0; k < ctop; ++k) {
        double bv = NINF;
        int bc = 0x7fffffff;
        if (!dead) {
            for (int c = lane; c < C; c += 64) {
                const double v = z[c];
                const bool after = v < pv || (v == pv && c > pc);
                if (after && (v > bv || (v == bv && c < bc))) { bv = v; bc = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = shfl_xor_f64(bv, o);
                const int oc = __shfl_xor(bc, o, 64);
                if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            }
        }
        if (k == 0) { M = bv; dead = !(M > GMMIV_ZERO_LLK); }
        if (dead) { bc = k; bv = NINF; } // zero-likelihood frame: the lowest indices, likelihoods 0
        pv = bv; pc = bc;
        st += dead ? 0.0 : gexp(bv - M);
        snsw -= w[bc];
        if (lane == 0) {
            if (idx_out) idx_out[t * ctop + k] = bc;
            if (lk_out) lk_out[t * ctop + k] = dead ? 0.0 : exp(bv);
        }
    }
    double sr = 0.0; // the remainder = every entry after the last pick
    if (!dead)
        for (int c = lane; c < C; c += 64) {
            const double v = z[c];
            sr += (v < pv || (v == pv && c > pc)) ? gexp(v - M) : 0.0;
        }
    sr = wave_sum_f64(sr);
    if (lane == 0) {
        const double rest_llk = (sr > 0.0 && !dead) ? M + log(sr) : NINF;
        if (nontop_llk) nontop_llk[t] = rest_llk;
        if (nontop_lk) nontop_lk[t] = (sr > 0.0 && !dead) ? exp(rest_llk) : 0.0;
        if (llk_out) {
            const double tot = complete ? st + sr : st;
            llk_out[t] = dead ? lo : fmin(fmax(M + log(tot), lo), hi);
        }
        if (nontop_w) nontop_w[t] = snsw; // 1 - the selected weights, subtracted in selection order (TopGauss.cpp:183-186)
    }
}

// USE_TOP_DISTRIBS with more than 64 selected Gaussians: one wave per frame, the candidates in rounds of 64 (two sweeps: the
// largest logit, then the sum relative to it).  Indices outside the model are skipped, never dereferenced.
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_use_big(const void *__restrict__ x, long T, long ldx, int D,
                                                      const double *__restrict__ mean, const double *__restrict__ iv,
                                                      const double *__restrict__ lwc, int C, int ctop,
                                                      const int *__restrict__ idx, const double *__restrict__ nontop_llk,
                                                      int complete, double lo, double hi, double *__restrict__ llk_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= T) return;
    const double NINF = -__builtin_inf();
    auto logit = [&](int k) {
        const int c = k < ctop ? idx[t * ctop + k] : -1;
        if ((unsigned)c >= (unsigned)C) return NINF;
        double acc = 0.0;
        for (int d = 0; d < D; ++d) {
            const double dx = feat_load<XT>::get(x, t * ldx + d) - mean[(size_t)c * D + d];
            acc = __builtin_fma(dx * dx, iv[(size_t)c * D + d], acc);
        }
        const double z = __builtin_fma(-0.5, acc, lwc[c]);
        return z == z ? z : NINF;
    };
    const double r = (complete && nontop_llk) ? nontop_llk[t] : NINF;
    double M = r;
    for (int k0 = 0; k0 < ctop; k0 += 64) M = fmax(M, logit(k0 + lane));
    M = wave_max_f64(M);
    double s = 0.0;
    for (int k0 = 0; k0 < ctop; k0 += 64) { const double z = logit(k0 + lane); s += z > NINF ? gexp(z - M) : 0.0; }
    s = wave_sum_f64(s);
    if (lane == 0) {
        if (r > NINF) s += gexp(r - M);
        llk_out[t] = fmin(fmax(M + log(s), lo), hi);
    }
}

// K1u: USE_TOP_DISTRIBS for a client model; one wave per frame, lane j evaluates Gaussian idx[t][j].
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_use(const void *__restrict__ x, long T, long ldx, int D,
                                                  const double *__restrict__ mean, const double *__restrict__ iv,
                                                  const double *__restrict__ lwc, int C, int ctop,
                                                  const int *__restrict__ idx, const double *__restrict__ nontop_llk,
                                                  int complete, double lo, double hi, double *__restrict__ llk_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= T) return;
    double z = -__builtin_inf();
    const int c = lane < ctop ? idx[t * ctop + lane] : -1;
    const bool live = (unsigned)c < (unsigned)C; // indices outside the model are skipped, never dereferenced
    if (live) {
        double acc = 0.0;
        for (int d = 0; d < D; ++d) {
            const double dx = feat_load<XT>::get(x, t * ldx + d) - mean[(size_t)c * D + d];
            acc = __builtin_fma(dx * dx, iv[(size_t)c * D + d], acc);
        }
        z = __builtin_fma(-0.5, acc, lwc[c]);
    }
    double r = (complete && nontop_llk) ? nontop_llk[t] : -__builtin_inf();
    double M = wave_max_f64(fmax(z, r));
    double s = live ? gexp(z - M) : 0.0;
    s = wave_sum_f64(s);
    if (lane == 0) {
        if (r > -__builtin_inf()) s += gexp(r - M);
        llk_out[t] = fmin(fmax(M + log(s), lo), hi);
    }
}

// Posterior vector of every frame (computeAndAccumulateOcc + getOccVect): gamma[t][c] =
// exp(z_tc - lse_t) in the reference's direct form z = log(w cst) - 0.5 sum (x - mu)^2 iv.  One
// workgroup per frame, threads over Gaussians (coalesced reads of the transposed model).
template <typename XT>
__global__ __launch_bounds__(256) void k_posteriors(const void *__restrict__ x, long ldx, int D, int C, int Cp,
                                                    const double *__restrict__ meanT, const double *__restrict__ ivT,
                                                    const double *__restrict__ lwc, const double *__restrict__ lse,
                                                    double *__restrict__ gamma)
{
    extern __shared__ double xs[];
    const long t = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) xs[d] = feat_load<XT>::get(x, t * ldx + d);
    __syncthreads();
    double l = lse[t];
    if (!(l > -__builtin_inf())) l = __builtin_inf(); // a zero-likelihood frame: a row of zeros
    for (int c = threadIdx.x; c < C; c += 256) {
        double acc = 0.0;
        for (int d = 0; d < D; ++d) {
            const double dx = xs[d] - meanT[(size_t)d * Cp + c];
            acc = __builtin_fma(dx * dx, ivT[(size_t)d * Cp + c], acc);
        }
        const double z = __builtin_fma(-0.5, acc, lwc[c]);
        gamma[t * C + c] = z == z ? exp(z - l) : 0.0; // a NaN logit (weight 0: lwc = log 0 + ... may be NaN) is a term of likelihood 0, as in k_topc_determine_big
    }
}

// -------------------------------------------------------------------------------------------
// K4: frame moments sum x, sum x^2 (FrameAccGD).  HBM-bound stream; per-block partials.
// -------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void k_frame_moments(const void *__restrict__ x, long T, long ldx, int D,
                                                       double *__restrict__ partial)
{
    // thread -> (row slot rs, column d); columns padded to 64 per wave row
    __shared__ double red[2][4][64];
    const int d = threadIdx.x & 63, rs = threadIdx.x >> 6;
    double s = 0.0, ss = 0.0;
    for (int d0 = 0; d0 < D; d0 += 64) {
        s = 0.0; ss = 0.0;
        const int dd = d0 + d;
        if (dd < D)
            for (long t = (long)blockIdx.x * 4 + rs; t < T; t += (long)gridDim.x * 4) {
                const double v = feat_load<XT>::raw(x, t * ldx + dd); // the raw value: a NaN goes into the sums like in the reference
                s += v;
                ss = __builtin_fma(v, v, ss);
            }
        red[0][rs][d] = s;
        red[1][rs][d] = ss;
        __syncthreads();
        if (rs == 0 && dd < D) {
            partial[(size_t)blockIdx.x * 2 * D + dd] = red[0][0][d] + red[0][1][d] + red[0][2][d] + red[0][3][d];
            partial[(size_t)blockIdx.x * 2 * D + D + dd] = red[1][0][d] + red[1][1][d] + red[1][2][d] + red[1][3][d];
        }
        __syncthreads();
    }
}
// acc[e] += sum over the nb block partials, fixed order (deterministic): one workgroup per 4 columns, 64 lanes x 4 column
// slots walk the blocks (a single workgroup looping over 2048 partial rows took longer than the streaming pass itself)
__global__ __launch_bounds__(256) void k_moments_reduce(const double *__restrict__ partial, int nb, int n, double *__restrict__ acc)
{
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int e = blockIdx.x * 4 + slot;
    double s = 0.0;
    if (e < n)
        for (int b = lane; b < nb; b += 64) s += partial[(size_t)b * n + e];
    red[slot][lane] = s;
    __syncthreads();
    if (lane == 0 && e < n) {
        double t = 0.0;
        for (int l = 0; l < 64; ++l) t += red[slot][l];
        acc[e] += t;
    }
}

// -------------------------------------------------------------------------------------------
// Host-side launchers
// -------------------------------------------------------------------------------------------
#define HIPCHK(e)                                                         \
    do {                                                                  \
        hipError_t _e = (e);                                              \
        if (_e != hipSuccess) return (int)_e;                             \
    } while (0)

int gmmk_ks_for_dim(int D)
{
    if (D <= 16) return 4;
    if (D <= 32) return 8;
    if (D <= 60) return 15;
    if (D <= 80) return 20;
    return D <= GMMK_MAX_DIM ? GMMK_KS_GENERIC : 0; // no MFMA instantiation: the generic (VALU logits + fp64 GEMM statistics) paths of capi_gmm.hip
}
int gmmk_rl_for_ks(int KS) { return ((4 * KS + 2 + 31) / 32) * 32; }

int gmmk_pack_model(hipStream_t st, int C, int D, int KS, int nct, int Cp64, const double *w, const double *mean,
                    const double *iv, double *a, double *lwc, double *Pt, double *meanT, double *ivT)
{
    const int Cp = Cp64 > nct * 16 ? Cp64 : nct * 16;
    k_gmm_const<<<(Cp + 255) / 256, 256, 0, st>>>(C, Cp, D, w, mean, iv, a, lwc);
    long total = (long)nct * (2 * KS + 2) * 64;
    if (KS != GMMK_KS_GENERIC) k_gmm_pack<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(C, D, KS, nct, mean, iv, a, Pt);
    long tt = (long)D * Cp64;
    k_gmm_transpose<<<(unsigned)((tt + 255) / 256), 256, 0, st>>>(C, Cp64, D, mean, iv, meanT, ivT);
    return (int)hipGetLastError();
}

template <int KS, typename XT, int NW, int MODE>
static int launch_llk(hipStream_t st, const void *x, long T, long ldx, int D, const double *Pt, int nct,
                      double *lse, int use_glds, double *zbuf = nullptr, long nfb = 0, int *eit = nullptr, double *inv = nullptr,
                      int *efin = nullptr)
{
    constexpr int NR = 2 * KS + 2;
    const size_t lds = 2 * 2 * NR * 64 * sizeof(double) + GEXP_TAB_N * sizeof(double) + (MODE == 2 ? NW * 32 * sizeof(int) : 0); // two model stages + the exp table (+ TC: candidate counters)
    HIPCHK((gmmiv_lds_attr<k_llk_mfma<KS, XT, NW, MODE>>(lds))); // per (device, kernel): lds_attr.h
    const unsigned grid = (unsigned)((T + NW * 32 - 1) / (NW * 32));
    k_llk_mfma<KS, XT, NW, MODE><<<grid, NW * 64, lds, st>>>(x, T, ldx, D, Pt, nct, lse, use_glds & 1, use_glds >> 8, zbuf, nfb, eit, inv, efin);
    return (int)hipGetLastError();
}

// wg_waves: 4 -> two independent 4-wave workgroups per CU (their MFMA and VALU phases drift apart
// and overlap), 8 -> one 8-wave workgroup per CU (its two waves per SIMD run in barrier lockstep)
int gmmk_llk(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct,
             double *lse, int use_glds, int wg_waves)
{
    if (T <= 0) return 0;
    if (T <= 32768 && gmmiv_kopts_cur().short_calls) wg_waves = 4; // short calls: one round of 4-wave workgroups, half the time per stage (see gmmk_llk_topc)
#define CASE(K)                                                                                      \
    case K:                                                                                          \
        if (wg_waves == 8)                                                                           \
            return x_f64 ? launch_llk<K, double, 8, 0>(st, x, T, ldx, D, Pt, nct, lse, use_glds) \
                         : launch_llk<K, float, 8, 0>(st, x, T, ldx, D, Pt, nct, lse, use_glds); \
        return x_f64 ? launch_llk<K, double, 4, 0>(st, x, T, ldx, D, Pt, nct, lse, use_glds)     \
                     : launch_llk<K, float, 4, 0>(st, x, T, ldx, D, Pt, nct, lse, use_glds);
    switch (KS) {
        CASE(4) CASE(8) CASE(15) CASE(20)
    }
#undef CASE
    return -1;
}

// the same kernel, additionally leaving the scaled likelihoods of frames [0, T) in zbuf (nct * nfb
// blocks of 2 KB, nfb = 16 * ceil(T / 256): whole workgroups are written), the running exponents in
// eit[(nct / 2) * nfb * 16] and 1 / S_t, Efin per frame in inv[T], efin[T]; 8-wave workgroups
int gmmk_llk_z(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct,
               double *lse, int use_glds, double *zbuf, long nfb, int *eit, double *inv, int *efin)
{
    if (T <= 0) return 0;
    const bool small = T <= 32768 && gmmiv_kopts_cur().short_calls; // one round of 4-wave workgroups: half the time per stage (see gmmk_llk_topc); same blocks, same values
#define CASE(K)                                                                                                              \
    case K:                                                                                                                  \
        if (small)                                                                                                           \
            return x_f64 ? launch_llk<K, double, 4, 1>(st, x, T, ldx, D, Pt, nct, lse, use_glds, zbuf, nfb, eit, inv, efin) \
                         : launch_llk<K, float, 4, 1>(st, x, T, ldx, D, Pt, nct, lse, use_glds, zbuf, nfb, eit, inv, efin); \
        return x_f64 ? launch_llk<K, double, 8, 1>(st, x, T, ldx, D, Pt, nct, lse, use_glds, zbuf, nfb, eit, inv, efin)   \
                     : launch_llk<K, float, 8, 1>(st, x, T, ldx, D, Pt, nct, lse, use_glds, zbuf, nfb, eit, inv, efin);
    switch (KS) {
        CASE(4) CASE(8) CASE(15)
    }
#undef CASE
    return -1;
}

// TC: the world pass of ComputeTest; cand = 2 * TOPC_CAP doubles per frame for ceil(T / 256) * 256 frames, cnt / theta / slow /
// efin per frame (see k_llk_mfma, MODE 2).  Returns -1 when no instantiation serves KS.
int gmmk_topc_cap(void) { return TOPC_CAP; }
int gmmk_llk_topc(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct, int use_glds,
                  int ctop, double *cand, int *cnt, double *theta, double *slow, int *efin)
{
    if (T <= 0) return 0;
    if (ctop < 1 || ctop > 16) return -1;
    use_glds = (use_glds & 1) | (ctop << 8); // the kernel's dbg argument carries ctop in this mode
    // A workgroup walks the whole model for its frames (64 stages of ~10 us with two waves per SIMD): a call of up to one round of
    // workgroups takes 0.65 ms whatever its length -- ComputeTest's segments of a few thousand frames.  Short calls run 4-wave
    // workgroups (128 frames, one wave per SIMD: half the time per stage); per-frame results do not depend on the workgroup shape.
    const bool small = T <= 32768 && gmmiv_kopts_cur().short_calls;
#define CASE(K)                                                                                                                    \
    case K:                                                                                                                        \
        if (small)                                                                                                                 \
            return x_f64 ? launch_llk<K, double, 4, 2>(st, x, T, ldx, D, Pt, nct, theta, use_glds, cand, 0, cnt, slow, efin)        \
                         : launch_llk<K, float, 4, 2>(st, x, T, ldx, D, Pt, nct, theta, use_glds, cand, 0, cnt, slow, efin);        \
        return x_f64 ? launch_llk<K, double, 8, 2>(st, x, T, ldx, D, Pt, nct, theta, use_glds, cand, 0, cnt, slow, efin)            \
                     : launch_llk<K, float, 8, 2>(st, x, T, ldx, D, Pt, nct, theta, use_glds, cand, 0, cnt, slow, efin);
    switch (KS) {
        CASE(4) CASE(8) CASE(15)
    }
#undef CASE
    return -1;
}

int gmmk_llk_finalize(hipStream_t st, const double *lse, long T, double lo, double hi, double *llk_out,
                      double *partial /* >= 3*256 doubles */, double scale_c, double scale_r, double *dst_clamped,
                      double *dst_raw, double scale_n, double *dst_count)
{
    if (T <= 0) return 0;
    int nb = (int)((T + 255) / 256);
    if (nb > 256) nb = 256;
    k_llk_finalize<<<nb, 256, 0, st>>>(lse, T, lo, hi, llk_out, partial);
    k_reduce_partials<<<1, 64, 0, st>>>(partial, nb, 3, scale_c, scale_r, scale_n, dst_clamped, dst_raw, dst_count);
    return (int)hipGetLastError();
}

// dst[g][e] = sum of the rows [rb[g], rb[g + 1]) of src (row stride n), e < n: the pieces of a split utterance back into its row
__global__ void k_rows_sum_groups(long n, const int *__restrict__ rb, const double *__restrict__ src, double *__restrict__ dst)
{
    const int g = blockIdx.y, r0 = rb[g], r1 = rb[g + 1];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < (size_t)n; e += (size_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += src[(size_t)r * n + e];
        dst[(size_t)g * n + e] = s;
    }
}
int gmmk_rows_sum_groups(hipStream_t st, long n, int ngroups, const int *rb, const double *src, double *dst)
{
    if (n <= 0 || ngroups <= 0) return 0;
    const unsigned bx = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    k_rows_sum_groups<<<dim3(bx, (unsigned)ngroups), 256, 0, st>>>(n, rb, src, dst);
    return (int)hipGetLastError();
}

// Segment sums of per-frame values (ComputeTest's mean log-likelihood per segment, ComputeTest.cpp:181-199): work item i covers
// elements [item[3 i + 1], item[3 i + 2]) of row item[3 i] -- at most 8192 of them -- and leaves its sum in part[i]; a pair (row,
// segment) owns the items [pair_off[p], pair_off[p + 1]) and adds them up in item order: the result does not depend on the launch.
__global__ __launch_bounds__(256) void k_segment_partials(const double *__restrict__ v, long ld, const long *__restrict__ item,
                                                          double *__restrict__ part)
{
    __shared__ double red[4];
    const long i = blockIdx.x;
    const double *row = v + item[3 * i] * ld;
    double s = 0.0;
    for (long t = item[3 * i + 1] + threadIdx.x; t < item[3 * i + 2]; t += 256) s += row[t];
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[i] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_segment_finish(const double *__restrict__ part, const long *__restrict__ pair_off, const long *__restrict__ pair_len, long npair,
                                 double *__restrict__ out)
{
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npair) return;
    double s = 0.0;
    for (long i = pair_off[p]; i < pair_off[p + 1]; ++i) s += part[i];
    out[p] = pair_len[p] > 0 ? s / (double)pair_len[p] : 0.0;
}
int gmmk_segment_means(hipStream_t st, const double *v, long ld, const long *item, long nitem, double *part, const long *pair_off,
                       const long *pair_len, long npair, double *out)
{
    if (npair <= 0) return 0;
    if (nitem > 0) k_segment_partials<<<(unsigned)nitem, 256, 0, st>>>(v, ld, item, part);
    k_segment_finish<<<(unsigned)((npair + 127) / 128), 128, 0, st>>>(part, pair_off, pair_len, npair, out);
    return (int)hipGetLastError();
}

// TopGauss::compute, selection by likelihood mass (topGauss < 1, TopGauss.cpp:162-192): per frame, Gaussians of the sorted top
// list are taken until their cumulative likelihood passes mass * exp(llk) (test BEFORE each addition, like the reference's loop);
// a fixed count (topGauss >= 1) when mass_count > 0.  snsw = 1 - sum of the selected weights, snsl = exp(llk) - sum of the
// selected likelihoods floored at EPS_LK, both subtracted in list order.  Entries of idx past the count become -1 (never
// dereferenced by the USE kernels).  capped += frames whose mass was not reached inside the list.
__global__ __launch_bounds__(256) void k_topgauss_select(long T, int cap, double mass, int fixed_count, const double *__restrict__ w,
                                                         int *__restrict__ idx, const double *__restrict__ lk, const double *__restrict__ llk,
                                                         int *__restrict__ count, double *__restrict__ snsw, double *__restrict__ snsl,
                                                         unsigned long long *__restrict__ capped)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const double lk_tot = exp(llk[t]);
    int *ix = idx + t * cap;
    const double *lv = lk + t * cap;
    int n = 0;
    if (fixed_count > 0) n = fixed_count;
    else {
        double val = 0.0;
        for (int j = 0; j < cap; ++j) {
            if (val > mass * lk_tot) break;
            val += lv[j];
            ++n;
        }
        if (n == cap && !(val > mass * lk_tot)) atomicAdd(capped, 1ULL);
    }
    double sw = 1.0, sl = lk_tot;
    for (int j = 0; j < n; ++j) { sw -= w[ix[j]]; sl -= lv[j]; }
    for (int j = n; j < cap; ++j) ix[j] = -1;
    count[t] = n;
    snsw[t] = sw;
    snsl[t] = sl < 1e-200 ? 1e-200 : sl; // EPS_LK, TopGauss.cpp:67,190
}
int gmmk_topgauss_select(hipStream_t st, long T, int cap, double mass, int fixed_count, const double *w, int *idx, const double *lk,
                         const double *llk, int *count, double *snsw, double *snsl, unsigned long long *capped)
{
    if (T <= 0) return 0;
    k_topgauss_select<<<(unsigned)((T + 255) / 256), 256, 0, st>>>(T, cap, mass, fixed_count, w, idx, lk, llk, count, snsw, snsl, capped);
    return (int)hipGetLastError();
}

// Screening of the features (include/gmmiv.h, "degenerate inputs"): a frame with a value that is NaN, infinite or beyond 1e18 in
// magnitude (its square would leave the range in which the expanded logits are meaningful) is unusable.  flag[t] = 1 for such
// frames, *any = 1 if there is one.  One pass over x at HBM speed; element-parallel, 16-byte loads on contiguous float rows.
template <typename XT>
__global__ __launch_bounds__(256) void k_flag_frames(const XT *__restrict__ x, long T, long ldx, int D, unsigned char *__restrict__ flag,
                                                     int *__restrict__ any)
{
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    if (ldx == D && sizeof(XT) == 4 && (D & 3) == 0 && (((size_t)x) & 15) == 0) {
        const long nv = T * D / 4;
        for (long e = i0; e < nv; e += stride) {
            const float4 v = ((const float4 *)x)[e];
            const bool bad = !(fabsf(v.x) <= 1e18f) || !(fabsf(v.y) <= 1e18f) || !(fabsf(v.z) <= 1e18f) || !(fabsf(v.w) <= 1e18f);
            if (bad) { flag[(4 * e) / D] = 1; hit = true; } // D % 4 == 0: the four values belong to one frame
        }
    } else {
        const long n = T * D;
        for (long e = i0; e < n; e += stride) {
            const long t = e / D;
            const double v = (double)x[t * ldx + (e - t * D)];
            if (!(fabs(v) <= 1e18)) { flag[t] = 1; hit = true; }
        }
    }
    if (hit) *any = 1;
}
int gmmk_flag_frames(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, unsigned char *flag, int *any)
{
    if (T <= 0) return 0;
    const long n = T * D;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256 + 1);
    if (x_f64) k_flag_frames<double><<<blocks, 256, 0, st>>>((const double *)x, T, ldx, D, flag, any);
    else k_flag_frames<float><<<blocks, 256, 0, st>>>((const float *)x, T, ldx, D, flag, any);
    return (int)hipGetLastError();
}

// chunk table on the device: dst[i] = min(i * per, n), i = 0 .. nseg -- the frame ranges of the statistics kernels.  Written by a kernel so
// that a frame-consuming call with device pointers ENQUEUES only (a host table would have to be copied from memory that outlives the call).
__global__ void k_fill_chunks(long *__restrict__ dst, int nseg, long per, long n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= nseg) { const long b = (long)i * per; dst[i] = b < n ? b : n; }
}
int gmmk_fill_chunks(hipStream_t st, long *dst, int nseg, long per, long n)
{
    k_fill_chunks<<<(unsigned)(nseg + 1 + 255) / 256, 256, 0, st>>>(dst, nseg, per, n);
    return (int)hipGetLastError();
}

// number of flagged frames added to a device counter (the "screened_frames" option): one atomic per workgroup that saw one
__global__ __launch_bounds__(256) void k_count_flags(const unsigned char *__restrict__ flag, long T, unsigned long long *__restrict__ cnt)
{
    __shared__ unsigned int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    unsigned int n = 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (long)gridDim.x * blockDim.x) n += flag[t] ? 1u : 0u;
    if (n) atomicAdd(&s, n);
    __syncthreads();
    if (threadIdx.x == 0 && s) atomicAdd(cnt, (unsigned long long)s);
}
int gmmk_count_flags(hipStream_t st, const unsigned char *flag, long T, unsigned long long *cnt)
{
    if (T <= 0) return 0;
    const unsigned blocks = (unsigned)((T + 255) / 256 > 1024 ? 1024 : (T + 255) / 256);
    k_count_flags<<<blocks, 256, 0, st>>>(flag, T, cnt);
    return (int)hipGetLastError();
}

int gmmk_add_scalar(hipStream_t st, double *dst, double v)
{
    k_add_scalar<<<1, 1, 0, st>>>(dst, v);
    return (int)hipGetLastError();
}

template <int KS, bool SQ, typename XT, int NW, bool PRUNE>
static int launch_stats_p(hipStream_t st, const void *x, long ldx, int D, int C, const double *Pt, int nct,
                        const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0,
                        double *out1, int mode, double prune_arg)
{
    constexpr int RL = ((4 * KS + 2 + 31) / 32) * 32;
    const size_t lds = (2 * 64 * (RL + 32) + 32) * sizeof(double);
    HIPCHK((gmmiv_lds_attr<k_stats_mfma<KS, SQ, XT, NW, PRUNE>>(lds))); // per (device, kernel): lds_attr.h
    const int ngrp = (nct + NW - 1) / NW;
    const unsigned grid = (unsigned)(ngrp * 8 * ((nseg + 7) / 8));
    const unsigned magicD = gmmiv_div_magic(D); // floor(e/D) == umulhi(e, magicD) for e < 2^16
    k_stats_mfma<KS, SQ, XT, NW, PRUNE><<<grid, NW * 64, lds, st>>>(x, ldx, D, C, Pt, nct, lse, lse_shift, seg_begin, nseg, ngrp,
                                                     out0, out1, mode, magicD, prune_arg);
    return (int)hipGetLastError();
}

template <int KS, bool SQ, typename XT, int NW>
static int launch_stats(hipStream_t st, const void *x, long ldx, int D, int C, const double *Pt, int nct,
                        const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0,
                        double *out1, int mode, double prune_arg)
{
    if (prune_arg > -1.0e300)
        return launch_stats_p<KS, SQ, XT, NW, true>(st, x, ldx, D, C, Pt, nct, lse, lse_shift, seg_begin, nseg, out0, out1, mode, prune_arg);
    return launch_stats_p<KS, SQ, XT, NW, false>(st, x, ldx, D, C, Pt, nct, lse, lse_shift, seg_begin, nseg, out0, out1, mode, prune_arg);
}

#define STATS_ARGS st, x, ldx, D, C, Pt, nct, lse, lse_shift, seg_begin, nseg, out0, out1, mode, prune_arg
template <int KS, int NW>
static int dispatch_stats(int sq, int x_f64, hipStream_t st, const void *x, long ldx, int D, int C, const double *Pt, int nct,
                          const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0, double *out1,
                          int mode, double prune_arg)
{
    if (sq) return x_f64 ? launch_stats<KS, true, double, NW>(STATS_ARGS) : launch_stats<KS, true, float, NW>(STATS_ARGS);
    return x_f64 ? launch_stats<KS, false, double, NW>(STATS_ARGS) : launch_stats<KS, false, float, NW>(STATS_ARGS);
}

int gmmk_stats(hipStream_t st, int KS, int sq, int x_f64, const void *x, long ldx, int D, int C, const double *Pt,
               int nct, const double *lse, double lse_shift, const long *seg_begin, int nseg, double *out0,
               double *out1, int mode, int wg_waves, double prune_arg)
{
    if (nseg <= 0) return 0;
#define CASE(K)                                                                       \
    case K:                                                                           \
        return wg_waves == 8 ? dispatch_stats<K, 8>(sq, x_f64, STATS_ARGS)            \
                             : dispatch_stats<K, 4>(sq, x_f64, STATS_ARGS);
    switch (KS) {
        CASE(4) CASE(8) CASE(15) CASE(20)
    }
#undef CASE
    return -1;
}

int gmmk_em_reduce(hipStream_t st, const double *part, int nseg, int C, int Cp, int D, int KS, double *acc)
{
    const long n = (long)C * (1 + 2 * D);
    k_em_reduce<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(part, nseg, C, Cp, D, 4 * KS, gmmk_rl_for_ks(KS), acc);
    return (int)hipGetLastError();
}

int gmmk_em_get(hipStream_t st, int C, int D, const double *acc, const double *prev_mean, const double *prev_cov,
                double *w, double *mean, double *cov)
{
    const long n = (long)C * D;
    k_em_get<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(C, D, acc, prev_mean, prev_cov, w, mean, cov);
    return (int)hipGetLastError();
}

int gmmk_gather_frames(hipStream_t st, int x_f64, const void *x, long ldx, int D, const long *idx, long n, void *out)
{
    if (n <= 0) return 0;
    const long tot = n * D;
    const unsigned blocks = (unsigned)((tot + 255) / 256 > 8192 ? 8192 : (tot + 255) / 256);
    if (x_f64) k_gather_frames<double><<<blocks, 256, 0, st>>>((const double *)x, ldx, D, idx, n, (double *)out);
    else k_gather_frames<float><<<blocks, 256, 0, st>>>((const float *)x, ldx, D, idx, n, (float *)out);
    return (int)hipGetLastError();
}

int gmmk_gather_runs(hipStream_t st, int x_f64, const void *x, long ldx, int D, const long *runs, long nrun, void *out)
{
    if (nrun <= 0) return 0;
    const long wgs = (nrun + 3) / 4;
    const unsigned blocks = (unsigned)(wgs > 16384 ? 16384 : wgs);
    if (x_f64) k_gather_runs<double><<<blocks, 256, 0, st>>>((const double *)x, ldx, D, runs, nrun, (double *)out);
    else k_gather_runs<float><<<blocks, 256, 0, st>>>((const float *)x, ldx, D, runs, nrun, (float *)out);
    return (int)hipGetLastError();
}

int gmmk_variance_control(hipStream_t st, int C, int D, double *cov, double flooring, double ceiling,
                          const double *cov_signal, unsigned long long *counts)
{
    const long n = (long)C * D;
    k_variance_control<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(C, D, cov, flooring, ceiling, cov_signal, counts);
    return (int)hipGetLastError();
}
int gmmk_reciprocal(hipStream_t st, long n, const double *in, double *out)
{
    k_reciprocal<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(n, in, out);
    return (int)hipGetLastError();
}

int gmmk_topc_frames_per_block(int Cp64, int D)
{
    // LDS budget 160 KiB: FT * (Cp + D) doubles
    if ((size_t)8 * (Cp64 + D) * 8 <= 150 * 1024) return 8;
    if ((size_t)4 * (Cp64 + D) * 8 <= 150 * 1024) return 4;
    return 0;
}

template <int FT, typename XT>
static int launch_topc(hipStream_t st, const void *x, long T, long ldx, int D, int C, int Cp, const double *meanT,
                       const double *ivT, const double *lwc, const double *w, int ctop, int complete, double lo,
                       double hi, int *idx, double *lk, double *nlk, double *nllk, double *nw, double *llk)
{
    const size_t lds = (size_t)FT * (Cp + D) * sizeof(double);
    HIPCHK((gmmiv_lds_attr<k_topc_determine<FT, XT>>(lds))); // per (device, kernel): lds_attr.h
    const unsigned grid = (unsigned)((T + FT - 1) / FT);
    k_topc_determine<FT, XT><<<grid, 256, lds, st>>>(x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi,
                                                     idx, lk, nlk, nllk, nw, llk);
    return (int)hipGetLastError();
}

int gmmk_topc_determine(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp,
                        const double *meanT, const double *ivT, const double *lwc, const double *w, int ctop,
                        int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                        double *nw, double *llk)
{
    if (T <= 0) return 0;
    const int ft = gmmk_topc_frames_per_block(Cp, D);
    if (ft == 8)
        return x_f64 ? launch_topc<8, double>(st, x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk)
                     : launch_topc<8, float>(st, x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk);
    if (ft == 4)
        return x_f64 ? launch_topc<4, double>(st, x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk)
                     : launch_topc<4, float>(st, x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk);
    return -1;
}

// ---- generic statistics (a vectSize without an MFMA instantiation): S = gamma^T [x | 1 | x^2] by the fp64 GEMM ----
// Xa[t] = [x_t(D) | 1 | x_t^2(D) | 0] (sq, NC = 2 D + 2) or [x_t(D) | 1] padded to an even NC = D + 2 - (D & 1) ... the caller passes NC
template <typename XT>
__global__ void k_build_xa(const void *__restrict__ x, long ldx, int D, long n, int sq, int NC, double *__restrict__ Xa)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * NC) return;
    const long t = e / NC;
    const int j = (int)(e - t * NC);
    double v = 0.0;
    if (j < D) v = feat_load<XT>::get(x, t * ldx + j);
    else if (j == D) v = 1.0;
    else if (sq && j <= 2 * D) { const double xv = feat_load<XT>::get(x, t * ldx + (j - D - 1)); v = xv * xv; }
    Xa[e] = v;
}
int gmmk_build_xa(hipStream_t st, int x_f64, const void *x, long ldx, int D, long n, int sq, int NC, double *Xa)
{
    if (n <= 0) return 0;
    const long tot = n * NC;
    if (x_f64) k_build_xa<double><<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(x, ldx, D, n, sq, NC, Xa);
    else k_build_xa<float><<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(x, ldx, D, n, sq, NC, Xa);
    return (int)hipGetLastError();
}
// acc (flat EM accumulator: occ | sum x | sum x^2 | ...) += scale * S, S [C x NC] = [sum g x | sum g | sum g x^2 | 0]
__global__ void k_scatter_em(int C, int D, int NC, const double *__restrict__ S, double scale, double *__restrict__ acc)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * (2 * D + 1)) return;
    const int c = (int)(e / (2 * D + 1)), j = (int)(e - (long)c * (2 * D + 1));
    const double v = scale * S[(size_t)c * NC + j];
    if (j < D) acc[(size_t)C + (size_t)c * D + j] += v;
    else if (j == D) acc[c] += v;
    else acc[(size_t)C + (size_t)C * D + (size_t)c * D + (j - D - 1)] += v;
}
int gmmk_scatter_em(hipStream_t st, int C, int D, int NC, const double *S, double scale, double *acc)
{
    const long tot = (long)C * (2 * D + 1);
    k_scatter_em<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(C, D, NC, S, scale, acc);
    return (int)hipGetLastError();
}
// Nrow[c] = S[c][D], Frow[c * D + d] = S[c][d]  (rows are overwritten, like the MFMA path)
__global__ void k_scatter_nf(int C, int D, int NC, const double *__restrict__ S, double *__restrict__ Nrow, double *__restrict__ Frow)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)C * (D + 1)) return;
    const int c = (int)(e / (D + 1)), j = (int)(e - (long)c * (D + 1));
    const double v = S[(size_t)c * NC + j];
    if (j < D) Frow[(size_t)c * D + j] = v;
    else Nrow[c] = v;
}
int gmmk_scatter_nf(hipStream_t st, int C, int D, int NC, const double *S, double *Nrow, double *Frow)
{
    const long tot = (long)C * (D + 1);
    k_scatter_nf<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(C, D, NC, S, Nrow, Frow);
    return (int)hipGetLastError();
}

// zs: gmmk_topc_big_scratch_doubles(T, Cp) doubles of device scratch (the logit rows of the frames in flight)
size_t gmmk_topc_big_scratch_doubles(long T, int Cp) { return (size_t)((T + 3) / 4) * 4 * (size_t)Cp; }
int gmmk_topc_determine_big(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp,
                            const double *meanT, const double *ivT, const double *lwc, const double *w, int ctop,
                            int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                            double *nw, double *llk, double *zs)
{
    if (T <= 0) return 0;
    if (ctop < 1 || ctop > C) return -2; // the selection loop indexes w[] with every pick: it must find ctop entries (not left to the callers)
    const size_t lds = (size_t)4 * D * sizeof(double);
    if (lds > 150 * 1024) return -1;
    const unsigned grid = (unsigned)((T + 3) / 4);
    if (x_f64) {
        HIPCHK((gmmiv_lds_attr<k_topc_determine_big<double>>(lds)));
        k_topc_determine_big<double><<<grid, 256, lds, st>>>(x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk, zs);
    } else {
        HIPCHK((gmmiv_lds_attr<k_topc_determine_big<float>>(lds)));
        k_topc_determine_big<float><<<grid, 256, lds, st>>>(x, T, ldx, D, C, Cp, meanT, ivT, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk, zs);
    }
    return (int)hipGetLastError();
}
int gmmk_topc_use_big(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean,
                      const double *iv, const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete,
                      double lo, double hi, double *llk)
{
    if (T <= 0) return 0;
    const unsigned grid = (unsigned)((T + 3) / 4);
    if (x_f64) k_topc_use_big<double><<<grid, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    else k_topc_use_big<float><<<grid, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    return (int)hipGetLastError();
}

int gmmk_topc_use(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean,
                  const double *iv, const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete,
                  double lo, double hi, double *llk)
{
    if (T <= 0) return 0;
    const unsigned grid = (unsigned)((T + 3) / 4);
    if (x_f64)
        k_topc_use<double><<<grid, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    else
        k_topc_use<float><<<grid, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    return (int)hipGetLastError();
}

int gmmk_posteriors(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, int C, int Cp, const double *meanT,
                    const double *ivT, const double *lwc, const double *lse, double *gamma)
{
    if (T <= 0) return 0;
    if (x_f64) k_posteriors<double><<<(unsigned)T, 256, D * sizeof(double), st>>>(x, ldx, D, C, Cp, meanT, ivT, lwc, lse, gamma);
    else k_posteriors<float><<<(unsigned)T, 256, D * sizeof(double), st>>>(x, ldx, D, C, Cp, meanT, ivT, lwc, lse, gamma);
    return (int)hipGetLastError();
}

// Streaming variant for dense rows (ldx == D) whose length is a multiple of the 16-byte vector (4 floats / 2 doubles): the
// matrix is one flat stream of 16-byte vectors, G = D / V per row.  Thread tid of block b starts at vector 256 b + tid and advances by
// gridDim.x * 256 -- the launcher makes the grid a multiple of G, so a thread always sees the same V columns (phase (256 b + tid) % G)
// and keeps V sums and V sums of squares in registers, while every wave-load is 1 KB on a 1 KB boundary (the first version used
// G * (256 / G) = 255 threads per block at D = 60: every load straddled one more cache line than it needed).
// (The scalar kernel has one 4-byte load per lane and iteration.)
template <typename XT>
__global__ __launch_bounds__(256) void k_frame_moments_vec(const XT *__restrict__ x, long nvec, int D, int nthr, double *__restrict__ partial)
{
    constexpr int V = 16 / sizeof(XT);
    typedef XT vec_t __attribute__((ext_vector_type(V)));
    __shared__ double red[256][2 * V];
    const int tid = threadIdx.x;
    double s[V], ss[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s[k] = 0.0; ss[k] = 0.0; }
    {
        const vec_t *xv = (const vec_t *)x;
        const long stride = (long)gridDim.x * 256;
        long j = (long)blockIdx.x * 256 + tid;
        // ONE load per trip: hipcc unrolls and software-pipelines this loop better than a hand-batched "4 (or 8) loads, then their
        // arithmetic" body does (tools/hbm_read_probe.hip: 6.7 against 5.9 TB/s for exactly this arithmetic)
        for (; j < nvec; j += stride) {
            const vec_t a = __builtin_nontemporal_load(xv + j);
#pragma unroll
            for (int k = 0; k < V; ++k) { const double va = (double)a[k]; s[k] += va; ss[k] = __builtin_fma(va, va, ss[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) { red[tid][k] = s[k]; red[tid][V + k] = ss[k]; }
    __syncthreads();
    // column c = V g + k lives in the threads t with (256 b + t) % G == g   (G = D / V vectors per row)
    const int G = D / V;
    (void)nthr;
    const int base = (int)(((long)blockIdx.x * 256) % G);
    for (int e = tid; e < 2 * D; e += 256) {
        const int sq = e >= D, c = sq ? e - D : e, g = c / V, k = c - g * V;
        double acc = 0.0;
        for (int t = (g - base + G) % G; t < 256; t += G) acc += red[t][sq * V + k];
        partial[(size_t)blockIdx.x * 2 * D + e] = acc;
    }
}

int gmmk_frame_moments(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, double *partial,
                       int max_blocks, double *acc)
{
    if (T <= 0) return 0;
    int nb = (int)((T + 3) / 4);
    if (nb > max_blocks) nb = max_blocks;
    const int V = x_f64 ? 2 : 4;
    if (ldx == D && D % V == 0 && D / V <= 256 && ((size_t)x % 16) == 0) { // dense rows: the flat-stream kernel
        const int G = D / V, nthr = 256;
        const long nvec = T * G;
        long want = (nvec + (long)nthr * 8 - 1) / ((long)nthr * 8); // >= 8 vectors per lane
        if (want < nb) nb = (int)(want < 1 ? 1 : want);
        nb = nb >= G ? nb / G * G : G;                              // the grid stride must be a multiple of the row length in vectors
        if (nb > max_blocks) nb = max_blocks / G * G;
        if (x_f64) k_frame_moments_vec<double><<<nb, 256, 0, st>>>((const double *)x, nvec, D, nthr, partial);
        else k_frame_moments_vec<float><<<nb, 256, 0, st>>>((const float *)x, nvec, D, nthr, partial);
    } else if (x_f64) k_frame_moments<double><<<nb, 256, 0, st>>>(x, T, ldx, D, partial);
    else k_frame_moments<float><<<nb, 256, 0, st>>>(x, T, ldx, D, partial);
    k_moments_reduce<<<(2 * D + 3) / 4, 256, 0, st>>>(partial, nb, 2 * D, acc);
    return (int)hipGetLastError();
}
