// llk_pc.hip -- K1 as a PRODUCER / CONSUMER pipeline (round 5; option "k1_pc").
//
// k_llk_mfma (gmm_kernels.hip) lets every wave alternate between its logit MFMAs and the exponentials of its own tile; the two
// waves of a SIMD are phase-shifted ("late" waves), but each of them still spends a fifth of its time in VALU code and both
// compete for the matrix pipe the rest of the time.  tools/mfma_probe showed what the hardware can do when the two kinds of work
// come from DIFFERENT waves that never change roles: 4 MFMA waves + 4 exp waves per CU finish in 12.7 ms where the same counts
// interleaved inside each wave need 26.5 and one after the other 16.3 (profiles/r01/mfma_probe.txt).  This kernel is that
// structure applied to the log-likelihood pass:
//   waves 0..3 (one per SIMD): PRODUCERS.  32 frames each as MFMA A operands in registers (x, x^2), the packed model streamed through
//       the double-buffered LDS stage exactly like k_llk_mfma; after the 120 MFMAs of a stage (32 Gaussians) the 16 logits of every
//       lane go to an LDS hand-off block (8 x ds_write_b128, conflict-free) -- no VALU work, no VMEM after the prologue;
//   waves 4..7 (their SIMD partners): CONSUMERS.  They take the previous stage's logits from the hand-off block (the MFMA result
//       layout, lane for lane), run the online log-sum-exp epilogue of the WZ / shared-exponent variant unchanged (table-driven
//       exp, one running exponent per frame row), store the scaled likelihoods / exponents, and issue the LDS-DMA of the next
//       model stage (they have the slack);
//   one workgroup barrier per stage; hand-off blocks double-buffered (written in stage k, read in k + 1, rewritten in k + 2).
// A workgroup covers 128 frames (four producer waves) instead of 256: the model is staged twice as often per frame -- 32 KB per
// ~8 k cycles and CU, nothing against L2.  LDS: 64 KB model stages + 16 KB exp table + 64 KB hand-off = 144 KB, one workgroup per CU.
// Results: the same arithmetic on the same values in the same order per frame row as k_llk_mfma<WZ> -- bitwise identical outputs
// (tests/test_gpu_gmm.py::test_k1_producer_consumer_is_bitwise_the_alternating_kernel).
#include "devutil.h"
#include "gmm_kernels.h"
#include "lds_attr.h"

typedef double d2w __attribute__((ext_vector_type(2)));

#define PC_MIN_FRAME_EXP (-1076) // = GMMIV_MIN_FRAME_EXP of gmm_kernels.hip: largest w_c lk_c below 2^-1075 -> zero-likelihood frame

// K1PC_ABL (timing experiments only, results wrong when != 0; tools/k1_pc_ablate.sh): 1 = consumers skip the exponentials and stores,
// 2 = producers skip the hand-off writes, 4 = producers skip the MFMAs, 8 = consumers skip the stores only
#ifndef K1PC_ABL
#define K1PC_ABL 0
#endif
#ifndef K1PC_PRIO
#define K1PC_PRIO 1 // producers run at raised wave priority: the matrix pipe never waits behind the partner's VALU stream
#endif

template <int KS, typename XT, int MODE>
__global__ __launch_bounds__(512, 2) void k_llk_pc(const void *__restrict__ x, long T, long ldx, int D, const double *__restrict__ Pt, int nct,
                                                   double *__restrict__ lse_out, double *__restrict__ zbuf, long nfb, int *__restrict__ eit,
                                                   double *__restrict__ inv_out, int *__restrict__ efin_out)
{
    constexpr bool WZ = MODE == 1;
    constexpr int NR = 2 * KS + 2;
    constexpr int GT = 2;
    constexpr int TILE_D = GT * NR * 64;      // doubles per LDS stage
    constexpr int PIECES = TILE_D * 8 / 1024; // 1 KiB pieces per stage
    constexpr int HAND_D = 4 * 16 * 64;       // doubles per hand-off buffer: 4 producer waves x 16 logits x 64 lanes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *buf0 = (double *)smem;
    double *buf1 = buf0 + TILE_D;
    double *etab = buf1 + TILE_D;
    double *hand = etab + GEXP_TAB_N; // [2][4][8][64] x 16 bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const bool producer = wave < 4;
    const int pw = wave & 3;
    const long tb = (long)blockIdx.x * 128 + pw * 32;
    gexp_tab_init(etab, tid, 512);
    const int ntiles = nct / GT;

    auto stage_by = [&](double *dst, int tile, int w, int nw) __attribute__((always_inline)) {
        const char *src = (const char *)(Pt + (size_t)tile * TILE_D);
#pragma unroll
        for (int j = 0; j < PIECES / 4; ++j) {
            const int p = w + nw * j;
            if (p < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)((char *)dst + p * 1024), 16, 0, 0);
        }
    };
    // first model stage: all eight waves (PIECES / 8 pieces each; PIECES is 2 NR = a multiple of 4, the guard covers the rest)
    {
        const char *src = (const char *)Pt;
        for (int p = wave; p < PIECES; p += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)((char *)buf0 + p * 1024), 16, 0, 0);
    }

    if (producer) {
        // ---------------- PRODUCER: logits of 32 frames x 32 Gaussians per stage, handed over through LDS ----------------
        double A[2][2 * KS];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long t = tb + h * 16 + i16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = 4 * s + q;
                double v = 0.0;
                if (t < T && k < D) v = feat_load<XT>::get(x, t * ldx + k);
                A[h][s] = v;
                A[h][KS + s] = v * v;
            }
        }
        __syncthreads(); // table + first stage (vmcnt(0) inside)
        if (K1PC_PRIO) __builtin_amdgcn_s_setprio(3);
        for (int tl = 0; tl < ntiles; ++tl) {
            const double *cur = (tl & 1) ? buf1 : buf0;
            d4 acc[GT][2];
#pragma unroll
            for (int g = 0; g < GT; ++g) {
                const double a = cur[(g * NR + 2 * KS) * 64 + lane];
                acc[g][0] = (d4){a, a, a, a};
                acc[g][1] = acc[g][0];
            }
            if (!(K1PC_ABL & 4))
#pragma unroll
            for (int s = 0; s < 2 * KS; ++s) {
                const double b0 = cur[(0 * NR + s) * 64 + lane];
                const double b1 = cur[(1 * NR + s) * 64 + lane];
                acc[0][0] = MFMA_F64(A[0][s], b0, acc[0][0]);
                acc[1][0] = MFMA_F64(A[0][s], b1, acc[1][0]);
                acc[0][1] = MFMA_F64(A[1][s], b0, acc[0][1]);
                acc[1][1] = MFMA_F64(A[1][s], b1, acc[1][1]);
            }
            // hand-off: block j = (g, h, half) of this wave, 16 bytes per lane, lanes contiguous (conflict-free ds_write_b128)
            d2w *hw = (d2w *)(hand + (size_t)(tl & 1) * HAND_D + (size_t)pw * (16 * 64)) + lane;
            if (K1PC_ABL & 2) {
#pragma unroll
                for (int g = 0; g < GT; ++g)
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(acc[g][h]));
            } else
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    hw[((g * 2 + h) * 2 + 0) * 64] = (d2w){acc[g][h][0], acc[g][h][1]};
                    hw[((g * 2 + h) * 2 + 1) * 64] = (d2w){acc[g][h][2], acc[g][h][3]};
                }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory"); // the consumers' last iteration
        return;
    }

    // ---------------- CONSUMER: online log-sum-exp of the partner's logits, one stage behind ----------------
    double sacc[2][4];
    int E[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) { sacc[h][r] = 0.0; E[h][r] = -(1 << 30); }
    __syncthreads();
    for (int tl = 0; tl <= ntiles; ++tl) {
        if (tl + 1 < ntiles) stage_by((tl & 1) ? buf0 : buf1, tl + 1, pw, 4); // the producers read tile tl from the other buffer
        if (tl == 0) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            continue;
        }
        const int te = tl - 1;
        d4 acc[GT][2];
        {
            const d2w *hr = (const d2w *)(hand + (size_t)(te & 1) * HAND_D + (size_t)pw * (16 * 64)) + lane;
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const d2w lo = hr[((g * 2 + h) * 2 + 0) * 64], hi = hr[((g * 2 + h) * 2 + 1) * 64];
                    acc[g][h] = (d4){lo[0], lo[1], hi[0], hi[1]};
                }
        }
        // the epilogue of k_llk_mfma<WZ> (gmm_kernels.hip), statement for statement: argument reduction of the 16 exponentials, the
        // row's binary exponent from their integer parts (DPP row maximum), rescale when it grows by 64 or more, finish, accumulate
        if (K1PC_ABL & 1) {
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(acc[g][h]));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            continue;
        }
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
                nm[h][r] = km >> GEXP_TAB_BITS; // the row maximum only behind the branch, like k_llk_mfma
                grow |= nm[h][r] - E[h][r] >= 64;
            }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    nm[h][r] = row_max_i32(nm[h][r]);
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
                const double e0 = gexp_tab_finish(k0[h][r], acc[0][h][r], E[h][r], etab);
                const double e1 = gexp_tab_finish(k1[h][r], acc[1][h][r], E[h][r], etab);
                sacc[h][r] += e0 + e1;
                acc[0][h][r] = e0;
                acc[1][h][r] = e1;
            }
        // the DMA of the next model stage (issued at the top) and the stores of the previous iteration have had a whole stage to
        // land: wait HERE, before this stage's stores go out, so that those stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WZ && !(K1PC_ABL & 8)) {
            double *zw = zbuf + ((((size_t)(te * GT)) * nfb + (tb >> 4)) * 64 + lane) * 4;
#pragma unroll
            for (int g = 0; g < GT; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h) __builtin_nontemporal_store(acc[g][h], (d4 *)(zw + ((size_t)g * nfb + h) * 256));
            if (i16 == 0) {
                int *ew = eit + (size_t)te * (nfb * 16) + tb + q;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ew[h * 16 + 4 * r] = E[h][r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
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
            if (i16 == 0 && t < T) {
                const bool ok = sv > 0.0 && sv < __builtin_inf() && Em > PC_MIN_FRAME_EXP; // zero-likelihood frame rule of k_llk_mfma
                lse_out[t] = ok ? log(sv) + (double)Em * 0.693147180559945309417 : -__builtin_inf();
                if (WZ) { inv_out[t] = ok ? 1.0 / sv : 0.0; efin_out[t] = ok ? Em : 0; }
            }
        }
}

#define HIPCHK(e)                             \
    do {                                      \
        hipError_t _e = (e);                  \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

template <int KS, typename XT, int MODE>
static int launch_pc(hipStream_t st, const void *x, long T, long ldx, int D, const double *Pt, int nct, double *lse, double *zbuf, long nfb,
                     int *eit, double *inv, int *efin)
{
    constexpr int NR = 2 * KS + 2;
    const size_t lds = (size_t)2 * 2 * NR * 64 * sizeof(double) + GEXP_TAB_N * sizeof(double) + (size_t)2 * 4 * 16 * 64 * sizeof(double);
    HIPCHK((gmmiv_lds_attr<k_llk_pc<KS, XT, MODE>>(lds)));
    const unsigned grid = (unsigned)((T + 127) / 128);
    k_llk_pc<KS, XT, MODE><<<grid, 512, lds, st>>>(x, T, ldx, D, Pt, nct, lse, zbuf, nfb, eit, inv, efin);
    return (int)hipGetLastError();
}

// zbuf == NULL: the plain log-likelihood (nothing stored); else the stored-likelihood variant.  -1: no instantiation serves KS.
int gmmk_llk_pc(hipStream_t st, int KS, int x_f64, const void *x, long T, long ldx, int D, const double *Pt, int nct, double *lse,
                double *zbuf, long nfb, int *eit, double *inv, int *efin)
{
    if (T <= 0) return 0;
#define CASE(K)                                                                                                          \
    case K:                                                                                                              \
        if (zbuf)                                                                                                        \
            return x_f64 ? launch_pc<K, double, 1>(st, x, T, ldx, D, Pt, nct, lse, zbuf, nfb, eit, inv, efin)            \
                         : launch_pc<K, float, 1>(st, x, T, ldx, D, Pt, nct, lse, zbuf, nfb, eit, inv, efin);            \
        return x_f64 ? launch_pc<K, double, 0>(st, x, T, ldx, D, Pt, nct, lse, nullptr, 0, nullptr, nullptr, nullptr)    \
                     : launch_pc<K, float, 0>(st, x, T, ldx, D, Pt, nct, lse, nullptr, 0, nullptr, nullptr, nullptr);
    switch (KS) {
        CASE(15)
    }
#undef CASE
    return -1;
}
