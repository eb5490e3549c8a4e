// topc_z.hip -- DETERMINE_TOP_DISTRIBS (ComputeTest's world-model pass: LIA_SpkDet/ComputeTest/src/ComputeTest.cpp:163-167,
// LIA_SpkTools/src/TopGauss.cpp:167-193) from the STORED scaled likelihoods of k_llk_mfma<WZ>.
//
// The first version evaluated every logit in the reference's own form a_c - 1/2 sum (x - mu)^2 iv on the VALU (180 fp64
// instructions per frame-Gaussian pair) because the selection needs an exact ordering: 37 G pairs/s, six times slower than
// the MFMA log-likelihood kernel next to it.  Only the handful of Gaussians that end up selected need that care:
//   1. k_llk_mfma<WZ> evaluates all logits on the matrix cores and leaves e = exp(z) 2^-E in HBM (the EM path's kernel,
//      unchanged; |z_mfma - z_exact| ~ 1e-12);
//   2. this kernel, one workgroup per 8 frames: the frame's C values v = e 2^(E - Efin) go to REGISTERS (32 per lane); ctop + 4
//      rounds of wave-wide arg-max pick the candidates by v; the candidates' logits are recomputed in the direct form and ranked on
//      those (ties: lowest index, like the reference's stable order); the best NON-candidate must lie 1e-6 below the weakest
//      selected logit -- a million times the MFMA error -- or the call is redone with the direct-form kernel (flag);
//   3. the non-selected remainder is the plain sum of the other v (no total - top cancellation) plus the rejected candidates.
#include "devutil.h"
#include "gmm_kernels.h"

typedef double d2 __attribute__((ext_vector_type(2)));

template <typename XT>
__global__ __launch_bounds__(256, 2) void k_topc_from_z(const void *__restrict__ x, long n, long ldx, int D, int C, int nct,
                                                        const double *__restrict__ zbuf, long nfb, const int *__restrict__ eit,
                                                        const int *__restrict__ efin, const double *__restrict__ mean,
                                                        const double *__restrict__ iv, const double *__restrict__ lwc,
                                                        const double *__restrict__ w, int ctop, int complete, double lo, double hi,
                                                        int *__restrict__ idx_out, double *__restrict__ lk_out,
                                                        double *__restrict__ nontop_lk, double *__restrict__ nontop_llk,
                                                        double *__restrict__ nontop_w, double *__restrict__ llk_out, int *__restrict__ flag)
{
    // Workgroup = 8 frames of one 16-frame likelihood block: wave w takes the frames of lane group q0 = w, registers
    // r = 2 half + j (j = 0, 1).  A frame's 2048 values sit in the 16 lanes (i16) of that group of each tile's block; here lane
    // (i16, qq) of the wave loads them for the tiles 4 m + qq: 32 values per lane and frame, IN REGISTERS -- the selection rounds
    // are register compares + one wave arg-max, nothing is re-read (the first version kept the values in LDS and re-scanned
    // them every round: one LDS latency per element and round, 56 ms per 10^6 frames, all of it in the selection).
    __shared__ double xs[8][64 + 1];
    __shared__ int ord[4][64];
    __shared__ __attribute__((aligned(16))) double lmx[4][64];
    __shared__ double thx[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, qq = lane >> 4;
    const long fb = blockIdx.x >> 1;
    const int half = blockIdx.x & 1, q0 = wave;
    for (int e = tid; e < 8 * D; e += 256) { // frame rows for the exact logits: local frame 2 w + j
        const int lf = e / D, d = e - lf * D;
        const long t = fb * 16 + (lf >> 1) + 4 * (2 * half + (lf & 1));
        xs[lf][d] = t < n ? feat_load<XT>::get(x, t * ldx + d) : 0.0;
    }
    long tj[2];
    int ef[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        tj[j] = fb * 16 + q0 + 4 * (2 * half + j);
        ef[j] = tj[j] < n ? efin[tj[j]] : 0;
    }
    double val[2][32];
    // Loads first (tile index clamped), in two batches of 16 tiles, each closed by a compiler barrier: without it hipcc moves
    // every load into the `ct < nct` arm that consumes it -- 32 basic blocks, each waiting for its own load.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        d2 e2[16];
        int e0[16], e1[16];
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            const int ct = 4 * (16 * h + mm) + qq, cc = ct < nct ? ct : nct - 1;
            e2[mm] = *(const d2 *)(zbuf + (((size_t)cc * nfb + fb) * 64 + 16 * q0 + i16) * 4 + 2 * half);
            const int *ep = eit + (size_t)(cc >> 1) * (nfb * 16) + fb * 16 + q0 + 8 * half;
            e0[mm] = ep[0];
            e1[mm] = ep[4];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            const int m = 16 * h + mm, ct = 4 * m + qq;
            const double a0 = __builtin_ldexp(e2[mm][0], e0[mm] - ef[0]), a1 = __builtin_ldexp(e2[mm][1], e1[mm] - ef[1]);
            val[0][m] = ct < nct ? a0 : -1.0;
            val[1][m] = ct < nct ? a1 : -1.0;
        }
    }
    __syncthreads();

    const double NINF = -__builtin_inf();
    const int K = ctop + 4 < C ? ctop + 4 : C; // candidates (host: K <= 64)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long t = tj[j];
        if (t >= n) continue;
        const int lf = 2 * wave + j;
        // Candidates without rescanning: theta = the K-th largest of the 64 per-lane maxima (K arg-max rounds over ONE value
        // per lane).  At least K values are >= theta, so the K largest of the frame all are: every value >= theta is a
        // candidate (typically K .. 2K of them), compacted into LDS with ballots.
        double lm;
        {
            double t16[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t16[i] = fmax(val[j][i], val[j][i + 16]);
#pragma unroll
            for (int i = 0; i < 8; ++i) t16[i] = fmax(t16[i], t16[i + 8]);
#pragma unroll
            for (int i = 0; i < 4; ++i) t16[i] = fmax(t16[i], t16[i + 4]);
            lm = fmax(fmax(t16[0], t16[2]), fmax(t16[1], t16[3]));
        }
        // K-th largest of the 64 lane maxima: every lane ranks its own value against all 64 (LDS broadcast reads, ~200
        // instructions; K rounds of wave arg-max were 1400)
        lmx[wave][lane] = lm;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int lrank = 0;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const double o = lmx[wave][i];
            lrank += (o > lm || (o == lm && i < lane)) ? 1 : 0;
        }
        if (lrank == K - 1) thx[wave] = lm;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double theta = thx[wave];
        __builtin_amdgcn_wave_barrier();
        if (!(theta > 0.0)) theta = 4.9e-324; // fewer than K lanes carry mass: every positive value is a candidate
        int nc = 0;
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const bool hit = val[j][m] >= theta;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            if (mask) {
                if (hit) {
                    const int pos = nc + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    if (pos < 64) ord[wave][pos] = 16 * (4 * m + qq) + i16;
                }
                nc += __builtin_popcountll(mask);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nc > 64) { if (lane == 0) atomicExch(flag, 1); continue; } // a pile-up of equal values: the direct-form kernel redoes the call
        // fewer stored likelihoods above 0 than Gaussians to select (a zero-likelihood frame; a frame whose other Gaussians lie more
        // than 2^-1074 below its best): the ranks beyond nc would stay unwritten -- the direct-form kernel ranks on the logits instead
        if (nc < (ctop < C ? ctop : C)) { if (lane == 0) atomicExch(flag, 1); continue; }
        const int ci = lane < nc ? ord[wave][lane] : 0x7fffffff;
        __builtin_amdgcn_wave_barrier();
        const double vnext = theta; // every non-candidate is below theta
        // exact logits of the candidates
        const bool cand = lane < nc && ci < C;
        double zc = NINF;
        if (cand) {
            // the candidate's own rows of the row-major model: contiguous, all loads of a pass in flight together (the transposed
            // copy would be D dependent strided gathers per candidate)
            const double *mu = mean + (size_t)ci * D, *vi = iv + (size_t)ci * D;
            double acc = 0.0;
            int d = 0;
            for (; d + 8 <= D; d += 8) {
                double m8[8], v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { m8[u] = mu[d + u]; v8[u] = vi[d + u]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double dx = xs[lf][d + u] - m8[u];
                    acc = __builtin_fma(dx * dx, v8[u], acc);
                }
            }
            for (; d < D; ++d) {
                const double dx = xs[lf][d] - mu[d];
                acc = __builtin_fma(dx * dx, vi[d], acc);
            }
            zc = __builtin_fma(-0.5, acc, lwc[ci]);
        }
        int rank = 0;
        for (int jj = 0; jj < nc; ++jj) {
            const double zj = readlane_f64u(zc, jj); // jj is wave-uniform: v_readlane, not a permute through LDS
            const int cj = __builtin_amdgcn_readlane(ci, jj);
            if (jj != lane && (zj > zc || (zj == zc && cj < ci))) ++rank;
        }
        const bool sel = cand && rank < ctop;
        const double M = wave_max_f64_dpp(zc);
        // weakest selected logit vs the best non-candidate (from the stored likelihoods: exp(z) = v 2^Efin)
        const double zmin = wave_min_f64_dpp(sel ? zc : __builtin_inf());
        const double lnE = (double)ef[j] * 0.6931471805599453;
        if (nc > ctop && log(vnext) + lnE > zmin - 1e-6) { if (lane == 0) atomicExch(flag, 1); }
        // remainder: non-candidates straight from the stored values, rejected candidates from their exact logits
        double sr = 0.0;
#pragma unroll
        for (int m = 0; m < 32; ++m) sr += (val[j][m] > 0.0 && val[j][m] < theta) ? val[j][m] : 0.0;
        sr = wave_sum_f64_dpp(sr);
        double srel = sr > 0.0 ? exp(log(sr) + lnE - M) : 0.0;
        srel += wave_sum_f64_dpp((cand && !sel) ? gexp(zc - M) : 0.0);
        const double st = wave_sum_f64_dpp(sel ? gexp(zc - M) : 0.0);
        if (sel) {
            idx_out[t * ctop + rank] = ci;
            if (lk_out) lk_out[t * ctop + rank] = exp(zc);
            ord[wave][rank] = ci;
        }
        if (lane == 0) {
            const double rest_llk = srel > 0.0 ? M + log(srel) : NINF;
            if (nontop_llk) nontop_llk[t] = rest_llk;
            if (nontop_lk) nontop_lk[t] = exp(rest_llk);
            if (llk_out) {
                const double tot = complete ? st + srel : st;
                llk_out[t] = fmin(fmax(M + log(tot), lo), hi);
            }
        }
        if (nontop_w) { // 1 - sum of selected weights, subtracted in selection order (TopGauss.cpp:183-186)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {
                double snsw = 1.0;
                for (int k = 0; k < ctop; ++k) snsw -= w[ord[wave][k]];
                nontop_w[t] = snsw;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

#define HIPCHK(e)                                                         \
    do {                                                                  \
        hipError_t _e = (e);                                              \
        if (_e != hipSuccess) return (int)_e;                             \
    } while (0)

// 1 when the kernel applies (a lane holds 32 values per frame: at most 128 Gaussian tiles; frame rows of at most 64 dims)
size_t gmmk_topc_z_lds(int nct, int D) { return (nct <= 128 && D <= 64) ? 1 : 0; }

int gmmk_topc_from_z(hipStream_t st, int x_f64, const void *x, long n, long ldx, int D, int C, int nct, const double *zbuf, long nfb,
                     const int *eit, const int *efin, const double *mean, const double *iv, const double *lwc,
                     const double *w, int ctop, int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                     double *nw, double *llk, int *flag)
{
    if (n <= 0) return 0;
    if (!gmmk_topc_z_lds(nct, D)) return -1;
    const size_t lds = 0;
    const unsigned grid = (unsigned)(2 * ((n + 15) / 16));
    if (x_f64)
        k_topc_from_z<double><<<grid, 256, lds, st>>>(x, n, ldx, D, C, nct, zbuf, nfb, eit, efin, mean, iv, lwc, w, ctop, complete,
                                                      lo, hi, idx, lk, nlk, nllk, nw, llk, flag);
    else
        k_topc_from_z<float><<<grid, 256, lds, st>>>(x, n, ldx, D, C, nct, zbuf, nfb, eit, efin, mean, iv, lwc, w, ctop, complete, lo,
                                                     hi, idx, lk, nlk, nllk, nw, llk, flag);
    return (int)hipGetLastError();
}

// ---- ranking of the candidates collected by k_llk_mfma<TC> (gmm_kernels.hip, MODE 2) -----------------------------------------
// One wave per frame.  The frame's list holds every Gaussian whose MFMA logit reached the running threshold of its time, in the
// order the appends happened to win their LDS counter -- first of all it is put into a CANONICAL order (ascending Gaussian index,
// through a bitmap of the indices in LDS), so that everything below, sums included, is bitwise reproducible.  Then, like
// k_topc_from_z: the survivors of the FINAL threshold theta (a superset of the C' largest: at least 16 logits reach theta) get
// their logit in the reference's direct form and are ranked on it (ties: lowest index); theta must lie 1e-6 below the weakest
// selected logit or the call is redone by the direct-form kernel (flag).  Remainder = the likelihoods k_llk_mfma never appended
// (slow 2^Efin) + the appended ones below theta (from their MFMA logits) + the rejected survivors (direct form).
#define TOPC_CAP 256
template <typename XT>
__device__ __forceinline__ void topc_rank_frame(const long t, const void *__restrict__ x, long n, long ldx, int D, int C, const double *__restrict__ cand,
                                                   const int *__restrict__ cnt, const double *__restrict__ theta,
                                                   const double *__restrict__ slow, const int *__restrict__ efin,
                                                   const double *__restrict__ mean, const double *__restrict__ iv,
                                                   const double *__restrict__ lwc, const double *__restrict__ w, int ctop, int complete,
                                                   double lo, double hi, int *__restrict__ idx_out, double *__restrict__ lk_out,
                                                   double *__restrict__ nontop_lk, double *__restrict__ nontop_llk,
                                                   double *__restrict__ nontop_w, double *__restrict__ llk_out, int *__restrict__ flag, long *__restrict__ redo,
                                                   int stats)
{
    // flag[0] = number of frames handed to the direct-form kernel (their indices in redo[]); flag[1..4] = reasons (list overflow,
    // more than 64 survivors, fewer than ctop, margin); with `stats & 1` also flag[5] / [6,7] = max / sum of the list lengths,
    // flag[8] / [10,11] of the survivor counts and flag[12] = frames whose survivors were re-evaluated in the direct form (global
    // atomics from every wave: measurement runs only)
    __shared__ double xs[4][64 + 1];
    __shared__ unsigned bmap[4][64];      // Gaussians 32 l .. 32 l + 31 of the frame's list (C <= 2048)
    __shared__ double sz[4][TOPC_CAP];    // logits in canonical order
    __shared__ int si[4][TOPC_CAP];
    __shared__ int ord[4][64];
    __shared__ double ordz[4][64];        // MFMA logits of the survivors, in the order of ord
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool force_direct = (stats & 2) != 0; // option "topc_rank_direct": every frame's survivors in the direct form (round 2)
    const bool live = t >= 0 && t < n;
    for (int d = lane; d < D; d += 64) xs[wave][d] = live ? feat_load<XT>::get(x, t * ldx + d) : 0.0;
    bmap[wave][lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!live) return;
    const double NINF = -__builtin_inf();
    const int ncr = cnt[t];
    if ((stats & 1) && lane == 0) { atomicMax(&flag[5], ncr); atomicAdd((unsigned long long *)&flag[6], (unsigned long long)ncr); }
    if (ncr > TOPC_CAP) { if (lane == 0) { redo[atomicAdd(&flag[0], 1)] = t; atomicAdd(&flag[1], 1); } return; } // list overflow: the direct-form kernel redoes the frame
    // the list length as a SCALAR: the four 64-record slices below are skipped wave-uniformly when the list ends before them (mean
    // length 68 of 256: slices 2 and 3 almost never exist, and a skipped slice costs a branch instead of its predicated
    // instructions -- the exponentials of the rejected records above all)
    const int nc = __builtin_amdgcn_readfirstlane(ncr);
    double zl[4];
    int cl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane + 64 * j;
        zl[j] = NINF; cl[j] = -1;
        if (64 * j < nc && k < nc) {
            const d2 rec = *(const d2 *)(cand + 2 * ((size_t)t * TOPC_CAP + k));
            zl[j] = rec[0];
            cl[j] = (int)__double_as_longlong(rec[1]);
            atomicOr(&bmap[wave][(cl[j] >> 5) & 63], 1u << (cl[j] & 31));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // exclusive prefix of the word populations -> canonical position of every record
    const unsigned myw = bmap[wave][lane];
    int pre = __builtin_popcount(myw);
    {
        int inc = pre;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
        pre = inc - pre;
    }
    ord[wave][lane] = pre; // reused as the prefix table until the survivors are compacted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (64 * j < nc && cl[j] >= 0) {
            const int wd = (cl[j] >> 5) & 63;
            const int pos = ord[wave][wd] + __builtin_popcount(bmap[wave][wd] & ((1u << (cl[j] & 31)) - 1u));
            sz[wave][pos] = zl[j];
            si[wave][pos] = cl[j];
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double th = theta[t];
    const int ef = efin[t];
    const double lnE = (double)ef * 0.6931471805599453;
    // canonical order from here on: lane l looks at records l, l + 64, ...; survivors (logit >= theta) are compacted in that order
    int ns = 0;
    double zrej[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        zrej[j] = NINF;
        if (64 * j >= nc) continue; // wave-uniform
        const int k = lane + 64 * j;
        const double z = k < nc ? sz[wave][k] : NINF;
        const bool hit = k < nc && z >= th;
        zrej[j] = (k < nc && !hit) ? z : NINF;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        if (hit) {
            const int pos = ns + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (pos < 64) { ord[wave][pos] = si[wave][k]; ordz[wave][pos] = z; }
        }
        ns += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if ((stats & 1) && lane == 0) { atomicMax(&flag[8], ns); atomicAdd((unsigned long long *)&flag[10], (unsigned long long)ns); }
    if (ns > 64 || ns < (ctop < C ? ctop : C)) { if (lane == 0) { redo[atomicAdd(&flag[0], 1)] = t; atomicAdd(&flag[ns > 64 ? 2 : 3], 1); } return; } // a pile-up of near-equal logits
    const int ci = lane < ns ? ord[wave][lane] : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    const bool cd = lane < ns && ci < C;
    // Ranking on the MFMA logits first (|z_mfma - z_direct| ~ 1e-12): the order -- and with it the selection -- is the order of the
    // direct form whenever no survivor lies within 1e-6 of a SELECTED one; only then (ties, duplicated Gaussians: ~1e-5 of the
    // frames of real data) are the survivors re-evaluated in the reference's direct form below.  Round 2 did that for every frame:
    // 15 survivors x 960 B of model rows gathered from L2 per frame were most of this kernel's 2.3 ms per 10^6 frames.
    double zc = cd ? ordz[wave][lane] : NINF;
    int rank = 0;
    bool near = false;
    for (int jj = 0; jj < ns; ++jj) {
        const double zj = readlane_f64u(zc, jj);
        const int cj = __builtin_amdgcn_readlane(ci, jj);
        if (jj != lane && (zj > zc || (zj == zc && cj < ci))) ++rank;
        near |= jj != lane && __builtin_fabs(zj - zc) <= 1e-6;
    }
    if (force_direct || __builtin_amdgcn_ballot_w64(cd && rank < ctop && near) != 0) { // wave-uniform
    // The survivors' logits in the reference's direct form, FOUR lanes per survivor (16 survivors per pass): lane `sub` of a
    // group takes the dimension pairs sub, sub + 4, ... of the survivor's own rows of the row-major model (16-byte loads), the
    // group's partial sums meet through two quad exchanges.  (One lane per survivor left 3/4 of the wave idle behind 120 loads.)
    for (int p0 = 0; p0 < ns; p0 += 16) {
        const int cj = p0 + (lane >> 2), sub = lane & 3;
        const int c = cj < ns ? ord[wave][cj] : 0x7fffffff;
        double acc = 0.0;
        if (c < C) {
            const double *mu = mean + (size_t)c * D, *vi = iv + (size_t)c * D, *xr = xs[wave];
            if ((D & 1) == 0) {
                for (int pr = sub; pr < (D >> 1); pr += 4) {
                    const d2 m2 = *(const d2 *)(mu + 2 * pr), v2 = *(const d2 *)(vi + 2 * pr);
                    const double dx0 = xr[2 * pr] - m2[0], dx1 = xr[2 * pr + 1] - m2[1];
                    acc = __builtin_fma(dx0 * dx0, v2[0], acc);
                    acc = __builtin_fma(dx1 * dx1, v2[1], acc);
                }
            } else {
                for (int d = sub; d < D; d += 4) {
                    const double dx = xr[d] - mu[d];
                    acc = __builtin_fma(dx * dx, vi[d], acc);
                }
            }
        }
        acc += shfl_xor_f64(acc, 1);
        acc += shfl_xor_f64(acc, 2);
        if (sub == 0 && cj < ns) sz[wave][cj] = c < C ? __builtin_fma(-0.5, acc, lwc[c]) : NINF; // sz is free again: reused for the exact logits
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    zc = cd ? sz[wave][lane] : NINF;
    rank = 0;
    for (int jj = 0; jj < ns; ++jj) {
        const double zj = readlane_f64u(zc, jj);
        const int cj = __builtin_amdgcn_readlane(ci, jj);
        if (jj != lane && (zj > zc || (zj == zc && cj < ci))) ++rank;
    }
    if ((stats & 1) && lane == 0) atomicAdd(&flag[12], 1);
    }
    const bool sel = cd && rank < ctop;
    const double M = wave_max_f64_dpp(zc);
    const double zmin = wave_min_f64_dpp(sel ? zc : __builtin_inf());
    // every Gaussian outside the survivors has an MFMA logit below theta: it must not be able to overtake the weakest selected
    if (ns > ctop && th > zmin - 1e-6) { if (lane == 0) { redo[atomicAdd(&flag[0], 1)] = t; atomicAdd(&flag[4], 1); } return; } // uniform: th, zmin are wave-wide
    // remainder relative to M: never-appended likelihoods + appended below theta + rejected survivors
    double srel = 0.0;
    {
        const double sl = slow[t];
        if (sl > 0.0) srel = exp(log(sl) + lnE - M);
    }
    double sr = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (64 * j < nc) sr += zrej[j] > NINF ? gexp(zrej[j] - M) : 0.0; // wave-uniform skip
    srel += wave_sum_f64_dpp(sr);
    srel += wave_sum_f64_dpp((cd && !sel) ? gexp(zc - M) : 0.0);
    const double st = wave_sum_f64_dpp(sel ? gexp(zc - M) : 0.0);
    const bool dead = !(M > GMMIV_ZERO_LLK); // zero-likelihood frame (include/gmmiv.h): the lowest indices, likelihoods 0, llk = lo
    if (sel) {
        idx_out[t * ctop + rank] = dead ? rank : ci;
        if (lk_out) lk_out[t * ctop + rank] = dead ? 0.0 : exp(zc);
        ord[wave][rank] = dead ? rank : ci;
    }
    if (lane == 0) {
        const double rest_llk = (srel > 0.0 && !dead) ? M + log(srel) : NINF;
        if (nontop_llk) nontop_llk[t] = rest_llk;
        if (nontop_lk) nontop_lk[t] = dead ? 0.0 : exp(rest_llk);
        if (llk_out) {
            const double tot = complete ? st + srel : st;
            llk_out[t] = dead ? lo : fmin(fmax(M + log(tot), lo), hi);
        }
    }
    if (nontop_w) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            double snsw = 1.0;
            for (int k = 0; k < ctop; ++k) snsw -= w[ord[wave][k]];
            nontop_w[t] = snsw;
        }
    }
}

// one wave per frame: frame blockIdx.x * 4 + wave
#define TOPC_RANK_PARAMS                                                                                                              \
    const void *__restrict__ x, long n, long ldx, int D, int C, const double *__restrict__ cand, const int *__restrict__ cnt,          \
        const double *__restrict__ theta, const double *__restrict__ slow, const int *__restrict__ efin,                               \
        const double *__restrict__ mean, const double *__restrict__ iv, const double *__restrict__ lwc, const double *__restrict__ w,  \
        int ctop, int complete, double lo, double hi, int *__restrict__ idx_out, double *__restrict__ lk_out,                          \
        double *__restrict__ nontop_lk, double *__restrict__ nontop_llk, double *__restrict__ nontop_w, double *__restrict__ llk_out,  \
        int *__restrict__ flag, long *__restrict__ redo, int stats
#define TOPC_RANK_FWD                                                                                                                  \
    x, n, ldx, D, C, cand, cnt, theta, slow, efin, mean, iv, lwc, w, ctop, complete, lo, hi, idx_out, lk_out, nontop_lk, nontop_llk,    \
        nontop_w, llk_out, flag, redo, stats
template <typename XT> __global__ __launch_bounds__(256) void k_topc_rank(TOPC_RANK_PARAMS)
{
    topc_rank_frame<XT>((long)blockIdx.x * 4 + (threadIdx.x >> 6), TOPC_RANK_FWD);
}
// the same routine on the frames list[0 .. *list_n) -- the ones k_topc_rank2 passed on (more than 128 records or more than 32
// survivors) -- a small fixed grid striding over them.  (A kernel of its own: with the loop around it the routine takes 138 VGPRs
// instead of 28, which cost the every-frame form half its speed when the two shared one kernel.)
template <typename XT> __global__ __launch_bounds__(256) void k_topc_rank_list(TOPC_RANK_PARAMS, const long *__restrict__ list, const int *__restrict__ list_n)
{
    const int wave = threadIdx.x >> 6;
    const long cntl = *list_n;
    for (long i0 = (long)blockIdx.x * 4; i0 < cntl; i0 += (long)gridDim.x * 4) { // workgroup-uniform trip count
        const long i = i0 + wave;
        topc_rank_frame<XT>(i < cntl ? list[i] : -1, TOPC_RANK_FWD);
        __syncthreads(); // keep the waves of a workgroup in step between frames
    }
}

// Two frames per wave (lanes 0..31 / 32..63), eight per workgroup: k_topc_rank above is bound by VALU issue -- ~1170 vector
// instructions per one-frame wave (PMC) for ~70 records and ~15 survivors, i.e. most lanes idle most of the time.  Same steps, same
// canonical order, same rules; what differs: records in four slices of 32 (lists of more than 128 records), survivors one per lane
// (more than 32) -- such frames go to `wide` (count in flag[13]) and k_topc_rank does them in list mode right behind this kernel, no
// host round trip; the reductions run per half wave (row rotations + 4 v_readlane, a select per half); the rank loop reads survivor jj
// of both halves (4 v_readlane, a select).  A frame that is handed on or fails a check only clears
// `alive` for its half: the other half carries on.  The direct form is taken by BOTH halves when either needs it (wave-uniform).
#ifndef TOPC_CAP2
#define TOPC_CAP2 128
#endif
#define TOPC_NSL2 (TOPC_CAP2 / 32)
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_rank2(const void *__restrict__ x, long n, long ldx, int D, int C, const double *__restrict__ cand,
                                                    const int *__restrict__ cnt, const double *__restrict__ theta,
                                                    const double *__restrict__ slow, const int *__restrict__ efin,
                                                    const double *__restrict__ mean, const double *__restrict__ iv,
                                                    const double *__restrict__ lwc, const double *__restrict__ w, int ctop, int complete,
                                                    double lo, double hi, int *__restrict__ idx_out, double *__restrict__ lk_out,
                                                    double *__restrict__ nontop_lk, double *__restrict__ nontop_llk,
                                                    double *__restrict__ nontop_w, double *__restrict__ llk_out, int *__restrict__ flag, long *__restrict__ redo,
                                                    int stats, long *__restrict__ wide)
{
    __shared__ double xs[8][64 + 1];
    __shared__ unsigned bmap[8][64];
    __shared__ double sz[8][TOPC_CAP2];
    __shared__ int si[8][TOPC_CAP2];
    __shared__ int ord[8][64];
    __shared__ double ordz[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31;
    const bool hi2 = lane >= 32;
    const int f = wave * 2 + (hi2 ? 1 : 0);
    const bool force_direct = (stats & 2) != 0;
    const long t = (long)blockIdx.x * 8 + f;
    const bool live = t < n;
    const long tc = live ? t : 0; // clamped: dead halves read frame 0's scalars and never write
    for (int d = l32; d < D; d += 32) xs[f][d] = live ? feat_load<XT>::get(x, t * ldx + d) : 0.0;
    bmap[f][l32] = 0u;
    bmap[f][l32 + 32] = 0u;
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    wsync();
    const double NINF = -__builtin_inf();
    const int ncr = live ? cnt[tc] : 0;
    bool alive = live;
    if ((stats & 1) && live && l32 == 0) { atomicMax(&flag[5], ncr); atomicAdd((unsigned long long *)&flag[6], (unsigned long long)ncr); }
    if (alive && ncr > TOPC_CAP2) { // long list: the one-frame kernel (it hands lists beyond its own capacity to the direct-form kernel)
        if (l32 == 0) wide[atomicAdd(&flag[13], 1)] = t;
        alive = false;
    }
    const int nc = alive ? ncr : 0;
    const int nc0 = __builtin_amdgcn_readlane(nc, 0), nc1 = __builtin_amdgcn_readlane(nc, 32);
    const int ncmax = nc0 > nc1 ? nc0 : nc1; // scalar: slices beyond both lists are skipped wave-uniformly
    double zl[TOPC_NSL2];
    int cl[TOPC_NSL2];
#pragma unroll
    for (int j = 0; j < TOPC_NSL2; ++j) {
        const int k = l32 + 32 * j;
        zl[j] = NINF; cl[j] = -1;
        if (32 * j < ncmax && k < nc) {
            const d2 rec = *(const d2 *)(cand + 2 * ((size_t)t * TOPC_CAP + k));
            zl[j] = rec[0];
            cl[j] = (int)__double_as_longlong(rec[1]);
            atomicOr(&bmap[f][(cl[j] >> 5) & 63], 1u << (cl[j] & 31));
        }
    }
    wsync();
    // exclusive prefix of the word populations (two words per lane) -> canonical position of every record
    {
        const unsigned w0 = bmap[f][2 * l32], w1 = bmap[f][2 * l32 + 1];
        const int p0 = __builtin_popcount(w0), p1 = __builtin_popcount(w1);
        int inc = p0 + p1;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const int v = __shfl_up(inc, o, 32); if (l32 >= o) inc += v; }
        ord[f][2 * l32] = inc - p0 - p1; // reused as the prefix table until the survivors are compacted
        ord[f][2 * l32 + 1] = inc - p1;
    }
    wsync();
#pragma unroll
    for (int j = 0; j < TOPC_NSL2; ++j)
        if (32 * j < ncmax && cl[j] >= 0) {
            const int wd = (cl[j] >> 5) & 63;
            const int pos = ord[f][wd] + __builtin_popcount(bmap[f][wd] & ((1u << (cl[j] & 31)) - 1u));
            sz[f][pos] = zl[j];
            si[f][pos] = cl[j];
        }
    wsync();
    const double th = theta[tc];
    const int ef = efin[tc];
    const double lnE = (double)ef * 0.6931471805599453;
    // canonical order from here on: lane l looks at records l, l + 32, ...; survivors (logit >= theta) are compacted in that order
    int ns = 0;
    double zrej[TOPC_NSL2];
#pragma unroll
    for (int j = 0; j < TOPC_NSL2; ++j) {
        zrej[j] = NINF;
        if (32 * j >= ncmax) continue; // wave-uniform
        const int k = l32 + 32 * j;
        const double z = k < nc ? sz[f][k] : NINF;
        const bool hit = k < nc && z >= th;
        zrej[j] = (k < nc && !hit) ? z : NINF;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        const unsigned m32 = hi2 ? (unsigned)(mask >> 32) : (unsigned)mask;
        if (hit) {
            const int pos = ns + __builtin_popcount(m32 & ((1u << l32) - 1u));
            if (pos < 32) { ord[f][pos] = si[f][k]; ordz[f][pos] = z; }
        }
        ns += __builtin_popcount(m32);
    }
    wsync();
    if ((stats & 1) && alive && l32 == 0) { atomicMax(&flag[8], ns); atomicAdd((unsigned long long *)&flag[10], (unsigned long long)ns); }
    if (alive && ns > 32) { // more survivors than lanes: the one-frame kernel
        if (l32 == 0) wide[atomicAdd(&flag[13], 1)] = t;
        alive = false;
    }
    if (alive && ns < (ctop < C ? ctop : C)) {
        if (l32 == 0) { redo[atomicAdd(&flag[0], 1)] = t; atomicAdd(&flag[3], 1); }
        alive = false;
    }
    const int nsv = alive ? ns : 0;
    const int ns0 = __builtin_amdgcn_readlane(nsv, 0), ns1 = __builtin_amdgcn_readlane(nsv, 32);
    const int nsmax = ns0 > ns1 ? ns0 : ns1;
    if (nsmax == 0) return; // both frames handed on (or dead): wave-uniform
    const int ci = l32 < nsv ? ord[f][l32] : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    const bool cd = l32 < nsv && ci < C;
    // rank on the MFMA logits; ties go to the earlier survivor = the lower Gaussian index (canonical order)
    double zc = cd ? ordz[f][l32] : NINF;
    int rank = 0;
    bool near = false;
    for (int jj = 0; jj < nsmax; ++jj) {
        const double za = readlane_f64u(zc, jj), zb = readlane_f64u(zc, jj + 32);
        const double zj = hi2 ? zb : za;
        if (jj != l32 && (zj > zc || (zj == zc && jj < l32))) ++rank;
        near |= jj != l32 && __builtin_fabs(zj - zc) <= 1e-6;
    }
    if (force_direct || __builtin_amdgcn_ballot_w64(cd && rank < ctop && near) != 0) { // wave-uniform: both frames
        // the survivors' logits in the reference's direct form, four lanes per survivor (8 survivors per half and pass)
        for (int p0 = 0; p0 < nsmax; p0 += 8) {
            const int cj = p0 + (l32 >> 2), sub = l32 & 3;
            const int c = cj < nsv ? ord[f][cj] : 0x7fffffff;
            double acc = 0.0;
            if (c < C) {
                const double *mu = mean + (size_t)c * D, *vi = iv + (size_t)c * D, *xr = xs[f];
                if ((D & 1) == 0) {
                    for (int pr = sub; pr < (D >> 1); pr += 4) {
                        const d2 m2 = *(const d2 *)(mu + 2 * pr), v2 = *(const d2 *)(vi + 2 * pr);
                        const double dx0 = xr[2 * pr] - m2[0], dx1 = xr[2 * pr + 1] - m2[1];
                        acc = __builtin_fma(dx0 * dx0, v2[0], acc);
                        acc = __builtin_fma(dx1 * dx1, v2[1], acc);
                    }
                } else {
                    for (int d = sub; d < D; d += 4) {
                        const double dx = xr[d] - mu[d];
                        acc = __builtin_fma(dx * dx, vi[d], acc);
                    }
                }
            }
            acc += shfl_xor_f64(acc, 1);
            acc += shfl_xor_f64(acc, 2);
            if (sub == 0 && cj < nsv) ordz[f][cj] = c < C ? __builtin_fma(-0.5, acc, lwc[c]) : NINF;
        }
        wsync();
        zc = cd ? ordz[f][l32] : NINF;
        rank = 0;
        for (int jj = 0; jj < nsmax; ++jj) {
            const double za = readlane_f64u(zc, jj), zb = readlane_f64u(zc, jj + 32);
            const double zj = hi2 ? zb : za;
            if (jj != l32 && (zj > zc || (zj == zc && jj < l32))) ++rank;
        }
        if ((stats & 1) && alive && l32 == 0) atomicAdd(&flag[12], 1);
    }
    const bool sel = cd && rank < ctop;
    const double M = half_max_f64_dpp(zc, hi2);
    const double zmin = half_min_f64_dpp(sel ? zc : __builtin_inf(), hi2);
    // every Gaussian outside the survivors has an MFMA logit below theta: it must not be able to overtake the weakest selected
    if (alive && ns > ctop && th > zmin - 1e-6) {
        if (l32 == 0) { redo[atomicAdd(&flag[0], 1)] = t; atomicAdd(&flag[4], 1); }
        alive = false;
    }
    // remainder relative to M: never-appended likelihoods + appended below theta + rejected survivors
    double srel = 0.0;
    {
        const double sl = slow[tc];
        if (sl > 0.0) srel = exp(log(sl) + lnE - M);
    }
    double sr = 0.0;
#pragma unroll
    for (int j = 0; j < TOPC_NSL2; ++j)
        if (32 * j < ncmax) sr += zrej[j] > NINF ? gexp(zrej[j] - M) : 0.0; // wave-uniform skip
    srel += half_sum_f64_dpp(sr, hi2);
    srel += half_sum_f64_dpp((cd && !sel) ? gexp(zc - M) : 0.0, hi2);
    const double st = half_sum_f64_dpp(sel ? gexp(zc - M) : 0.0, hi2);
    const bool dead = !(M > GMMIV_ZERO_LLK); // zero-likelihood frame (include/gmmiv.h): the lowest indices, likelihoods 0, llk = lo
    if (sel && alive) {
        idx_out[t * ctop + rank] = dead ? rank : ci;
        if (lk_out) lk_out[t * ctop + rank] = dead ? 0.0 : exp(zc);
        ord[f][32 + rank] = dead ? rank : ci; // the upper half of the row is free (survivors use 0..31)
    }
    if (l32 == 0 && alive) {
        const double rest_llk = (srel > 0.0 && !dead) ? M + log(srel) : NINF;
        if (nontop_llk) nontop_llk[t] = rest_llk;
        if (nontop_lk) nontop_lk[t] = dead ? 0.0 : exp(rest_llk);
        if (llk_out) {
            const double tot = complete ? st + srel : st;
            llk_out[t] = dead ? lo : fmin(fmax(M + log(tot), lo), hi);
        }
    }
    if (nontop_w) {
        wsync();
        if (l32 == 0 && alive) {
            double snsw = 1.0;
            for (int k = 0; k < ctop; ++k) snsw -= w[ord[f][32 + k]];
            nontop_w[t] = snsw;
        }
    }
}

// results of the frames redone by the direct-form kernel (rows i of the s* arrays) -> rows redo[i] of the caller's arrays
__global__ void k_topc_scatter(long n, int ctop, const long *__restrict__ redo, const int *__restrict__ sidx, const double *__restrict__ slk,
                               const double *__restrict__ snlk, const double *__restrict__ snllk, const double *__restrict__ snw,
                               const double *__restrict__ sllk, int *__restrict__ idx, double *__restrict__ lk, double *__restrict__ nlk,
                               double *__restrict__ nllk, double *__restrict__ nw, double *__restrict__ llk)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * ctop) return;
    const long i = e / ctop, t = redo[i];
    const int k = (int)(e - i * ctop);
    idx[t * ctop + k] = sidx[e];
    if (lk) lk[t * ctop + k] = slk[e];
    if (k == 0) {
        if (nlk) nlk[t] = snlk[i];
        if (nllk) nllk[t] = snllk[i];
        if (nw) nw[t] = snw[i];
        if (llk) llk[t] = sllk[i];
    }
}
int gmmk_topc_scatter(hipStream_t st, long n, int ctop, const long *redo, const int *sidx, const double *slk, const double *snlk,
                      const double *snllk, const double *snw, const double *sllk, int *idx, double *lk, double *nlk, double *nllk, double *nw,
                      double *llk)
{
    if (n <= 0) return 0;
    k_topc_scatter<<<(unsigned)((n * ctop + 255) / 256), 256, 0, st>>>(n, ctop, redo, sidx, slk, snlk, snllk, snw, sllk, idx, lk, nlk, nllk, nw, llk);
    return (int)hipGetLastError();
}

int gmmk_topc_rank(hipStream_t st, int x_f64, const void *x, long n, long ldx, int D, int C, const double *cand, const int *cnt,
                   const double *theta, const double *slow, const int *efin, const double *mean, const double *iv, const double *lwc,
                   const double *w, int ctop, int complete, double lo, double hi, int *idx, double *lk, double *nlk, double *nllk,
                   double *nw, double *llk, int *flag, long *redo, int stats, long *wide)
{
    // stats: bit 0 = list / survivor statistics (global atomics), bit 1 = direct form for every frame, bit 2 = one frame per wave only;
    // wide = n longs of device memory for the frames the two-frame kernel hands to the one-frame kernel (NULL: one frame per wave)
    if (n <= 0) return 0;
    if (C > 2048 || D > 64 || ctop > 16) return -1; // the index bitmap / frame row / threshold rule of this path
    const bool two = wide && !(stats & 4);
    stats &= 3;
#define TOPC_RANK_ARGS x, n, ldx, D, C, cand, cnt, theta, slow, efin, mean, iv, lwc, w, ctop, complete, lo, hi, idx, lk, nlk, nllk, nw, llk, flag, redo, stats
    if (two) {
        const unsigned grid2 = (unsigned)((n + 7) / 8);
        if (x_f64) k_topc_rank2<double><<<grid2, 256, 0, st>>>(TOPC_RANK_ARGS, wide);
        else k_topc_rank2<float><<<grid2, 256, 0, st>>>(TOPC_RANK_ARGS, wide);
        // the frames it passed on (count in flag[13], known to the device only): a small fixed grid strides over the list
        if (x_f64) k_topc_rank_list<double><<<128, 256, 0, st>>>(TOPC_RANK_ARGS, wide, flag + 13);
        else k_topc_rank_list<float><<<128, 256, 0, st>>>(TOPC_RANK_ARGS, wide, flag + 13);
    } else {
        const unsigned grid = (unsigned)((n + 3) / 4);
        if (x_f64) k_topc_rank<double><<<grid, 256, 0, st>>>(TOPC_RANK_ARGS);
        else k_topc_rank<float><<<grid, 256, 0, st>>>(TOPC_RANK_ARGS);
    }
#undef TOPC_RANK_ARGS
    return (int)hipGetLastError();
}

// USE_TOP_DISTRIBS (client models on the world's top-C' indices, TopGauss.cpp:224-316 / ComputeTest.cpp:170-207), 16 lanes per
// frame: lane k of a frame's DPP row evaluates candidate k from the row-major model (16-byte loads, 8 dimensions in flight at
// a time), the frame's log-sum is two DPP row reductions.  The first kernel gave a whole wave to one frame (10 busy lanes, one
// 8-byte load per dimension and lane, permute-based reductions): 190 M frames/s per client model -- and ComputeTest runs this
// once per CLIENT of every test segment.
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_use16(const void *__restrict__ x, long T, long ldx, int D, const double *__restrict__ mean,
                                                    const double *__restrict__ iv, const double *__restrict__ lwc, int C, int ctop,
                                                    const int *__restrict__ idx, const double *__restrict__ nontop_llk, int complete,
                                                    double lo, double hi, double *__restrict__ llk_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xs = (double *)smem; // [16][D + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = lane & 15, f = wave * 4 + (lane >> 4);
    const long tb = (long)blockIdx.x * 16, t = tb + f;
    const int Dp = D + 1;
    for (int e = tid; e < 16 * D; e += 256) {
        const int ff = e / D, d = e - ff * D;
        xs[ff * Dp + d] = tb + ff < T ? feat_load<XT>::get(x, (tb + ff) * ldx + d) : 0.0;
    }
    __syncthreads();
    const double NINF = -__builtin_inf();
    double z = NINF;
    const int c = (t < T && k < ctop) ? idx[t * ctop + k] : -1;
    const bool live = (unsigned)c < (unsigned)C; // an index outside the model (a caller's stale / mis-strided vector) is skipped, never dereferenced
    if (live) {
        const double *mu = mean + (size_t)c * D, *vi = iv + (size_t)c * D, *xr = xs + f * Dp;
        double acc = 0.0;
        int d = 0;
        for (; d + 8 <= D; d += 8) {
            d2 m4[4], v4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { m4[u] = *(const d2 *)(mu + d + 2 * u); v4[u] = *(const d2 *)(vi + d + 2 * u); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double dx0 = xr[d + 2 * u] - m4[u][0], dx1 = xr[d + 2 * u + 1] - m4[u][1];
                acc = __builtin_fma(dx0 * dx0, v4[u][0], acc);
                acc = __builtin_fma(dx1 * dx1, v4[u][1], acc);
            }
        }
        for (; d < D; ++d) {
            const double dx = xr[d] - mu[d];
            acc = __builtin_fma(dx * dx, vi[d], acc);
        }
        z = __builtin_fma(-0.5, acc, lwc[c]);
    }
    const double r = (complete && nontop_llk && t < T) ? nontop_llk[t] : NINF;
    double M = fmax(z, r);
    M = fmax(M, dpp_f64_0x128(M)); M = fmax(M, dpp_f64_0x124(M)); M = fmax(M, dpp_f64_0x122(M)); M = fmax(M, dpp_f64_0x121(M));
    double s = live ? gexp(z - M) : 0.0;
    s += dpp_f64_0x128(s); s += dpp_f64_0x124(s); s += dpp_f64_0x122(s); s += dpp_f64_0x121(s);
    if (k == 0 && t < T) {
        if (r > NINF) s += gexp(r - M);
        llk_out[t] = fmin(fmax(M + log(s), lo), hi);
    }
}

// The same with FOUR lanes per candidate and one frame per wave (16 candidate slots x 4): k_topc_use16 is bound by the address
// path, not by bytes -- every load instruction of a wave touches 40 different 128-byte lines (40 busy lanes, one row each, 16
// bytes at a time), 600 line look-ups per frame.  Here the four lanes of a candidate read 64 contiguous bytes of its row per
// instruction: 10 half lines per instruction, 160 per frame; the partial sums of a candidate meet with two quad exchanges.
// 1.22 -> 0.97 ms per 10^6 frames and client model.  (Measured and dropped: two lanes per candidate with two frames per wave --
// quarter lines, back on the address path: 1.25 ms; four lanes with four frames per wave one after the other and ONE tail for the
// four -- fewer waves in flight cost more than the shared reductions / exponentials / logarithm save: 1.04 ms.)
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_use4(const void *__restrict__ x, long T, long ldx, int D, const double *__restrict__ mean,
                                                   const double *__restrict__ iv, const double *__restrict__ lwc, int C, int ctop,
                                                   const int *__restrict__ idx, const double *__restrict__ nontop_llk, int complete,
                                                   double lo, double hi, double *__restrict__ llk_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 2, sub = lane & 3;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= T) return; // wave-uniform
    const double NINF = -__builtin_inf();
    const int c = slot < ctop ? idx[t * ctop + slot] : -1;
    const bool live = (unsigned)c < (unsigned)C; // an index outside the model (a caller's stale / mis-strided vector) is skipped, never dereferenced
    const int cc = live ? c : 0;
    const d2 *mu = (const d2 *)(mean + (size_t)cc * D), *vi = (const d2 *)(iv + (size_t)cc * D);
    const int np = D >> 1; // dimension pairs
    double acc = 0.0;
    for (int p0 = 0; p0 < np; p0 += 16) {
        d2 m[4], v[4];
        double x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { // all loads of the batch first
            const int p = p0 + sub + 4 * u, pc = p < np ? p : np - 1;
            m[u] = mu[pc];
            v[u] = vi[pc];
            x0[u] = feat_load<XT>::get(x, t * ldx + 2 * pc);
            x1[u] = feat_load<XT>::get(x, t * ldx + 2 * pc + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = p0 + sub + 4 * u < np;
            const double dx0 = x0[u] - m[u][0], dx1 = x1[u] - m[u][1];
            const double a1 = __builtin_fma(dx1 * dx1, v[u][1], __builtin_fma(dx0 * dx0, v[u][0], acc));
            acc = ok ? a1 : acc;
        }
    }
    acc += __hiloint2double(dpp_i32<0xB1>(__double2hiint(acc)), dpp_i32<0xB1>(__double2loint(acc))); // quad_perm [1 0 3 2]
    acc += __hiloint2double(dpp_i32<0x4E>(__double2hiint(acc)), dpp_i32<0x4E>(__double2loint(acc))); // quad_perm [2 3 0 1]
    const double z = live ? __builtin_fma(-0.5, acc, lwc[cc]) : NINF;
    const double r = (complete && nontop_llk) ? nontop_llk[t] : NINF;
    const double M = wave_max_f64_dpp(fmax(z, r));
    const double s0 = wave_sum_f64_dpp((live && sub == 0) ? gexp(z - M) : 0.0);
    if (lane == 0) {
        const double s = r > NINF ? s0 + gexp(r - M) : s0;
        llk_out[t] = fmin(fmax(M + log(s), lo), hi);
    }
}

// k_topc_use4 for SEVERAL client models in one launch (blockIdx.y = client): ComputeTest scores every test segment against all
// its clients with the same world indices; one launch per client costs a launch, a copy back and a host synchronisation each --
// 3000-frame segments run at a fifth of the kernel's rate that way.  Model pointers come from a device array; llk_out is
// [n_clients][T].  Same arithmetic as k_topc_use4, statement for statement.
struct TopcClient { const double *mean, *iv, *lwc; long C; };
template <typename XT>
__global__ __launch_bounds__(256) void k_topc_use4_multi(const void *__restrict__ x, long T, long ldx, int D, const TopcClient *__restrict__ clients,
                                                         int ctop, const int *__restrict__ idx, const double *__restrict__ nontop_llk,
                                                         int complete, double lo, double hi, double *__restrict__ llk_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 2, sub = lane & 3;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= T) return; // wave-uniform
    const TopcClient cl = clients[blockIdx.y];
    const double NINF = -__builtin_inf();
    const int c = slot < ctop ? idx[t * ctop + slot] : -1;
    const bool live = c >= 0 && (long)c < cl.C;
    const int cc = live ? c : 0;
    const d2 *mu = (const d2 *)(cl.mean + (size_t)cc * D), *vi = (const d2 *)(cl.iv + (size_t)cc * D);
    const int np = D >> 1;
    double acc = 0.0;
    for (int p0 = 0; p0 < np; p0 += 16) {
        d2 m[4], v[4];
        double x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + sub + 4 * u, pc = p < np ? p : np - 1;
            m[u] = mu[pc];
            v[u] = vi[pc];
            x0[u] = feat_load<XT>::get(x, t * ldx + 2 * pc);
            x1[u] = feat_load<XT>::get(x, t * ldx + 2 * pc + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = p0 + sub + 4 * u < np;
            const double dx0 = x0[u] - m[u][0], dx1 = x1[u] - m[u][1];
            const double a1 = __builtin_fma(dx1 * dx1, v[u][1], __builtin_fma(dx0 * dx0, v[u][0], acc));
            acc = ok ? a1 : acc;
        }
    }
    acc += __hiloint2double(dpp_i32<0xB1>(__double2hiint(acc)), dpp_i32<0xB1>(__double2loint(acc)));
    acc += __hiloint2double(dpp_i32<0x4E>(__double2hiint(acc)), dpp_i32<0x4E>(__double2loint(acc)));
    const double z = live ? __builtin_fma(-0.5, acc, cl.lwc[cc]) : NINF;
    const double r = (complete && nontop_llk) ? nontop_llk[t] : NINF;
    const double M = wave_max_f64_dpp(fmax(z, r));
    const double s0 = wave_sum_f64_dpp((live && sub == 0) ? gexp(z - M) : 0.0);
    if (lane == 0) {
        const double s = r > NINF ? s0 + gexp(r - M) : s0;
        llk_out[(size_t)blockIdx.y * T + t] = fmin(fmax(M + log(s), lo), hi);
    }
}
// clients: DEVICE array of n_clients {mean, iv, lwc, C} records (host-side layout: three pointers and a long, 32 bytes)
int gmmk_topc_use4_multi(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const void *clients, int n_clients, int ctop,
                         const int *idx, const double *nllk, int complete, double lo, double hi, double *llk)
{
    if (T <= 0 || n_clients <= 0) return 0;
    if (ctop > 16 || D % 2 != 0 || n_clients > 65535) return -1;
    const dim3 grid((unsigned)((T + 3) / 4), (unsigned)n_clients);
    if (x_f64) k_topc_use4_multi<double><<<grid, 256, 0, st>>>(x, T, ldx, D, (const TopcClient *)clients, ctop, idx, nllk, complete, lo, hi, llk);
    else k_topc_use4_multi<float><<<grid, 256, 0, st>>>(x, T, ldx, D, (const TopcClient *)clients, ctop, idx, nllk, complete, lo, hi, llk);
    return (int)hipGetLastError();
}

int gmmk_topc_use16(hipStream_t st, int x_f64, const void *x, long T, long ldx, int D, const double *mean, const double *iv,
                    const double *lwc, int C, int ctop, const int *idx, const double *nllk, int complete, double lo, double hi, double *llk,
                    int four)
{
    if (T <= 0) return 0;
    if (ctop > 16 || D % 2 != 0) return -1; // the caller keeps the one-wave-per-frame kernel
    if (four == 4) { // four lanes per candidate, one frame per wave
        const unsigned grid4 = (unsigned)((T + 3) / 4);
        if (x_f64) k_topc_use4<double><<<grid4, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
        else k_topc_use4<float><<<grid4, 256, 0, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
        return (int)hipGetLastError();
    }
    const unsigned grid = (unsigned)((T + 15) / 16);
    const size_t lds = (size_t)16 * (D + 1) * sizeof(double);
    if (x_f64) k_topc_use16<double><<<grid, 256, lds, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    else k_topc_use16<float><<<grid, 256, lds, st>>>(x, T, ldx, D, mean, iv, lwc, C, ctop, idx, nllk, complete, lo, hi, llk);
    return (int)hipGetLastError();
}

// Posterior vectors gamma[t][c] (MixtureGDStat::computeAndAccumulateOcc / getOccVect) from the stored scaled likelihoods:
// gamma = e 2^(E - Efin) / S_t.  One wave per 16-frame block and group of 8 tiles; register r of a lane is frame q + 4 r,
// Gaussian 16 ct + i16: for every r the 16 lanes of a row write 128 contiguous bytes.  Memory-bound (16 KB per frame each way);
// the direct-form kernel it replaces evaluated every logit on the VALU (16 G pairs/s).
__global__ __launch_bounds__(256) void k_post_from_z(long n, int C, int nct, const double *__restrict__ zbuf, long nfb,
                                                     const int *__restrict__ eit, const double *__restrict__ inv,
                                                     const int *__restrict__ efin, double *__restrict__ gamma)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, q = lane >> 4;
    const long fb = blockIdx.x;
    double fs[4];
    int ef[4];
    long tr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        tr[r] = fb * 16 + q + 4 * r;
        const long tc = tr[r] < n ? tr[r] : n - 1;
        fs[r] = inv[tc];
        ef[r] = efin[tc];
    }
    for (int ct = blockIdx.y * 4 + wave; ct < nct; ct += gridDim.y * 4) {
        const d2 *pz = (const d2 *)(zbuf + (((size_t)ct * nfb + fb) * 64 + lane) * 4);
        const d2 a = __builtin_nontemporal_load(pz), b = __builtin_nontemporal_load(pz + 1);
        const double e[4] = {a[0], a[1], b[0], b[1]};
        const int *ep = eit + (size_t)(ct >> 1) * (nfb * 16) + fb * 16 + q;
        const int c = 16 * ct + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (tr[r] < n && c < C) gamma[(size_t)tr[r] * C + c] = __builtin_ldexp(e[r] * fs[r], ep[4 * r] - ef[r]);
    }
}

int gmmk_post_from_z(hipStream_t st, long n, int C, int nct, const double *zbuf, long nfb, const int *eit, const double *inv,
                     const int *efin, double *gamma)
{
    if (n <= 0) return 0;
    const unsigned gy = nct >= 32 ? 8 : 1; // 8 tile groups per frame block: enough workgroups for short inputs
    k_post_from_z<<<dim3((unsigned)((n + 15) / 16), gy), 256, 0, st>>>(n, C, nct, zbuf, nfb, eit, inv, efin, gamma);
    return (int)hipGetLastError();
}
