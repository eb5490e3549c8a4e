/*
 * oracle.c -- CPU restatement (plain C, fp64) of the LIA_RAL / ALIZE GMM + i-vector hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lia_ral_amd/ (the product) may include, link or call
 * this file; it is imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity status: the per-frame arithmetic (rows 1-6 of SURVEY.md 8(a)) lives in the external,
 * un-vendored alize-core library; its semantics are pinned here by the reference's own fixtures
 * KAT-1 (ComputeTest/test/test1.validate.res), KAT-2 (TrainTarget/test/test1.validate.gmm) and
 * KAT-4 (NormFeat/test/test1.validate.prm) -- see tests/test_oracle_kat.py.  The i-vector / PLDA
 * rows (11-19) restate self-contained loops of LIA_SpkTools 1:1 but the reference ships no test
 * for them: "parity unpinned" for those rows.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * All matrices row-major, all arithmetic double, like the reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EPS_LK 1e-200 /* LIA_SpkTools/src/TopGauss.cpp:67 (EPS_LK) */
#define ORC_PI 3.14159265358979323846

/* ---------------------------------------------------------------------------------------------
 * Rows 1-3: DistribGD::computeLK, MixtureGDStat::computeAndAccumulateLLK, top-C state
 * ------------------------------------------------------------------------------------------- */

/* cst_c = (2 pi)^(-D/2) det_c^(-1/2), det_c = prod_d cov_cd (RAW GMM file fields, verified on
 * LIA_SpkDet/ComputeTest/test/wld; DistribGD::computeAll() is called after every setCov,
 * LIA_SpkTools/src/TrainTools.cpp:582). */
void orc_gmm_cst(int C, int D, const double *covinv, double *cst, double *det)
{
    for (int c = 0; c < C; ++c) {
        double d = 1.0;
        for (int k = 0; k < D; ++k) d *= 1.0 / covinv[(size_t)c * D + k];
        det[c] = d;
        cst[c] = pow(2.0 * ORC_PI, -0.5 * D) / sqrt(d);
    }
}

/* lk_c(x) = cst_c exp(-1/2 sum_d (x_d - mu_cd)^2 covInv_cd): DistribGD::computeLK as used at
 * LIA_SpkTools/src/TopGauss.cpp:254, FactorAnalysis.cpp:289, GeneralTools.cpp:768. */
static double distrib_lk(int D, const double *x, const double *mean, const double *covinv, double cst)
{
    double m = 0.0;
    for (int k = 0; k < D; ++k) {
        double d = x[k] - mean[k];
        m += d * d * covinv[k];
    }
    return cst * exp(-0.5 * m);
}

static double clamp_llk(double lk, double min_llk, double max_llk)
{
    /* EPS_LK floor then [minLLK,maxLLK] clamp: TopGauss.cpp:190-192,258-260 */
    if (lk < ORC_EPS_LK) lk = ORC_EPS_LK;
    double llk = log(lk);
    if (llk < min_llk) llk = min_llk;
    if (llk > max_llk) llk = max_llk;
    return llk;
}

/* computeAndAccumulateLLK(f, 1.0, TOP_DISTRIBS_NO_ACTION) per frame:
 * call sites LIA_SpkTools/src/AccumulateStat.cpp:77, AccumulateTVStat.cpp:1644-1648. */
void orc_llk(int C, int D, const double *w, const double *mean, const double *covinv,
             const double *x, long T, double min_llk, double max_llk, double *llk)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    for (long t = 0; t < T; ++t) {
        double s = 0.0;
        for (int c = 0; c < C; ++c)
            s += w[c] * distrib_lk(D, x + t * D, mean + (size_t)c * D, covinv + (size_t)c * D, cst[c]);
        llk[t] = clamp_llk(s, min_llk, max_llk);
    }
    free(cst); free(det);
}

typedef struct { double lk; long idx; } lkpair;
static int cmp_desc(const void *a, const void *b)
{
    const lkpair *x = a, *y = b;
    if (x->lk > y->lk) return -1;
    if (x->lk < y->lk) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx); /* tie-break: lower index first (assumed, U4) */
}

/* DETERMINE_TOP_DISTRIBS: LIA_SpkDet/ComputeTest/src/ComputeTest.cpp:163, TopGauss.cpp:167-193.
 * Sort w_c lk_c descending, keep ctop (idx, lk); sumNonTopLK = total - sum top (TopGauss.cpp:183-187),
 * sumNonTopWeights = 1 - sum top weights.  World llk: COMPLETE -> log(total);
 * PARTIAL -> log(sum top) (assumed). */
void orc_llk_determine_top(int C, int D, const double *w, const double *mean, const double *covinv,
                           const double *x, long T, int ctop, int complete,
                           double min_llk, double max_llk,
                           long *idx, double *lk, double *nontop_lk, double *nontop_w, double *llk)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    lkpair *v = malloc(sizeof(lkpair) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    if (ctop > C) ctop = C;
    for (long t = 0; t < T; ++t) {
        double tot = 0.0;
        for (int c = 0; c < C; ++c) {
            v[c].lk = w[c] * distrib_lk(D, x + t * D, mean + (size_t)c * D, covinv + (size_t)c * D, cst[c]);
            v[c].idx = c;
            tot += v[c].lk;
        }
        qsort(v, C, sizeof(lkpair), cmp_desc);
        double top = 0.0, tw = 0.0;
        for (int j = 0; j < ctop; ++j) {
            idx[t * ctop + j] = v[j].idx;
            lk[t * ctop + j] = v[j].lk;
            top += v[j].lk;
            tw += w[v[j].idx];
        }
        /* non-top remainder summed directly over the non-selected components (same quantity as
         * total - top, without the cancellation) */
        double rest = 0.0;
        for (int j = ctop; j < C; ++j) rest += v[j].lk;
        nontop_lk[t] = rest;
        nontop_w[t] = 1.0 - tw;
        llk[t] = clamp_llk(complete ? top + rest : top, min_llk, max_llk);
        (void)tot;
    }
    free(cst); free(det); free(v);
}

/* USE_TOP_DISTRIBS for a (client) model: ComputeTest.cpp:166-167.
 * lk = sum_{c in top} w_c lk_c(client) [+ sumNonTopLK(world) if COMPLETE]. */
void orc_llk_use_top(int C, int D, const double *w, const double *mean, const double *covinv,
                     const double *x, long T, int ctop, const long *idx, const double *nontop_lk,
                     int complete, double min_llk, double max_llk, double *llk)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    for (long t = 0; t < T; ++t) {
        double s = 0.0;
        for (int j = 0; j < ctop; ++j) {
            long c = idx[t * ctop + j];
            s += w[c] * distrib_lk(D, x + t * D, mean + (size_t)c * D, covinv + (size_t)c * D, cst[c]);
        }
        if (complete) s += nontop_lk[t];
        llk[t] = clamp_llk(s, min_llk, max_llk);
    }
    free(cst); free(det);
}

/* TopGauss::compute (LIA_SpkTools/src/TopGauss.cpp:136-198) on the frames x[T]: per frame the DETERMINE_TOP_DISTRIBS list of
 * length cap ("topDistribsCount ... should be high enough"), lk_tot = exp(llk) (:160), then
 *   topD < 1: `if (val > topD*lk_tot) break; val += topV[j].lk; _nbg[t]++` over the sorted list (:163-167; bounded by the list
 *             length here, by the mixture size in the reference);  topD >= 1: _nbg[t] = (unsigned long)topD (:170);
 *   snsw = 1 - sum UBM.weight(idx), snsl = lk_tot - sum lk, floored at EPS_LK (:183-192).
 * Outputs: nbg[T], idx_flat (sum nbg entries, frame after frame: _idx), snsw[T], snsl[T], llk[T]; returns _nbgcnt. */
long orc_topgauss_compute(int C, int D, const double *w, const double *mean, const double *covinv, const double *x, long T,
                          int cap, double topD, int complete, double min_llk, double max_llk,
                          long *nbg, long *idx_flat, double *snsw, double *snsl, double *llk)
{
    long *tidx = malloc(sizeof(long) * (size_t)T * cap);
    double *tlk = malloc(sizeof(double) * (size_t)T * cap), *nl = malloc(sizeof(double) * T), *nw = malloc(sizeof(double) * T);
    orc_llk_determine_top(C, D, w, mean, covinv, x, T, cap, complete, min_llk, max_llk, tidx, tlk, nl, nw, llk);
    long cnt = 0;
    for (long t = 0; t < T; ++t) {
        const double lk_tot = exp(llk[t]);
        double val = 0.0;
        nbg[t] = 0;
        if (topD < 1.0) {
            for (int j = 0; j < cap; ++j) {
                if (val > topD * lk_tot) break;
                val += tlk[t * cap + j];
                nbg[t]++;
            }
        } else nbg[t] = (long)topD;
        double sw = 1.0, sl = lk_tot;
        for (long j = 0; j < nbg[t]; ++j) {
            idx_flat[cnt + j] = tidx[t * cap + j];
            sw -= w[tidx[t * cap + j]];
            sl -= tlk[t * cap + j];
        }
        cnt += nbg[t];
        snsw[t] = sw;
        snsl[t] = sl < ORC_EPS_LK ? ORC_EPS_LK : sl;
    }
    free(tidx); free(tlk); free(nl); free(nw);
    return cnt;
}

/* TopGauss::get (TopGauss.cpp:275-316): per frame setTopDistribIndexVector(index, snsw, snsl) + computeAndAccumulateLLK(f, 1.0,
 * USE_TOP_DISTRIBS) on the stored, variable-length selection; llk[T] out (its mean is what get() returns). */
void orc_topgauss_get(int C, int D, const double *w, const double *mean, const double *covinv, const double *x, long T,
                      const long *nbg, const long *idx_flat, const double *snsl, int complete, double min_llk, double max_llk, double *llk)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    long b = 0;
    for (long t = 0; t < T; ++t) {
        double s = 0.0;
        for (long j = 0; j < nbg[t]; ++j) {
            const long c = idx_flat[b + j];
            s += w[c] * distrib_lk(D, x + t * D, mean + (size_t)c * D, covinv + (size_t)c * D, cst[c]);
        }
        if (complete) s += snsl[t];
        llk[t] = clamp_llk(s, min_llk, max_llk);
        b += nbg[t];
    }
    free(cst); free(det);
}

/* ---------------------------------------------------------------------------------------------
 * Rows 4-6: EM accumulators, occupation vector, FrameAccGD
 * ------------------------------------------------------------------------------------------- */

/* MixtureGDStat::computeAndAccumulateOcc + getOccVect: full posterior gamma_c = lk_c / sum lk
 * (call site AccumulateTVStat.cpp:334-335).  Returns the linear frame likelihood. */
static double occ_frame(int C, int D, const double *w, const double *mean, const double *covinv,
                        const double *cst, const double *x, double *gamma)
{
    double s = 0.0;
    for (int c = 0; c < C; ++c) {
        gamma[c] = w[c] * distrib_lk(D, x, mean + (size_t)c * D, covinv + (size_t)c * D, cst[c]);
        s += gamma[c];
    }
    if (s > 0.0)
        for (int c = 0; c < C; ++c) gamma[c] /= s;
    else
        for (int c = 0; c < C; ++c) gamma[c] = 0.0; /* U1: assumed (FactorAnalysis.cpp:293) */
    return s;
}

void orc_occ(int C, int D, const double *w, const double *mean, const double *covinv,
             const double *x, long T, double *gamma /* T x C */)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    for (long t = 0; t < T; ++t) occ_frame(C, D, w, mean, covinv, cst, x + t * D, gamma + t * C);
    free(cst); free(det);
}

/* accumulateStatEM frame loop: llkAcc += log(emAcc.computeAndAccumulateEM(f[,weight]))
 * LIA_SpkTools/src/AccumulateStat.cpp:103-114 (weighted variant :143-152, :206).
 * occ_c += w g_c ; sx_c += w g_c x ; sxx_c += w g_c x^2 ; count += w.  Accumulating (no reset). */
double orc_em_accumulate(int C, int D, const double *w, const double *mean, const double *covinv,
                         const double *x, long T, double weight,
                         double *occ, double *sx, double *sxx, double *count)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    double *g = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    double llk_acc = 0.0;
    for (long t = 0; t < T; ++t) {
        const double *f = x + t * D;
        double lk = occ_frame(C, D, w, mean, covinv, cst, f, g);
        llk_acc += log(lk);
        for (int c = 0; c < C; ++c) {
            double gc = g[c] * weight;
            occ[c] += gc;
            for (int k = 0; k < D; ++k) {
                sx[(size_t)c * D + k] += gc * f[k];
                sxx[(size_t)c * D + k] += gc * f[k] * f[k];
            }
        }
        *count += weight;
    }
    free(cst); free(det); free(g);
    return llk_acc;
}

/* MixtureStat::getEM(): w_c = occ_c/count ; mu = sx/occ ; cov = sxx/occ - mu^2
 * (weights/means verified by KAT-2 through computeMAPOccDep, TrainTools.cpp:445-489, where
 * alpha = client.weight(c)*frameCount = occ_c; variance formula = standard ML, U3 assumed).
 * Components with occ == 0 keep the previous parameters (assumed). */
void orc_em_get(int C, int D, const double *occ, const double *sx, const double *sxx, double count,
                double *w, double *mean, double *cov)
{
    for (int c = 0; c < C; ++c) {
        w[c] = occ[c] / count;
        if (occ[c] <= 0.0) continue;
        for (int k = 0; k < D; ++k) {
            double m = sx[(size_t)c * D + k] / occ[c];
            mean[(size_t)c * D + k] = m;
            cov[(size_t)c * D + k] = sxx[(size_t)c * D + k] / occ[c] - m * m;
        }
    }
}

/* setItParameter: LIA_SpkTools/src/TrainTools.cpp:560-564 */
double orc_set_it_parameter(double begin, double end, int nb_it, int it)
{
    if (nb_it < 2) return begin;
    double it_val = (begin - end) / ((double)nb_it - 1);
    return begin - it_val * it;
}

/* varianceControl: LIA_SpkTools/src/TrainTools.cpp:567-587 (floor then ceiling, in this order) */
void orc_variance_control(int C, int D, double *cov, double flooring, double ceiling,
                          const double *cov_signal, long *n_floor, long *n_ceil)
{
    long nf = 0, nc = 0;
    for (int c = 0; c < C; ++c)
        for (int v = 0; v < D; ++v) {
            double cv = cov[(size_t)c * D + v];
            if (cv <= flooring * cov_signal[v]) { cv = flooring * cov_signal[v]; nf++; }
            if (cv >= ceiling * cov_signal[v]) { cv = ceiling * cov_signal[v]; nc++; }
            cov[(size_t)c * D + v] = cv;
        }
    if (n_floor) *n_floor = nf;
    if (n_ceil) *n_ceil = nc;
}

/* computeMAPOccDep, mean-only: LIA_SpkTools/src/TrainTools.cpp:445-466 (pinned by KAT-2).
 * alpha_c = occ_c ; a = alpha/(alpha+r) ; mu = (1-a) mu_world + a mu_ML. */
void orc_map_occdep_mean(int C, int D, const double *mean_world, const double *w_ml,
                         const double *mean_ml, double frame_count, double reg, double *mean_out)
{
    for (int c = 0; c < C; ++c) {
        double alpha = w_ml[c] * frame_count;
        double a = alpha / (alpha + reg);
        for (int k = 0; k < D; ++k)
            mean_out[(size_t)c * D + k] =
                (1 - a) * mean_world[(size_t)c * D + k] + a * mean_ml[(size_t)c * D + k];
    }
}

/* computeMAP (LIA_SpkTools/src/TrainTools.cpp:543-556) and the four methods it dispatches to.  init = the a-priori model, client =
 * the ML estimate (in) / the adapted model (out); method 0 MAPOccDep (:445-489) and 3 MAPModelBased (:491-536) -- the same
 * statements --, 1 MAPConst (:356-384), 2 MAPConst2 (:390-420).  flags: bit 0 mean, 1 variance, 2 weight adaptation; reg[3] the
 * regulation factors, alpha_mean the constant of the two Const methods (their variance / weight branches are TODO in the reference:
 * the result carries the init model's variances and weights). */
void orc_compute_map(int method, int C, int D, const double *w0, const double *mean0, const double *cov0, double *w, double *mean, double *cov,
                     double frame_count, int flags, const double *reg, double alpha_mean)
{
    const size_t CD = (size_t)C * D;
    double *tw = malloc(sizeof(double) * C), *tm = malloc(sizeof(double) * CD), *tc = malloc(sizeof(double) * CD);
    memcpy(tw, w0, sizeof(double) * C); memcpy(tm, mean0, sizeof(double) * CD); memcpy(tc, cov0, sizeof(double) * CD);
    if (method == 0 || method == 3) {
        if (flags & 3)
            for (int c = 0; c < C; ++c) {
                const double alpha = w[c] * frame_count;
                if (flags & 1) {
                    const double a = alpha / (alpha + reg[0]);
                    for (int k = 0; k < D; ++k) tm[(size_t)c * D + k] = ((1 - a) * mean0[(size_t)c * D + k]) + (a * mean[(size_t)c * D + k]);
                }
                if (flags & 2) {
                    const double a = alpha / (alpha + reg[1]);
                    for (int k = 0; k < D; ++k)
                        tc[(size_t)c * D + k] = ((1 - a) * cov0[(size_t)c * D + k]) + (a * cov[(size_t)c * D + k]) +
                                                (((1 - a) * a) * pow(mean0[(size_t)c * D + k] - mean[(size_t)c * D + k], 2));
                }
            }
        if (flags & 4) {
            double sum = 0.0;
            for (int c = 0; c < C; ++c) {
                const double alpha = w[c] * frame_count, a = alpha / (alpha + reg[2]);
                tw[c] = a * w[c] + ((1 - a) * w0[c]);
                sum += tw[c];
            }
            for (int c = 0; c < C; ++c) tw[c] /= sum;
        }
    } else if (method == 1) {
        if (flags & 1)
            for (size_t e = 0; e < CD; ++e) tm[e] = (alpha_mean * tm[e]) + ((1 - alpha_mean) * mean[e]);
    } else if (method == 2) {
        if (flags & 1)
            for (int c = 0; c < C; ++c)
                for (int k = 0; k < D; ++k) {
                    const size_t e = (size_t)c * D + k;
                    tm[e] = ((alpha_mean * tw[c] * tm[e]) + ((1 - alpha_mean) * w[c] * mean[e])) / (tw[c] * alpha_mean + w[c] * (1 - alpha_mean));
                }
    } else { free(tw); free(tm); free(tc); return; } /* unknown method: no adaptation (:555) */
    memcpy(w, tw, sizeof(double) * C); memcpy(mean, tm, sizeof(double) * CD); memcpy(cov, tc, sizeof(double) * CD);
    free(tw); free(tm); free(tc);
}

/* FrameAccGD::accumulate / getMeanVect / getCovVect: sum x, sum x^2, n; mean, BIASED diagonal
 * covariance sum x^2/n - mean^2 (AccumulateStat.cpp:387-396, TrainTools.cpp:593-601; pinned by KAT-4). */
void orc_frame_acc(int D, const double *x, long T, double *sum, double *sumsq, double *count)
{
    for (long t = 0; t < T; ++t) {
        for (int k = 0; k < D; ++k) {
            double v = x[t * D + k];
            sum[k] += v;
            sumsq[k] += v * v;
        }
        *count += 1.0;
    }
}
void orc_frame_mean_cov(int D, const double *sum, const double *sumsq, double count, double *mean, double *cov)
{
    for (int k = 0; k < D; ++k) {
        mean[k] = sum[k] / count;
        cov[k] = sumsq[k] / count - mean[k] * mean[k];
    }
}

/* ---------------------------------------------------------------------------------------------
 * Row 10: bagging (glibc rand) -- GeneralTools.cpp:309-314, 455-510; seed TrainTools.cpp:1070
 * ------------------------------------------------------------------------------------------- */
static int bagged_frame(double p) { return ((double)rand() / (double)RAND_MAX) < p; }

/* in: nseg segments (begin,len); out: up to max_out bagged segments. Returns number written
 * (or the needed count if larger than max_out). */
long orc_bagged_segments(unsigned seed, const long *seg_begin, const long *seg_len, long nseg,
                         double p, long min_len, long max_len,
                         long *out_begin, long *out_len, long *out_src, long max_out)
{
    srand(seed);
    long n = 0, s = 0;
    if (nseg == 0) return 0;
    long begin = seg_begin[0], len = seg_len[0];
    int end = 0;
    while (!end) {
        long verify = len;
        if (verify < min_len) verify = min_len;
        if (verify > max_len) verify = max_len;
        int move;
        long length;
        if (len <= verify) { move = 1; length = len; }
        else { move = 0; length = verify; }
        if (length > 0 && bagged_frame(p)) {
            if (n < max_out) { out_begin[n] = begin; out_len[n] = length; out_src[n] = s; }
            n++;
        }
        if (move) {
            s++;
            end = (s >= nseg);
            if (!end) { begin = seg_begin[s]; len = seg_len[s]; }
        } else {
            len -= length;
            begin += length;
        }
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------
 * Row 10 edges: componentReduction and normalizeModel of trainModelStream (TrainTools.cpp:1078-1099)
 * ------------------------------------------------------------------------------------------- */
typedef struct { double weight; unsigned long distrib; void *distribP; } orc_tab_weight_elem; /* GeneralTools.h:145-149 */
static int orc_comp_f(const void *op1, const void *op2) /* _compF, GeneralTools.cpp:277-280 */
{
    if (((const orc_tab_weight_elem *)op1)->weight > ((const orc_tab_weight_elem *)op2)->weight) return -1;
    else return 1;
}
/* TabWeight::_sortByWeight (GeneralTools.h:157-164): order[i] = i-th heaviest component (qsort: ties as the C library leaves them) */
void orc_sort_by_weight(int C, const double *w, long *order)
{
    orc_tab_weight_elem *tab = (orc_tab_weight_elem *)malloc((size_t)C * sizeof(*tab));
    for (int i = 0; i < C; ++i) { tab[i].weight = w[i]; tab[i].distrib = (unsigned long)i; tab[i].distribP = 0; }
    qsort(tab, (size_t)C, sizeof(*tab), orc_comp_f);
    for (int i = 0; i < C; ++i) order[i] = (long)tab[i].distrib;
    free(tab);
}
/* selectComponent(selectCompA, nbTop, inputM) + reduceModel + normalizeWeights (TrainTools.cpp:197-227): the nb_top heaviest
 * components, in their ORIGINAL order, weights renormalised to 1.  w / mean / cov are compacted in place; returns nb_top. */
int orc_reduce_model(int C, int D, double *w, double *mean, double *cov, int nb_top)
{
    long *order = (long *)malloc((size_t)C * sizeof(long));
    char *sel = (char *)calloc((size_t)C, 1);
    orc_sort_by_weight(C, w, order);
    for (int i = 0; i < nb_top; ++i) sel[order[i]] = 1;
    int o = 0;
    double tot = 0.0;
    for (int c = 0; c < C; ++c)
        if (sel[c]) {
            for (int k = 0; k < D; ++k) { mean[(size_t)o * D + k] = mean[(size_t)c * D + k]; cov[(size_t)o * D + k] = cov[(size_t)c * D + k]; }
            w[o] = w[c];
            tot += w[o];
            o++;
        }
    tot = 0.0; /* normalizeWeights recomputes the total over the output model (:223-228) */
    for (int c = 0; c < o; ++c) tot += w[c];
    for (int c = 0; c < o; ++c) w[c] /= tot;
    free(order); free(sel);
    return o;
}
/* normalizeMixture(mixt, fake, fake, zeroOne = true, nbIt, meanOnly) (TrainTools.cpp:287-315) with mixtureFusion / gaussianFusion
 * (:240-283): tmp = the single Gaussian carrying the mixture's moments (components folded in one by one), then every component
 * mean <- (mean - tmp.mean) / sqrt(tmp.cov), cov <- cov / tmp.cov (unless meanOnly). */
void orc_normalize_mixture(int C, int D, const double *w, double *mean, double *cov, int nb_it, int mean_only)
{
    double *tm = (double *)malloc((size_t)D * sizeof(double)), *tc = (double *)malloc((size_t)D * sizeof(double));
    for (int it = 0; it < nb_it; ++it) {
        double wtmp = w[0], wres = w[0];
        for (int k = 0; k < D; ++k) { tm[k] = mean[k]; tc[k] = cov[k]; }
        for (int i = 1; i < C; ++i) { /* gaussianFusion(mixt.getDistrib(i), mixt.weight(i), res, wtmp, res, wres) */
            const double w1 = w[i], w2 = wtmp;
            const double a1 = w1 / (w1 + w2), a2 = 1.0 - a1;
            for (int k = 0; k < D; ++k) {
                const double d = mean[(size_t)i * D + k] - tm[k];
                tc[k] = a1 * cov[(size_t)i * D + k] + a2 * tc[k] + a1 * a2 * d * d;
                tm[k] = (a1 * mean[(size_t)i * D + k]) + (a2 * tm[k]);
            }
            wres = w1 + w2;
            wtmp = wres;
        }
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < D; ++i) {
                double nm = mean[(size_t)c * D + i] - tm[i];
                nm /= sqrt(tc[i]);
                mean[(size_t)c * D + i] = nm;
                if (!mean_only) cov[(size_t)c * D + i] = cov[(size_t)c * D + i] / tc[i];
            }
    }
    free(tm); free(tc);
}

/* The multi-selection bagging of mixtureInit (GeneralTools.cpp:330-390): one walk over the segments, nb_bagged draws per chunk,
 * a chunk kept for component idx is written with label idx.  Seeds the generator itself (TrainTools.cpp:732). */
long orc_bagged_segments_multi(unsigned seed, const long *seg_begin, const long *seg_len, long nseg, long nb_bagged,
                               double p, long min_len, long max_len,
                               long *out_begin, long *out_len, long *out_label, long max_out)
{
    srand(seed);
    long n = 0, s = 0;
    if (nseg == 0) return 0;
    long begin = seg_begin[0], len = seg_len[0];
    int end = 0;
    while (!end) {
        long verify = len;
        if (verify < min_len) verify = min_len;
        if (verify > max_len) verify = max_len;
        const int move = len <= verify;
        const long length = move ? len : verify;
        if (length > 0)
            for (long idx = 0; idx < nb_bagged; ++idx)
                if (bagged_frame(p)) {
                    if (n < max_out) { out_begin[n] = begin; out_len[n] = length; out_label[n] = idx; }
                    n++;
                }
        if (move) {
            s++;
            end = (s >= nseg);
            if (!end) { begin = seg_begin[s]; len = seg_len[s]; }
        } else {
            len -= length;
            begin += length;
        }
    }
    return n;
}

/* mixtureInit, multi-stream form with one stream (TrainTools.cpp:674-766): per-component FrameAccGD over the picked frames;
 * mean[C x D] out (covariances = globalCov and weights = 1/C are the caller's), count[C] = frames picked per component.
 * x: [T x D] row-major.  Returns 0, or -1 if the scratch for the bagged list was too small. */
/* one input stream of the multi-stream mixtureInit (TrainTools.cpp:698-735): its bagging probability (:700-708), its seeds
 * srand((stream + 1) * 100 + baggedIt + 1) (:732), the picked frames added to the per-component sums / counts (ACCUMULATING:
 * the frame accumulators of a component are shared by all streams, :686-690). */
int orc_mixture_init_stream(long stream, int C, int D, const double *x, const long *seg_begin, const long *seg_len, long nseg,
                            double stream_weight, double nb_frame_to_select, long min_len, long max_len, double *sum, double *count)
{
    long total = 0;
    for (long s = 0; s < nseg; ++s) total += seg_len[s];
    double proba = (nb_frame_to_select * stream_weight) / (double)total;   /* :700 */
    long nb_it = 1;
    double tmp = proba;
    while (tmp > 1) { /* :703-706, as written: tmp runs 2, 6/p, 24/p^2, ... -- it never comes back under 1 for 1 < p < ~4.9 */
        nb_it++;
        tmp /= proba / (double)nb_it;
        if (nb_it > 64) return -2; /* the reference hangs here; the restatement reports it */
    }
    proba = tmp;
    const long cap = (total + nseg + 8) * C;
    long *ob = (long *)malloc(sizeof(long) * cap), *ol = (long *)malloc(sizeof(long) * cap), *lab = (long *)malloc(sizeof(long) * cap);
    int rc = 0;
    for (long it = 0; it < nb_it && !rc; ++it) {
        const long n = orc_bagged_segments_multi((unsigned)((stream + 1) * 100 + (it + 1)), seg_begin, seg_len, nseg, C, proba, min_len, max_len,
                                                 ob, ol, lab, cap);
        if (n > cap) { rc = -1; break; }
        for (long k = 0; k < n; ++k) {                 /* accumulateStatFrame(*frameAcc[seg->labelCode()], ...) */
            double *sc = sum + (size_t)lab[k] * D;
            for (long t = ob[k]; t < ob[k] + ol[k]; ++t) {
                for (int i = 0; i < D; ++i) sc[i] += x[(size_t)t * D + i];
                count[lab[k]] += 1.0;
            }
        }
    }
    free(ob); free(ol); free(lab);
    return rc;
}
int orc_mixture_init(int C, int D, const double *x, const long *seg_begin, const long *seg_len, long nseg, double stream_weight,
                     double nb_frame_to_select, long min_len, long max_len, double *mean, double *count)
{
    double *sum = (double *)calloc((size_t)C * D, sizeof(double));
    for (int c = 0; c < C; ++c) count[c] = 0.0;
    const int rc = orc_mixture_init_stream(0, C, D, x, seg_begin, seg_len, nseg, stream_weight, nb_frame_to_select, min_len, max_len, sum, count);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < D; ++i) mean[(size_t)c * D + i] = sum[(size_t)c * D + i] / count[c];
    free(sum);
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * Rows 11-17: TVAcc (total variability)
 * ------------------------------------------------------------------------------------------- */

/* computeAndAccumulateTVStatUnThreaded inner loops: LIA_SpkTools/src/AccumulateTVStat.cpp:332-348.
 * frame t belongs to statistics row utt[t]. N[u,c] += g_c ; F[u, c*D+i] += g_c x_i. */
void orc_tv_stats(int C, int D, const double *w, const double *mean, const double *covinv,
                  const double *x, long T, const long *utt, double *N, double *F)
{
    double *cst = malloc(sizeof(double) * C), *det = malloc(sizeof(double) * C);
    double *g = malloc(sizeof(double) * C);
    orc_gmm_cst(C, D, covinv, cst, det);
    size_t SV = (size_t)C * D;
    for (long t = 0; t < T; ++t) {
        const double *f = x + t * D;
        occ_frame(C, D, w, mean, covinv, cst, f, g);
        long u = utt[t];
        for (int k = 0; k < C; ++k) {
            N[u * C + k] += g[k];
            for (int i = 0; i < D; ++i) F[u * SV + (size_t)k * D + i] += g[k] * f[i];
        }
    }
    free(cst); free(det); free(g);
}

/* substractMUnThreaded: AccumulateTVStat.cpp:1088-1105 */
void orc_tv_subtract_m(long U, int C, int D, const double *N, double *F, const double *ubm_means)
{
    size_t SV = (size_t)C * D;
    for (long u = 0; u < U; ++u)
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j)
                F[u * SV + (size_t)i * D + j] -= ubm_means[(size_t)i * D + j] * N[u * C + i];
}

/* estimateTETtUnThreaded: AccumulateTVStat.cpp:777-805. TETt[c] is R x R (full, mirrored). */
void orc_tv_tett(int C, int D, int R, const double *Tm /* R x SV */, const double *invvar, double *TETt)
{
    size_t SV = (size_t)C * D;
    for (int d = 0; d < C; ++d) {
        double *o = TETt + (size_t)d * R * R;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
                for (int k = 0; k < D; ++k)
                    s += Tm[i * SV + (size_t)d * D + k] * invvar[(size_t)d * D + k] * Tm[j * SV + (size_t)d * D + k];
                o[(size_t)i * R + j] = s;
            }
        for (int i = 0; i < R; ++i)
            for (int j = i + 1; j < R; ++j) o[(size_t)i * R + j] = o[(size_t)j * R + i];
    }
}

/* DoubleSquareMatrix::invert (alize-core, algorithm unknown: U5). Gauss-Jordan with partial
 * pivoting; returns 0 on success. */
int orc_invert(int n, const double *a_in, double *inv)
{
    double *a = malloc(sizeof(double) * n * n);
    memcpy(a, a_in, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) inv[(size_t)i * n + j] = (i == j);
    for (int c = 0; c < n; ++c) {
        int p = c;
        double best = fabs(a[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r)
            if (fabs(a[(size_t)r * n + c]) > best) { best = fabs(a[(size_t)r * n + c]); p = r; }
        if (best == 0.0) { free(a); return 1; }
        if (p != c)
            for (int j = 0; j < n; ++j) {
                double t = a[(size_t)c * n + j]; a[(size_t)c * n + j] = a[(size_t)p * n + j]; a[(size_t)p * n + j] = t;
                t = inv[(size_t)c * n + j]; inv[(size_t)c * n + j] = inv[(size_t)p * n + j]; inv[(size_t)p * n + j] = t;
            }
        double d = 1.0 / a[(size_t)c * n + c];
        for (int j = 0; j < n; ++j) { a[(size_t)c * n + j] *= d; inv[(size_t)c * n + j] *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            double f = a[(size_t)r * n + c];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) {
                a[(size_t)r * n + j] -= f * a[(size_t)c * n + j];
                inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
            }
        }
    }
    free(a);
    return 0;
}

/* DoubleSquareMatrix::upperCholesky (alize-core): upper factor Ch with R = Ch^T Ch (assumed; the
 * minimum-divergence update T <- Ch T, AccumulateTVStat.cpp:2070-2094, whitens w only for this
 * convention). Returns 0 on success. */
int orc_upper_cholesky(int n, const double *a, double *ch)
{
    memset(ch, 0, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = a[(size_t)i * n + j];
            for (int k = 0; k < i; ++k) s -= ch[(size_t)k * n + i] * ch[(size_t)k * n + j];
            if (i == j) {
                if (s <= 0.0) return 1;
                ch[(size_t)i * n + i] = sqrt(s);
            } else
                ch[(size_t)i * n + j] = s / ch[(size_t)i * n + i];
        }
    return 0;
}

/* L = I + sum_c N[u,c] TETt_c (lower, mirrored): AccumulateTVStat.cpp:2126-2146 */
static void build_L(int C, int R, const double *Nu, const double *TETt, double *L)
{
    memset(L, 0, sizeof(double) * R * R);
    for (int i = 0; i < R; ++i) L[(size_t)i * R + i] = 1.0;
    for (int c = 0; c < C; ++c) {
        const double *t = TETt + (size_t)c * R * R;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j <= i; ++j) L[(size_t)i * R + j] += t[(size_t)i * R + j] * Nu[c];
    }
    for (int i = 0; i < R; ++i)
        for (int j = i + 1; j < R; ++j) L[(size_t)i * R + j] = L[(size_t)j * R + i];
}

/* estimateWUnThreaded: AccumulateTVStat.cpp:2114-2169. F must already be centred (substractM). */
int orc_tv_estimate_w(long U, int C, int D, int R, const double *N, const double *F,
                      const double *Tm, const double *invvar, const double *TETt, double *W)
{
    size_t SV = (size_t)C * D;
    double *L = malloc(sizeof(double) * R * R), *Li = malloc(sizeof(double) * R * R);
    double *aux = malloc(sizeof(double) * R);
    int rc = 0;
    for (long u = 0; u < U; ++u) {
        build_L(C, R, N + u * C, TETt, L);
        if (orc_invert(R, L, Li)) { rc = 1; break; }
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (size_t k = 0; k < SV; ++k) s += F[u * SV + k] * invvar[k] * Tm[i * SV + k];
            aux[i] = s;
        }
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (int k = 0; k < R; ++k) s += aux[k] * Li[(size_t)i * R + k];
            W[u * R + i] = s;
        }
    }
    free(L); free(Li); free(aux);
    return rc;
}

/* estimateAandCUnthreaded: AccumulateTVStat.cpp:1702-1795.
 * Outputs: W[U x R], A[C x R*R], Cmx[R x SV], Rm[R x R], r[R], meanW[R] (= sum w / U). */
int orc_tv_estimate_a_and_c(long U, int C, int D, int R, const double *N, const double *F,
                            const double *Tm, const double *invvar, const double *TETt,
                            double *W, double *A, double *Cmx, double *Rm, double *r, double *meanW)
{
    size_t SV = (size_t)C * D, RR = (size_t)R * R;
    double *L = malloc(sizeof(double) * RR), *Li = malloc(sizeof(double) * RR);
    double *aux = malloc(sizeof(double) * R);
    memset(A, 0, sizeof(double) * C * RR);
    memset(Cmx, 0, sizeof(double) * R * SV);
    memset(Rm, 0, sizeof(double) * RR);
    memset(r, 0, sizeof(double) * R);
    memset(meanW, 0, sizeof(double) * R);
    int rc = 0;
    for (long u = 0; u < U; ++u) {
        build_L(C, R, N + u * C, TETt, L);
        if (orc_invert(R, L, Li)) { rc = 1; break; }
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (size_t k = 0; k < SV; ++k) s += F[u * SV + k] * invvar[k] * Tm[i * SV + k];
            aux[i] = s;
        }
        double *y = W + u * R;
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (int k = 0; k < R; ++k) s += aux[k] * Li[(size_t)i * R + k];
            y[i] = s;
        }
        for (int k = 0; k < R; ++k) meanW[k] += y[k];
        for (int i = 0; i < R; ++i) {
            for (int j = 0; j < R; ++j) {
                Li[(size_t)i * R + j] += y[i] * y[j];
                Rm[(size_t)i * R + j] += Li[(size_t)i * R + j];
            }
            r[i] += y[i];
        }
        for (int c = 0; c < C; ++c) {
            double n = N[u * C + c];
            double *a = A + (size_t)c * RR;
            for (size_t e = 0; e < RR; ++e) a[e] += Li[e] * n;
        }
        for (int i = 0; i < R; ++i)
            for (size_t j = 0; j < SV; ++j) Cmx[i * SV + j] += y[i] * F[u * SV + j];
    }
    for (int k = 0; k < R; ++k) meanW[k] /= (double)U;
    free(L); free(Li); free(aux);
    return rc;
}

/* updateTestimate: T_c = A_c^-1 Cmx_c, AccumulateTVStat.cpp:974-1005 */
int orc_tv_update_t(int C, int D, int R, const double *A, const double *Cmx, double *Tm)
{
    size_t SV = (size_t)C * D, RR = (size_t)R * R;
    double *Ai = malloc(sizeof(double) * RR);
    int rc = 0;
    for (int c = 0; c < C; ++c) {
        if (orc_invert(R, A + (size_t)c * RR, Ai)) { rc = 1; break; }
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < D; ++j) {
                double s = 0.0;
                for (int k = 0; k < R; ++k) s += Ai[(size_t)i * R + k] * Cmx[k * SV + (size_t)c * D + j];
                Tm[i * SV + (size_t)c * D + j] = s;
            }
    }
    free(Ai);
    return rc;
}

/* minDivergence: AccumulateTVStat.cpp:2056-2099. In/out: Rm, r (normalised in place like the
 * reference), ubm_means (+= T^T meanW), T (<- Ch T). */
int orc_tv_min_divergence(int C, int D, int R, double n_sessions, double *Rm, double *r,
                          const double *meanW, double *ubm_means, double *Tm)
{
    size_t SV = (size_t)C * D;
    for (int i = 0; i < R; ++i) r[i] /= n_sessions;
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) Rm[(size_t)i * R + j] = Rm[(size_t)i * R + j] / n_sessions - r[i] * r[j];
    double *ch = malloc(sizeof(double) * R * R);
    if (orc_upper_cholesky(R, Rm, ch)) { free(ch); return 1; }
    for (size_t j = 0; j < SV; ++j)
        for (int k = 0; k < R; ++k) ubm_means[j] += meanW[k] * Tm[k * SV + j];
    double *tmp = calloc((size_t)R * SV, sizeof(double));
    for (int i = 0; i < R; ++i)
        for (int k = 0; k < R; ++k) {
            double c = ch[(size_t)i * R + k];
            if (c == 0.0) continue;
            for (size_t j = 0; j < SV; ++j) tmp[i * SV + j] += c * Tm[k * SV + j];
        }
    memcpy(Tm, tmp, sizeof(double) * R * SV);
    free(tmp); free(ch);
    return 0;
}

/* ---- approximate i-vector extractors (IvExtractor --mode ubmWeight / eigenDecomposition) ----------
 * TVAcc::normStatisticsUnThreaded, LIA_SpkTools/src/AccumulateTVStat.cpp:1225-1242:
 * F[u,c,d] = (F[u,c,d] - mean[c,d] N[u,c]) * sqrt(invvar[c,d]) */
void orc_tv_norm_statistics(long U, int C, int D, const double *N, double *F, const double *ubm_means, const double *invvar)
{
    const size_t SV = (size_t)C * D;
    for (int i = 0; i < C; ++i)
        for (int j = 0; j < D; ++j) {
            const double s = sqrt(invvar[(size_t)i * D + j]);
            for (long u = 0; u < U; ++u) {
                F[u * SV + (size_t)i * D + j] -= ubm_means[(size_t)i * D + j] * N[u * C + i];
                F[u * SV + (size_t)i * D + j] *= s;
            }
        }
}

/* TVAcc::substractMplusTWUnThreaded + getMplusTW, AccumulateTVStat.cpp:1379-1399, 964-971:
 * F[u,c,d] -= (mean[c,d] + sum_i T[i, cD+d] W[u,i]) * N[u,c] */
void orc_tv_subtract_m_plus_tw(long U, int C, int D, int R, const double *N, double *F, const double *ubm_means,
                               const double *Tm, const double *W)
{
    const size_t SV = (size_t)C * D;
    double *m = malloc(sizeof(double) * SV);
    for (long u = 0; u < U; ++u) {
        for (size_t k = 0; k < SV; ++k) {
            double v = 0.0;
            for (int i = 0; i < R; ++i) v += Tm[(size_t)i * SV + k] * W[u * R + i];
            m[k] = ubm_means[k] + v;
        }
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j) F[u * SV + (size_t)i * D + j] -= m[(size_t)i * D + j] * N[u * C + i];
    }
    free(m);
}

/* TVAcc::normTMatrix, AccumulateTVStat.cpp:1600-1609: T[j,i] *= sqrt(invvar[i]) */
void orc_tv_norm_t(int C, int D, int R, double *Tm, const double *invvar)
{
    const size_t SV = (size_t)C * D;
    for (size_t i = 0; i < SV; ++i) {
        const double s = sqrt(invvar[i]);
        for (int j = 0; j < R; ++j) Tm[(size_t)j * SV + i] *= s;
    }
}

/* TVAcc::getWeightedCovUnThreaded, AccumulateTVStat.cpp:2837-2855: W = sum_c weight_c T_c T_c^T (R x R, symmetric) */
void orc_tv_weighted_cov(int C, int D, int R, const double *Tm, const double *weight, double *Wm)
{
    const size_t SV = (size_t)C * D;
    memset(Wm, 0, sizeof(double) * (size_t)R * R);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < R; ++i)
            for (int j = 0; j <= i; ++j) {
                double v = 0.0;
                for (int k = 0; k < D; ++k) v += Tm[(size_t)i * SV + (size_t)c * D + k] * Tm[(size_t)j * SV + (size_t)c * D + k];
                Wm[(size_t)i * R + j] += weight[c] * v;
            }
    for (int i = 0; i < R; ++i)
        for (int j = i; j < R; ++j) Wm[(size_t)i * R + j] = Wm[(size_t)j * R + i];
}

/* TVAcc::approximateTcTcUnThreaded, AccumulateTVStat.cpp:3116-3136: A_c = T_c^T Q (D x R);
 * Dm[c,i] += sum_k A_c[k,i]^2 (accumulating like the reference: zero Dm for a fresh result) */
void orc_tv_approximate_tctc(int C, int D, int R, const double *Tm, const double *Q, double *Dm)
{
    const size_t SV = (size_t)C * D;
    double *A = malloc(sizeof(double) * (size_t)D * R);
    for (int c = 0; c < C; ++c) {
        memset(A, 0, sizeof(double) * (size_t)D * R);
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < R; ++j)
                for (int k = 0; k < R; ++k) A[(size_t)i * R + j] += Tm[(size_t)k * SV + (size_t)c * D + i] * Q[(size_t)k * R + j];
        for (int i = 0; i < R; ++i)
            for (int k = 0; k < D; ++k) Dm[(size_t)c * R + i] += A[(size_t)k * R + i] * A[(size_t)k * R + i];
    }
    free(A);
}

/* TVAcc::estimateWUbmWeightUnThreaded, AccumulateTVStat.cpp:2348-2396: L_u = I + (sum_c N[u,c]) Wm;
 * aux = F[u] T^T (T pre-normalised, no invvar); W[u] += L_u^-1 aux.  Wout is accumulated into, like _W. */
int orc_tv_estimate_w_ubm_weight(long U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                                 const double *Wm, double *Wout)
{
    const size_t SV = (size_t)C * D;
    double *L = malloc(sizeof(double) * (size_t)R * R), *Li = malloc(sizeof(double) * (size_t)R * R);
    double *aux = malloc(sizeof(double) * R);
    int rc = 0;
    for (long u = 0; u < U && !rc; ++u) {
        double ns = 0.0;
        for (int c = 0; c < C; ++c) ns += N[u * C + c];
        for (int i = 0; i < R; ++i) {
            for (int j = 0; j < R; ++j) L[(size_t)i * R + j] = ns * Wm[(size_t)i * R + j];
            L[(size_t)i * R + i] += 1.0;
        }
        rc = orc_invert(R, L, Li);
        for (int i = 0; i < R; ++i) {
            double v = 0.0;
            for (size_t k = 0; k < SV; ++k) v += F[u * SV + k] * Tm[(size_t)i * SV + k];
            aux[i] = v;
        }
        for (int i = 0; i < R; ++i)
            for (int k = 0; k < R; ++k) Wout[u * R + i] += aux[k] * Li[(size_t)i * R + k];
    }
    free(L); free(Li); free(aux);
    return rc;
}

/* TVAcc::estimateWEigenDecompositionUnThreaded, AccumulateTVStat.cpp:2566-2609:
 * invL_i = 1 / (1 + sum_c N[u,c] Dm[c,i]); aux = F[u] T^T; appL = Q diag(invL) Q^T; W[u] += appL aux */
void orc_tv_estimate_w_eigen(long U, int C, int D, int R, const double *N, const double *F, const double *Tm,
                             const double *Dm, const double *Q, double *Wout)
{
    const size_t SV = (size_t)C * D;
    double *il = malloc(sizeof(double) * R), *aux = malloc(sizeof(double) * R), *app = malloc(sizeof(double) * (size_t)R * R);
    for (long u = 0; u < U; ++u) {
        for (int i = 0; i < R; ++i) {
            double t = 1.0;
            for (int c = 0; c < C; ++c) t += N[u * C + c] * Dm[(size_t)c * R + i];
            il[i] = 1 / t;
        }
        for (int i = 0; i < R; ++i) {
            double v = 0.0;
            for (size_t k = 0; k < SV; ++k) v += F[u * SV + k] * Tm[(size_t)i * SV + k];
            aux[i] = v;
        }
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j) {
                double v = 0.0;
                for (int k = 0; k < R; ++k) v += Q[(size_t)i * R + k] * il[k] * Q[(size_t)j * R + k];
                app[(size_t)i * R + j] = v;
            }
        for (int i = 0; i < R; ++i)
            for (int k = 0; k < R; ++k) Wout[u * R + i] += aux[k] * app[(size_t)i * R + k];
    }
    free(il); free(aux); free(app);
}

/* ---- i-vector back-end estimation on a development set (PldaDev, LIA_SpkTools/src/PldaTools.cpp) --------------
 * data X [dim x n] (one vector per column); sessions are grouped by speaker, sps[c] = sessions of speaker c
 * (PldaDev::_session_per_speaker; _class[s] is then the non-decreasing speaker index of session s).
 * PldaDev::computeAll, :353-387: global mean and per-speaker means. */
void orc_dev_means(int dim, long n, const double *X, long nspk, const long *sps, double *mean, double *spk_means /* dim x nspk */)
{
    memset(mean, 0, sizeof(double) * dim);
    memset(spk_means, 0, sizeof(double) * (size_t)dim * nspk);
    long s = 0;
    for (long c = 0; c < nspk; ++c)
        for (long e = 0; e < sps[c]; ++e, ++s)
            for (int k = 0; k < dim; ++k) { spk_means[(size_t)k * nspk + c] += X[(size_t)k * n + s]; mean[k] += X[(size_t)k * n + s]; }
    for (int k = 0; k < dim; ++k) {
        mean[k] /= (double)n;
        for (long c = 0; c < nspk; ++c) spk_means[(size_t)k * nspk + c] /= (double)sps[c];
    }
}
static long *dev_classes(long n, long nspk, const long *sps)
{
    long *cls = malloc(sizeof(long) * (n > 0 ? n : 1)), s = 0;
    for (long c = 0; c < nspk; ++c) for (long e = 0; e < sps[c]; ++e) cls[s++] = c;
    return cls;
}
/* PldaDev::computeCovMatUnThreaded, :527-566: total, within and between covariance, all divided by n_sessions */
void orc_dev_cov_mat(int dim, long n, const double *X, long nspk, const long *sps, double *Sigma, double *W, double *B)
{
    double *mean = malloc(sizeof(double) * dim), *sm = malloc(sizeof(double) * (size_t)dim * nspk);
    long *cls = dev_classes(n, nspk, sps);
    orc_dev_means(dim, n, X, nspk, sps, mean, sm);
    for (int i = 0; i < dim; ++i)
        for (int j = i; j < dim; ++j) {
            double sg = 0.0, w = 0.0, b = 0.0;
            for (long s = 0; s < n; ++s) {
                sg += (X[(size_t)i * n + s] - mean[i]) * (X[(size_t)j * n + s] - mean[j]);
                w += (X[(size_t)i * n + s] - sm[(size_t)i * nspk + cls[s]]) * (X[(size_t)j * n + s] - sm[(size_t)j * nspk + cls[s]]);
            }
            for (long c = 0; c < nspk; ++c) b += sps[c] * (sm[(size_t)i * nspk + c] - mean[i]) * (sm[(size_t)j * nspk + c] - mean[j]);
            Sigma[(size_t)i * dim + j] = Sigma[(size_t)j * dim + i] = sg / n;
            W[(size_t)i * dim + j] = W[(size_t)j * dim + i] = w / n;
            B[(size_t)i * dim + j] = B[(size_t)j * dim + i] = b / n;
        }
    free(mean); free(sm); free(cls);
}
/* PldaDev::computeWccnCholUnThreaded, :1124-1176: W = mean over speakers of the per-speaker covariance
 * (each divided by its session count); WCCN = upperCholesky(W^-1) */
int orc_dev_wccn_chol(int dim, long n, const double *X, long nspk, const long *sps, double *WCCN)
{
    double *mean = malloc(sizeof(double) * dim), *sm = malloc(sizeof(double) * (size_t)dim * nspk);
    double *W = calloc((size_t)dim * dim, sizeof(double)), *iW = malloc(sizeof(double) * (size_t)dim * dim);
    orc_dev_means(dim, n, X, nspk, sps, mean, sm);
    long s0 = 0;
    for (long c = 0; c < nspk; ++c) {
        for (int i = 0; i < dim; ++i)
            for (int j = i; j < dim; ++j) {
                double v = 0.0;
                for (long s = s0; s < s0 + sps[c]; ++s)
                    v += (X[(size_t)i * n + s] - sm[(size_t)i * nspk + c]) * (X[(size_t)j * n + s] - sm[(size_t)j * nspk + c]);
                W[(size_t)i * dim + j] += v / (double)sps[c];
            }
        s0 += sps[c];
    }
    for (int i = 0; i < dim; ++i)
        for (int j = i; j < dim; ++j) { W[(size_t)i * dim + j] /= nspk; W[(size_t)j * dim + i] = W[(size_t)i * dim + j]; }
    int rc = orc_invert(dim, W, iW);
    if (!rc) rc = orc_upper_cholesky(dim, iW, WCCN);
    free(mean); free(sm); free(W); free(iW);
    return rc;
}
/* PldaDev::computeScatterMatUnThreaded, :1610-1644, AS WRITTEN: SB = sum_c (m_c - m)(m_c - m)^T (no weights, no
 * normalisation); SW is ASSIGNED per speaker, so it ends as the last speaker's matrix, and that matrix is built from
 * the FIRST sps[c] sessions of the whole set (index s restarts at 0), each centred on its own speaker mean. */
void orc_dev_scatter_mat(int dim, long n, const double *X, long nspk, const long *sps, double *SB, double *SW)
{
    double *mean = malloc(sizeof(double) * dim), *sm = malloc(sizeof(double) * (size_t)dim * nspk);
    long *cls = dev_classes(n, nspk, sps);
    orc_dev_means(dim, n, X, nspk, sps, mean, sm);
    memset(SB, 0, sizeof(double) * (size_t)dim * dim);
    memset(SW, 0, sizeof(double) * (size_t)dim * dim);
    for (long c = 0; c < nspk; ++c)
        for (int i = 0; i < dim; ++i)
            for (int j = 0; j < dim; ++j) {
                SB[(size_t)i * dim + j] += (sm[(size_t)i * nspk + c] - mean[i]) * (sm[(size_t)j * nspk + c] - mean[j]);
                double t = 0.0;
                for (long s = 0; s < sps[c]; ++s)
                    t += (X[(size_t)i * n + s] - sm[(size_t)i * nspk + cls[s]]) * (X[(size_t)j * n + s] - sm[(size_t)j * nspk + cls[s]]);
                SW[(size_t)i * dim + j] = t / (double)sps[c];
            }
    free(mean); free(sm); free(cls);
}
/* cyclic Jacobi for a symmetric matrix: eigenvalues descending in val[rank], vect[k*rank + j] = component k of the
 * j-th eigenvector -- the layout PldaDev::computeEigenProblem (:1490-1535) fills from Eigen::EigenSolver */
void orc_sym_eigen(int n, const double *A, int rank, double *vect, double *val)
{
    double *a = malloc(sizeof(double) * (size_t)n * n), *v = calloc((size_t)n * n, sizeof(double));
    memcpy(a, A, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) { dg += a[(size_t)i * n + i] * a[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j]; }
        if (off <= 1e-30 * (dg + off)) break;
        for (int p = 0; p + 1 < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double th = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) { double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q]; a[(size_t)k * n + p] = cs * x - sn * y; a[(size_t)k * n + q] = sn * x + cs * y; }
                for (int k = 0; k < n; ++k) { double x = a[(size_t)p * n + k], y = a[(size_t)q * n + k]; a[(size_t)p * n + k] = cs * x - sn * y; a[(size_t)q * n + k] = sn * x + cs * y; }
                for (int k = 0; k < n; ++k) { double x = v[(size_t)k * n + p], y = v[(size_t)k * n + q]; v[(size_t)k * n + p] = cs * x - sn * y; v[(size_t)k * n + q] = sn * x + cs * y; }
            }
    }
    int *ord = malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    for (int i = 1; i < n; ++i) { int o = ord[i], j = i - 1; while (j >= 0 && a[(size_t)ord[j] * n + ord[j]] < a[(size_t)o * n + o]) { ord[j + 1] = ord[j]; --j; } ord[j + 1] = o; }
    for (int j = 0; j < rank; ++j) {
        val[j] = a[(size_t)ord[j] * n + ord[j]];
        for (int k = 0; k < n; ++k) vect[(size_t)k * rank + j] = v[(size_t)k * n + ord[j]];
    }
    free(a); free(v); free(ord);
}
/* EFR / sphNorm matrix of PldaDev::sphericalNuisanceNormalization, :1852-1902: (V diag(1/sqrt(lambda)))^T for the
 * eigen-decomposition of the covariance (Sigma for EFR, W for sphNorm) */
void orc_dev_efr_matrix(int dim, const double *Cov, double *Mout)
{
    double *vect = malloc(sizeof(double) * (size_t)dim * dim), *val = malloc(sizeof(double) * dim);
    orc_sym_eigen(dim, Cov, dim, vect, val);
    for (int j = 0; j < dim; ++j)
        for (int k = 0; k < dim; ++k) Mout[(size_t)j * dim + k] = vect[(size_t)k * dim + j] / sqrt(val[j]);
    free(vect); free(val);
}
/* PldaDev::computeLDA, :1381-1413: the ldaRank leading eigenvectors (unit norm, as Eigen::EigenSolver returns them)
 * of W^-1 B as the ROWS of ldaMat [rank x dim].  Solved through the symmetric form L^-1 B L^-T with W = L L^T. */
int orc_dev_lda(int dim, const double *W, const double *B, int rank, double *ldaMat, double *eigval)
{
    double *U = malloc(sizeof(double) * (size_t)dim * dim);
    int rc = orc_upper_cholesky(dim, W, U); /* W = U^T U, L = U^T */
    if (rc) { free(U); return rc; }
    double *Cm = malloc(sizeof(double) * (size_t)dim * dim), *T1 = malloc(sizeof(double) * (size_t)dim * dim);
    /* T1 = L^-1 B : forward substitution on columns of B with L = U^T */
    for (int j = 0; j < dim; ++j)
        for (int i = 0; i < dim; ++i) {
            double v = B[(size_t)i * dim + j];
            for (int k = 0; k < i; ++k) v -= U[(size_t)k * dim + i] * T1[(size_t)k * dim + j];
            T1[(size_t)i * dim + j] = v / U[(size_t)i * dim + i];
        }
    /* Cm = T1 L^-T  <=>  Cm^T = L^-1 T1^T */
    for (int j = 0; j < dim; ++j)
        for (int i = 0; i < dim; ++i) {
            double v = T1[(size_t)j * dim + i];
            for (int k = 0; k < i; ++k) v -= U[(size_t)k * dim + i] * Cm[(size_t)j * dim + k];
            Cm[(size_t)j * dim + i] = v / U[(size_t)i * dim + i];
        }
    for (int i = 0; i < dim; ++i) for (int j = i + 1; j < dim; ++j) { double m = 0.5 * (Cm[(size_t)i * dim + j] + Cm[(size_t)j * dim + i]); Cm[(size_t)i * dim + j] = Cm[(size_t)j * dim + i] = m; }
    double *vect = malloc(sizeof(double) * (size_t)dim * rank), *val = malloc(sizeof(double) * rank);
    orc_sym_eigen(dim, Cm, rank, vect, val);
    for (int j = 0; j < rank; ++j) { /* v = L^-T y = U^-1 y (back substitution), then unit norm */
        double nrm = 0.0;
        for (int i = dim - 1; i >= 0; --i) {
            double v = vect[(size_t)i * rank + j];
            for (int k = i + 1; k < dim; ++k) v -= U[(size_t)i * dim + k] * ldaMat[(size_t)j * dim + k];
            ldaMat[(size_t)j * dim + i] = v / U[(size_t)i * dim + i];
        }
        for (int i = 0; i < dim; ++i) nrm += ldaMat[(size_t)j * dim + i] * ldaMat[(size_t)j * dim + i];
        nrm = sqrt(nrm);
        for (int i = 0; i < dim; ++i) ldaMat[(size_t)j * dim + i] /= nrm;
        if (eigval) eigval[j] = val[j];
    }
    free(U); free(Cm); free(T1); free(vect); free(val);
    return 0;
}

static void mm(int M, int N, int K, const double *A, int ta, const double *B, int tb, double *Cm);
/* PldaModel::em_iteration = center(Delta) + computeCovMatEigen + getExpectedValuesUnThreaded + mStep,
 * LIA_SpkTools/src/PldaTools.cpp:2329-2343, 931-950, 2359-2484, 2790-2815.  X [dim x n] (centred in place by Delta,
 * like _Dev.center), sessions grouped by speaker; F [dim x rf], G [dim x rg], Sigma [dim x dim], Delta [dim] updated.
 * The reference builds M = V (n D + I)^-1 V^T from the eigen-decomposition of A; that is (n A + I)^-1, restated here
 * with the explicit inverse (no dependence on the eigen-solver's ordering). */
int orc_plda_em_iteration(int dim, long n, double *X, long nspk, const long *sps, int rf, int rg, double *F, double *G,
                          double *Sigma, double *Delta)
{
    const int rh = rf + rg;
    int rc = 0;
    for (long s = 0; s < n; ++s) for (int k = 0; k < dim; ++k) X[(size_t)k * n + s] -= Delta[k];
    double *sigObs = calloc((size_t)dim * dim, sizeof(double));
    for (int i = 0; i < dim; ++i)
        for (int j = i; j < dim; ++j) {
            double v = 0.0;
            for (long s = 0; s < n; ++s) v += X[(size_t)i * n + s] * X[(size_t)j * n + s];
            sigObs[(size_t)i * dim + j] = sigObs[(size_t)j * dim + i] = v;
        }
    /* preComputation */
    double *Si = malloc(sizeof(double) * (size_t)dim * dim), *Ftw = malloc(sizeof(double) * (size_t)rf * dim), *Gtw = malloc(sizeof(double) * (size_t)rg * dim);
    double *GtwG = malloc(sizeof(double) * (size_t)rg * rg), *FtwG = malloc(sizeof(double) * (size_t)rf * rg), *iGG = malloc(sizeof(double) * (size_t)rg * rg);
    double *FtwF = malloc(sizeof(double) * (size_t)rf * rf), *Sm = malloc(sizeof(double) * (size_t)rg * rf), *A = malloc(sizeof(double) * (size_t)rf * rf);
    double *t1 = malloc(sizeof(double) * (size_t)rf * rg);
    rc |= orc_invert(dim, Sigma, Si);
    mm(rf, dim, dim, F, 1, Si, 0, Ftw);
    mm(rg, dim, dim, G, 1, Si, 0, Gtw);
    mm(rg, rg, dim, Gtw, 0, G, 0, GtwG);
    mm(rf, rg, dim, Ftw, 0, G, 0, FtwG);
    for (int i = 0; i < rg; ++i) GtwG[(size_t)i * rg + i] += 1.0;
    rc |= orc_invert(rg, GtwG, iGG);
    mm(rf, rf, dim, Ftw, 0, F, 0, FtwF);
    mm(rg, rf, rg, iGG, 0, FtwG, 1, Sm);                 /* S = iGG FtwG^T  [rg x rf] */
    mm(rf, rg, rg, FtwG, 0, iGG, 0, t1);
    mm(rf, rf, rg, t1, 0, FtwG, 1, A);
    for (size_t i = 0; i < (size_t)rf * rf; ++i) A[i] = FtwF[i] - A[i];
    double *Ehh = calloc((size_t)rh * rh, sizeof(double)), *xh = calloc((size_t)dim * rh, sizeof(double)), *U = calloc(rh, sizeof(double));
    double *M = malloc(sizeof(double) * (size_t)rf * rf), *MsT = malloc(sizeof(double) * (size_t)rf * rg), *J = malloc(sizeof(double) * (size_t)rf * rf);
    double *tmpM = malloc(sizeof(double) * (size_t)rh * rh), *SMsT = malloc(sizeof(double) * (size_t)rg * rg);
    long cur = 0, s0 = 0;
    for (long spk = 0; spk < nspk && !rc; ++spk) {
        const long ns = sps[spk];
        if (ns != cur) {
            cur = ns;
            for (size_t i = 0; i < (size_t)rf * rf; ++i) J[i] = (double)ns * A[i];
            for (int i = 0; i < rf; ++i) J[(size_t)i * rf + i] += 1.0;
            rc |= orc_invert(rf, J, M);
            mm(rf, rg, rf, M, 0, Sm, 1, MsT);            /* M S^T */
            mm(rg, rg, rf, Sm, 0, MsT, 0, SMsT);
            for (int i = 0; i < rf; ++i) for (int j = 0; j < rf; ++j) tmpM[(size_t)i * rh + j] = M[(size_t)i * rf + j];
            for (int i = 0; i < rf; ++i) for (int j = 0; j < rg; ++j) { tmpM[(size_t)i * rh + rf + j] = -MsT[(size_t)i * rg + j]; tmpM[(size_t)(rf + j) * rh + i] = -MsT[(size_t)i * rg + j]; }
            for (int i = 0; i < rg; ++i) for (int j = 0; j < rg; ++j) tmpM[(size_t)(rf + i) * rh + rf + j] = iGG[(size_t)i * rg + j] + SMsT[(size_t)i * rg + j];
        }
        double *fi = malloc(sizeof(double) * (size_t)rf * ns), *gi = malloc(sizeof(double) * (size_t)rg * ns);
        double *f = calloc(rf, sizeof(double)), *g = calloc(rg, sizeof(double)), *h = malloc(sizeof(double) * rf), *Eh = malloc(sizeof(double) * (size_t)rh * ns);
        for (long j = 0; j < ns; ++j) {
            for (int r = 0; r < rf; ++r) { double v = 0.0; for (int k = 0; k < dim; ++k) v += Ftw[(size_t)r * dim + k] * X[(size_t)k * n + s0 + j]; fi[(size_t)r * ns + j] = v; f[r] += v; }
            for (int r = 0; r < rg; ++r) { double v = 0.0; for (int k = 0; k < dim; ++k) v += Gtw[(size_t)r * dim + k] * X[(size_t)k * n + s0 + j]; gi[(size_t)r * ns + j] = v; g[r] += v; }
        }
        for (int r = 0; r < rf; ++r) { double v = f[r]; for (int k = 0; k < rg; ++k) v -= Sm[(size_t)k * rf + r] * g[k]; fi[(size_t)r * ns] = v; } /* reuse column 0 of fi as (f - S^T g) */
        for (int r = 0; r < rf; ++r) { double v = 0.0; for (int k = 0; k < rf; ++k) v += M[(size_t)r * rf + k] * fi[(size_t)k * ns]; h[r] = v; }
        for (long j = 0; j < ns; ++j) {
            for (int r = 0; r < rf; ++r) Eh[(size_t)r * ns + j] = h[r];
            for (int r = 0; r < rg; ++r) {
                double v = 0.0;
                for (int k = 0; k < rg; ++k) v += iGG[(size_t)r * rg + k] * gi[(size_t)k * ns + j];
                for (int k = 0; k < rf; ++k) v -= Sm[(size_t)r * rf + k] * h[k];
                Eh[(size_t)(rf + r) * ns + j] = v;
            }
        }
        for (int i = 0; i < rh; ++i)
            for (int j = 0; j < rh; ++j) {
                double v = (double)ns * tmpM[(size_t)i * rh + j];
                for (long k = 0; k < ns; ++k) v += Eh[(size_t)i * ns + k] * Eh[(size_t)j * ns + k];
                Ehh[(size_t)i * rh + j] += v;
            }
        for (int i = 0; i < dim; ++i)
            for (int j = 0; j < rh; ++j) { double v = 0.0; for (long k = 0; k < ns; ++k) v += X[(size_t)i * n + s0 + k] * Eh[(size_t)j * ns + k]; xh[(size_t)i * rh + j] += v; }
        for (long k = 0; k < ns; ++k) for (int i = 0; i < rh; ++i) U[i] += Eh[(size_t)i * ns + k];
        free(fi); free(gi); free(f); free(g); free(h); free(Eh);
        s0 += ns;
    }
    /* mStep */
    double *iE = malloc(sizeof(double) * (size_t)rh * rh), *FG = malloc(sizeof(double) * (size_t)dim * rh), *SL = malloc(sizeof(double) * (size_t)dim * dim);
    rc |= orc_invert(rh, Ehh, iE);
    mm(dim, rh, rh, xh, 0, iE, 0, FG);
    mm(dim, dim, rh, FG, 0, xh, 1, SL);
    for (size_t i = 0; i < (size_t)dim * dim; ++i) Sigma[i] = (sigObs[i] - SL[i]) / (double)n;
    for (int i = 0; i < rh; ++i) U[i] /= (double)n;
    double *cF = malloc(sizeof(double) * (size_t)rf * rf), *cG = malloc(sizeof(double) * (size_t)rg * rg), *Rh = malloc(sizeof(double) * (size_t)rf * rf), *Rw = malloc(sizeof(double) * (size_t)rg * rg);
    for (int i = 0; i < rf; ++i) for (int j = 0; j < rf; ++j) cF[(size_t)i * rf + j] = Ehh[(size_t)i * rh + j] / (double)n - U[i] * U[j];
    for (int i = 0; i < rg; ++i) for (int j = 0; j < rg; ++j) cG[(size_t)i * rg + j] = Ehh[(size_t)(rf + i) * rh + rf + j] / (double)n - U[rf + i] * U[rf + j];
    rc |= orc_upper_cholesky(rf, cF, Rh);
    rc |= orc_upper_cholesky(rg, cG, Rw);
    for (int i = 0; i < dim; ++i) {   /* F = FGEst[:, :rf] Rh^T ; G = FGEst[:, rf:] Rw^T ; Delta += FGEst U */
        for (int j = 0; j < rf; ++j) { double v = 0.0; for (int k = 0; k < rf; ++k) v += FG[(size_t)i * rh + k] * Rh[(size_t)j * rf + k]; F[(size_t)i * rf + j] = v; }
        for (int j = 0; j < rg; ++j) { double v = 0.0; for (int k = 0; k < rg; ++k) v += FG[(size_t)i * rh + rf + k] * Rw[(size_t)j * rg + k]; G[(size_t)i * rg + j] = v; }
        double d = 0.0;
        for (int k = 0; k < rh; ++k) d += FG[(size_t)i * rh + k] * U[k];
        Delta[i] += d;
    }
    free(sigObs); free(Si); free(Ftw); free(Gtw); free(GtwG); free(FtwG); free(iGG); free(FtwF); free(Sm); free(A); free(t1);
    free(Ehh); free(xh); free(U); free(M); free(MsT); free(J); free(tmpM); free(SMsT); free(iE); free(FG); free(SL);
    free(cF); free(cG); free(Rh); free(Rw);
    return rc;
}

/* PldaModel::preComputation + the first lines of PldaTest::pldaNativeScoring, LIA_SpkTools/src/PldaTools.cpp:2950-2972,
 * 4494-4496.  F [dim x rf], G [dim x rg], Sigma [dim x dim] (row-major):
 *   FTJ  = F^T S^-1 - F^T S^-1 G (G^T S^-1 G + I)^-1 G^T S^-1      [rf x dim]
 *   FTJF = FTJ F                                                    [rf x rf]   */
static void mm(int M, int N, int K, const double *A, int ta, const double *B, int tb, double *Cm)
{   /* C[M x N] = op(A) op(B); op(A) is M x K */
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double v = 0.0;
            for (int k = 0; k < K; ++k) v += (ta ? A[(size_t)k * M + i] : A[(size_t)i * K + k]) * (tb ? B[(size_t)j * K + k] : B[(size_t)k * N + j]);
            Cm[(size_t)i * N + j] = v;
        }
}
int orc_plda_precompute(int dim, int rf, int rg, const double *F, const double *G, const double *Sigma, double *FTJ, double *FTJF)
{
    double *Si = malloc(sizeof(double) * (size_t)dim * dim), *Ftw = malloc(sizeof(double) * (size_t)rf * dim);
    int rc = orc_invert(dim, Sigma, Si);
    mm(rf, dim, dim, F, 1, Si, 0, Ftw);
    memcpy(FTJ, Ftw, sizeof(double) * (size_t)rf * dim);
    if (rg > 0 && !rc) {
        double *Gtw = malloc(sizeof(double) * (size_t)rg * dim), *GtwG = malloc(sizeof(double) * (size_t)rg * rg);
        double *Mi = malloc(sizeof(double) * (size_t)rg * rg), *FtwG = malloc(sizeof(double) * (size_t)rf * rg);
        double *t1 = malloc(sizeof(double) * (size_t)rf * rg), *t2 = malloc(sizeof(double) * (size_t)rf * dim);
        mm(rg, dim, dim, G, 1, Si, 0, Gtw);
        mm(rg, rg, dim, Gtw, 0, G, 0, GtwG);
        mm(rf, rg, dim, Ftw, 0, G, 0, FtwG);
        for (int i = 0; i < rg; ++i) GtwG[(size_t)i * rg + i] += 1.0;
        rc = orc_invert(rg, GtwG, Mi);
        mm(rf, rg, rg, FtwG, 0, Mi, 0, t1);
        mm(rf, dim, rg, t1, 0, Gtw, 0, t2);
        for (size_t i = 0; i < (size_t)rf * dim; ++i) FTJ[i] -= t2[i];
        free(Gtw); free(GtwG); free(Mi); free(FtwG); free(t1); free(t2);
    }
    mm(rf, rf, dim, FTJ, 0, F, 0, FTJF);
    free(Si); free(Ftw);
    return rc;
}

/* orthonormalizeT: classical Gram-Schmidt over the rows of T, AccumulateTVStat.cpp:1548-1596 */
void orc_tv_orthonormalize_t(int R, size_t SV, double *Tm)
{
    double *Q = calloc((size_t)R * SV, sizeof(double));
    double *v = malloc(sizeof(double) * SV);
    for (int j = 0; j < R; ++j) {
        memcpy(v, Tm + j * SV, sizeof(double) * SV);
        for (int i = 0; i < j; ++i) {
            double rij = 0.0;
            for (size_t k = 0; k < SV; ++k) rij += Q[i * SV + k] * Tm[j * SV + k];
            for (size_t k = 0; k < SV; ++k) v[k] -= rij * Q[i * SV + k];
        }
        double nv = 0.0;
        for (size_t k = 0; k < SV; ++k) nv += v[k] * v[k];
        double rjj = sqrt(nv);
        if (rjj == 0.0) for (size_t k = 0; k < SV; ++k) Q[j * SV + k] = 0;
        else for (size_t k = 0; k < SV; ++k) Q[j * SV + k] = v[k] / rjj;
    }
    memcpy(Tm, Q, sizeof(double) * R * SV);
    free(Q); free(v);
}

/* ---------------------------------------------------------------------------------------------
 * Row 19: i-vector scoring (PldaTest). models[dim x M], segments[dim x S] column = vector, like
 * the reference's _models/_segments; scores[M x S]; trials mask (NULL = all).
 * ------------------------------------------------------------------------------------------- */

/* PldaTest::center -> rotateLeft -> lengthNorm: LIA_SpkTools/src/PldaTools.cpp:3754-3767, 3770-3790,
 * 3706-3751 (one iteration of sphericalNuisanceNormalization :3793-3839).  X[din x n] vectors as
 * columns; mean / M may be NULL; Y[dout x n]. */
void orc_iv_normalize(int din, int dout, long n, const double *X, const double *mean, const double *Mx,
                      int length_norm, double *Y)
{
    double *c = malloc(sizeof(double) * din * n);
    for (int k = 0; k < din; ++k)
        for (long s = 0; s < n; ++s) c[k * n + s] = X[k * n + s] - (mean ? mean[k] : 0.0);
    for (int i = 0; i < dout; ++i)
        for (long s = 0; s < n; ++s) {
            double a = 0.0;
            if (Mx) for (int k = 0; k < din; ++k) a += Mx[(size_t)i * din + k] * c[k * n + s];
            else a = c[i * n + s];
            Y[i * n + s] = a;
        }
    if (length_norm)
        for (long s = 0; s < n; ++s) {
            double t = 0.0;
            for (int k = 0; k < dout; ++k) t += Y[k * n + s] * Y[k * n + s];
            t = sqrt(t);
            for (int k = 0; k < dout; ++k) Y[k * n + s] /= t;
        }
    free(c);
}

/* cosineDistance: LIA_SpkTools/src/PldaTools.cpp:3842-3879 */
void orc_score_cosine(int dim, long M, long S, const double *models, const double *segs,
                      const unsigned char *trials, double *scores)
{
    double *nm = calloc(M, sizeof(double)), *ns = calloc(S, sizeof(double));
    for (int k = 0; k < dim; ++k) {
        for (long m = 0; m < M; ++m) nm[m] += models[k * M + m] * models[k * M + m];
        for (long s = 0; s < S; ++s) ns[s] += segs[k * S + s] * segs[k * S + s];
    }
    for (long m = 0; m < M; ++m) nm[m] = sqrt(nm[m]);
    for (long s = 0; s < S; ++s) ns[s] = sqrt(ns[s]);
    for (long m = 0; m < M; ++m)
        for (long s = 0; s < S; ++s) {
            if (trials && !trials[m * S + s]) continue;
            double v = 0.0;
            for (int k = 0; k < dim; ++k) v += models[k * M + m] * segs[k * S + s];
            scores[m * S + s] = v / (nm[m] * ns[s]);
        }
    free(nm); free(ns);
}

/* mahalanobisDistance: PldaTools.cpp:3882-3909. score = -1/2 (m-s)^T Mah (m-s) */
void orc_score_mahalanobis(int dim, long M, long S, const double *models, const double *segs,
                           const double *Mah, const unsigned char *trials, double *scores)
{
    double *d = malloc(sizeof(double) * dim), *t = malloc(sizeof(double) * dim);
    for (long m = 0; m < M; ++m)
        for (long s = 0; s < S; ++s) {
            if (trials && !trials[m * S + s]) continue;
            for (int k = 0; k < dim; ++k) d[k] = models[k * M + m] - segs[k * S + s];
            for (int k = 0; k < dim; ++k) {
                double a = 0.0;
                for (int i = 0; i < dim; ++i) a += -0.5 * d[i] * Mah[(size_t)i * dim + k];
                t[k] = a;
            }
            double v = 0.0;
            for (int i = 0; i < dim; ++i) v += t[i] * d[i];
            scores[m * S + s] = v;
        }
    free(d); free(t);
}

static void matmul(int n, const double *a, const double *b, double *c)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += a[(size_t)i * n + k] * b[(size_t)k * n + j];
            c[(size_t)i * n + j] = s;
        }
}

/* two-covariance model matrices G, H from W, B: PldaTools.cpp:4089-4125 */
int orc_twocov_model(int dim, const double *Wm, const double *Bm, double *G, double *H)
{
    size_t nn = (size_t)dim * dim;
    double *iW = malloc(sizeof(double) * nn), *iB = malloc(sizeof(double) * nn);
    double *sG = malloc(sizeof(double) * nn), *sH = malloc(sizeof(double) * nn);
    double *tG = malloc(sizeof(double) * nn), *tH = malloc(sizeof(double) * nn);
    double *t2 = malloc(sizeof(double) * nn);
    int rc = orc_invert(dim, Wm, iW) | orc_invert(dim, Bm, iB);
    for (size_t e = 0; e < nn; ++e) { sG[e] = iB[e] + 2 * iW[e]; sH[e] = iB[e] + iW[e]; }
    rc |= orc_invert(dim, sG, tG) | orc_invert(dim, sH, tH);
    matmul(dim, iW, tG, t2); matmul(dim, t2, iW, G);
    matmul(dim, iW, tH, t2); matmul(dim, t2, iW, H);
    free(iW); free(iB); free(sG); free(sH); free(tG); free(tH); free(t2);
    return rc;
}

/* twoCovScoring scores given G,H: PldaTools.cpp:4127-4171 + :3923-3949.
 * score = (m+s)^T G (m+s) - m^T H m - s^T H s   (all pairs, no mask in the reference) */
void orc_score_twocov(int dim, long M, long S, const double *models, const double *segs,
                      const double *G, const double *H, double *scores)
{
    double *md = calloc(M, sizeof(double)), *sd = calloc(S, sizeof(double));
    double *a = malloc(sizeof(double) * dim), *d = malloc(sizeof(double) * dim);
    for (long m = 0; m < M; ++m) {
        for (int k = 0; k < dim; ++k) {
            double s = 0.0;
            for (int i = 0; i < dim; ++i) s += models[i * M + m] * H[(size_t)i * dim + k];
            a[k] = s;
        }
        for (int j = 0; j < dim; ++j) md[m] += a[j] * models[j * M + m];
    }
    for (long s = 0; s < S; ++s) {
        for (int k = 0; k < dim; ++k) {
            double v = 0.0;
            for (int i = 0; i < dim; ++i) v += segs[i * S + s] * H[(size_t)i * dim + k];
            a[k] = v;
        }
        for (int j = 0; j < dim; ++j) sd[s] += a[j] * segs[j * S + s];
    }
    for (long m = 0; m < M; ++m)
        for (long s = 0; s < S; ++s) {
            for (int j = 0; j < dim; ++j) d[j] = models[j * M + m] + segs[j * S + s];
            double v = 0.0;
            for (int k = 0; k < dim; ++k) {
                double c = 0.0;
                for (int i = 0; i < dim; ++i) c += d[i] * G[(size_t)i * dim + k];
                v += c * d[k];
            }
            scores[m * S + s] = v - (md[m] + sd[s]);
        }
    free(md); free(sd); free(a); free(d);
}

static double logdet_spd(int n, const double *a)
{
    /* alpha = 2 sum log diag(chol(K)): PldaTools.cpp:4236-4241 */
    double *ch = malloc(sizeof(double) * n * n);
    double s = 0.0;
    if (orc_upper_cholesky(n, a, ch) == 0)
        for (int i = 0; i < n; ++i) s += log(ch[(size_t)i * n + i]);
    else
        s = NAN;
    free(ch);
    return 2.0 * s;
}

/* PLDA native scoring on vectors ALREADY projected by FTJ (rank rf):
 * pldaScoringUnThreaded PldaTools.cpp:4186-4271 (+ K_one/alpha_one from :4506-4516).
 * models[rf x M] are the per-speaker SUMS of nsess[m] enrolment vectors.
 * K_n = (n FTJF + I)^-1, alpha_n = log det K_n,
 * score = ((s+m)^T K_{L+1} (s+m) - m^T K_L m - s^T K_1 s)/2 + (alpha_{L+1} - alpha_L - alpha_1)/2 */
int orc_score_plda(int rf, long M, long S, const double *models, const long *nsess,
                   const double *segs, const double *FTJF, double *scores)
{
    size_t nn = (size_t)rf * rf;
    double *K1 = malloc(sizeof(double) * nn), *KL = malloc(sizeof(double) * nn), *KL1 = malloc(sizeof(double) * nn);
    double *tmp = malloc(sizeof(double) * nn), *v = malloc(sizeof(double) * rf);
    double *s1 = malloc(sizeof(double) * S);
    int rc = 0;
    for (size_t e = 0; e < nn; ++e) tmp[e] = FTJF[e];
    for (int i = 0; i < rf; ++i) tmp[(size_t)i * rf + i] += 1.0;
    rc |= orc_invert(rf, tmp, K1);
    double alpha1 = logdet_spd(rf, K1);
    for (long s = 0; s < S; ++s) {
        double q = 0.0;
        for (int i = 0; i < rf; ++i) {
            double a = 0.0;
            for (int j = 0; j < rf; ++j) a += K1[(size_t)i * rf + j] * segs[j * S + s];
            q += segs[i * S + s] * a;
        }
        s1[s] = q;
    }
    long cur = -1;
    double cst = 0.0;
    for (long m = 0; m < M; ++m) {
        if (nsess[m] != cur) {
            cur = nsess[m];
            for (size_t e = 0; e < nn; ++e) tmp[e] = cur * FTJF[e];
            for (int i = 0; i < rf; ++i) tmp[(size_t)i * rf + i] += 1.0;
            rc |= orc_invert(rf, tmp, KL);
            for (size_t e = 0; e < nn; ++e) tmp[e] = (cur + 1) * FTJF[e];
            for (int i = 0; i < rf; ++i) tmp[(size_t)i * rf + i] += 1.0;
            rc |= orc_invert(rf, tmp, KL1);
            cst = (logdet_spd(rf, KL1) - logdet_spd(rf, KL) - alpha1) / 2.0;
        }
        double s2 = 0.0;
        for (int i = 0; i < rf; ++i) {
            double a = 0.0;
            for (int j = 0; j < rf; ++j) a += KL[(size_t)i * rf + j] * models[j * M + m];
            s2 += models[i * M + m] * a;
        }
        for (long s = 0; s < S; ++s) {
            for (int i = 0; i < rf; ++i) v[i] = segs[i * S + s] + models[i * M + m];
            double s3 = 0.0;
            for (int i = 0; i < rf; ++i) {
                double a = 0.0;
                for (int j = 0; j < rf; ++j) a += KL1[(size_t)i * rf + j] * v[j];
                s3 += v[i] * a;
            }
            scores[m * S + s] = (s3 - s2 - s1[s]) / 2.0 + cst;
        }
    }
    free(K1); free(KL); free(KL1); free(tmp); free(v); free(s1);
    return rc;
}

/* ---- JFA: LIA_SpkTools/src/AccumulateJFAStat.cpp -------------------------------------------------------------------------
 * M_{s,h} = m + V y_s + U x_h + D z_s.  N [nspk x C] / F_X [nspk x SV] per speaker, N_h / F_X_h per session. */

/* estimateAndInverseLUnThreaded_EV (:1970-1996) + estimateYandVUnThreaded (:2467-2511) -- the same two loops serve the
 * eigenchannel side (_EC :2137-2163, estimateXandU :3040-3083) with per-session statistics.  A is FULL [C x R x R]. */
int orc_jfa_estimate_y_and_v(long nspk, int C, int D, int R, const double *N, const double *F, const double *V, const double *invvar,
                             const double *VEVT /* C x R x R */, double *Y, double *A, double *Cmx)
{
    const size_t SV = (size_t)C * D;
    double *L = (double *)malloc(sizeof(double) * R * R), *invl = (double *)malloc(sizeof(double) * R * R);
    double *aux = (double *)calloc(R, sizeof(double));
    int rc = 0;
    for (long spk = 0; spk < nspk && !rc; ++spk) {
        for (int i = 0; i < R * R; ++i) L[i] = 0.0;                                     /* :1974-1975 */
        for (int i = 0; i < R; ++i) L[i * R + i] = 1.0;
        for (int dis = 0; dis < C; ++dis)                                             /* :1979-1986 */
            for (int i = 0; i < R; ++i)
                for (int j = 0; j <= i; ++j) L[i * R + j] += VEVT[(size_t)dis * R * R + i * R + j] * N[spk * C + dis];
        for (int i = 0; i < R; ++i)                                                   /* :1988-1992 */
            for (int j = i + 1; j < R; ++j) L[i * R + j] = L[j * R + i];
        if (orc_invert(R, L, invl)) { rc = 1; break; }                                /* :1995 */
        for (int i = 0; i < R; ++i) {                                                 /* :2478-2482 */
            aux[i] = 0.0;
            for (size_t k = 0; k < SV; ++k) aux[i] += F[spk * SV + k] * invvar[k] * V[i * SV + k];
        }
        for (int i = 0; i < R; ++i) {                                                 /* :2484-2488 */
            Y[spk * R + i] = 0.0;
            for (int k = 0; k < R; ++k) Y[spk * R + i] += aux[k] * invl[i * R + k];
        }
        for (int i = 0; i < R; ++i)                                                   /* :2489-2493 */
            for (int j = 0; j < R; ++j) invl[i * R + j] += Y[spk * R + i] * Y[spk * R + j];
        for (int dis = 0; dis < C; ++dis)                                             /* :2494-2500 */
            for (int i = 0; i < R * R; ++i) A[(size_t)dis * R * R + i] += invl[i] * N[spk * C + dis];
        for (int i = 0; i < R; ++i)                                                   /* :2501-2505 */
            for (size_t j = 0; j < SV; ++j) Cmx[i * SV + j] += Y[spk * R + i] * F[spk * SV + j];
    }
    free(L); free(invl); free(aux);
    return rc;
}

/* F[r] -= N[r] (.) (means + W[o] T + Dm (.) Z[o]), o = owner ? owner[r] : r -- one loop for substractMplusDZ (:3805-3822,
 * getMplusDZ :1860-1867), substractMplusVY (:3988-4005, getMplusVY :1880-1889), substractMplusVYplusDZ (:4400-4422: rows are
 * sessions, owner = their speaker).  NULL terms are absent. */
void orc_jfa_subtract(long rows, int C, int D, const double *N, double *F, const long *owner, const double *means, int R,
                      const double *Tm, const double *W, const double *Dm, const double *Z)
{
    const size_t SV = (size_t)C * D;
    double *v = (double *)malloc(sizeof(double) * SV);
    for (long r = 0; r < rows; ++r) {
        const long o = owner ? owner[r] : r;
        for (size_t k = 0; k < SV; ++k) v[k] = 0.0;
        if (Tm)
            for (size_t k = 0; k < SV; ++k)
                for (int j = 0; j < R; ++j) v[k] += Tm[j * SV + k] * W[o * R + j];     /* :4046-4048 */
        if (means)
            for (size_t k = 0; k < SV; ++k) v[k] += means[k];                          /* :4050 */
        if (Dm)
            for (size_t k = 0; k < SV; ++k) v[k] += Dm[k] * Z[o * SV + k];             /* :3866 */
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j) F[r * SV + i * D + j] -= v[i * D + j] * N[r * C + i]; /* :3817-3821 */
    }
    free(v);
}

/* substractUXUnThreaded (:4152-4172): every session's channel term leaves the statistics of its speaker */
void orc_jfa_subtract_sessions(long nspk, const long *sess_begin, int C, int D, const double *N_h, double *F_X, int R, const double *Um,
                               const double *X)
{
    const size_t SV = (size_t)C * D;
    double *ux = (double *)malloc(sizeof(double) * SV);
    for (long spk = 0; spk < nspk; ++spk)
        for (long h = sess_begin[spk]; h < sess_begin[spk + 1]; ++h) {
            for (size_t i = 0; i < SV; ++i) {                                          /* getUX :1803-1817 */
                ux[i] = 0.0;
                for (int j = 0; j < R; ++j) ux[i] += Um[j * SV + i] * X[h * R + j];
            }
            for (int k = 0; k < C; ++k)
                for (int i = 0; i < D; ++i) F_X[spk * SV + k * D + i] -= N_h[h * C + k] * ux[i + k * D]; /* :4166-4168 */
        }
    free(ux);
}

/* estimateZ (:3550-3573, tau < 0) / estimateZMAP (:3576-3594, tau >= 0) */
void orc_jfa_estimate_z(long nspk, int C, int D, const double *N, const double *F, const double *invvar, const double *Dm, double tau,
                        double *Z)
{
    const size_t SV = (size_t)C * D;
    for (long spk = 0; spk < nspk; ++spk)
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j) {
                const size_t k = (size_t)i * D + j;
                if (tau < 0.0) {
                    const double L = 1 + N[spk * C + i] * invvar[k] * Dm[k] * Dm[k];
                    Z[spk * SV + k] = F[spk * SV + k] * invvar[k] * Dm[k] / L;
                } else Z[spk * SV + k] = (tau / (tau + N[spk * C + i])) * Dm[k] * invvar[k] * F[spk * SV + k];
            }
}

/* estimateZandD (:3480-3516) */
void orc_jfa_estimate_z_and_d(long nspk, int C, int D, const double *N, const double *F, const double *invvar, double *Dm, double *Z)
{
    const size_t SV = (size_t)C * D;
    double *aux1 = (double *)calloc(SV, sizeof(double)), *aux2 = (double *)calloc(SV, sizeof(double));
    double *L = (double *)malloc(sizeof(double) * SV);
    for (long spk = 0; spk < nspk; ++spk) {
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j) {
                const size_t k = (size_t)i * D + j;
                L[k] = 1 + N[spk * C + i] * invvar[k] * Dm[k] * Dm[k];
                Z[spk * SV + k] = F[spk * SV + k] * invvar[k] * Dm[k] / L[k];
            }
        for (int i = 0; i < C; ++i)
            for (int j = 0; j < D; ++j) {
                const size_t k = (size_t)i * D + j;
                aux1[k] += (1 / L[k] + Z[spk * SV + k] * Z[spk * SV + k]) * N[spk * C + i];
                aux2[k] += Z[spk * SV + k] * F[spk * SV + k];
            }
    }
    for (size_t i = 0; i < SV; ++i) Dm[i] = aux2[i] / aux1[i];
    free(aux1); free(aux2); free(L);
}

/* substractMplusUX (:4336-4364): the SPEAKER statistics lose, for every session of the speaker, N_h (.) (m + U x_h) */
void orc_jfa_subtract_m_plus_ux(long nspk, const long *sess_begin, int C, int D, const double *N_h, double *F_X, const double *means,
                                int R, const double *Um, const double *X)
{
    const size_t SV = (size_t)C * D;
    double *mux = (double *)malloc(sizeof(double) * SV);
    for (long spk = 0; spk < nspk; ++spk)
        for (long h = sess_begin[spk]; h < sess_begin[spk + 1]; ++h) {
            for (size_t i = 0; i < SV; ++i) {                                          /* getMplusUX :1902-1911 */
                mux[i] = 0.0;
                for (int j = 0; j < R; ++j) mux[i] += Um[j * SV + i] * X[h * R + j];
                mux[i] += means[i];
            }
            for (int k = 0; k < C; ++k)
                for (int i = 0; i < D; ++i) F_X[spk * SV + k * D + i] -= N_h[h * C + k] * mux[i + k * D]; /* :4350-4354 */
        }
    free(mux);
}

/* TVAcc::initT, randomInitLaw "normal" (LIA_SpkTools/src/AccumulateTVStat.cpp:729-748) with boxMullerGenerator
 * (LIA_SpkTools/src/ScoreWarp.cpp:68-81): x1 = rand()/(float)RAND_MAX is kept as the phase of the next draw,
 * y = sqrt(-2 log x1) cos(2 pi x2), NaN / Inf values are redrawn, T(i,j) = y * (sum_k invvar_k) * 0.001.  glibc rand();
 * the reference never seeds (default seed 1): `seed` is applied with srand() first. */
void orc_tv_init_t(int R, long SV, const double *invvar, unsigned seed, double *Tm)
{
    const double PI2 = 3.14159265358979323846 * 2;
    double norm = 0.0;
    for (long k = 0; k < SV; ++k) norm += invvar[k];
    srand(seed);
    double x1 = (rand() / (float)RAND_MAX), x2;
    for (long e = 0; e < (long)R * SV; ++e) {
        double val;
        do {
            x2 = x1;
            x1 = (rand() / (float)RAND_MAX);
            val = sqrt(-2.0 * log(x1)) * cos(PI2 * x2);
        } while (isnan(val) || isinf(val));
        Tm[e] = val * norm * 0.001;
    }
}

