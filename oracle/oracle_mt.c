/*
 * oracle_mt.c -- pthread driver around the oracle's EM / TV-stat loops, used ONLY by bench.py's
 * cpu_baseline leg (TEST INFRASTRUCTURE).  It reproduces the reference's CPU partitioning:
 *   - EM: workers take frame ranges and own a private accumulator, merged at the end
 *     (LIA_SpkTools/src/AccumulateStat.cpp:170-212 EMthread, :286-292 addAccEM merge);
 *   - TV stats / i-vectors: contiguous utterance ranges per thread, disjoint output rows
 *     (LIA_SpkTools/src/AccumulateTVStat.cpp:498-507, :2282-2300).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

double orc_em_accumulate(int C, int D, const double *w, const double *mean, const double *covinv,
                         const double *x, long T, double weight,
                         double *occ, double *sx, double *sxx, double *count);
int orc_tv_estimate_w(long U, int C, int D, int R, const double *N, const double *F,
                      const double *Tm, const double *invvar, const double *TETt, double *W);

typedef struct {
    int C, D; const double *w, *mean, *covinv, *x; long T; double weight;
    double *occ, *sx, *sxx; double count, llk;
} em_job;

static void *em_worker(void *p)
{
    em_job *j = p;
    j->llk = orc_em_accumulate(j->C, j->D, j->w, j->mean, j->covinv, j->x, j->T, j->weight,
                               j->occ, j->sx, j->sxx, &j->count);
    return NULL;
}

double orc_em_accumulate_mt(int nthreads, int C, int D, const double *w, const double *mean,
                            const double *covinv, const double *x, long T, double weight,
                            double *occ, double *sx, double *sxx, double *count)
{
    if (nthreads < 1) nthreads = 1;
    em_job *jobs = calloc(nthreads, sizeof(em_job));
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    size_t CD = (size_t)C * D;
    long per = (T + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; ++i) {
        long b = i * per, e = b + per > T ? T : b + per;
        if (b > T) b = e = T;
        em_job *j = &jobs[i];
        j->C = C; j->D = D; j->w = w; j->mean = mean; j->covinv = covinv;
        j->x = x + b * D; j->T = e - b; j->weight = weight;
        j->occ = calloc(C, sizeof(double)); j->sx = calloc(CD, sizeof(double)); j->sxx = calloc(CD, sizeof(double));
        pthread_create(&th[i], NULL, em_worker, j);
    }
    double llk = 0.0;
    for (int i = 0; i < nthreads; ++i) {
        pthread_join(th[i], NULL);
        em_job *j = &jobs[i];
        for (int c = 0; c < C; ++c) occ[c] += j->occ[c];
        for (size_t k = 0; k < CD; ++k) { sx[k] += j->sx[k]; sxx[k] += j->sxx[k]; }
        *count += j->count;
        llk += j->llk;
        free(j->occ); free(j->sx); free(j->sxx);
    }
    free(jobs); free(th);
    return llk;
}

typedef struct {
    long U; int C, D, R; const double *N, *F, *Tm, *invvar, *TETt; double *W; int rc;
} w_job;

static void *w_worker(void *p)
{
    w_job *j = p;
    j->rc = orc_tv_estimate_w(j->U, j->C, j->D, j->R, j->N, j->F, j->Tm, j->invvar, j->TETt, j->W);
    return NULL;
}

int orc_tv_estimate_w_mt(int nthreads, long U, int C, int D, int R, const double *N, const double *F,
                         const double *Tm, const double *invvar, const double *TETt, double *W)
{
    if (nthreads < 1) nthreads = 1;
    w_job *jobs = calloc(nthreads, sizeof(w_job));
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    size_t SV = (size_t)C * D;
    long per = (U + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; ++i) {
        long b = i * per, e = b + per > U ? U : b + per;
        if (b > U) b = e = U;
        w_job *j = &jobs[i];
        j->U = e - b; j->C = C; j->D = D; j->R = R; j->N = N + b * C; j->F = F + b * SV;
        j->Tm = Tm; j->invvar = invvar; j->TETt = TETt; j->W = W + b * R;
        pthread_create(&th[i], NULL, w_worker, j);
    }
    int rc = 0;
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); rc |= jobs[i].rc; }
    free(jobs); free(th);
    return rc;
}

/* IvExtractor end to end (LIA_SpkDet/IvExtractor/src/IvExtractor.cpp:70-148) with the reference's thread split: contiguous
 * ranges of statistics rows per thread, for the statistics (AccumulateTVStat.cpp:498-507) and for estimateW (:2282-2300).  Every
 * worker runs both phases on its own rows: computeAndAccumulateTVStat, substractM, estimateW (TETt is computed once by the
 * caller, like estimateTETt before the threads start).  x [T x D] fp64 (Feature::getDataVector), utt_begin [U + 1]. */
void orc_tv_stats(int C, int D, const double *w, const double *mean, const double *covinv,
                  const double *x, long T, const long *utt, double *N, double *F);
void orc_tv_subtract_m(long U, int C, int D, const double *N, double *F, const double *ubm_means);

typedef struct {
    long U; int C, D, R; const double *w, *mean, *covinv, *x; const long *utt_begin;
    const double *Tm, *invvar, *TETt; double *W; int rc;
} iv_job;

static void *iv_worker(void *p)
{
    iv_job *j = p;
    if (j->U <= 0) { j->rc = 0; return NULL; }
    const size_t SV = (size_t)j->C * j->D;
    const long t0 = j->utt_begin[0], T = j->utt_begin[j->U] - t0;
    double *N = calloc((size_t)j->U * j->C, sizeof(double)), *F = calloc((size_t)j->U * SV, sizeof(double));
    long *utt = malloc(sizeof(long) * (T > 0 ? T : 1));
    for (long u = 0; u < j->U; ++u)
        for (long t = j->utt_begin[u]; t < j->utt_begin[u + 1]; ++t) utt[t - t0] = u;
    orc_tv_stats(j->C, j->D, j->w, j->mean, j->covinv, j->x + (size_t)t0 * j->D, T, utt, N, F);
    orc_tv_subtract_m(j->U, j->C, j->D, N, F, j->mean);
    j->rc = orc_tv_estimate_w(j->U, j->C, j->D, j->R, N, F, j->Tm, j->invvar, j->TETt, j->W);
    free(N); free(F); free(utt);
    return NULL;
}

int orc_iv_extract_mt(int nthreads, long U, int C, int D, int R, const double *w, const double *mean, const double *covinv,
                      const double *x, const long *utt_begin, const double *Tm, const double *invvar, const double *TETt, double *W)
{
    if (nthreads < 1) nthreads = 1;
    iv_job *jobs = calloc(nthreads, sizeof(iv_job));
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    long per = (U + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; ++i) {
        long b = i * per, e = b + per > U ? U : b + per;
        if (b > U) b = e = U;
        iv_job *j = &jobs[i];
        j->U = e - b; j->C = C; j->D = D; j->R = R; j->w = w; j->mean = mean; j->covinv = covinv; j->x = x;
        j->utt_begin = utt_begin + b; j->Tm = Tm; j->invvar = invvar; j->TETt = TETt; j->W = W + (size_t)b * R;
        pthread_create(&th[i], NULL, iv_worker, j);
    }
    int rc = 0;
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); rc |= jobs[i].rc; }
    free(jobs); free(th);
    return rc;
}

/* IvTest Mahalanobis scoring in the reference's per-pair form (PldaTest::mahalanobisDistance, PldaTools.cpp:3882-3909: an O(dim^2)
 * matrix-vector product per (model, segment) pair) with the model rows split over threads.  The reference runs this loop on ONE
 * thread; the split exists so that bench.py can quote a many-core figure next to the single-thread one (BASELINE.md section 3,
 * config 5).  models [dim x M], segs [dim x S], scores [M x S]. */
void orc_score_mahalanobis(int dim, long M, long S, const double *models, const double *segs,
                           const double *Mah, const unsigned char *trials, double *scores);

typedef struct { int dim; long M, S; const double *models, *segs, *Mah; double *scores; } mah_job;

static void *mah_worker(void *p)
{
    mah_job *j = p;
    if (j->M > 0) orc_score_mahalanobis(j->dim, j->M, j->S, j->models, j->segs, j->Mah, NULL, j->scores);
    return NULL;
}

void orc_score_mahalanobis_mt(int nthreads, int dim, long M, long S, const double *models, const double *segs, const double *Mah,
                              double *scores)
{
    if (nthreads < 1) nthreads = 1;
    mah_job *jobs = calloc(nthreads, sizeof(mah_job));
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    double **blk = calloc(nthreads, sizeof(double *));
    const long per = (M + nthreads - 1) / nthreads;
    for (int i = 0; i < nthreads; ++i) {
        long b = i * per, e = b + per > M ? M : b + per;
        if (b > M) b = e = M;
        mah_job *j = &jobs[i];
        const long n = e - b;
        /* the scalar routine addresses models as [dim x M] with stride M: hand every worker a compact [dim x n] copy of its columns */
        blk[i] = malloc(sizeof(double) * (size_t)dim * (n > 0 ? n : 1));
        for (int k = 0; k < dim; ++k)
            for (long m = 0; m < n; ++m) blk[i][(size_t)k * n + m] = models[(size_t)k * M + b + m];
        j->dim = dim; j->M = n; j->S = S; j->models = blk[i]; j->segs = segs; j->Mah = Mah; j->scores = scores + (size_t)b * S;
        pthread_create(&th[i], NULL, mah_worker, j);
    }
    for (int i = 0; i < nthreads; ++i) { pthread_join(th[i], NULL); free(blk[i]); }
    free(jobs); free(th); free(blk);
}
