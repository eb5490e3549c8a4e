/*
 * oracle_mt.c -- pthread drivers around the oracle's EM / TV loops: TEST INFRASTRUCTURE, used only by bench.py's
 * cpu_baseline legs and by tests/test_oracle_selfcheck.py (which holds them to the single-thread oracle); never linked into,
 * loaded by or called from the product.  They reproduce the reference's CPU partitioning:
 *   - EM: workers take frame ranges and own a private accumulator, merged at the end
 *     (LIA_SpkTools/src/AccumulateStat.cpp:170-212 EMthread, :286-292 addAccEM merge);
 *   - TV stats / i-vectors: contiguous utterance ranges per thread, disjoint output rows
 *     (LIA_SpkTools/src/AccumulateTVStat.cpp:498-507, :2282-2300);
 *   - one whole TotalVariability iteration: estimateTETtThreaded (:826-950), estimateAandCThreaded (:1831-2052: utterance ranges,
 *     private R / r / meanW, A and C shared under two mutexes), updateTestimate (:974-1005), minDivergence (:2056-2099).
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

/* ---- One TotalVariability iteration with the reference's CPU threading --------------------
 *   estimateTETtThreaded   AccumulateTVStat.cpp:826-950   Gaussian ranges per thread
 *   estimateAandCThreaded  :1831-2052  utterance ranges per thread (offset = U / nthreads, the first U % nthreads threads take one
 *                          more, :1991-2000); every thread owns its R / r / meanW (:1975-1986, summed after the join :2035-2044) and
 *                          adds into the SHARED A (all C blocks, full R x R) and C under two mutexes (:1916-1935).  The reference
 *                          re-initialises both mutexes at the top of every thread (:1833-1834 -- undefined while another thread
 *                          holds one); here they are initialised once, which is what that code means to do.
 *   updateTestimate        :974-1005   SINGLE-threaded in the reference.  upd_threads > 1 splits the Gaussians over threads --
 *                          a faster CPU program than the reference's; bench.py says which was timed.
 *   minDivergence          :2056-2099  single-threaded in the reference; the T <- Ch T product is split over columns when
 *                          upd_threads > 1 (same remark).
 * F must already be centred (substractM).  Tm [R x SV] and ubm_means [SV] are updated in place; W [U x R] receives the i-vectors
 * of the E-step; phase_s[4] the seconds of TETt / estimateAandC / updateTestimate / minDivergence.  Returns 0, 1 on a singular
 * matrix, 2 when the 2 x C x R x R doubles of TETt and A cannot be allocated. */
void orc_tv_tett(int C, int D, int R, const double *Tm, const double *invvar, double *TETt);
int orc_invert(int n, const double *a_in, double *inv);
int orc_upper_cholesky(int n, const double *a, double *ch);
#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct { int c0, c1, C, D, R; const double *Tm, *invvar; double *TETt; } tett_job;
static void *tett_worker(void *p)
{
    tett_job *j = p;   /* TETthread: the Gaussians [c0, c1) */
    size_t SV = (size_t)j->C * j->D;
    for (int d = j->c0; d < j->c1; ++d) {
        double *o = j->TETt + (size_t)d * j->R * j->R;
        for (int i = 0; i < j->R; ++i)
            for (int k2 = 0; k2 <= i; ++k2) {
                double s = 0.0;
                for (int k = 0; k < j->D; ++k)
                    s += j->Tm[i * SV + (size_t)d * j->D + k] * j->invvar[(size_t)d * j->D + k] * j->Tm[k2 * SV + (size_t)d * j->D + k];
                o[(size_t)i * j->R + k2] = s;
            }
        for (int i = 0; i < j->R; ++i)
            for (int k2 = i + 1; k2 < j->R; ++k2) o[(size_t)i * j->R + k2] = o[(size_t)k2 * j->R + i];
    }
    return NULL;
}

typedef struct {
    long u0, u1; int C, D, R; const double *N, *F, *Tm, *invvar, *TETt;
    double *W, *A, *Cmx, *Rm, *r, *meanW; pthread_mutex_t *mutexA, *mutexC; int rc;
} ac_job;
static void *ac_worker(void *p)
{
    ac_job *j = p;     /* estimateAandCTthread, :1831-1940 */
    const int C = j->C, R = j->R;
    const size_t SV = (size_t)C * j->D, RR = (size_t)R * R;
    double *L = malloc(sizeof(double) * RR), *Li = malloc(sizeof(double) * RR), *aux = malloc(sizeof(double) * R);
    for (long u = j->u0; u < j->u1; ++u) {
        memset(L, 0, sizeof(double) * RR);
        for (int i = 0; i < R; ++i) L[(size_t)i * R + i] = 1.0;
        for (int d = 0; d < C; ++d) {
            const double *t = j->TETt + (size_t)d * RR, n = j->N[u * C + d];
            for (int i = 0; i < R; ++i)
                for (int k = 0; k <= i; ++k) L[(size_t)i * R + k] += t[(size_t)i * R + k] * n;
        }
        for (int i = 0; i < R; ++i)
            for (int k = i + 1; k < R; ++k) L[(size_t)i * R + k] = L[(size_t)k * R + i];
        if (orc_invert(R, L, Li)) { j->rc = 1; break; }
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (size_t k = 0; k < SV; ++k) s += j->F[u * SV + k] * j->invvar[k] * j->Tm[i * SV + k];
            aux[i] = s;
        }
        double *y = j->W + u * R;
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
            for (int k = 0; k < R; ++k) s += aux[k] * Li[(size_t)i * R + k];
            y[i] = s;
        }
        for (int k = 0; k < R; ++k) j->meanW[k] += y[k];
        for (int i = 0; i < R; ++i) {
            for (int k = 0; k < R; ++k) {
                Li[(size_t)i * R + k] += y[i] * y[k];
                j->Rm[(size_t)i * R + k] += Li[(size_t)i * R + k];
            }
            j->r[i] += y[i];
        }
        pthread_mutex_lock(j->mutexA);
        for (int d = 0; d < C; ++d) {
            const double n = j->N[u * C + d];
            double *a = j->A + (size_t)d * RR;
            for (size_t e = 0; e < RR; ++e) a[e] += Li[e] * n;
        }
        pthread_mutex_unlock(j->mutexA);
        pthread_mutex_lock(j->mutexC);
        for (int i = 0; i < R; ++i)
            for (size_t k = 0; k < SV; ++k) j->Cmx[i * SV + k] += y[i] * j->F[u * SV + k];
        pthread_mutex_unlock(j->mutexC);
    }
    free(L); free(Li); free(aux);
    return NULL;
}

typedef struct { int c0, c1, C, D, R; const double *A, *Cmx; double *Tm; int rc; } upd_job;
static void *upd_worker(void *p)
{
    upd_job *j = p;    /* updateTestimate's loop body (:981-999) on the Gaussians [c0, c1) */
    const int R = j->R, D = j->D;
    const size_t SV = (size_t)j->C * D, RR = (size_t)R * R;
    double *Ai = malloc(sizeof(double) * RR);
    for (int c = j->c0; c < j->c1; ++c) {
        if (orc_invert(R, j->A + (size_t)c * RR, Ai)) { j->rc = 1; break; }
        for (int i = 0; i < R; ++i)
            for (int d = 0; d < D; ++d) {
                double s = 0.0;
                for (int k = 0; k < R; ++k) s += Ai[(size_t)i * R + k] * j->Cmx[k * SV + (size_t)c * D + d];
                j->Tm[i * SV + (size_t)c * D + d] = s;
            }
    }
    free(Ai);
    return NULL;
}

typedef struct { size_t k0, k1, SV; int R; const double *ch, *Tin; double *Tout; } md_job;
static void *md_worker(void *p)
{
    md_job *j = p;     /* tmpV(i, k) = sum_l Ch(i, l) T(l, k) on the columns [k0, k1) (:2082-2088) */
    for (int i = 0; i < j->R; ++i) {
        double *o = j->Tout + (size_t)i * j->SV;
        for (size_t k = j->k0; k < j->k1; ++k) o[k] = 0.0;
        for (int l = 0; l < j->R; ++l) {
            const double c = j->ch[(size_t)i * j->R + l];
            if (c == 0.0) continue;
            const double *t = j->Tin + (size_t)l * j->SV;
            for (size_t k = j->k0; k < j->k1; ++k) o[k] += c * t[k];
        }
    }
    return NULL;
}

int orc_tv_em_iteration_mt(int nthreads, int upd_threads, long U, int C, int D, int R, const double *N, const double *F, double *Tm,
                           const double *invvar, double *ubm_means, double *W, double *phase_s)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > U) nthreads = (int)U;            /* :1949 */
    if (upd_threads < 1) upd_threads = 1;
    const size_t SV = (size_t)C * D, RR = (size_t)R * R;
    double *TETt = malloc(sizeof(double) * C * RR), *A = calloc((size_t)C * RR, sizeof(double));
    if (!TETt || !A) { free(TETt); free(A); return 2; }
    double *Cmx = calloc((size_t)R * SV, sizeof(double)), *Rm = calloc(RR, sizeof(double)), *r = calloc(R, sizeof(double));
    double *meanW = calloc(R, sizeof(double));
    pthread_t *th = malloc(sizeof(pthread_t) * (nthreads > upd_threads ? nthreads : upd_threads));
    int rc = 0;
    /* estimateTETtThreaded */
    double t0 = now_s();
    {
        tett_job *jobs = calloc(nthreads, sizeof(tett_job));
        int per = (C + nthreads - 1) / nthreads;
        for (int i = 0; i < nthreads; ++i) {
            int b = i * per, e = b + per > C ? C : b + per;
            if (b > C) b = e = C;
            tett_job tj = {b, e, C, D, R, Tm, invvar, TETt};
            jobs[i] = tj;
            pthread_create(&th[i], NULL, tett_worker, &jobs[i]);
        }
        for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
        free(jobs);
    }
    double t1 = now_s();
    /* estimateAandCThreaded */
    {
        pthread_mutex_t mutexA, mutexC;
        pthread_mutex_init(&mutexA, NULL); pthread_mutex_init(&mutexC, NULL);
        ac_job *jobs = calloc(nthreads, sizeof(ac_job));
        memset(W, 0, sizeof(double) * U * R);
        long offset = U / nthreads, re = U - (long)nthreads * offset, bottom = 0;
        for (int i = 0; i < nthreads; ++i) {
            long up = bottom + offset + (i < re ? 1 : 0);
            ac_job *j = &jobs[i];
            j->u0 = bottom; j->u1 = up; j->C = C; j->D = D; j->R = R; j->N = N; j->F = F; j->Tm = Tm; j->invvar = invvar; j->TETt = TETt;
            j->W = W; j->A = A; j->Cmx = Cmx; j->mutexA = &mutexA; j->mutexC = &mutexC;
            j->Rm = calloc(RR, sizeof(double)); j->r = calloc(R, sizeof(double)); j->meanW = calloc(R, sizeof(double));
            pthread_create(&th[i], NULL, ac_worker, j);
            bottom = up;
        }
        for (int i = 0; i < nthreads; ++i) {
            pthread_join(th[i], NULL);
            ac_job *j = &jobs[i];
            rc |= j->rc;
            for (size_t e = 0; e < RR; ++e) Rm[e] += j->Rm[e];
            for (int k = 0; k < R; ++k) { r[k] += j->r[k]; meanW[k] += j->meanW[k]; }
            free(j->Rm); free(j->r); free(j->meanW);
        }
        for (int k = 0; k < R; ++k) meanW[k] /= (double)U;
        pthread_mutex_destroy(&mutexA); pthread_mutex_destroy(&mutexC);
        free(jobs);
    }
    free(TETt);
    double t2 = now_s();
    /* updateTestimate */
    if (!rc) {
        upd_job *jobs = calloc(upd_threads, sizeof(upd_job));
        int per = (C + upd_threads - 1) / upd_threads;
        for (int i = 0; i < upd_threads; ++i) {
            int b = i * per, e = b + per > C ? C : b + per;
            if (b > C) b = e = C;
            upd_job uj = {b, e, C, D, R, A, Cmx, Tm, 0};
            jobs[i] = uj;
            pthread_create(&th[i], NULL, upd_worker, &jobs[i]);
        }
        for (int i = 0; i < upd_threads; ++i) { pthread_join(th[i], NULL); rc |= jobs[i].rc; }
        free(jobs);
    }
    double t3 = now_s();
    /* minDivergence */
    if (!rc) {
        const double n_sessions = (double)U;
        for (int i = 0; i < R; ++i) r[i] /= n_sessions;
        for (int i = 0; i < R; ++i)
            for (int k = 0; k < R; ++k) Rm[(size_t)i * R + k] = Rm[(size_t)i * R + k] / n_sessions - r[i] * r[k];
        double *ch = malloc(sizeof(double) * RR);
        if (orc_upper_cholesky(R, Rm, ch)) rc = 1;
        else {
            for (size_t k = 0; k < SV; ++k)
                for (int l = 0; l < R; ++l) ubm_means[k] += meanW[l] * Tm[(size_t)l * SV + k];
            double *tmp = malloc(sizeof(double) * R * SV);
            md_job *jobs = calloc(upd_threads, sizeof(md_job));
            size_t per = (SV + upd_threads - 1) / upd_threads;
            for (int i = 0; i < upd_threads; ++i) {
                size_t b = i * per, e = b + per > SV ? SV : b + per;
                if (b > SV) b = e = SV;
                md_job mj = {b, e, SV, R, ch, Tm, tmp};
                jobs[i] = mj;
                pthread_create(&th[i], NULL, md_worker, &jobs[i]);
            }
            for (int i = 0; i < upd_threads; ++i) pthread_join(th[i], NULL);
            memcpy(Tm, tmp, sizeof(double) * R * SV);
            free(tmp); free(jobs);
        }
        free(ch);
    }
    double t4 = now_s();
    if (phase_s) { phase_s[0] = t1 - t0; phase_s[1] = t2 - t1; phase_s[2] = t3 - t2; phase_s[3] = t4 - t3; }
    free(A); free(Cmx); free(Rm); free(r); free(meanW); free(th);
    return rc;
}
